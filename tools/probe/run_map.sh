#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests -m gpu -q -k "config3 or mc_ or svt or large_orders or kernels or vamp or eig or mex" 2>&1 | tail -4
bash tools/prof_cmd.sh r05b_cfg3 tools/bench_cfg3.py 1024 | tail -12
