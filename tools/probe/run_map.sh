#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 1 0; do echo "== J128_ORDER=$k"; JSTSP_J128_ORDER=$k python3 tools/probe/svt128_err.py 2>&1 | tail -1; JSTSP_J128_ORDER=$k python3 tools/bench_cfg3.py 1024 2>&1 | tail -4 | head -3; done
