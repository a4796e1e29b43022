#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests -m gpu -q -x -k "sparse_admm or config3" 2>&1 | tail -5
for f in 0 1; do echo "== OVERLAP=$f"; JSTSP_SADMM_OVERLAP=$f python3 tools/bench_cfg3.py 1024 2>&1 | tail -1; JSTSP_SADMM_OVERLAP=$f python3 tools/bench_cfg3.py 256 2>&1 | tail -1;  done
