#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for m in 0 1; do echo "== MAP=$m"; JSTSP_HGEMM_MAP=$m python3 tools/bench_cfg1_omp.py 2>&1 | tail -6; JSTSP_HGEMM_MAP=$m python3 tools/bench_cfg3.py 2>&1 | tail -12; done
python3 -m pytest tests -m gpu -q -x -k "config5 or cfg5 or shared or baselines or toeplitz" 2>&1 | tail -3
