#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests -m gpu -q -x -k "omp or OMP or mmv" 2>&1 | tail -3
bash tools/prof_cmd.sh r05b_cfg1_omp tools/probe/omp_b1.py | tail -6
