#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
JSTSP_KPACK=0 python3 tools/probe/cfg5_angles.py 32 2>&1 | tail -2
JSTSP_KPACK=1 python3 tools/probe/cfg5_angles.py 32 2>&1 | tail -2
JSTSP_KPACK=1 bash tools/prof_cmd.sh r05c_cfg5_kpack tools/probe/cfg5_angles.py 32 | tail -7 | cut -c1-150
