#!/bin/bash
# On the GPU box: kernel trace of one 256-trial solve, window statistics + per-kernel summary
# usage: tools/prof_windows.sh <tag> [env assignments...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_$tag -o $tag -- python3 $R/tools/run_proposed_once.py 256 60 1 > $R/gpurun_out/${tag}.log 2>&1
tail -2 $R/gpurun_out/${tag}.log
python3 $R/tools/kstats_db.py /tmp/prof_$tag/${tag}_results.db 0.5 > $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/tools/kwindows_db.py /tmp/prof_$tag/${tag}_results.db 100 8 > $R/gpurun_out/${tag}_windows.txt 2>&1
head -40 $R/gpurun_out/${tag}_windows.txt
