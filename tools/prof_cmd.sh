#!/bin/bash
# On the GPU box: kernel-trace profile of an arbitrary python tool; summary to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_cmd.sh <tag> <script.py> [args...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_$tag -o $tag -- python3 $script "$@" > $R/gpurun_out/${tag}.log 2>&1
python3 $R/tools/kstats_db.py /tmp/prof_$tag/${tag}_results.db 0.5 > $R/gpurun_out/${tag}_kernel_stats.csv
tail -5 $R/gpurun_out/${tag}.log
python3 $R/tools/ktimeline_db.py /tmp/prof_$tag/${tag}_results.db > $R/gpurun_out/${tag}_timeline.txt 2>&1
python3 $R/tools/kstats_db.py /tmp/prof_$tag/${tag}_results.db 0.2 split > $R/gpurun_out/${tag}_kernel_stats_split.csv
cat $R/gpurun_out/${tag}_kernel_stats.csv
