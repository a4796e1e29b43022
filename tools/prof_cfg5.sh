#!/bin/bash
# On the GPU box: kernel-trace profile of tools/probe/cfg5_angles.py (configs[4]); summary to gpurun_out/<tag>_cfg5_kernel_stats.csv
# usage: tools/prof_cfg5.sh <tag> [batch] [Imax]
tag=$1; batch=${2:-32}; imax=${3:-8}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_$tag -o $tag -- python3 $R/tools/probe/cfg5_angles.py $batch $imax > $R/gpurun_out/${tag}_cfg5.txt 2> $R/gpurun_out/${tag}_cfg5.err
cat $R/gpurun_out/${tag}_cfg5.txt
python3 $R/tools/kstats_db.py /tmp/prof_$tag/${tag}_results.db 0.2 split > $R/gpurun_out/${tag}_cfg5_kernel_stats.csv
cut -c1-180 $R/gpurun_out/${tag}_cfg5_kernel_stats.csv | head -16
