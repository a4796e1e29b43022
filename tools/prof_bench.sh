#!/bin/bash
# On the GPU box: kernel-trace profile of a short bench run; the per-kernel summary goes to gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_bench.sh <tag> [bench args...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_$tag -o $tag -- python3 $R/bench.py "$@" > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
python3 $R/tools/kstats_db.py /tmp/prof_$tag/${tag}_results.db 0.2 > $R/gpurun_out/${tag}_kernel_stats.csv
cat $R/gpurun_out/${tag}_kernel_stats.csv
python3 $R/tools/ktimeline_db.py /tmp/prof_$tag/${tag}_results.db > $R/gpurun_out/${tag}_timeline.txt 2>&1
python3 -c "
import json,sys
d=json.loads(open('$R/gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step'],'frac',d['roofline']['frac'] if d.get('roofline') else None, 'parity', d.get('parity'))"
