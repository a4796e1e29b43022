#!/bin/bash
# On the GPU box: HBM traffic counters of one bench step, separate passes per counter as the microarch guide prescribes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass); summaries to gpurun_out/<tag>_pmc_<counter>.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$tag_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o $tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/${tag}_pmc_$c.err
  f=$(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $f fused_pass > $R/gpurun_out/${tag}_pmc_$c.txt
  python3 $R/tools/pmc_summary.py $f hgemm >> $R/gpurun_out/${tag}_pmc_$c.txt
  python3 $R/tools/pmc_summary.py $f "cgemm_kernel<128, 0, false, false, 2>" >> $R/gpurun_out/${tag}_pmc_$c.txt
  python3 $R/tools/pmc_summary.py $f hgram >> $R/gpurun_out/${tag}_pmc_$c.txt
  cat $R/gpurun_out/${tag}_pmc_$c.txt
done
