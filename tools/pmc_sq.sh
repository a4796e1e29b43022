#!/bin/bash
# On the GPU box: SQ counters (one pass, 8 slots) of a python tool, per-kernel averages to gpurun_out/<tag>_pmc_sq.txt
# usage: tools/pmc_sq.sh <tag> <script.py> [args...]     (kernels matched: fused_pass, hgemm, jacobi128, lanczos, omp_step)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/gpurun_out"
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/pmcsq_$tag; rm -rf $d
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $d -o $tag -- python3 $script "$@" > /dev/null 2> $R/gpurun_out/${tag}_pmc_sq.err
f=$(find $d -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then echo "no counter csv"; tail -5 $R/gpurun_out/${tag}_pmc_sq.err; exit 1; fi
: > $R/gpurun_out/${tag}_pmc_sq.txt
for k in fused_pass hgemm jacobi128 lanczos omp_step hgram; do python3 $R/tools/pmc_summary.py $f $k >> $R/gpurun_out/${tag}_pmc_sq.txt; done
cat $R/gpurun_out/${tag}_pmc_sq.txt
