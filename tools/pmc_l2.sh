#!/bin/bash
# On the GPU box: L2 (TCC) and vector-L1 (TCP) counters of a python tool in two passes, per-kernel averages per dispatch to
# gpurun_out/<tag>_pmc_l2.txt        usage: tools/pmc_l2.sh <tag> <script.py> [args...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/gpurun_out"
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/${tag}_pmc_l2.txt; : > $out
pass() {
    d=/tmp/pmcl2_${tag}_$1; rm -rf $d
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -o $tag -- python3 $script $ARGS > /dev/null 2> $R/gpurun_out/${tag}_pmc_l2_$name.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -z "$f" ]; then echo "pass $name: no counter csv" >> $out; tail -5 $R/gpurun_out/${tag}_pmc_l2_$name.err >> $out; return; fi
    echo "# pass $name: $*" >> $out
    for k in fused_pass hgram3 hgram_kernel cgemm_kernel reduce_parts hgemm_kernel step_v lanczos jacobi2 pack_as; do python3 $R/tools/pmc_summary.py $f $k >> $out; done
}
ARGS="$*"
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
cat $out
