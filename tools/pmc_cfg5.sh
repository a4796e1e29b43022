#!/bin/bash
# On the GPU box: SQ counters (pass 1) and L2 counters (pass 2) of the two big contractions at configs[4] (tools/probe/cfg5_angles.py),
# per-kernel averages to gpurun_out/<tag>_pmc_cfg5.txt         usage: tools/pmc_cfg5.sh <tag> [batch] [Imax]
tag=$1; batch=${2:-32}; imax=${3:-4}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/${tag}_pmc_cfg5.txt; : > $out
pass() {
    d=/tmp/pmccfg5_${tag}_$1; rm -rf $d
    shift_ctrs="${@:2}"
    rocprofv3 --kernel-trace --pmc $shift_ctrs --output-format csv -d $d -o $tag -- python3 $R/tools/probe/cfg5_angles.py $batch $imax > /dev/null 2> $R/gpurun_out/${tag}_pmc_cfg5_$1.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -z "$f" ]; then echo "pass $1: no counter csv" >> $out; tail -5 $R/gpurun_out/${tag}_pmc_cfg5_$1.err >> $out; return; fi
    echo "# pass $1: $shift_ctrs" >> $out
    python3 $R/tools/pmc_summary.py $f hgemm >> $out
}
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
cat $out
