#!/bin/bash
# On the GPU box: every profile of round 5 in one call - the bench line, its kernel table and timeline, the PMC traffic of the
# dominant kernel, the strict-fp32 table, tables of configs[0], [2], [4] (batch 32), the two-output and sweep lines, host-path
# rates.  Outputs under gpurun_out/r05_*.   usage: JSTSP_GIT_SHA=<sha> tools/prof_round5.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -c 400 gpurun_out/r05_bench_default.json; echo
timeout 600 bash tools/prof_bench.sh r05_bench --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 | tail -26
timeout 900 bash tools/pmc_bench.sh r05 | tail -3
JSTSP_H2=0 timeout 600 bash tools/prof_bench.sh r05_strict_fp32 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 | tail -14
JSTSP_OVERLAP=0 timeout 600 bash tools/prof_bench.sh r05_bench_serial --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 | tail -22
timeout 300 python3 bench.py --steps 10 --warmup 2 --no-ce --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 > gpurun_out/r05_bench_two_outputs.json 2>/dev/null; tail -c 300 gpurun_out/r05_bench_two_outputs.json | head -c 300; echo
timeout 600 python3 bench.py --sweep > gpurun_out/r05_bench_sweep_config3.json 2>/dev/null; head -c 400 gpurun_out/r05_bench_sweep_config3.json; echo
timeout 300 bash tools/prof_cmd.sh r05_cfg1_omp tools/bench_cfg1_omp.py | tail -12
timeout 600 bash tools/prof_cmd.sh r05_cfg3 tools/bench_cfg3.py 1024 | tail -24
timeout 900 bash tools/prof_cmd.sh r05_cfg5_b32 tools/probe/cfg5_angles.py 32 | tail -24
timeout 600 python3 tools/host_path_rate.py 256 c64 > gpurun_out/r05_host_path.txt 2>&1; grep -v amdgpu gpurun_out/r05_host_path.txt
timeout 300 python3 tools/parity_fixture_check.py --n 640 --out gpurun_out/r05_rv_refresh1.json "" "JSTSP_RV_REFRESH=1" "JSTSP_RV_REFRESH=2" 2>&1 | grep -v amdgpu | cut -c1-200
for acc in 1 3 0; do echo "PASS_ACC=$acc"; JSTSP_PASS_ACC=$acc timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['parity']['whole_batch']['max_abs_dNMSE'], d['parity']['whole_batch']['rms_dNMSE'])"; done > gpurun_out/r05_pass_acc_speed.txt 2>&1; cat gpurun_out/r05_pass_acc_speed.txt
