#!/bin/bash
# On the GPU box: first measurement of round 5 - new tests, the bench line, its kernel table, configs[2] timings with both
# register budgets of the order-128 Lanczos kernel, host-path rates, and the accumulation variants of the pass on the fixture
# the defaults are chosen on (seed 20190913: NOT the held-out one).  Outputs under gpurun_out/r05a_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_lanczos_warm.py tests/test_gpu_toeplitz_dictionary.py tests/test_gpu_parity_proposed.py -q > gpurun_out/r05a_tests.log 2>&1; tail -25 gpurun_out/r05a_tests.log
timeout 600 python3 bench.py --steps 10 --warmup 2 > gpurun_out/r05a_bench_default.json 2> gpurun_out/r05a_bench_default.err; tail -c 2500 gpurun_out/r05a_bench_default.json
timeout 600 bash tools/prof_bench.sh r05a --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 | tail -32
timeout 900 python3 tools/parity_fixture_check.py --out gpurun_out/r05a_acc_setA.json "" "JSTSP_PASS_ACC=1" "JSTSP_PASS_ACC=2" "JSTSP_PASS_ACC=1,JSTSP_FUSED_PARTS=8" "JSTSP_FUSED_PARTS=8" > gpurun_out/r05a_acc_setA.txt 2>&1; cat gpurun_out/r05a_acc_setA.txt
for acc in 0 1 2; do echo "PASS_ACC=$acc"; JSTSP_PASS_ACC=$acc timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --no-strict-fp32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['parity']['whole_batch']['max_abs_dNMSE'], d['parity']['whole_batch']['rms_dNMSE'])"; done > gpurun_out/r05a_acc_speed.txt 2>&1; cat gpurun_out/r05a_acc_speed.txt
for occ in 2 1; do echo "LZ128_OCC=$occ"; JSTSP_LZ128_OCC=$occ timeout 300 python3 tools/bench_cfg3.py 1024; done > gpurun_out/r05a_cfg3.txt 2>&1; cat gpurun_out/r05a_cfg3.txt
echo "cold lanczos:"; JSTSP_LANCZOS_WARM=0 timeout 300 python3 tools/bench_cfg3.py 1024 2>&1 | tail -1 | tee -a gpurun_out/r05a_cfg3.txt
timeout 600 python3 tools/host_path_rate.py 256 c64 > gpurun_out/r05a_host_path.txt 2>&1; cat gpurun_out/r05a_host_path.txt
JSTSP_HOST_COMPACT=0 timeout 600 python3 tools/host_path_rate.py 256 c64 > gpurun_out/r05a_host_path_nocompact.txt 2>&1; cat gpurun_out/r05a_host_path_nocompact.txt
