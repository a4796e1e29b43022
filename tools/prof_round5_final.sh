#!/bin/bash
# On the GPU box: the end-of-round evidence on the final build - smoke, the default bench line, its kernel table / timeline, the
# PMC traffic of the dominant kernel, configs[4] at batch 32 with the two shapes of hgemm_kernel separated.
# usage: JSTSP_GIT_SHA=<sha> tools/prof_round5_final.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -c 300 gpurun_out/r05_bench_default.json; echo
timeout 600 bash tools/prof_bench.sh r05_bench --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 | tail -24
timeout 900 bash tools/pmc_bench.sh r05 | tail -3
timeout 900 bash tools/prof_cmd.sh r05_cfg5_b32 tools/probe/cfg5_angles.py 32 | tail -4; cat gpurun_out/r05_cfg5_b32_kernel_stats_split.csv | head -12
timeout 300 python3 bench.py --steps 10 --warmup 2 --no-ce --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 > gpurun_out/r05_bench_two_outputs.json 2>/dev/null
