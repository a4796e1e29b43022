#!/bin/bash
# On the GPU box: every profile of round 4 in one call - kernel tables of the bench (BASELINE configs[1]), of configs[0]
# (dense OMP), configs[2] (128 x 128: svt / mc_svt / sparse_admm at batch 1024), configs[4] (N=64, G2=4096: angles + VAMP),
# the PMC traffic of the bench's dominant kernel, and the bench line itself.  Outputs under gpurun_out/r04_*.
# usage: JSTSP_GIT_SHA=<sha> tools/prof_round4.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; tail -c 600 gpurun_out/r04_bench_default.json
bash tools/prof_bench.sh r04_bench --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 | tail -30
bash tools/pmc_bench.sh r04 | tail -3
bash tools/prof_cmd.sh r04_cfg1_omp tools/bench_cfg1_omp.py | tail -25
bash tools/prof_cmd.sh r04_cfg3 tools/bench_cfg3.py 1024 | tail -30
bash tools/prof_cmd.sh r04_cfg5 tools/run_cfg5_shape.py 2 | tail -30
