#!/bin/bash
# On the GPU box: the end-of-round evidence on the final build - smoke, the default bench line, its kernel table / timeline, the
# PMC traffic of the dominant kernel, the two-output and sweep lines, configs[2] shapes, the whole -m gpu suite (which also writes
# gpurun_out/measured_tolerances.json).   usage: tools/prof_round6_final.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 400 gpurun_out/r06_bench_default.json; echo
timeout 600 bash tools/prof_bench.sh r06_bench --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 | tail -24
timeout 900 bash tools/pmc_bench.sh r06 | tail -3
timeout 300 python3 bench.py --steps 10 --warmup 2 --no-ce --no-cpu-baseline --no-host-path --no-strict-fp32 --no-configs4 > gpurun_out/r06_bench_two_outputs.json 2>/dev/null
timeout 600 python3 bench.py --sweep > gpurun_out/r06_bench_sweep_config3.json 2>/dev/null
timeout 600 python3 tools/bench_cfg3.py 1024 > gpurun_out/r06_cfg3_shapes.txt 2>&1; tail -6 gpurun_out/r06_cfg3_shapes.txt
timeout 2400 python3 -m pytest tests -m gpu -q --timeout 1500 --durations=12 > gpurun_out/r06_gpu_tests.txt 2>&1; tail -22 gpurun_out/r06_gpu_tests.txt
python3 - <<'PY'
import json
for n in ("r06_bench_default", "r06_bench_two_outputs", "r06_bench_sweep_config3"):
    try:
        j = json.load(open("gpurun_out/%s.json" % n))
        print(n, j["value"], j["ms_per_step"], (j.get("roofline") or {}).get("avg_launch_ms"), (j.get("end_to_end") or {}).get("value"), (j.get("host_path") or {}).get("value"))
    except Exception as e:
        print(n, "ERR", e)
PY
