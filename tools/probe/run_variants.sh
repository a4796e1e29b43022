cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_baselines.py -x -q -k omp 2>&1 | tail -3
python3 tools/bench_cfg1_omp.py 2>&1 | grep -v amdgpu
