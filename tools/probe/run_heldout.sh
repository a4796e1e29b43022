cd $GRAFT_REPO_ROOT
python3 tools/bench_cfg1_omp.py 2>&1 | grep -v amdgpu
python3 tools/parity_fixture_check.py --fixture fullsize_port_heldout --group sweep_proposed --out gpurun_out/r05_heldout_proposed.json "" 2>&1 | grep -v amdgpu.ids
python3 tools/parity_fixture_check.py --fixture fullsize_port_heldout --group sweep_proposed --two-output --out gpurun_out/r05_heldout_proposed_two_output.json "" 2>&1 | grep -v amdgpu.ids
python3 tools/parity_fixture_check.py --fixture fullsize_port_heldout --group sweep_angles --out gpurun_out/r05_heldout_angles.json "" 2>&1 | grep -v amdgpu.ids
python3 tools/parity_fixture_check.py --fixture fullsize_port_heldout --group sweep_angles --two-output --out gpurun_out/r05_heldout_angles_two_output.json "" 2>&1 | grep -v amdgpu.ids
python3 tools/parity_fixture_check.py --group sweep_angles --out gpurun_out/r05_setA_angles.json "" 2>&1 | grep -v amdgpu.ids
python3 tools/parity_fixture_check.py --group bench_proposed --out gpurun_out/r05_setA_bench.json "" 2>&1 | grep -v amdgpu.ids
