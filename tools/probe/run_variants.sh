cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity_proposed.py tests/test_gpu_hgemm.py tests/test_gpu_toeplitz_dictionary.py tests/test_gpu_baselines.py -x -q 2>&1 | tail -5
python3 tools/parity_fixture_check.py --out gpurun_out/r05c_variants.json "" "JSTSP_PASS_ACC=1" "JSTSP_PASS_ACC=1,JSTSP_INV2=0" "JSTSP_PASS_ACC=1,JSTSP_H2=0" 2>&1 | grep -v amdgpu.ids | cut -c1-330
for acc in 0 1; do echo "PASS_ACC=$acc"; JSTSP_PASS_ACC=$acc timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-path --no-strict-fp32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['parity']['whole_batch']['max_abs_dNMSE'], d['parity']['whole_batch']['rms_dNMSE'])"; done
