#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in "" "JSTSP_FUSED_PARTS=2" "JSTSP_FUSED_PARTS=8"; do env $v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-strict-fp32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['roofline']['avg_launch_ms'], d['parity']['whole_batch']['max_abs_dNMSE'], d['parity']['whole_batch']['rms_dNMSE'])"; done
python3 tools/parity_fixture_check.py --group sweep_proposed --n 640 "" "JSTSP_FUSED_PARTS=2" 2>&1 | grep -v amdgpu | cut -c1-260
