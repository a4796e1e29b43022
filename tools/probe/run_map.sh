#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 tools/parity_fixture_check.py --group sweep_proposed "" "JSTSP_RV_COMP=1,JSTSP_RV_REFRESH=1000" "JSTSP_RV_COMP=1" "JSTSP_RV_COMP=1,JSTSP_RV_REFRESH=8" 2>&1 | grep -v amdgpu | cut -c1-330
python3 tools/parity_fixture_check.py --group bench_proposed "" "JSTSP_RV_COMP=1,JSTSP_RV_REFRESH=1000" 2>&1 | grep -v amdgpu | cut -c1-330
for v in "" "JSTSP_RV_COMP=1 JSTSP_RV_REFRESH=1000"; do env $v python3 bench.py --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['parity'].get('whole_batch'))"; done
