#!/bin/bash
# round 5: sigma_z^2 of a cell's winner carried through LDS instead of a scattered cold-record fetch: parity, time, traffic
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_pipeline_gpu.py tests/test_tiling_gpu.py -q -x 2>&1 | tail -2
python scripts/c4_ab.py "" 2>/dev/null | tail -1
bash scripts/pmc_traffic_run.sh c4v gpurun_out/pmc_c4v.json --workload c4 > /dev/null 2>&1
python3 scripts/pmc_traffic.py c4v gpurun_out/pmct_c4v gpurun_out/pmc_c4v.json 2>/dev/null | tail -3
python3 -c "
import json
d=json.load(open('gpurun_out/pmc_c4v.json'))
for w,e in d.items():
    for k,v in e.items(): print(w,k,round(v['hbm_bytes_per_launch']/1e6,1),'MB')
"
