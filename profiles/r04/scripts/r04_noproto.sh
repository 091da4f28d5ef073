#!/bin/bash
# k_mbatch without any in-launch wait (scouts always one launch ahead): batch tests, c2/c3 lines, a short soak of each profile
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_noproto
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_tbatch_gpu.py -m gpu -q -x 2>&1 | tail -4
for W in c2 c3; do
timeout 600 python bench.py --workload $W --no-cpu-baseline --no-host-legs --no-large > $O/bench_$W.json 2>$O/err_$W.txt || tail -3 $O/err_$W.txt
python - $W <<'PY'
import json,sys
w=sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/r04_noproto/bench_{w}.json') if l.startswith('{')][-1]); print(w, 'value', d['value'], 'us/scan', d['timed_region_us_per_scan_hip_events'], 'frac', d['roofline']['frac'], 'cache_resident', d.get('cache_resident',{}).get('us_per_scan_hip_events'), 'latency', d['latency_path']['us_per_scan'])
PY
done
for P in small p2 tbatch tiled; do
timeout 400 python scripts/soak_r04.py ${SOAK_S:-60} 5 no $P 2>&1 | tail -3 | tee -a $O/soak.jsonl
done
