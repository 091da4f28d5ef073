#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_waves
mkdir -p $O
cd $R
for rep in 1 2; do
for L in cur w5 w7; do
  if [ $L = cur ]; then unset FDM_ENGINE_LIB; else export FDM_ENGINE_LIB=$R/fastdem_amd/lib/libfdm_engine_$L.so; fi
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-host-legs --no-large > $O/bench_c2_$L.json 2>$O/err.txt || tail -3 $O/err.txt
  python - $L <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r04_waves/bench_c2_{sys.argv[1]}.json') if l.startswith('{')][-1]); print(sys.argv[1], 'us/scan', d['timed_region_us_per_scan_hip_events'], 'frac', d['roofline']['frac'], 'cache_resident', d.get('cache_resident',{}).get('us_per_scan_hip_events'))
PY
done
done
