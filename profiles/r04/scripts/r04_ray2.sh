#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ray
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_batch_ray_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_batch_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2; do
for L in base cur; do
  if [ $L = base ]; then export FDM_ENGINE_LIB=$R/fastdem_amd/lib/libfdm_engine_base.so; else unset FDM_ENGINE_LIB; fi
  timeout 600 python bench.py --workload c2 --no-cpu-baseline --no-host-legs --no-large > $O/bench_c2_$L.json 2>$O/err_c2.txt || tail -3 $O/err_c2.txt
  python - $L <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r04_ray/bench_c2_{sys.argv[1]}.json') if l.startswith('{')][-1]); print(sys.argv[1], 'c2 value', d['value'], 'us/scan', d['timed_region_us_per_scan_hip_events'], 'frac', d['roofline']['frac'])
PY
done
done
