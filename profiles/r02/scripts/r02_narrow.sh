#!/bin/bash
# configs[3]: light tiles walked by two wavefronts (option narrow_max = chunk-count threshold)
for NM in "$@"; do
  python bench.py --workload c4 --steps 1000 --warmup 100 --no-cpu-baseline --no-large --no-host-legs --profile-steps 30 --set narrow_max=$NM 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('narrow_max $NM', round(d['ms_per_step']*1e3,2), {a:round(v['ms']*1e3,2) for a,v in k.items() if isinstance(v,dict) and 'ms' in v})"
done
