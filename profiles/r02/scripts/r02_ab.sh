#!/bin/bash
# configs[3] A/B of one engine option: r02_ab.sh <option> <values...>
OPT=$1; shift
for V in "$@"; do
  python bench.py --workload c4 --steps 1000 --warmup 100 --no-cpu-baseline --no-large --no-host-legs --profile-steps 30 --set $OPT=$V 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$OPT $V', round(d['ms_per_step']*1e3,2), {a:round(v['ms']*1e3,2) for a,v in k.items() if isinstance(v,dict) and 'ms' in v})"
done
