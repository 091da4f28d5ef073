#!/bin/bash
# configs[3]: tiles per update group (fewer, longer-running update blocks leave slots to the bin half)
for SP in 0 2 3 4; do
  python bench.py --workload c4 --steps 1000 --warmup 100 --no-cpu-baseline --no-large --no-host-legs --profile-steps 30 --set dbg_span=$SP 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('span $SP', round(d['ms_per_step']*1e3,2), {a:round(v['ms']*1e3,2) for a,v in k.items() if isinstance(v,dict) and 'ms' in v})"
done
