#!/bin/bash
# configs[3] fused launch vs the occupancy the tile kernels are compiled for (rebuilds the library on the box)
for W in 7 8 6; do
  make -s -C fastdem_amd/csrc clean >/dev/null; make -s -C fastdem_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DFDM_UPD_WAVES=$W" 2>&1 | grep -E "error" | head -3
  python bench.py --workload c4 --steps 1000 --warmup 100 --no-cpu-baseline --no-large --no-host-legs --profile-steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('waves $W', round(d['ms_per_step']*1e3,2), {a:round(v['ms']*1e3,2) for a,v in k.items() if isinstance(v,dict) and 'ms' in v})"
done
