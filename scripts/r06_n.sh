#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for L in "" "_mhash" "" "_mhash"; do
  timeout 300 python scripts/lib_bench.py $R/fastdem_amd/lib/libfdm_engine$L.so --no-cpu-baseline --no-host-legs --no-large > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('c2 lib$L', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
done
for L in "" "_mhash"; do
  timeout 300 python scripts/lib_bench.py $R/fastdem_amd/lib/libfdm_engine$L.so --workload c3 --steps 2000 --warmup 200 --no-cpu-baseline --no-host-legs --no-large > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('c3 lib$L', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
done
