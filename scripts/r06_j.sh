#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_batch_gpu.py tests/test_batch_ray_gpu.py -x -q -m gpu 2>&1 | tail -4
for V in "batch_max=16" "batch_max=32" "batch_max=24"; do
  T=$(echo $V | tr ' =' '__')
  timeout 200 python3 scripts/timeline_batch.py $V > $O/timeline_c2_$T.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/timeline_c2_$T.json')); print('$V', d['grid'], d['span_us'], 'upd', d['update']['end'], 'bin', d['bin']['end'], 'dur', d['bin']['dur'])"
done
for V in "batch_max=16" "batch_max=32" "batch_max=32,batch_walk=1" "batch_max=24"; do
  A=""; for kv in $(echo $V | tr ',' ' '); do A="$A --set $kv"; done
  timeout 300 python bench.py --no-cpu-baseline --no-host-legs --no-large $A > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('$V', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
done
for BM in 16 32; do
  timeout 300 python bench.py --workload c3 --steps 2000 --warmup 200 --no-cpu-baseline --no-host-legs --no-large --set batch_max=$BM > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('c3', $BM, round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
done
