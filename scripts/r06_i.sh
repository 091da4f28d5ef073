#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
for V in "batch_max=16" "batch_max=32" "batch_max=32 batch_walk=1" "batch_max=24"; do
  T=$(echo $V | tr ' =' '__')
  timeout 200 python3 scripts/timeline_batch.py $V > $O/timeline_c2_$T.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/timeline_c2_$T.json')); print('$V', d['grid'], d['span_us'], 'upd', d['update']['end'], 'bin', d['bin']['end'], 'dur', d['bin']['dur']); print('   ', [(b['k'], b['start50'], b['end50']) for b in d['bin_by_scan']][::3])"
done
for V in "batch_max=32,batch_walk=1" "batch_max=24" "batch_max=20"; do
  A=""; for kv in $(echo $V | tr ',' ' '); do A="$A --set $kv"; done
  timeout 300 python bench.py --no-cpu-baseline --no-host-legs --no-large $A > $O/b.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$O/b.json')); print('$V', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
done
