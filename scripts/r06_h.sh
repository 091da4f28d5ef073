#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_batch_gpu.py tests/test_batch_ray_gpu.py tests/test_long_horizon_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_h.txt
cat $O/pytest_h.txt
for BM in 16 32; do
  timeout 300 python bench.py --no-cpu-baseline --no-host-legs --no-large --set batch_max=$BM > $O/bench_bm$BM.json 2>$O/bench_bm$BM.err || tail -3 $O/bench_bm$BM.err
  python3 -c "
import json; d=json.load(open('$O/bench_bm$BM.json')); print($BM, round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['config'].get('scans_per_launch'), d.get('raycasting_on'))"
done
for BM in 16 32; do
  timeout 300 python bench.py --workload c3 --steps 2000 --warmup 200 --no-cpu-baseline --no-host-legs --no-large --set batch_max=$BM > $O/bench_c3_bm$BM.json 2>$O/bench_c3_bm$BM.err || tail -3 $O/bench_c3_bm$BM.err
  python3 -c "
import json; d=json.load(open('$O/bench_c3_bm$BM.json')); print('c3', $BM, round(d['value']), d['ms_per_step'], d['roofline']['frac'])"
done
