#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_batch_gpu.py tests/test_batch_ray_gpu.py tests/test_long_horizon_gpu.py tests/test_perf_guard_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-host-legs --no-large > $O/b.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/b.json')); print('c2 auto', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['config']['scans_per_launch'], d.get('raycasting_on'))"
timeout 300 python bench.py --workload c3 --steps 2000 --warmup 200 --no-cpu-baseline --no-host-legs --no-large > $O/b.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/b.json')); print('c3 auto', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['config']['scans_per_launch'], d.get('raycasting_on'))"
