#!/bin/bash
# round 5: the evidence the first pass does not hold — the driver-style line three times, the raycasting stage's kernel
# trace and times per config, the stencil stages
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r05x; mkdir -p $O
timeout 900 python -m pytest tests/test_raycast_gpu.py tests/test_batch_ray_gpu.py -x -q 2>&1 | tail -2
for k in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-large --no-cpu-baseline --no-host-legs 2>/dev/null | tail -1 > $O/bench_steps20_$k.json; python3 -c "import json; d=json.load(open('$O/bench_steps20_$k.json')); print('steps20', d['value'], d['ms_per_step'], d.get('repeats'), d.get('ms_per_step_min'), d.get('ms_per_step_max'))"; done
rm -f $O/ray_bench.jsonl
for W in c2 c3 c4; do timeout 600 python3 scripts/ray_bench.py $W --cpu-iters 2 2>/dev/null | tail -1 >> $O/ray_bench.jsonl; done
cut -c1-330 $O/ray_bench.jsonl
bash scripts/prof_ray.sh c3 c4 2>&1 | tail -34
cp gpurun_out/prof_ray_c3/c3_kernel_stats.csv $O/rocprof_ray_c3_kernel_stats.csv
cp gpurun_out/prof_ray_c4/c4_kernel_stats.csv $O/rocprof_ray_c4_kernel_stats.csv
rm -f $O/stage_bench.jsonl
for W in c2 c4; do timeout 900 python scripts/stage_bench.py $W --cpu-iters 1 2>/dev/null | grep '^{' >> $O/stage_bench.jsonl; done
python3 -c "
import json
for l in open('$O/stage_bench.jsonl'):
    d=json.loads(l); print(d['workload'], d['stage'][:70], d.get('gpu_ms'))
"
timeout 600 python3 bench.py 2>/dev/null | tail -1 > $O/bench_default.json
python3 -c "import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['roofline']['frac'], d['large']['roofline']['frac'], d['large']['raycasting_on'], d['raycasting_on']['us_per_scan_hip_events'], d.get('host_points4_pool_ms_per_scan'))"
timeout 300 python3 bench.py --workload c3 --steps 2000 --warmup 200 --no-large --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_c3.json
python3 -c "import json; d=json.load(open('$O/bench_c3.json')); print('c3', d['value'], d['roofline']['frac'], d['raycasting_on'])"
