#!/bin/bash
# round 5: the sector-window ray walk (fdm_raywedge.hpp): parity tests, stage times, kernel trace of the stage
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r05_ray; mkdir -p $O
timeout 1500 python -m pytest tests/test_raycast_gpu.py -x -q 2>&1 | tail -8
for W in c3 c4; do timeout 600 python3 scripts/ray_bench.py $W --cpu-iters 1 2>/dev/null | tail -1 | cut -c1-400; done
for W in c3 c4; do timeout 600 python3 scripts/ray_bench.py $W --cpu-iters 1 --set ray_wedge=0 2>/dev/null | tail -1 | cut -c1-200; done
bash scripts/prof_ray.sh c3 c4 2>&1 | tail -32
timeout 300 python3 bench.py --workload c5 --routed 0 --steps 300 --warmup 50 --no-large --no-cpu-baseline --no-host-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 plain us/step', d['ms_per_step']*1e3, 'frac', d['roofline']['frac'])"
timeout 300 python3 bench.py --workload c4 --steps 300 --warmup 50 --no-large --no-cpu-baseline --no-host-legs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 us/step', d['ms_per_step']*1e3, 'frac', d['roofline']['frac'])"
