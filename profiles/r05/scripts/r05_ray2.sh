#!/bin/bash
# round 5: quick look at the raycasting stage after a kernel change: parity of the sector-window tests, stage times,
# kernel trace.  usage: r05_ray2.sh [ray_bench --set args...]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_raycast_gpu.py -x -q -k "Sector or c4 or populated or wrapped" 2>&1 | tail -3
for W in c3 c4; do timeout 600 python3 scripts/ray_bench.py $W --cpu-iters 1 "$@" 2>/dev/null | tail -1 | cut -c1-260; done
bash scripts/prof_ray.sh c3 c4 2>&1 | grep -E "^c[34]|k_ray|k_rs_scatter|compact|resolve"
