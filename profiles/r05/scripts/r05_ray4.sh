#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_raycast_gpu.py -x -q 2>&1 | tail -3
for D in 0 4096 8192 16384; do
  echo "dbg_ray $D: $(timeout 600 python3 scripts/ray_bench.py c4 --cpu-iters 1 --steps 10 --set dbg_ray=$D 2>/dev/null | tail -1 | cut -c1-100)"
done
timeout 600 python3 scripts/ray_bench.py c3 --cpu-iters 1 2>/dev/null | tail -1 | cut -c1-260
bash scripts/prof_ray.sh c3 c4 2>&1 | grep -E "^c[34]|k_ray|k_rs_scatter|compact|resolve"
