#!/bin/bash
# round 5: where the sector-window walk's time goes (measurement switches dbg_ray 4096 / 8192 / 16384: results are wrong)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for D in 0 4096 8192 12288 16384 64; do
  echo "dbg_ray $D: $(timeout 600 python3 scripts/ray_bench.py c4 --cpu-iters 1 --steps 10 --set dbg_ray=$D 2>/dev/null | tail -1 | cut -c1-100)"
done
