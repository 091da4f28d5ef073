#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for D in 0 16384 49152 8192; do
echo "dbg $D"
RB_OPTS="dbg_ray=$D" bash scripts/r04_rayprof.sh 2>&1 | grep -E "us_per_scan|k_rb_ray_lds" | grep -v '^"'
done
