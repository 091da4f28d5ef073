#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for D in 0 256 512 768; do
echo "dbg $D"
RB_OPTS="dbg_ray=$D" bash scripts/r04_rayprof.sh 2>&1 | grep -E "k_rb_count|k_rb_scatter|k_rb_mark" | grep -v '^"'
done
