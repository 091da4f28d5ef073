#!/bin/bash
# round 5: A/B of engine BUILDS (fastdem_amd/lib/libfdm_engine_<name>.so, `make variant`) at configs[3]:
# fused launch per scan, and the two halves as separate launches (overlap 0: bin + update per scan)
# usage: r05_libab.sh name1 name2 ...   ("" = the shipped library)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for N in "$@"; do
  L=$R/fastdem_amd/lib/libfdm_engine${N:+_$N}.so
  echo "== ${N:-shipped}"
  timeout 400 python scripts/c4_ab.py --lib=$L "" "overlap=0" 2>/dev/null | tail -1
done
