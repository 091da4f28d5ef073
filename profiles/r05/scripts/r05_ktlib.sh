#!/bin/bash
# round 5: rocprofv3 kernel durations of the large-scan kernels (separate launches) for several engine BUILDS
R=${GRAFT_REPO_ROOT:-$PWD}
for N in "$@"; do
  L=$R/fastdem_amd/lib/libfdm_engine${N:+_$N}.so
  FDM_ENGINE_LIB=$L bash $R/scripts/r05_kt.sh lib_${N:-shipped} "--overlap 0" | sed "s/^/${N:-shipped} /"
done
