#!/bin/bash
# usage: r05_libab2.sh "<variants of c4_ab>" name1 name2 ...
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
V=$1; shift
for N in "$@"; do
  L=$R/fastdem_amd/lib/libfdm_engine${N:+_$N}.so
  echo "== ${N:-shipped}"
  timeout 600 python scripts/c4_ab.py --lib=$L $V 2>/dev/null | tail -1
done
