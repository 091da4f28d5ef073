#!/bin/bash
# round 5: configs[1] batch launch (k_mbatch) for several engine builds: bench.py default line, value / ms_per_step
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for N in "$@"; do
  L=$R/fastdem_amd/lib/libfdm_engine${N:+_$N}.so
  echo "== ${N:-shipped}"
  python3 - $L <<'PY'
import json, subprocess, sys, os
from fastdem_amd import capi
PY
  timeout 600 python scripts/lib_bench.py $L --no-large --no-cpu-baseline --no-host-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','repeats')}, d['roofline']['frac'], d['roofline']['avg_kernel_us'])"
done
