#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for N in "" r8 c32 r2 c8; do
  L=$R/fastdem_amd/lib/libfdm_engine${N:+_$N}.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rw_${N:-base} -o c -- python3 $R/scripts/lib_ray.py $L c4 --steps 10 --cpu-iters 1 > /dev/null 2>&1
  python3 - $R/gpurun_out/rw_${N:-base}/c_kernel_stats.csv "${N:-base}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ray_wedge" in r["Name"]:
        print(sys.argv[2], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
