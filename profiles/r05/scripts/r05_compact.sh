#!/bin/bash
# round 5: what the ray queue builder (k_ray_compact) costs at configs[3]: measurement switches dbg_ray 128 (no evidence
# atomics); results are wrong with them
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for D in 0 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cmp_$D -o c -- python3 $R/scripts/ray_bench.py c4 --steps 10 --cpu-iters 1 --set dbg_ray=$D > /dev/null 2>&1
  python3 - $R/gpurun_out/cmp_$D/c_kernel_stats.csv $D <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ray_compact" in r["Name"] or "k_ray_wedge" in r["Name"]:
        print("dbg", sys.argv[2], r["Name"].split("(")[0][-30:], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
