#!/bin/bash
# rocprofv3 kernel-trace of integrate()+raycasting for the given workloads (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ray_$W -o $W -- python3 $R/scripts/ray_bench.py $W --steps 20 --cpu-iters 1 > $R/gpurun_out/prof_ray_$W.log 2>&1
done
cd $R
python3 - "$@" <<'PY'
import csv, sys
for w in sys.argv[1:]:
    print(w)
    rows = list(csv.DictReader(open(f"gpurun_out/prof_ray_{w}/{w}_kernel_stats.csv")))
    for r in rows[:14]:
        print("  %-60s calls=%5s avg=%9.1f us  total=%9.1f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
