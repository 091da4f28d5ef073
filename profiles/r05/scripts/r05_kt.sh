#!/bin/bash
# round 5: kernel durations of the large-scan kernels, separate launches (--overlap 0) and fused, by rocprofv3 --stats
# usage: r05_kt.sh <tag> "<bench args>" ...
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  TAG=$1; ARGS=$2; shift 2
  O=$R/gpurun_out/kt_$TAG; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/bench.py --no-large --no-cpu-baseline --no-host-legs --steps 100 --warmup 10 --profile-steps 5 --workload c4 $ARGS > $O/log.txt 2>&1 || tail -3 $O/log.txt
  python3 - $O $TAG <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Name"]
        if "fdm::k_t" in n:
            print(sys.argv[2], n.split("(")[0][:60], row["Calls"], round(float(row["AverageNs"]) / 1e3, 2), "us min", round(float(row["MinNs"]) / 1e3, 2))
PY
done
