#!/bin/bash
# kernel durations (rocprofv3 --stats) of the large-scan kernels, separate launches and fused; phase stamps
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for TAG in sep fused; do
  ARGS=""; [ $TAG = sep ] && ARGS="--overlap 0"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$TAG -o t -- python3 $R/bench.py --no-large --no-cpu-baseline --no-host-legs --steps 100 --warmup 10 --profile-steps 5 --workload c4 $ARGS > $O/kt_$TAG.log 2>&1 || tail -3 $O/kt_$TAG.log
  python3 - $O/kt_$TAG $TAG <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Name"]
        if "fdm::k_t" in n:
            print(sys.argv[2], n.split("(")[0][:60], row["Calls"], round(float(row["AverageNs"]) / 1e3, 2), "us min", round(float(row["MinNs"]) / 1e3, 2))
PY
done
cd $R
timeout 300 python3 scripts/phases_tiled.py c4 > $O/phases_c4.json 2>$O/phases_c4.err || tail -3 $O/phases_c4.err
cat $O/phases_c4.json
