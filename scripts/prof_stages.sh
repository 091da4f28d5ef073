#!/bin/bash
# rocprofv3 kernel-trace of the egress / ingest stages (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stage_$W -o $W -- python3 $R/scripts/stage_bench.py $W --iters 20 --cpu-iters 1 > $R/gpurun_out/prof_stage_$W.log 2>&1
  grep '^{' $R/gpurun_out/prof_stage_$W.log
done
cd $R
python3 - "$@" <<'PY'
import csv, sys
for w in sys.argv[1:]:
    print(w)
    for r in csv.DictReader(open(f"gpurun_out/prof_stage_{w}/{w}_kernel_stats.csv")):
        if any(k in r["Name"] for k in ("k_pack", "k_ingest", "k_inpaint", "k_median", "k_fusion", "k_features")):
            print("  %-40s calls=%5s avg=%9.1f us" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3))
PY
