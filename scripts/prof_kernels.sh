#!/bin/bash
# rocprofv3 kernel-trace of the two scan kernels for the given workloads (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -o $W -- python3 $R/scripts/ab_kernels.py $W --order azimuth --rounds 1 --steps 40 > $R/gpurun_out/prof_$W.log 2>&1
done
cd $R
python3 - "$@" <<'PY'
import csv, sys
for w in sys.argv[1:]:
    print(w)
    for r in csv.DictReader(open(f"gpurun_out/prof_{w}/{w}_kernel_stats.csv")):
        if "k_bin" in r["Name"] or "k_update" in r["Name"]:
            print("  %-30s calls=%4s avg=%9.1f us  min=%8.1f max=%8.1f" % (r["Name"].split("(")[0][-30:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
