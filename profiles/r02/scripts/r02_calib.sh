#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-byte kernels (separate --pmc passes, no tracing)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02/calib
mkdir -p $O
$R/scripts/ubench/pmc_calib.bin > $O/known.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -o p -- $R/scripts/ubench/pmc_calib.bin > $O/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p2 -o p -- $R/scripts/ubench/pmc_calib.bin > $O/p2.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/p3 -o p -- $R/scripts/ubench/pmc_calib.bin > $O/p3.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum --output-format csv -d $O/p4 -o p -- $R/scripts/ubench/pmc_calib.bin > $O/p4.log 2>&1
cd $R
python3 scripts/pmc_calib.py $O $O/known.json $O/pmc_calibration.json
python3 - $O <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/p[34]/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]] = float(row["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k, {a: int(b) for a, b in sorted(d.items())})
PY
