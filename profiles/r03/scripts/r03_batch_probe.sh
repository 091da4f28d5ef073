#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03_batch_probe
mkdir -p $O
cd $R
python scripts/batch_probe.py "$@" | tee $O/probe.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -o p -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 640 --warmup 64 --profile-steps 4 > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $O/pmc2 -o p -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 640 --warmup 64 --profile-steps 4 > $O/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc3 -o p -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 640 --warmup 64 --profile-steps 4 > $O/pmc3.log 2>&1
cd $R
python3 - $O <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_m" in n:
            agg[n.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
