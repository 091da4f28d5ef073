#!/bin/bash
# SQ counters of selected kernels for an arbitrary python script (run on the GPU box; --pmc only, no tracing).
# usage: pmc_cmd.sh <tag> <kernel-substr[,substr...]> <script.py> [args...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; KER=$2; SCRIPT=$3; shift 3
O=$R/gpurun_out/pmc_cmd_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/$SCRIPT "$@" > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
cd $R
python3 - $O $KER <<'PY'
import collections, csv, glob, sys
keys = sys.argv[2].split(",")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        for key in keys:
            if key in n:
                agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                break
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
