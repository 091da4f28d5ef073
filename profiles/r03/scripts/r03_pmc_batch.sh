#!/bin/bash
# (every pass under its own `timeout`: a counter set the hardware refuses makes rocprofv3 abort and then hang)
# PMC passes (own runs, --pmc only) of the default bench command: counters of the batch launch k_mbatch
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03_pmc_batch
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --no-host-legs --no-cpu-baseline --no-large --steps 640 --warmup 64 --profile-steps 4 "$@" > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
cd $R
python3 - $O <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_mbatch" in n:
            agg[n.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
