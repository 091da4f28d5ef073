#!/bin/bash
# round 5: SQ counters of k_mbatch at configs[2] (what bounds a launch of 5.5 rounds of bin blocks)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/sq_c3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --workload c3 --no-host-legs --no-cpu-baseline --no-large --steps 64 --warmup 16 --profile-steps 4 > $O/p$i.log 2>&1 || tail -2 $O/p$i.log
done
cd $R
python3 - $O <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "fdm::k_mbatch<" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("sq_c3 k_mbatch", {c: round(sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
