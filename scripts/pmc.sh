#!/bin/bash
# PMC passes (each its own rocprofv3 run, --pmc only — never combined with tracing).
# usage: scripts/pmc.sh <tag> <python args...>   (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
SETS=("TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
      "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
      "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
      "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64")
if [ -n "$PMC_SETS" ]; then IFS=',' read -ra PICK <<< "$PMC_SETS"; else PICK=(0 1 2 3 4 5); fi
for si in "${PICK[@]}"; do
  SET=${SETS[$si]}
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -o p -- python3 $R/"$@" > $R/gpurun_out/pmc_$TAG.p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_$TAG/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:40]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    if not k.startswith(("void fdm::k_bin", "fdm::k_update")): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
PY
