#!/bin/bash
# LDS / instruction counters of the bin kernel for a workload and point order (run on the GPU box).
# usage: pmc_lds.sh c4 ring
R=${GRAFT_REPO_ROOT:-$PWD}
W=$1; ORDER=$2
O=$R/gpurun_out/pmc_lds_${W}_${ORDER}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --workload $W --order $ORDER --overlap 0 --no-large --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 5 > $O/p$i.log 2>&1
done
cd $R
python3 - $O <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        k = "k_bin" if "fdm::k_bin" in n else ("k_update" if "fdm::k_update" in n else None)
        if k: agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
