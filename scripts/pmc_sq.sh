#!/bin/bash
# SQ / LDS counters of the scan kernels (run on the GPU box; --pmc only, no tracing).
# usage: pmc_sq.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/pmc_sq_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --overlap 0 --no-large --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 5 "$@" > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
cd $R
python3 - $O <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        k = None
        for key in ("k_tbin", "k_tupdate", "k_bin4", "k_update", "k_bin"):
            if "fdm::" + key in n:
                k = key
                break
        if k: agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
