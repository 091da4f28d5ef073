#!/bin/bash
# SQ counters of the default-radius stencil kernels (scripts/post_ab.py on the configs[3] map): separate --pmc passes, no tracing
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06ev
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/sq_post/p$i -o p -- python3 $R/scripts/post_ab.py c4 --iters 3 > $O/sq_post.p$i.log 2>&1 || tail -2 $O/sq_post.p$i.log
done
cd $R
python3 - $O <<'PY' > $O/pmc_sq_post.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/sq_post/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        for key in ("k_fusion_f64_tiled<29, true>", "k_fusion_net32_tiled", "k_features_tiled<6, 7, true>", "k_features_tiled<8, 8, false>"):
            if key in name:
                agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                break
for k, d in agg.items():
    print("sq_post", k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
cut -c1-700 $O/pmc_sq_post.txt
