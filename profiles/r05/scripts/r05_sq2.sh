#!/bin/bash
# round 5: VALU / SALU / LDS instruction counts of the large-scan kernels per variant (two --pmc passes each)
# usage: r05_sq2.sh <tag> "<bench args>" [<tag2> "<...>"] ...
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  TAG=$1; ARGS=$2; shift 2
  O=$R/gpurun_out/sq2_$TAG; mkdir -p $O
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64"; do
    i=$((i+1))
    rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --overlap 0 --no-large --no-cpu-baseline --no-host-legs --steps 20 --warmup 5 --profile-steps 5 --workload c4 $ARGS > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
  done
  python3 - $O $TAG <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        for key in ("k_tupdate_tbin", "k_tbin", "k_tupdate"):
            if "fdm::" + key + "<" in n:
                agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                break
for k, d in agg.items():
    w = max(1.0, sum(d.get("SQ_WAVES", [1])) / max(1, len(d.get("SQ_WAVES", [1]))))
    print(sys.argv[2], k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())},
          "per wave:", {c: round(sum(v) / len(v) / w, 1) for c, v in sorted(d.items()) if c.startswith("SQ_INSTS")})
PY
done
