#!/bin/bash
# where do k_mbatch's LDS bank conflicts come from — the bin half's cell table or the update half's event
# exchange?  With batch_fuse 0 the two halves leave as separate k_mbatch launches ([bin | crop] and [update]), told apart
# by their grid size in the counter CSV.  (--pmc only, every pass under its own timeout.)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_pmc_mbatch
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES"; do
  for FUSE in 0 1; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --scans 4 --no-host-legs --no-cpu-baseline --no-large --steps 200 --warmup 16 --profile-steps 4 --set batch_fuse=$FUSE > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
  done
done
cd $R
python3 - $O <<'PY' | tee $O/summary.txt
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    tag = f.split("/")[-2]
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "k_mbatch" in n:
            agg[(tag, row.get("Grid_Size", "?"))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    m = {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}
    n = len(next(iter(d.values())))
    conf = m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1))
    print(k, "launches", n, m, "conflict_share_of_lds_active %.2f" % conf)
PY
