#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_ph
mkdir -p $O
cd $R
timeout 600 python scripts/c4_ab.py "upd_prio=0" "upd_prio=1" "upd_prio=1,upd_blocks=1444" "upd_prio=1,upd_blocks=256" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json; tail -2 $O/c4_ab.err
for V in "upd_prio=0" "upd_prio=1"; do
timeout 300 python scripts/phases_tiled.py c4 $V > $O/ph_$V.json 2> $O/ph.err || tail -3 $O/ph.err
python3 - $O/ph_$V.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1], d["span_us"]); print("bin", d["bin"]); print("bin_first", d["bin_first_round"]); print("bin_late", d["bin_late"]); print("upd", d["update_all"]["dur"], d["update_heavy"].get("dur"))
PY
done
