#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_ab3
mkdir -p $O
cd $R
timeout 300 python scripts/phases_tiled.py c4 > $O/phases_single.json 2>> $O/err.txt || tail -3 $O/err.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_ab3/phases_single.json"))
print("span",d["span_us"]); print(" upd", d["update_all"]); print(" bin", d["bin"]); print(" bin_first", d["bin_first_round"]); print(" bin_late", d["bin_late"])
PY
