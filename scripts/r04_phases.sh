#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_phases
mkdir -p $O
cd $R
timeout 300 python scripts/phases_tiled.py c4 tbatch_max=4 batch_fuse=0 > $O/phases_k4_nofuse.json 2>> $O/err.txt || tail -3 $O/err.txt
timeout 300 python scripts/phases_tiled.py c4 tbatch_max=4 tb_groups=2000 > $O/phases_k4_g1444.json 2>> $O/err.txt || tail -3 $O/err.txt
timeout 300 python scripts/phases_tiled.py c4 tbatch=0 > $O/phases_single.json 2>> $O/err.txt || tail -3 $O/err.txt
timeout 600 python scripts/c4_ab.py "tbatch=0" "tbatch_max=4,batch_fuse=0" "tb_groups=2000" "tbatch_max=8,tb_groups=2000"> $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json
python - <<'PY'
import json
for f in ("phases_k4_nofuse","phases_k4_g1444","phases_single"):
    d=json.load(open(f"gpurun_out/r04_phases/{f}.json"))
    print("==",f,"span",d["span_us"],"blocks",d["blocks"],"upd",d["update_groups"])
    print(" upd", d["update_all"])
    print(" bin", d["bin"])
    for k,v in d.get("bin_by_start",{}).items(): print("  ",k,v)
PY
