#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_phases5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_tbatch_gpu.py tests/test_batch_gpu.py -m gpu -q 2>&1 | tail -12
timeout 300 python scripts/phases_tiled.py c4 tbatch_max=4 batch_fuse=0 > $O/phases_k4_nofuse.json 2>> $O/err.txt || tail -3 $O/err.txt
timeout 300 python scripts/phases_tiled.py c4 tbatch_max=4 > $O/phases_k4.json 2>> $O/err.txt || tail -3 $O/err.txt
timeout 300 python scripts/phases_tiled.py c4 tbatch_max=8 > $O/phases_k8.json 2>> $O/err.txt || tail -3 $O/err.txt
timeout 600 python scripts/c4_ab.py "tbatch=0" "" "tbatch_max=4,batch_fuse=0" "tbatch_max=8" "tbatch_max=8,tb_groups=384" "tbatch_max=8,tb_groups=768" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json
python - <<'PY'
import json
for f in ("phases_k4_nofuse","phases_k4","phases_k8"):
    d=json.load(open(f"gpurun_out/r04_phases5/{f}.json"))
    print("==",f,"span",d["span_us"],"blocks",d["blocks"],"upd",d["update_groups"])
    print(" upd", {a:d["update_all"].get(a) for a in ("n","dur","last_end")})
    print(" bin", d["bin"])
    for k,v in d.get("bin_by_start",{}).items(): print("  ",k,{a:v[a] for a in v if a in("n","init_candidate","flush","dur")})
PY
