#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_third
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_tbatch_gpu.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python scripts/timeline.py c4 --set tbatch_max=4 > $O/timeline_k4.json 2> $O/timeline.err || tail -3 $O/timeline.err
timeout 600 python scripts/c4_ab.py "tbatch=0" "" "tbatch_max=8" "tb_groups=384" "tb_groups=640" > $O/c4_ab.json 2> $O/c4_ab.err
cat $O/c4_ab.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_third/timeline_*.json')):
    d=json.load(open(f))
    print(f, {k:d[k] for k in d if k not in('resident_by_us',)})
    print([ (r['t'],r['update'],r['bin']) for r in d['resident_by_us'][::10]])
PY
