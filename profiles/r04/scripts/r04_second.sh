#!/bin/bash
# round 4: parity of both batch pipelines (cell ids off: the calls really leave in batch launches), then the block
# timeline of the tile-batch launch at configs[3]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_second
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_tbatch_gpu.py tests/test_batch_gpu.py -m gpu -q 2>&1 | tail -60 > $O/pytest.txt
cat $O/pytest.txt
for v in "" "tbatch_max=8"; do
  timeout 300 python scripts/timeline.py c4 --set tbatch_max=4 ${v:+--set $v} > $O/timeline_${v:-k4}.json 2> $O/timeline.err || tail -3 $O/timeline.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_second/timeline_*.json')):
    d=json.load(open(f))
    print(f, {k:d[k] for k in d if k not in('resident_by_us',)})
    print([ (r['t'],r['update'],r['bin']) for r in d['resident_by_us'][::5]])
PY
