#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
time (python3 bench.py > $O/bench_default.json 2> $O/bench_default.err)
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_final/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype')}, d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['large']['roofline']['frac'], d['raycasting_on']['us_per_scan_hip_events'])
PY
