#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
time (python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err)
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_final/bench_steps20.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['large']['roofline']['frac'], d['raycasting_on']['us_per_scan_hip_events'])
PY
