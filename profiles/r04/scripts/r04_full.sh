#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_full
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_full/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','device_value')}, d['roofline']['frac'], d.get('latency_path'), d['large']['roofline']['avg_kernel_us'], d['large']['roofline']['frac'])
print({k:v for k,v in d.items() if k.startswith('host_')})
PY
