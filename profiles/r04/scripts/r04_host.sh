#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04_host
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_cpp_host_api.py tests/test_batch_gpu.py tests/test_tiling_gpu.py -m gpu -q -x 2>&1 | tail -15
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_host/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','device_value')}, 'frac', d['roofline']['frac'], 'cache_resident', d.get('cache_resident'))
print(d['config']['inputs'])
print({k:v for k,v in d.items() if k.startswith('host_')})
print('large', d['large']['roofline']['avg_kernel_us'], d['large']['roofline']['frac'], {k:v for k,v in d['large'].items() if k.startswith('host_')})
PY
bash scripts/r04_pmc_mbatch.sh 2>&1 | tail -12
