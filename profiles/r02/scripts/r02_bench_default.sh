#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py "$@" > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo rc=$?
tail -c 800 gpurun_out/bench_default.err | grep -v amdgpu.ids
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","device_value","ms_per_step","steps")}, d["roofline"]["avg_kernel_us"], d["roofline"]["frac"])
print({k:v for k,v in d.items() if k.startswith("host_")})
if "large" in d:
    print("large", d["large"]["value"], d["large"]["ms_per_step"], d["large"]["roofline"]["frac"], {a:round(v["ms"]*1e3,2) for a,v in d["large"]["kernels"].items() if isinstance(v,dict) and "ms" in v})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["march_native_value"], d["cpu_baseline"]["parity"])
PY
