#!/bin/bash
# round-2 GPU pass: parity of the tiled pipeline + C4 A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "tiled_all" > gpurun_out/t_tiled.log 2>&1
echo "tiled_all parity rc=$?" ; tail -15 gpurun_out/t_tiled.log
run() {  # name, args...
  name=$1; shift
  timeout 600 python bench.py --workload c4 --steps 2000 --warmup 200 --no-cpu-baseline --profile-steps 50 "$@" > gpurun_out/c4_$name.json 2> gpurun_out/c4_$name.err
  echo "c4 $name rc=$?"; tail -2 gpurun_out/c4_$name.err | grep -v amdgpu.ids
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/c4_$name.json").read().strip().splitlines()[-1])
    print("  value", round(d["value"]), "us/step", round(d["ms_per_step"]*1e3,2), "roof us", d.get("roofline",{}).get("avg_kernel_us"), "frac", d.get("roofline",{}).get("frac"))
    print("  ", {k:(round(v["ms"]*1e3,2) if isinstance(v,dict) else v) for k,v in d.get("kernels",{}).items() if isinstance(v,dict) and "ms" in v})
except Exception as e: print("parse fail", e)
PY
}
for a in "$@"; do
  run "$(echo $a | tr '=, ' '___')" $(echo $a | tr ',' ' ' | sed 's/\([a-z_]*=[0-9]*\)/--set \1/g')
done
