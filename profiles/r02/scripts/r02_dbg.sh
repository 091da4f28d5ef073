#!/bin/bash
# phase timing of the tiled kernels through the measurement-only exits
mkdir -p gpurun_out
run() {
  name=$1; shift
  timeout 600 python bench.py --workload c4 --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 40 "$@" > gpurun_out/d_$name.json 2> gpurun_out/d_$name.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/d_$name.json").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print("$name", "rc=$rc", "step us", round(d["ms_per_step"]*1e3,2), {a:round(v["ms"]*1e3,2) for a,v in k.items() if isinstance(v,dict) and "ms" in v})
except Exception as e: print("$name rc=$rc parse fail", e); print(open("gpurun_out/d_$name.err").read()[-400:])
PY
}
for a in "$@"; do
  run "$(echo $a | tr '=, ' '___')" $(echo $a | tr ',' ' ' | sed 's/\([a-z_]*=[0-9]*\)/--set \1/g')
done
