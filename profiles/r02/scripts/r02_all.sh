#!/bin/bash
# all four GPU configs, short runs (no CPU baseline)
mkdir -p gpurun_out
for W in c2 c3 c4 c5; do
  S=2000; [ $W = c2 ] && S=10000; [ $W = c5 ] && S=300
  timeout 900 python bench.py --workload $W --steps $S --warmup 200 --no-cpu-baseline --no-large --profile-steps 50 "$@" > gpurun_out/all_$W.json 2> gpurun_out/all_$W.err
  rc=$?
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/all_$W.json").read().strip().splitlines()[-1])
    k=d.get("kernels",{})
    print("$W rc=$rc", "Mpts/s", round(d["value"]), "us/step", round(d["ms_per_step"]*1e3,2), "frac", round(d.get("roofline",{}).get("frac") or 0,4), {a:round(v["ms"]*1e3,2) for a,v in k.items() if isinstance(v,dict) and "ms" in v})
except Exception as e: print("$W rc=$rc parse fail", e); print(open("gpurun_out/all_$W.err").read()[-600:])
PY
done
