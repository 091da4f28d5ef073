#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for D in 0 1; do
python - $D <<'PY'
import sys, json
sys.path.insert(0, '.')
import torch, bench
from fastdem_amd import synth
wl = synth.make("c2", n_scans=64)
r = bench.Resident(wl, 0)
r.eng.set_option("dbg_batch", int(sys.argv[1]))
w, _ = r.batch(0, 160); assert r.eng.integrate_device_batch_timed(w) == 0
out = []
for rep in range(3):
    b, _ = r.batch(160, 1600); assert r.eng.integrate_device_batch_timed(b) == 0
    out.append(round(r.eng.timer_ms() / 100 * 1e3, 2))
print("dbg_batch", sys.argv[1], "us per 16-scan launch", out)
PY
done
