#!/usr/bin/env python3
"""Can two raycasting stages share the GPU?  Two independent engines (own streams) stream scans with raycasting on: one
alone, then both with their enqueue-only calls interleaved (chunks of `chunk` scans).  If the pair takes about as long as
one alone, the stage's small kernels leave the chip to each other.   python scripts/ray_overlap_probe.py c3 [scans] [chunk]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import synth
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wl = synth.make(name)
engs = []
for _ in range(2):
    r = bench.Resident(wl, 0)
    cfg = r.eng.cfg
    cfg.raycast_enabled = 1
    r.eng.set_config(cfg)
    w, _ = r.batch(0, 8)
    assert r.eng.integrate_device_batch(w) == 0
    r.eng.sync()
    engs.append(r)


def run(which):
    batches = {k: [engs[k].batch(8 + c, chunk)[0] for c in range(0, n, chunk)] for k in which}
    for k in which:
        engs[k].eng.sync()
    t0 = time.perf_counter()
    for c in range(len(batches[which[0]])):
        for k in which:
            assert engs[k].eng.integrate_device_batch(batches[k][c]) == 0
    for k in which:
        engs[k].eng.sync()
    return (time.perf_counter() - t0) / n * 1e6


out = {"workload": name, "scans": n, "chunk": chunk}
out["one_engine_us_per_scan"] = round(min(run([0]) for _ in range(3)), 2)
out["two_engines_us_per_scan_of_either"] = round(min(run([0, 1]) for _ in range(3)), 2)
print(json.dumps(out))
