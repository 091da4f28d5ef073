#!/usr/bin/env python3
"""A/B of engine options on the batch pipeline, same box, interleaved repetitions (configs[1], 16-scan batches).
usage: batch_ab.py "opt=val,opt=val" "opt=val" ...   (each argument one variant; "" = defaults)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import synth
import bench

variants = sys.argv[1:] or [""]
wl = synth.make("c2", n_scans=8)
res = {}
for rep in range(3):
    for v in variants:
        r = bench.Resident(wl, 0)
        for kv in [x for x in v.split(",") if x]:
            r.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        for kk in range(4000):
            r.pose(kk)
        w, _ = r.batch(0, 640)
        assert r.eng.integrate_device_batch_timed(w) == 0
        b, pts = r.batch(640, 3200)
        assert r.eng.integrate_device_batch_timed(b) == 0
        res.setdefault(v or "default", []).append(round(r.eng.timer_ms() / 3200 * 1e3, 3))
        del r
print(json.dumps(res))
