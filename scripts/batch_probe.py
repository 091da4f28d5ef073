#!/usr/bin/env python3
"""Device time per scan of the batch pipeline on configs[1] for several batch sizes (HIP events around one
fdm_engine_integrate_device_batch call).  usage: batch_probe.py [key=value ...] (engine options)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import synth
import bench

opts = [a for a in sys.argv[1:] if "=" in a]
wl = synth.make("c2", n_scans=8)
out = {}
for bm in (0, 2, 4, 8, 16):
    res = bench.Resident(wl, 0)
    for kv in opts:
        res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if bm == 0:
        res.eng.set_option("batch", 0)
    else:
        res.eng.set_option("batch_max", bm)
    for kk in range(4000):
        res.pose(kk)
    w, _ = res.batch(0, 640)
    assert res.eng.integrate_device_batch_timed(w) == 0
    b, pts = res.batch(640, 3200)
    assert res.eng.integrate_device_batch_timed(b) == 0
    us = res.eng.timer_ms() / 3200 * 1e3
    out[f"batch_max_{bm}"] = {"us_per_scan": round(us, 3), "Mpts_per_s": round(pts / 3200 / us, 1)}
    del res
print(json.dumps(out))
