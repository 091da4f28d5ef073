#!/usr/bin/env python3
"""configs[1] with raycasting on through the batch entry point: `n` scans in one fdm_engine_integrate_device_batch call
(what rocprofv3 wraps for profiles/r04/rocprof_ray_batch_c2_kernel_stats.csv).  python3 scripts/ray_batch_run.py [n] [key=val ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import synth
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
wl = synth.make("c2")
res = bench.Resident(wl, 0)
cfg = res.eng.cfg
cfg.raycast_enabled = 1
res.eng.set_config(cfg)
for kv in sys.argv[2:]:
    res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
w, _ = res.batch(0, 64)
assert res.eng.integrate_device_batch_timed(w) == 0
out = []
for rep in range(3):
    b, _ = res.batch(64 + rep * n, n)
    assert res.eng.integrate_device_batch_timed(b) == 0
    out.append(round(res.eng.timer_ms() / n * 1e3, 3))
print(json.dumps({"workload": wl.name, "scans_per_call": n, "raycasting": 1, "us_per_scan": out, "batch_launches": res.eng.batch_launches()}))
