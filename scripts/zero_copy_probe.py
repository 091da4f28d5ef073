#!/usr/bin/env python3
"""Probe: scans read straight from PINNED host memory by the bin kernel (no staging copy) against
the staged async path (fdm_engine_integrate_async with zero_copy = 0) and the in-place path
(bin kernel reads pinned memory + writes it through to HBM; "direct" = no write-through).  Prints ms/scan for both and checks that the
two maps are identical."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fastdem_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
wl = synth.make(name)
out = {"workload": name}
maps = {}
for mode in ("staged", "in_place", "direct"):
    res = bench.Resident(wl, 0)
    res.eng.set_option("zero_copy", 0 if mode == "staged" else 1 << 22)
    pins = []
    for s in wl.scans:
        pin = {c: torch.from_numpy(s[c]).pin_memory() for c in ("x", "y", "z", "intensity", "rgb") if s.get(c) is not None}
        pins.append((s["x"].size, pin, {c: C.c_void_p(t.data_ptr()) for c, t in pin.items()}))

    def step(k):
        n, _, hp = pins[k % len(pins)]
        if mode != "direct":
            return res.eng.integrate_async_raw(n, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(k), hp.get("intensity"), hp.get("rgb"))
        return res.eng.integrate_device_raw(n, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(k), hp.get("intensity"), hp.get("rgb"))

    for k in range(50):
        step(k)
    res.eng.sync()
    t0 = time.perf_counter()
    for k in range(50, 50 + steps):
        step(k)
    res.eng.sync()
    out[mode + "_ms_per_scan"] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    maps[mode] = {l: res.eng.layer(l) for l in ("elevation", "variance", "n_points", "elevation_max")}
out["identical"] = all(all(np.array_equal(maps["staged"][l], maps[m][l], equal_nan=True) for l in maps["staged"]) for m in ("in_place", "direct"))
print(json.dumps(out))
