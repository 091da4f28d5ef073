#!/usr/bin/env python3
"""One-off stall early in a process's life?  Per-step host time stamps of enqueue-only c5 steps (raw C calls, a sync
every 25 steps) from the first step on."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastdem_amd import Engine, capi, synth
t_start = time.perf_counter()
wl = synth.global_map(n_scans=2)
eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()), device=0)
mine = [{c: torch.from_numpy(s[c]).cuda() for c in ("x", "y", "z", "intensity")} for s in wl.scans]
torch.cuda.synchronize()
tb = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(wl.T_base_sensor, dtype=np.float64).T).reshape(16))
marks = []
t0 = time.perf_counter()
for k in range(600):
    d = mine[k % 2]
    tw = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(wl.pose(k), dtype=np.float64).T).reshape(16))
    eng.integrate_device_raw(wl.n_points, d["x"].data_ptr(), d["y"].data_ptr(), d["z"].data_ptr(), tb, tw, dint=d["intensity"].data_ptr())
    if k % 150 == 149:
        eng.sync()
        marks.append(round((time.perf_counter() - t0) * 1e3, 2))
seg = [round(b - a, 2) for a, b in zip([0.0] + marks[:-1], marks)]
print(json.dumps({"setup_s": round(t0 - t_start, 2), "ms_per_150_steps_no_sync_inside": seg}))
