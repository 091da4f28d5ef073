import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from fastdem_amd import synth
wl = synth.make("c2")
res = bench.Resident(wl, 0)
for k in range(200): res.pose(k)
for k in range(50): res.step(k)
res.eng.sync()
# host-side cost: enqueue 2000 scans as fast as possible; the GPU (6 us/scan) is slower than the host if host < 6 us
t0 = time.perf_counter()
for k in range(2000): res.step(50 + k % 100)
t_enq = time.perf_counter() - t0
res.eng.sync()
t_all = time.perf_counter() - t0
print("enqueue-only per call %.2f us, incl. drain %.2f us" % (t_enq / 2000 * 1e6, t_all / 2000 * 1e6))
# the same for the host entry point on pinned arrays (hipPointerGetAttributes per channel + one launch)
s = wl.scans[0]
pin = {c: torch.from_numpy(s[c]).pin_memory() for c in ("x", "y", "z", "intensity")}
hp = {c: C.c_void_p(t.data_ptr()) for c, t in pin.items()}
for k in range(50):
    res.eng.integrate_async_raw(s["x"].size, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(k), hp["intensity"])
res.eng.sync()
t0 = time.perf_counter()
for k in range(2000):
    res.eng.integrate_async_raw(s["x"].size, hp["x"], hp["y"], hp["z"], res.tbs, res.pose(50 + k % 100), hp["intensity"])
t_enq = time.perf_counter() - t0
res.eng.sync()
t_all = time.perf_counter() - t0
print("pinned host arrays: enqueue-only per call %.2f us, incl. drain %.2f us" % (t_enq / 2000 * 1e6, t_all / 2000 * 1e6))
