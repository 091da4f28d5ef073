#!/usr/bin/env python3
"""Where does a one-rank c5 step spend its time?  (bench_global measured 0.44 ms per step in most runs and 0.037 in some:
none of the per-step host calls probed here — it was ONE 50 ms stall of the first engine call behind the first RCCL
barrier of the process, whose communicator set-up the barrier only enqueues; python bench.py --workload c5 --trace-steps
shows the per-step host times.  bench.py now warms the communicator and synchronises behind every barrier.)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastdem_amd import Engine, capi, synth
from fastdem_amd import halo as halo_c
wl = synth.global_map(n_scans=2)
eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()), device=0)
rows = cols = int(round(float(np.float32(wl.width)) / float(np.float32(wl.resolution))))
mine = [{c: torch.from_numpy(s[c]).cuda() for c in ("x", "y", "z", "intensity")} for s in wl.scans]
native = halo_c.NativeRoutedScan(eng, 0, 1, rows, cols, 6, wl.n_points, comm=None)
out = {}
def run(name, fn, steps=100):
    for k in range(10): fn(k)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(10, 10 + steps): fn(k)
    t1 = time.perf_counter()
    eng.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[name] = {"enqueue_ms_per_step": round((t1 - t0) / steps * 1e3, 4), "total_ms_per_step": round((t2 - t0) / steps * 1e3, 4)}
def direct(k):
    d = mine[k % 2]
    eng.integrate_device(d["x"], d["y"], d["z"], wl.T_base_sensor, wl.pose(k), intensity=d["intensity"])
def routed(k):
    d = mine[k % 2]
    native.integrate(d["x"], d["y"], d["z"], wl.T_base_sensor, wl.pose(k), intensity=d["intensity"], sensors=True, want_matrix=False)
import ctypes as C
def raw(k):  # no torch event per step
    d = mine[k % 2]
    tb = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(wl.T_base_sensor, dtype=np.float64).T).reshape(16))
    tw = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(wl.pose(k), dtype=np.float64).T).reshape(16))
    eng.integrate_device_raw(wl.n_points, d["x"].data_ptr(), d["y"].data_ptr(), d["z"].data_ptr(), tb, tw, dint=d["intensity"].data_ptr())
def timed_kernels(name, fn, steps=40):
    """per-step device time through HIP events on the engine stream (timer_start/stop around each call)"""
    ms = []
    for k in range(200, 200 + steps):
        eng.timer_start(); fn(k); eng.timer_stop(); ms.append(eng.timer_ms())
    out[name + "_device_ms"] = {"median": round(float(np.median(ms)), 4), "max": round(float(np.max(ms)), 4), "first5": [round(m, 4) for m in ms[:5]]}
orig_wait = eng.wait_torch
def ev_always():
    if getattr(eng, "_ev_in", None) is None:
        eng._ev_in = torch.cuda.Event()
    eng._ev_in.record()
    from fastdem_amd.engine import _ck
    _ck(eng._lib.fdm_engine_wait_event(eng._h, C.c_void_p(eng._ev_in.cuda_event)))
def query_only():
    torch.cuda.current_stream().query()
variants = {"none": (lambda: None), "query_only": query_only, "event_always": ev_always, "shipped": orig_wait}
for rep in range(3):
    for name, w in variants.items():
        eng.wait_torch = w
        run(f"direct[{name}]_{rep}", direct)
    run(f"raw_{rep}", raw)
print(json.dumps({k: v["total_ms_per_step"] for k, v in out.items()}, indent=0))
