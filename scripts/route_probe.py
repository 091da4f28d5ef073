#!/usr/bin/env python3
"""Where a routed step's time goes (1 rank, configs[4] map): phases timed with a device sync after each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as dist
from fastdem_amd import Engine, capi, synth, tiling
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
wl = synth.global_map(n_scans=2)
rows = cols = 8000
plan = tiling.make_plan(0, 1, rows, cols, 6)
eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()))
router = tiling.RoutedScan(eng, plan, "cuda:0", max_points=wl.n_points)
d = [{c: torch.from_numpy(s[c]).cuda() for c in ("x", "y", "z", "intensity")} for s in wl.scans]
def sync():
    eng.sync(); torch.cuda.synchronize()
for k in range(5):
    router.integrate(d[k % 2]["x"], d[k % 2]["y"], d[k % 2]["z"], wl.T_base_sensor, wl.pose(k), dist, intensity=d[k % 2]["intensity"], sensors=True)
sync()
T = {}
def lap(name, t0):
    sync(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
N = 30
for k in range(N):
    dd = d[k % 2]
    t0 = time.perf_counter(); eng.route_scan(router.rp, dd["x"], dd["y"], dd["z"], wl.T_base_sensor, wl.pose(k), router.send, router.counts, intensity=dd["intensity"]); lap("route", t0)
    t0 = time.perf_counter(); c = router.counts.cpu().numpy(); lap("counts_readback", t0)
    n_in = int(c[0])
    t0 = time.perf_counter(); router._recv_buffer(n_in)[:n_in].copy_(router.send[:n_in]); lap("self_copy", t0)
    t0 = time.perf_counter(); eng.integrate_points4_device(router.recv, n_in, wl.T_base_sensor, wl.pose(k), True, True); lap("integrate_points4", t0)
    t0 = time.perf_counter(); eng.integrate_device(dd["x"], dd["y"], dd["z"], wl.T_base_sensor, wl.pose(k), intensity=dd["intensity"]); lap("plain_integrate_device", t0)
t0 = time.perf_counter()
for k in range(N):
    dd = d[k % 2]
    router.integrate(dd["x"], dd["y"], dd["z"], wl.T_base_sensor, wl.pose(k), dist, intensity=dd["intensity"], sensors=True)
sync(); T["whole_step"] = time.perf_counter() - t0
print({k: round(v / N * 1e6, 1) for k, v in T.items()})
ts = []
for k in range(12):
    dd = d[k % 2]
    t0 = time.perf_counter()
    router.integrate(dd["x"], dd["y"], dd["z"], wl.T_base_sensor, wl.pose(k), dist, intensity=dd["intensity"], sensors=True)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e6), round((t2 - t1) * 1e6)))
print("per step (host us, then wait us):", ts)
ts = []
for k in range(12):
    dd = d[k % 2]
    t0 = time.perf_counter()
    router.integrate(dd["x"], dd["y"], dd["z"], wl.T_base_sensor, wl.pose(k), dist, intensity=dd["intensity"], sensors=False)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e6), round((t2 - t1) * 1e6)))
print("slices mode:", ts)
