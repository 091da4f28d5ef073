#!/usr/bin/env python3
"""Round-4 soak (every profile: now and then the call goes in through
fdm_engine_integrate_host_batch on pageable host arrays on side A).  Otherwise the round-3 soak: the new paths against the old ones, engine against engine, bit for bit, for a few minutes.
  A: defaults — batch launches (fdm_multi.hpp), the sort-free voxel filter (k_vs_*), the own radix sort
  B: batch 0, voxel_small 0 — one launch per scan, every voxel filter through the radix sort
Random scans (1 .. 70 000 points: both sides of the 64 K limit of the sort-free filter; now and then every point
filtered, a dense cluster, duplicates), random poses with LOCAL-mode moves, raycasting on for a third of the scans
(the batch path takes the others), every layer compared exactly every few calls.   python scripts/soak_r03.py [seconds]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastdem_amd import capi
from fastdem_amd.engine import Engine

F32 = np.float32
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
every = int(sys.argv[2]) if len(sys.argv) > 2 else 5   # compare every so many calls
with_oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"  # the CPU oracle in lock-step (slow): who is right
profile = sys.argv[4] if len(sys.argv) > 4 else "small"
# small: 16 m map, Kalman, LOCAL       A = defaults, B = batch 0 + voxel_small 0
# tiled: 60 m map (351 tiles), scans up to 300 K points     A = defaults (record-pool pipeline), B = tiled 0
# p2   : 12 m GLOBAL map, P2 quantile estimator, colour + intensity     A = defaults, B = batch 0 + voxel_small 0
# ray   : 16 m map, raycasting on in every call      A = the stage inside the batches (fdm_rbatch.hpp), B = batch_ray 0 (one scan per launch)
# rayp2 : the same with the P2 estimator + colour
# walk  : 16 m LOCAL map, Kalman     A = the walker block walks every batch's chain of moves one launch ahead (batch_walk 1), B = batch_walk 0
# rayw  : 40 m map, raycasting on in every call, scans up to 200 K points     A = every scan's ray walk on the sector window
#         in LDS (fdm_raywedge.hpp: ray_large_min 1), B = one lane per ray on memory-side atomics (ray_wedge 0)
SIZE = {"small": 16.0, "tiled": 60.0, "p2": 12.0, "ray": 16.0, "rayp2": 12.0, "walk": 16.0, "rayw": 40.0}[profile]
BIG = {"small": 70000, "tiled": 300000, "p2": 40000, "ray": 70000, "rayp2": 40000, "walk": 30000, "rayw": 200000}[profile]
rng = np.random.default_rng(2026)


def T(x, y, yaw):
    M = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    M[:2, :2] = [[c, -s], [s, c]]
    M[0, 3], M[1, 3] = x, y
    return M


def col16(M):
    return (C.c_double * 16)(*np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(16))


def make(raycast):
    cfg = capi.default_config()
    cfg.z_min, cfg.z_max, cfg.range_min, cfg.range_max = -2.0, 4.0, 0.2, 12.0
    cfg.raycast_enabled = raycast
    if profile in ("p2", "rayp2"):
        cfg.mode = 1 if profile == "p2" else 0
        cfg.estimation_type = 1
        cfg.sensor_type = 2
    if profile in ("ray", "rayp2", "rayw"):  # ghosts get cleared now and then
        cfg.rc_log_odds_ghost, cfg.rc_clear_threshold, cfg.rc_height_conflict_threshold = 0.9, -0.5, 0.02
    return cfg


def cloud(n):
    half = SIZE / 2 + 1.0
    x = rng.uniform(-half, half, n).astype(F32)
    y = rng.uniform(-half, half, n).astype(F32)
    z = (rng.uniform(-1.0, 0.4, n) - 1.2).astype(F32)
    kind = rng.integers(0, 12)
    if kind == 0:
        z += 40.0  # every point filtered
    elif kind == 1 and n > 30:  # a dense cluster with exact duplicates
        m = n // 2
        x[:m] = (1.0 + rng.uniform(0, 0.5, m)).astype(F32)
        y[:m] = (-2.0 + rng.uniform(0, 0.5, m)).astype(F32)
        x[3:m:5], y[3:m:5], z[3:m:5] = x[2], y[2], z[2]
    a = rng.uniform(0, 1, n).astype(F32)
    if kind == 2:
        a[::7] = np.nan
        z[::5] = 0.0 - 1.2
    return x, y, z, a


Tbs = np.eye(4)
Tbs[2, 3] = 1.2
A = Engine(SIZE, SIZE, 0.1, make(0))
B = Engine(SIZE, SIZE, 0.1, make(0))
if profile == "tiled":
    B.set_option("tiled", 0)
elif profile in ("ray", "rayp2"):
    B.set_option("batch_ray", 0)
elif profile == "rayw":
    for e in (A, B):
        e.set_option("ray_large_min", 1)
        e.set_option("batch_ray", 0)
    B.set_option("ray_wedge", 0)
elif profile == "walk":
    A.set_option("batch_walk", 1)
    B.set_option("batch_walk", 0)
else:
    B.set_option("batch", 0)
    B.set_option("voxel_small", 0)
Rf = None
if with_oracle:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fdm_ref_py as R  # (the checker)
    rcfg = R.default_config()
    rcfg.z_min, rcfg.z_max, rcfg.range_min, rcfg.range_max = -2.0, 4.0, 0.2, 12.0
    if profile in ("p2", "rayp2"):
        rcfg.mode, rcfg.estimation_type, rcfg.sensor_type = (1 if profile == "p2" else 0), 1, 2
    if profile in ("ray", "rayp2", "rayw"):
        rcfg.rc_log_odds_ghost, rcfg.rc_clear_threshold, rcfg.rc_height_conflict_threshold = 0.9, -0.5, 0.02
    Rf = R.RefEngine(SIZE, SIZE, 0.1, rcfg)
t0 = time.perf_counter()
scans = calls = compares = 0
px = py = 0.0
alive = []  # device arrays stay alive until both engines have synchronised (the calls only enqueue)
while time.perf_counter() - t0 < budget:
    ray = 1 if profile in ("ray", "rayp2", "rayw") else int(rng.integers(0, 3) == 0)
    for e in (A, B) + ((Rf,) if Rf else ()):
        cfg = e.cfg
        cfg.raycast_enabled = ray
        e.set_config(cfg)
    count = int(rng.integers(1, 40))
    sizes = [int(rng.integers(1, BIG)) if rng.integers(0, 4) == 0 else int(rng.integers(1, 6000)) for _ in range(count)]
    keep, arr = [], (capi.FdmDeviceScan * count)()
    alive.append(keep)
    for k, n in enumerate(sizes):
        x, y, z, a = cloud(n)
        rgb = rng.integers(0, 1 << 24, n, dtype=np.uint32) if profile in ("p2", "rayp2") else None
        d = [torch.from_numpy(v).cuda() for v in (x, y, z, a)] + ([torch.from_numpy(rgb.view(np.int32)).cuda()] if rgb is not None else [])
        keep.append(d)
        host = (x, y, z, a)
        px += float(rng.uniform(-0.3, 0.4))
        py += float(rng.uniform(-0.2, 0.2))
        arr[k].n = n
        arr[k].x, arr[k].y, arr[k].z, arr[k].intensity = (t.data_ptr() for t in d[:4])
        arr[k].rgb = d[4].data_ptr() if rgb is not None else None
        arr[k].sigma_z2 = None
        arr[k].T_base_sensor = col16(Tbs)
        arr[k].T_world_base = col16(T(px, py, 0.01 * scans))
        if Rf:
            Rf.integrate(host[0], host[1], host[2], Tbs, T(px, py, 0.01 * scans), intensity=host[3], rgb=rgb)
        scans += 1
    torch.cuda.synchronize()
    how = int(rng.integers(0, 8))
    if how == 0 and profile not in ("p2", "rayp2"):  # the same scans one by one (enqueue-only calls) ...
        for d, sc in zip(keep, arr):
            Tb, Tw = np.array(sc.T_base_sensor).reshape(4, 4).T, np.array(sc.T_world_base).reshape(4, 4).T
            for e in (A, B):
                e.integrate_device(d[0], d[1], d[2], Tb, Tw, intensity=d[3])
    elif how == 1 and profile not in ("p2", "rayp2"):  # ... or the first as a synchronous host call, the rest as a batch
        d, sc = keep[0], arr[0]
        Tb, Tw = np.array(sc.T_base_sensor).reshape(4, 4).T, np.array(sc.T_world_base).reshape(4, 4).T
        hx, hy, hz, ha = (t.cpu().numpy() for t in d[:4])
        ra, rb = A.integrate(hx, hy, hz, Tb, Tw, intensity=ha), B.integrate(hx, hy, hz, Tb, Tw, intensity=ha)
        assert ra == rb, (ra, rb)
        if count > 1:
            rest = (capi.FdmDeviceScan * (count - 1))(*list(arr)[1:])
            assert A.integrate_device_batch(rest) == 0
            assert B.integrate_device_batch(rest) == 0
    elif how == 2:  # side A from pageable HOST arrays in one call (staged), side B from the device arrays
        harr = (capi.FdmDeviceScan * count)()
        hkeep = []
        for k, d in enumerate(keep):
            hs = [t.cpu().numpy() for t in d]
            hkeep.append(hs)
            harr[k].n = arr[k].n
            harr[k].x, harr[k].y, harr[k].z, harr[k].intensity = (h.ctypes.data for h in hs[:4])
            harr[k].rgb = hs[4].ctypes.data if len(hs) > 4 else None
            harr[k].sigma_z2 = None
            harr[k].T_base_sensor = arr[k].T_base_sensor
            harr[k].T_world_base = arr[k].T_world_base
        A.integrate_host_batch(harr)
        assert B.integrate_device_batch(arr) == 0
    else:
        assert A.integrate_device_batch(arr) == 0
        assert B.integrate_device_batch(arr) == 0
    calls += 1
    if calls % every == 0:
        A.sync()
        B.sync()
        dirty = A.debug_batch_dirty()
        if dirty != (0, 0, 0):
            print(json.dumps({"dirty_scratch": dirty, "call": calls, "scans": scans, "sizes": sizes, "raycast": ray}))
            raise SystemExit("batch scratch not clean")
        assert A.last_stats() == B.last_stats(), (A.last_stats(), B.last_stats())
        assert sorted(A.layers()) == sorted(B.layers())
        for name in A.layers():
            la, lb = A.layer(name), B.layer(name)
            same = (la.view(np.uint32) == lb.view(np.uint32)) | (np.isnan(la) & np.isnan(lb))
            if not same.all():
                bad = np.argwhere(~same)
                if Rf:
                    lr = Rf.layer(name)
                    print("oracle", lr[~same][:5].tolist(), "A wrong" if not np.array_equal(lr[~same], la[~same], equal_nan=True) else "A ok",
                          "B wrong" if not np.array_equal(lr[~same], lb[~same], equal_nan=True) else "B ok")
                print(json.dumps({"layer": name, "cells": int((~same).sum()), "scans": scans, "call": calls, "raycast": ray,
                                  "sizes": sizes, "first": bad[:5].tolist(), "A": la[~same][:5].tolist(), "B": lb[~same][:5].tolist(),
                                  "rows": [int(bad[:, 0].min()), int(bad[:, 0].max())], "cols": [int(bad[:, 1].min()), int(bad[:, 1].max())],
                                  "geomA": str(A.geometry()), "geomB": str(B.geometry()), "statsA": A.last_stats(), "last_batch": A.last_batch()}))
                raise SystemExit(f"layer {name} differs in {(~same).sum()} cells after {scans} scans")
        compares += 1
        alive.clear()
print(json.dumps({"profile": profile, "seconds": round(time.perf_counter() - t0, 1), "scans": scans, "batch_calls": calls, "exact_compares": compares,
                  "layers": len(A.layers()), "finite_cells": int(np.isfinite(A.layer("elevation")).sum())}))
