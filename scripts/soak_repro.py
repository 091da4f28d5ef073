#!/usr/bin/env python3
"""Replay of scripts/soak_r03.py's random sequence up to a given call, the last call truncated to its first m scans:
where does the batch path leave the oracle?   python scripts/soak_repro.py <call> <row> <col> [batch_max]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from fastdem_amd import capi
from fastdem_amd.engine import Engine
import fdm_ref_py as R

F32 = np.float32
call_stop, row, col = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bmax = int(sys.argv[4]) if len(sys.argv) > 4 else 16


def T(x, y, yaw):
    M = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    M[:2, :2] = [[c, -s], [s, c]]
    M[0, 3], M[1, 3] = x, y
    return M


def col16(M):
    return (C.c_double * 16)(*np.ascontiguousarray(np.asarray(M, dtype=np.float64).T).reshape(16))


def run(m_last):
    rng = np.random.default_rng(2026)

    def cloud(n):
        x = rng.uniform(-9.0, 9.0, n).astype(F32)
        y = rng.uniform(-9.0, 9.0, n).astype(F32)
        z = (rng.uniform(-1.0, 0.4, n) - 1.2).astype(F32)
        kind = rng.integers(0, 12)
        if kind == 0:
            z += 40.0
        elif kind == 1 and n > 30:
            m = n // 2
            x[:m] = (1.0 + rng.uniform(0, 0.5, m)).astype(F32)
            y[:m] = (-2.0 + rng.uniform(0, 0.5, m)).astype(F32)
            x[3:m:5], y[3:m:5], z[3:m:5] = x[2], y[2], z[2]
        a = rng.uniform(0, 1, n).astype(F32)
        return x, y, z, a

    cfg = capi.default_config()
    cfg.z_min, cfg.z_max, cfg.range_min, cfg.range_max = -2.0, 4.0, 0.2, 12.0
    cfg.raycast_enabled = 0
    A = Engine(16.0, 16.0, 0.1, cfg)
    A.set_option("batch_max", bmax)
    rcfg = R.default_config()
    rcfg.z_min, rcfg.z_max, rcfg.range_min, rcfg.range_max = -2.0, 4.0, 0.2, 12.0
    Rf = R.RefEngine(16.0, 16.0, 0.1, rcfg)
    Tbs = np.eye(4)
    Tbs[2, 3] = 1.2
    scans = calls = 0
    px = py = 0.0
    trace = []
    while calls < call_stop:
        ray = int(rng.integers(0, 3) == 0)
        for e in (A, Rf):
            c = e.cfg
            c.raycast_enabled = ray
            e.set_config(c)
        count = int(rng.integers(1, 40))
        sizes = [int(rng.integers(1, 70000)) if rng.integers(0, 4) == 0 else int(rng.integers(1, 6000)) for _ in range(count)]
        last = calls + 1 == call_stop
        use = min(count, m_last) if last else count
        keep, arr = [], (capi.FdmDeviceScan * count)()
        for k, n in enumerate(sizes):
            x, y, z, a = cloud(n)
            px += float(rng.uniform(-0.3, 0.4))
            py += float(rng.uniform(-0.2, 0.2))
            if k < use:
                d = [torch.from_numpy(v).cuda() for v in (x, y, z, a)]
                keep.append(d)
                arr[k].n = n
                arr[k].x, arr[k].y, arr[k].z, arr[k].intensity = (t.data_ptr() for t in d)
                arr[k].rgb = None
                arr[k].sigma_z2 = None
                arr[k].T_base_sensor = col16(Tbs)
                arr[k].T_world_base = col16(T(px, py, 0.01 * scans))
                rc, st = Rf.integrate(x, y, z, Tbs, T(px, py, 0.01 * scans), intensity=a)
                if last:
                    trace.append((k, n, rc, st["n_in_map"], st["shift_rows"], st["shift_cols"], float(Rf.layer("obstacle")[row, col])))
            scans += 1
        torch.cuda.synchronize()
        assert A.integrate_device_batch(arr, use) == 0
        A.sync()
        calls += 1
    la, lr = A.layer("obstacle"), Rf.layer("obstacle")
    same = (la.view(np.uint32) == lr.view(np.uint32)) | (np.isnan(la) & np.isnan(lr))
    return int((~same).sum()), float(la[row, col]), float(lr[row, col]), trace, np.argwhere(~same)[:4].tolist()


full = run(10 ** 9)
print("full call: differing cells", full[0], "A", full[1], "oracle", full[2], full[4])
for t in full[3]:
    print("  scan", t)
for m in range(1, len(full[3]) + 1):
    r = run(m)
    print("first", m, "scans: differing cells", r[0], "A", r[1], "oracle", r[2], r[4])
