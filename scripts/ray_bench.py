#!/usr/bin/env python3
"""Raycasting stage: device time per scan next to the CPU oracle.
   python scripts/ray_bench.py c2 [--steps 30] [--cpu-iters 5]
Prints one JSON line: HIP-event ms of the whole stage (voxel keys + sort + rays + resolve), the
integrate() wall time per scan with the stage on and off, and the oracle's per-scan cost of the
same stage (integrate with raycasting minus integrate without, same scans, one thread)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--cpu-iters", type=int, default=5)
ap.add_argument("--dbg-ray", type=int, default=0)
ap.add_argument("--order", default="azimuth", choices=["azimuth", "ring"])
ap.add_argument("--set", action="append", default=[])
a = ap.parse_args()

wl = synth.make(a.workload, **({"order": a.order} if a.workload in ("c2", "c4") else {}))


def run(raycast):
    res = bench.Resident(wl, 0)
    cfg = res.eng.cfg
    cfg.raycast_enabled = raycast
    res.eng.set_config(cfg)
    res.eng.set_option("dbg_ray", a.dbg_ray)
    for kv in a.set:
        res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    for k in range(10):
        res.step(k)
    res.eng.sync()
    t0 = time.perf_counter()
    for k in range(10, 10 + a.steps):
        res.step(k)
    res.eng.sync()
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    ray = 0.0
    if raycast:
        res.eng.enable_profile(True)
        for k in range(10 + a.steps, 10 + 2 * a.steps):
            res.step(k)
            ray += res.eng.last_ray_ms()
        res.eng.enable_profile(False)
        ray /= a.steps
    # the batch entry point (fdm_engine_integrate_device_batch): small scans leave in batches of 16, raycasting included
    w, _ = res.batch(0, 64)
    assert res.eng.integrate_device_batch_timed(w) == 0
    n_b = 640 if wl.n_points < 65536 else 32
    b, _ = res.batch(64, n_b)
    launches = sum(res.eng.batch_launches())
    assert res.eng.integrate_device_batch_timed(b) == 0
    batch_us = res.eng.timer_ms() / n_b * 1e3
    batched = sum(res.eng.batch_launches()) > launches
    return wall, ray, batch_us, batched


wall_off, _, batch_off, _ = run(0)
wall_on, ray_ms, batch_on, batched = run(1)

import fdm_ref_py as R  # noqa: E402  (checker / CPU baseline only)


def cpu(raycast):
    cfg = wl.apply_to(R.default_config())
    cfg.raycast_enabled = raycast
    ref = R.RefEngine(wl.width, wl.height, wl.resolution, cfg)
    s = wl.scans[0]
    poses = [wl.pose(k) for k in range(8)]
    ref.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, 2, intensity=s["intensity"], rgb=s["rgb"])
    return ref.time_integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, poses, a.cpu_iters,
                              intensity=s["intensity"], rgb=s["rgb"]) / a.cpu_iters * 1e3


cpu_off, cpu_on = cpu(0), cpu(1)
print(json.dumps({"workload": a.workload, "points": wl.n_points,
                  "gpu_ray_stage_ms": round(ray_ms, 4),
                  "gpu_integrate_ms": {"raycast_off": round(wall_off, 4), "raycast_on": round(wall_on, 4)},
                  "gpu_batch_call_us_per_scan": {"raycast_off": round(batch_off, 3), "raycast_on": round(batch_on, 3),
                                                 "raycast_on_took_batch_launches": batched},
                  "cpu_oracle_ms": {"raycast_off": round(cpu_off, 3), "raycast_on": round(cpu_on, 3),
                                    "stage": round(cpu_on - cpu_off, 3)},
                  "speedup_stage": round((cpu_on - cpu_off) / max(ray_ms, 1e-9), 1)}))
