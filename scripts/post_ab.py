#!/usr/bin/env python3
"""A/B of the default-radius stencil stages on the configs[3] map (1200 x 1200): wall time per call with the engine option
dbg_post selecting the kernel, and the layers of both compared bit for bit.
    python scripts/post_ab.py [c4] [--iters 50]
dbg_post: 0 = shipped; 32 = fusion with the 64-bit integer samples (round 2); 64 = features with the two-instruction
insertion chains (round 2)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="c4")
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
wl = synth.make(a.workload)
res = bench.Resident(wl, 0)
for k in range(12):
    res.step(k)
res.eng.sync()
eng = res.eng
saved = {n: eng.layer(n).copy() for n in ("upper_bound", "lower_bound")}


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def restore():
    for n, v in saved.items():
        eng.set_layer(n, v)


out = {}
for tag, opt in (("fusion_f64", 0), ("fusion_f64_branches", 128), ("fusion_u64", 32)):
    eng.set_option("dbg_post", opt)
    restore()
    eng.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3)
    out[tag] = {n: eng.layer(n).copy() for n in saved}
    restore()
    t = timed(lambda: eng.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3), a.iters)
    print(json.dumps({"stage": tag, "dbg_post": opt, "ms": round(t * 1e3, 4)}))
same = all(np.array_equal(out["fusion_f64"][n].view(np.uint32), out["fusion_u64"][n].view(np.uint32)) for n in saved)
print(json.dumps({"fusion_layers_identical": bool(same)}))
FEAT = ("step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z")
for tag, opt in (("features_med3", 0), ("features_med3_copy_first", 256), ("features_minmax", 64)):
    eng.set_option("dbg_post", opt)
    restore()
    eng.apply_feature_extraction(0.3, 4, 0.05, 0.95)
    out[tag] = {n: eng.layer(n).copy() for n in FEAT}
    t = timed(lambda: eng.apply_feature_extraction(0.3, 4, 0.05, 0.95), a.iters)
    print(json.dumps({"stage": tag, "dbg_post": opt, "ms": round(t * 1e3, 4)}))
same = all(np.array_equal(out["features_med3"][n].view(np.uint32), out["features_minmax"][n].view(np.uint32)) for n in FEAT)
print(json.dumps({"feature_layers_identical": bool(same)}))
