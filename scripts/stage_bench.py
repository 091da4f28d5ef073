#!/usr/bin/env python3
"""Device time of the stages either side of the hot path (SURVEY.md §8 f3 egress, f4 ingest), with the
algorithmic bytes each moves and the CPU oracle beside it.   python scripts/stage_bench.py [c2|c4]
One JSON line per stage.  Timing: HIP events on the engine's... the entry points are synchronous
(they return counts), so wall time of the call with inputs resident in HBM is reported, plus the
kernel-only time from rocprofv3 when run under scripts/prof_stages.sh."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402
from cloud2 import make_blob  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="c4")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cpu-iters", type=int, default=3)
a = ap.parse_args()
wl = synth.make(a.workload)
res = bench.Resident(wl, 0)
for k in range(12):
    res.step(k)
res.eng.sync()
eng = res.eng
HBM = 8.0e12


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


# ---- egress: toPointCloud2 compaction + packing, records stay in HBM ----
d_ptr, n_pts, step = eng.pack_cloud_device()
t = timed(lambda: eng.pack_cloud_device(), a.iters)
cells = eng.rows * eng.cols
nf = step // 4
alg = cells * 4 * 2 + n_pts * ((nf - 2) * 4 + step)  # elevation read by count+fill, layers read, records written
import fdm_ref_py as R  # noqa: E402  (CPU baseline only)
ref = R.RefEngine(wl.width, wl.height, wl.resolution, wl.apply_to(R.default_config()))
for name in eng.layers():
    ref.set_layer(name, eng.layer(name))
g = eng.geometry()
ref.set_position(g.position_x, g.position_y)
ref.set_start_index(g.start_row, g.start_col)
t0 = time.perf_counter()
for _ in range(a.cpu_iters):
    f_ref, s_ref, d_ref = ref.pack_cloud()
t_cpu = (time.perf_counter() - t0) / a.cpu_iters
assert d_ref.shape[0] == n_pts and s_ref == step
print(json.dumps({"stage": "egress_pack_cloud", "workload": a.workload, "cells": cells, "points_out": n_pts,
                  "point_step": step, "gpu_ms": round(t * 1e3, 4), "algorithmic_MB": round(alg / 1e6, 2),
                  "GBps": round(alg / t / 1e9, 1), "hbm_frac": round(alg / t / HBM, 4),
                  "cpu_oracle_ms": round(t_cpu * 1e3, 2), "speedup": round(t_cpu / t, 1)}))

# ---- ingest: PointCloud2 blob (32-byte records, x y z _ intensity ring time) already in HBM ----
s = wl.scans[0]
blob, lay = make_blob(s["x"], s["y"], s["z"], intensity=s["intensity"] if s["intensity"] is not None else np.zeros_like(s["x"]),
                      offsets=dict(x=0, y=4, z=8, intensity=16), point_step=32)
n = s["x"].size
d_blob = torch.from_numpy(np.ascontiguousarray(blob)).cuda()
layout = eng.cloud2_layout(lay.point_step, lay.off_x, lay.off_y, lay.off_z, lay.off_intensity, lay.intensity_type, lay.off_rgb)
import ctypes as C  # noqa: E402
nv = C.c_uint64(0)


def ingest():
    rc = eng._lib.fdm_engine_ingest_cloud2(eng._h, C.c_void_p(d_blob.data_ptr()), 1, n, C.byref(layout), C.byref(nv))
    assert rc == 0


t = timed(ingest, a.iters)
alg = n * 32 + n * 12 + nv.value * 16  # records read (count pass reads xyz only), SoA written
t0 = time.perf_counter()
for _ in range(a.cpu_iters):
    c_ref = R.from_cloud2(blob, n, lay)
t_cpu = (time.perf_counter() - t0) / a.cpu_iters
assert c_ref["x"].size == nv.value
print(json.dumps({"stage": "ingest_cloud2", "workload": a.workload, "points": n, "kept": nv.value, "point_step": 32,
                  "gpu_ms": round(t * 1e3, 4), "algorithmic_MB": round(alg / 1e6, 2), "GBps": round(alg / t / 1e9, 1),
                  "hbm_frac": round(alg / t / HBM, 4), "cpu_oracle_ms": round(t_cpu * 1e3, 2),
                  "speedup": round(t_cpu / t, 1)}))

# ---- message -> map in one call (from_impl + integrate, fdm_engine_integrate_cloud2): one decode kernel +
# the scan, no host round trip in between; blob in HBM / pinned message decoded over PCIe / pageable message
import fastdem_amd  # noqa: E402
eng2 = bench.Resident(wl, 0).eng
tbs = bench.colmajor16(wl.T_base_sensor)
twb = bench.colmajor16(wl.pose(0))
st2 = fastdem_amd.capi.FdmScanStats()
pin_msg = fastdem_amd.host_array(np.frombuffer(blob, dtype=np.uint8), dtype=np.uint8)
page_msg = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8)).copy()


def msg_call(ptr, on_dev):
    def f():
        rc = eng2._lib.fdm_engine_integrate_cloud2(eng2._h, C.c_void_p(ptr), on_dev, n, C.byref(layout), tbs, twb, C.byref(st2))
        assert rc == 0
    return f


t_hbm = timed(msg_call(d_blob.data_ptr(), 1), a.iters)
t_pin = timed(msg_call(pin_msg.array.ctypes.data, 0), a.iters)
t_page = timed(msg_call(page_msg.ctypes.data, 0), a.iters)
t0 = time.perf_counter()
for _ in range(a.cpu_iters):
    ref.integrate_cloud2(blob, n, lay, wl.T_base_sensor, wl.pose(0))
t_cpu = (time.perf_counter() - t0) / a.cpu_iters
print(json.dumps({"stage": "integrate_cloud2 (message -> map, synchronous)", "workload": a.workload, "points": n,
                  "point_step": 32, "gpu_ms_blob_in_hbm": round(t_hbm * 1e3, 4),
                  "gpu_ms_pinned_message": round(t_pin * 1e3, 4), "gpu_ms_pageable_message": round(t_page * 1e3, 4),
                  "cpu_oracle_ms": round(t_cpu * 1e3, 2), "speedup_pinned": round(t_cpu / t_pin, 1)}))

# ---- stencil post-processing on the mapped scene ----
for name, fn_gpu, fn_cpu, alg_bytes in (
        ("post_inpainting(3 passes)", lambda: eng.apply_inpainting(3, 2), lambda: ref.apply_inpainting(3, 2), cells * 4 * 2 * 4),
        ("post_median_3x3", lambda: eng.apply_spatial_smoothing("elevation_inpainted", 3, 5),
         lambda: ref.apply_spatial_smoothing("elevation_inpainted", 3, 5), cells * 4 * 3),
        ("post_uncertainty_fusion(r=0.15)", lambda: eng.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3),
         lambda: ref.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3), cells * 4 * 6),
        ("post_feature_extraction(r=0.3)", lambda: eng.apply_feature_extraction(0.3, 4, 0.05, 0.95),
         lambda: ref.apply_feature_extraction(0.3, 4, 0.05, 0.95), cells * 4 * 8)):
    t = timed(fn_gpu, max(3, a.iters // 4))
    t0 = time.perf_counter()
    fn_cpu()
    t_cpu = time.perf_counter() - t0
    print(json.dumps({"stage": name, "workload": a.workload, "cells": cells, "gpu_ms": round(t * 1e3, 4),
                      "algorithmic_MB": round(alg_bytes / 1e6, 2), "GBps": round(alg_bytes / t / 1e9, 1),
                      "cpu_oracle_ms": round(t_cpu * 1e3, 2), "speedup": round(t_cpu / t, 1)}))

# ---- the big neighbourhoods (k_median_sel / k_fusion_wave / k_features_sel; beyond the LDS: the pooled k_*_big), no CPU
# beside them; they exist so that no call the reference accepts is refused ----
res_cells = eng.rows * eng.cols
for name, fn_gpu, entries in (
        ("post_median_17x17 (k_median_sel)", lambda: eng.apply_spatial_smoothing("elevation_inpainted", 17, 5), 17 * 17),
        ("post_uncertainty_fusion(r=10 cells: 317 entries, k_fusion_wave)",
         lambda: eng.apply_uncertainty_fusion(True, 10.0 * wl.resolution + 1e-4, 0.05, 0.01, 0.99, 3), 317),
        ("post_feature_extraction(r=10 cells: 317 entries, k_features_sel)",
         lambda: eng.apply_feature_extraction(10.0 * wl.resolution + 1e-4, 4, 0.05, 0.95), 317)):
    t = timed(fn_gpu, 2)
    print(json.dumps({"stage": name, "workload": a.workload, "cells": res_cells, "entries_per_cell": entries,
                      "gpu_ms": round(t * 1e3, 3), "ns_per_cell": round(t * 1e9 / res_cells, 1)}))
