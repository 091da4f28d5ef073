#!/usr/bin/env python3
"""Where a block of the fused large-scan launch (k_tupdate_tbin) spends its time: three intermediate stamps per block
(measurement build, `make -C fastdem_amd/csrc phases`).   python scripts/phases_tiled.py [c4|c5] [key=val ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from fastdem_amd import capi  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "fastdem_amd", "lib", "libfdm_engine_phases.so")
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 and "=" not in sys.argv[1] else "c4"
wl = synth.make(w, n_scans=9)
res = bench.Resident(wl, 0)
res.eng.set_option("dbg_timeline", 1)
for kv in sys.argv[1:]:
    if "=" in kv:
        res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for k0, cnt in ((0, 40), (40, 12)):
    arr, _ = res.batch(k0, cnt)
    assert res.eng.integrate_device_batch(arr) == 0
    res.eng.sync()
t, nu = res.eng.debug_timeline()
t = t.astype(np.uint64)
wd = t[:, 1]
t0 = t[:, 0].min()
s = (t[:, 0] - t0).astype(np.float64) / 100.0
f = lambda sh: ((wd >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0
end, p0, p1, p2 = f(0), f(16), f(32), f(48)
def q(x): return [round(float(v), 2) for v in np.percentile(x, [10, 50, 90, 99])] if len(x) else []
def block(idx, names, mask=None):
    if mask is not None:
        idx = idx[mask[idx]]
    if len(idx) == 0:
        return {"n": 0}
    return {"n": int(len(idx)), "start": q(s[idx]), names[0]: q(p0[idx]), names[1]: q(p1[idx] - p0[idx]),
            names[2]: q(p2[idx] - p1[idx]), names[3]: q(end[idx] - p2[idx]), "dur": q(end[idx]),
            "last_end": round(float((s[idx] + end[idx]).max()), 2)}
U, B = np.arange(0, nu), np.arange(nu, len(t))
un = ["rt1_rows", "records_fold", "cells", "untouched_stores"]
out = {"workload": w, "blocks": int(len(t)), "update_groups": int(nu), "span_us": round(float((s + end).max()), 2),
       "update_all": block(U, un),
       "update_heavy": block(U, un, end > np.percentile(end[U], 90)) if nu else {},
       "update_light": block(U, un, end < np.percentile(end[U], 30)) if nu else {},
       "bin": block(B, ["init_candidate", "loads_transforms_index", "fold_compact", "flush"]),
       "bin_first_round": block(B, ["init_candidate", "loads_transforms_index", "fold_compact", "flush"], s < 2.0),
       "bin_late": block(B, ["init_candidate", "loads_transforms_index", "fold_compact", "flush"], s > 12.0)}
bn = ["init_candidate", "loads_transforms_index", "fold_compact", "flush"]
span = float((s + end).max())
if span > 60.0:  # a tile batch (fdm_tbatch.hpp): the bin blocks by when they start
    out["bin_by_start"] = {f"{int(a)}-{int(b)}us": block(B, bn, (s >= a) & (s < b))
                           for a, b in ((0, 2), (2, 30), (30, 60), (60, 90), (90, 120), (120, 150), (150, 1e9))}
print(json.dumps(out))
