#!/usr/bin/env python3
"""Where a k_mbatch block spends its time: three intermediate stamps per block (measurement build,
`make -C fastdem_amd/csrc phases`).   python scripts/phases_batch.py [batch_max] [key=val ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from fastdem_amd import capi  # noqa: E402
capi.LIB_PATH = os.path.join(ROOT, "fastdem_amd", "lib", "libfdm_engine_phases2.so" if "variant=2" in sys.argv else "libfdm_engine_phases.so")
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

bm = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
WL = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("workload=")), "c2")
wl = synth.make(WL, n_scans=8)
res = bench.Resident(wl, 0)
res.eng.set_option("dbg_timeline", 1)
res.eng.set_option("batch_max", bm)
if "raycast=1" in sys.argv:  # the variant whose update half resolves ray events
    cfg = res.eng.cfg
    cfg.raycast_enabled = 1
    res.eng.set_config(cfg)
for kv in sys.argv[1:]:
    if "=" in kv and not kv.startswith("variant") and not kv.startswith("workload") and not kv.startswith("raycast"):
        res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for kk in range(400):
    res.pose(kk)
arr, _ = res.batch(0, 10 * bm)
assert res.eng.integrate_device_batch(arr) == 0
res.eng.sync()
arr2, _ = res.batch(10 * bm, 5 * bm)   # last bin launch = update 3 | bin 4 | no crop; the one before it has a crop half
assert res.eng.integrate_device_batch(arr2) == 0
t, gx = res.eng.debug_timeline()
t = t.astype(np.uint64)
w = t[:, 1]
live = w > 0
t0 = t[live, 0].min()
s = (t[:, 0] - t0).astype(np.float64) / 100.0
end = (w & np.uint64(0xFFFF)).astype(np.float64) / 100.0
p0 = ((w >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0
p1 = ((w >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0
p2 = ((w >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.float64) / 100.0
nb = (wl.n_points + 511) // 512
cells = 64
nu = (int(res.eng.rows) * int(res.eng.cols) + cells - 1) // cells
ur = (nu + gx - 1) // gx
def q(x): return [round(float(v), 2) for v in np.percentile(x, [10, 50, 90])] if len(x) else []
def block(idx, names):
    idx = idx[live[idx]]
    return {"n": int(len(idx)), "start": q(s[idx]), names[0]: q(p0[idx]), names[1]: q(p1[idx] - p0[idx]),
            names[2]: q(p2[idx] - p1[idx]), names[3]: q(end[idx] - p2[idx]), "dur": q(end[idx]),
            "span_end": round(float((s[idx] + end[idx]).max()), 2)}
upd = np.arange(0, nu)
binr = np.concatenate([np.arange((ur + k) * gx, (ur + k) * gx + nb) for k in range(bm)])
names = ["init_chain", "loads_transforms", "index_fold", "compact_merge_flush"]
if os.path.basename(capi.LIB_PATH).endswith("phases2.so"):
    names = ["kernargs_issue", "loads_back", "chain", "rest"]
out = {"batch_max": bm, "grid": [int(gx), int(len(t) // gx)],
       "update": block(upd, ["rt1_keys", "rt2_obs", "apply", "stores"]),
       "bin": block(binr, names)}
for k in (0, 5, 10, 15):
    if k < bm:
        out[f"bin_row_{k}"] = block(np.arange((ur + k) * gx, (ur + k) * gx + nb), names)
print(json.dumps(out))
