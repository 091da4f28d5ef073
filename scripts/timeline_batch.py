#!/usr/bin/env python3
"""Block timeline of the batch launch (k_mbatch = update | bin | crop): who runs when.
   python scripts/timeline_batch.py [key=val ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

BM = 16  # scans per launch (the engine's default for Kalman; batch_max=N on the command line, up to kMaxBatch = 32)
for kv in sys.argv[1:]:
    if kv.startswith("batch_max="):
        BM = int(kv.split("=")[1])
wl = synth.make("c2", n_scans=8)
res = bench.Resident(wl, 0)
res.eng.set_option("dbg_timeline", 1)
for kv in sys.argv[1:]:
    if kv == "raycast=1":  # the k_mbatch variant whose update half resolves ray events (fdm_rbatch.hpp ran in between)
        cfg = res.eng.cfg
        cfg.raycast_enabled = 1
        res.eng.set_config(cfg)
        continue
    res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for kk in range(30 * BM):
    res.pose(kk)
arr, _ = res.batch(0, 10 * BM)
assert res.eng.integrate_device_batch(arr) == 0
res.eng.sync()
arr2, _ = res.batch(10 * BM, 5 * BM)
assert res.eng.integrate_device_batch(arr2) == 0   # 5 batches; the timeline holds the LAST launch with a bin half: update 3 | bin 4
t, gx = res.eng.debug_timeline()   # rows of gx blocks: row 0 update tiles, then one row per scan (bin), then crop rows
t = t.astype(np.int64)
live = t[:, 1] > 0
t0 = t[live, 0].min()
s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
d = e - s
nrow = len(t) // gx
def q(x): return [round(float(v), 2) for v in np.percentile(x, [0, 10, 50, 90, 99, 100])] if len(x) else []
out = {"grid": [int(gx), int(nrow)], "span_us": round(float(e[live].max()), 2)}
nb = (28800 + 511) // 512
def role(rows, width):
    idx = np.concatenate([np.arange(r * gx, r * gx + width) for r in rows]) if rows else np.array([], dtype=int)
    idx = idx[live[idx]] if len(idx) else idx
    return {"n": int(len(idx)), "start": q(s[idx]), "end": q(e[idx]), "dur": q(d[idx])}
cpb = 64 if BM <= 16 else 32   # cells per update block (fdm_multi.hpp upd_cells_per_block)
nu = (22500 + cpb - 1) // cpb
ur = (nu + gx - 1) // gx
upd_idx = np.arange(0, nu)
out["update"] = {"n": int(nu), "start": q(s[upd_idx]), "end": q(e[upd_idx]), "dur": q(d[upd_idx])}
out["bin"] = role(list(range(ur, ur + BM)), nb)
out["crop"] = role(list(range(ur + BM, nrow)), nb)
out["bin_by_scan"] = [{"k": k, "start50": round(float(np.median(s[(ur + k) * gx:(ur + k) * gx + nb])), 2),
                       "end50": round(float(np.median(e[(ur + k) * gx:(ur + k) * gx + nb])), 2)} for k in range(BM)]
print(json.dumps(out))
