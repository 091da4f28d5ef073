#!/usr/bin/env python3
"""Block timeline of the fused large-scan launch (k_tupdate_tbin): who runs when.
   python scripts/timeline.py c4 [--set key=val ...]
Streams the workload's scans with the engine option dbg_timeline on, then reads the per-block start / end
ticks (100 MHz) of the last launch and prints: launch span, when the update groups start / end (by duration
class), when the bin blocks start / end, and the occupancy of the chip over time by kind."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--set", action="append", default=[])
ap.add_argument("--scans", type=int, default=9)
ap.add_argument("--out", default="")
a = ap.parse_args()
wl = synth.make(a.workload, n_scans=a.scans)
res = bench.Resident(wl, 0)
res.eng.set_option("dbg_timeline", 1)
for kv in a.set:
    res.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for k0, cnt in ((0, 40), (40, 12)):
    arr, _ = res.batch(k0, cnt)
    rc = res.eng.integrate_device_batch(arr)
    assert rc == 0, rc
    res.eng.sync()
t, nu = res.eng.debug_timeline()
t = t.astype(np.int64)
t0 = t[:, 0].min()
s, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # us
d = e - s
U, B = slice(0, nu), slice(nu, len(t))
out = {"workload": a.workload, "blocks": int(len(t)), "update_groups": int(nu), "span_us": round(float(e.max()), 2)}
def q(x): return [round(float(v), 2) for v in np.percentile(x, [0, 10, 50, 90, 99, 100])]
out["update_start_us_pct"] = q(s[U]); out["update_end_us_pct"] = q(e[U]); out["update_dur_us_pct"] = q(d[U])
out["bin_start_us_pct"] = q(s[B]); out["bin_end_us_pct"] = q(e[B]); out["bin_dur_us_pct"] = q(d[B])
# bin blocks in grid order = scan order (a tile batch: K scans back to back): duration / start by eighth of the bin grid
nb = len(t) - nu
if nb >= 8:
    out["bin_by_eighth"] = [{"dur_med": round(float(np.median(d[nu + i * nb // 8: nu + (i + 1) * nb // 8])), 2),
                             "start_med": round(float(np.median(s[nu + i * nb // 8: nu + (i + 1) * nb // 8])), 2),
                             "end_max": round(float(e[nu + i * nb // 8: nu + (i + 1) * nb // 8].max()), 2)} for i in range(8)]
grid = np.arange(0.0, float(e.max()) + 1.0, 1.0)
out["resident_by_us"] = [{"t": float(g), "update": int(((s[U] <= g) & (e[U] > g)).sum()), "bin": int(((s[B] <= g) & (e[B] > g)).sum())} for g in grid]
heavy = np.argsort(-d[U])[:8]
out["longest_update_groups"] = [{"block": int(b), "start": round(float(s[b]), 2), "dur": round(float(d[b]), 2)} for b in heavy]
print(json.dumps(out))
if a.out:
    np.save(a.out, t)
