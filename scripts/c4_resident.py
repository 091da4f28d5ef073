#!/usr/bin/env python3
"""configs[3] with 9 / 2 / 1 distinct scans cycling through the call: 302 MB of input streams from HBM, 67 / 34 MB stay in the
Infinity Cache — the upper bound of what prefetching the next scan's points could buy (round 5: 31.0 / 28.2 / 28.2 us)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdem_amd import capi, synth
import bench
out = {}
for ns in (9, 2, 1):
    wl = synth.lidar128(n_scans=ns)
    for rep in range(2):
        r = bench.Resident(wl, 0)
        w, _ = r.batch(0, 100)
        assert r.eng.integrate_device_batch_timed(w) == 0
        b, pts = r.batch(100, 500)
        assert r.eng.integrate_device_batch_timed(b) == 0
        out.setdefault(str(ns), []).append(round(r.eng.timer_ms() / 500 * 1e3, 2))
        del r
print(json.dumps(out))
