#!/usr/bin/env python3
"""How much of the large-scan launch is the sensor model's sigma_z^2 (evaluated once per (bin block, cell) winner)?
configs[3] with the LiDAR model (as benchmarked) against the constant model (no covariance arithmetic).  Measurement only."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import synth
import bench
wl = synth.lidar128(n_scans=9)
res = {}
for rep in range(2):
    for st in (1, 0):
        r = bench.Resident(wl, 0)
        cfg = r.eng.cfg
        cfg.sensor_type = st
        r.eng.set_config(cfg)
        w, _ = r.batch(0, 100)
        assert r.eng.integrate_device_batch_timed(w) == 0
        b, _ = r.batch(100, 500)
        assert r.eng.integrate_device_batch_timed(b) == 0
        res.setdefault("lidar_model" if st == 1 else "constant_model", []).append(round(r.eng.timer_ms() / 500 * 1e3, 2))
        del r
print(json.dumps(res))
