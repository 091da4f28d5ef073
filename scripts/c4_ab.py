#!/usr/bin/env python3
"""A/B of engine options on configs[3] (2 M-point scans, fused k_tupdate_tbin), same box, interleaved repetitions.
usage: c4_ab.py [--lib=path/libfdm_engine_x.so] "opt=val,opt=val" "opt=val" ...   ("" = defaults)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import capi, synth
args = sys.argv[1:]
if args and args[0].startswith("--lib="):  # another build of the engine (make -C fastdem_amd/csrc variant NAME=x)
    capi.LIB_PATH = args.pop(0)[6:]
import bench

variants = args or [""]
wl = synth.lidar128(n_scans=9)
res = {}
for rep in range(3):
    for v in variants:
        r = bench.Resident(wl, 0)
        for kv in [x for x in v.split(",") if x]:
            r.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        w, _ = r.batch(0, 100)
        assert r.eng.integrate_device_batch_timed(w) == 0
        b, pts = r.batch(100, 500)
        assert r.eng.integrate_device_batch_timed(b) == 0
        res.setdefault(v or "default", []).append(round(r.eng.timer_ms() / 500 * 1e3, 2))
        del r
print(json.dumps(res))
