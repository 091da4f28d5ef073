#!/usr/bin/env python3
"""Round-6 probe at configs[3]: engine options x scan sizes, same box, interleaved repetitions.
usage: r06_probe.py [--lib=path] [--naz=16384,12288] "opt=val,opt=val" ...   ("" = defaults)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from fastdem_amd import capi, synth
args = sys.argv[1:]
naz = [16384]
while args and args[0].startswith("--"):
    a = args.pop(0)
    if a.startswith("--lib="):
        capi.LIB_PATH = a[6:]
    elif a.startswith("--naz="):
        naz = [int(v) for v in a[6:].split(",")]
import bench

variants = args or [""]
res = {}
for n in naz:
    wl = synth.lidar128(n_scans=9, n_az=n)
    for rep in range(3):
        for v in variants:
            r = bench.Resident(wl, 0)
            for kv in [x for x in v.split(",") if x]:
                r.eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
            w, _ = r.batch(0, 100)
            assert r.eng.integrate_device_batch_timed(w) == 0
            b, pts = r.batch(100, 500)
            assert r.eng.integrate_device_batch_timed(b) == 0
            res.setdefault(f"{n}:{v or 'default'}", []).append(round(r.eng.timer_ms() / 500 * 1e3, 2))
            del r
print(json.dumps(res))
