#!/usr/bin/env python3
"""A/B harness: event-timed k_bin / k_update for a workload under engine options.
   python scripts/ab_kernels.py c4 --opt wave_merge=0,1 --order azimuth,ring"""
import argparse, itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fastdem_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload")
ap.add_argument("--opt", action="append", default=[])     # key=v1,v2
ap.add_argument("--order", default="azimuth")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
opts = [(o.split("=")[0], [int(v) for v in o.split("=")[1].split(",")]) for o in a.opt]
for order in a.order.split(","):
    kw = {"order": order} if a.workload in ("c2", "c4") else {}
    wl = synth.make(a.workload, **kw)
    res = bench.Resident(wl, 0)
    k = 0
    for _ in range(10):
        res.step(k); k += 1
    for rnd in range(a.rounds):
        for combo in itertools.product(*[v for _, v in opts]) if opts else [()]:
            for (name, _), v in zip(opts, combo):
                res.eng.set_option(name, v)
            kern, roof = bench.measure_kernels(res, k, a.steps)
            k += a.steps
            print(json.dumps({"order": order, "opts": dict(zip([n for n, _ in opts], combo)),
                              "bin_us": round(kern["k_bin"]["ms"] * 1e3, 2),
                              "upd_us": round(kern["k_update"]["ms"] * 1e3, 2),
                              "bin_GBps": round(kern["k_bin"]["GBps"], 1),
                              "upd_GBps": round(kern["k_update"]["GBps"], 1),
                              "touched": kern["touched_cells_per_scan"]}), flush=True)
    del res
