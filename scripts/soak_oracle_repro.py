#!/usr/bin/env python3
"""Replay ONE seed of the oracle soak (scripts/soak_oracle.py) with diagnostics: what the last calls looked like and which cells
of which layers differ at the first mismatch.
    python scripts/soak_oracle_repro.py <seed> [scans] [tiled_all] [key=val ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import fastdem_amd as gpu
import fdm_ref_py as R
import test_long_horizon_gpu as T

seed = int(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
opts = {}
for a in sys.argv[3:]:
    if a == "tiled_all":
        opts.update({"tiled_min": 1, "ray_large_min": 1})
    elif a == "extra":
        pass
    elif "=" in a:
        opts[a.split("=")[0]] = int(a.split("=")[1])
if seed % 2 and "batch_max" not in opts:
    opts["batch_max"] = 32
gpu.Engine.default_options = opts
gpu.capi.load()
R.load()
T.N_SCANS[seed] = per
T.TRACE = []
if "extra" in sys.argv[3:]:   # (the soak's extra operations between the calls: the same generator)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import soak_oracle
    soak_oracle.install_extra(seed)
state = {}
orig = T.compare


def compare(eng, ref, what, names=None):
    state["eng"], state["ref"], state["what"] = eng, ref, what
    orig(eng, ref, what, None)   # (EVERY layer behind every call: the first divergence may be in a hidden layer)


T.compare = compare
# --at <call> <row> <col>: replay that call scan by scan (synchronous integrate on both sides) and print the cell's layers
# after every scan in which the two sides differ anywhere in the cell
if "--at" in sys.argv:
    k = sys.argv.index("--at")
    at_call, at_r, at_c = int(sys.argv[k + 1]), int(sys.argv[k + 2]), int(sys.argv[k + 3])

    def cell(o):
        return {n: float(o.layer(n)[at_r, at_c]) for n in sorted(o.layers())}

    def hook(call, eng, ref, scans, poses, Tbs):
        if call != at_call:
            return False
        eng.sync()
        print(json.dumps({"before_call": call, "engine": cell(eng), "oracle": cell(ref), "estimator": int(eng.cfg.estimation_type)}))
        for i, (s_, Twb) in enumerate(zip(scans, poses)):
            kw = {"intensity": s_["intensity"]}
            if s_["rgb"] is not None:
                kw["rgb"] = s_["rgb"]
            re_, se = eng.integrate(s_["x"], s_["y"], s_["z"], Tbs, Twb, **kw)
            rr, sr = ref.integrate(s_["x"], s_["y"], s_["z"], Tbs, Twb, **kw)
            ce, cr = cell(eng), cell(ref)
            same = all((np.isnan(ce[n]) and np.isnan(cr[n])) or ce[n] == cr[n] for n in cr)
            ge = eng.geometry()
            print(json.dumps({"scan": i, "n": int(s_["x"].size), "stats_e": se, "stats_r": sr, "same": same, "start": [ge.start_row, ge.start_col],
                              "engine": {n: ce[n] for n in ce if n.startswith("_p2_q") or n in ("n_points", "elevation")},
                              "oracle": {n: cr[n] for n in cr if n.startswith("_p2_q") or n in ("n_points", "elevation")}}))
        return True
    prev_hook = T.HOOK

    def both_hooks(call, eng, ref, scans, poses, Tbs):
        if prev_hook is not None:
            prev_hook(call, eng, ref, scans, poses, Tbs)
        return hook(call, eng, ref, scans, poses, Tbs)
    T.HOOK = both_hooks
try:
    T.test_thousands_of_scans_against_the_oracle(gpu, R, seed)
    print(json.dumps({"seed": seed, "ok": True, "calls": len(T.TRACE)}))
except AssertionError as e:
    eng, ref = state["eng"], state["ref"]
    out = {"seed": seed, "ok": False, "error": str(e)[:300], "where": state["what"], "options": opts, "last_calls": T.TRACE[-6:]}
    eng.sync()
    ge, gr = eng.geometry(), ref.geometry()
    out["geometry"] = {"engine": [ge.position_x, ge.position_y, ge.start_row, ge.start_col], "oracle": [gr.position_x, gr.position_y, gr.start_row, gr.start_col]}
    diffs = {}
    for n in ref.layers():
        if not eng.exists(n):
            diffs[n] = "missing in the engine"
            continue
        a, b = eng.layer(n), ref.layer(n)
        same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        if not same.all():
            bad = np.argwhere(~same)
            diffs[n] = {"cells": int((~same).sum()), "first": bad[:6].tolist(), "engine": a[~same][:6].tolist(), "oracle": b[~same][:6].tolist(),
                        "rows": [int(bad[:, 0].min()), int(bad[:, 0].max())], "cols": [int(bad[:, 1].min()), int(bad[:, 1].max())]}
    out["diffs"] = diffs
    print(json.dumps(out))
