#!/usr/bin/env python3
"""Long soak of the engine against the ORACLE: the generator of tests/test_long_horizon_gpu.py, more seeds, more scans.
    python scripts/soak_oracle.py [seconds] [first seed] [scans per seed] [tiled_all]
Every seed is a fresh engine / oracle pair; obstacle + elevation compared behind every call, every layer every 50 scans.
Prints one JSON line: seeds run, scans, the first failure (seed + message) if any."""
import json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fastdem_amd as gpu
import fdm_ref_py as R
import test_long_horizon_gpu as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
per = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
if len(sys.argv) > 4 and sys.argv[4] == "tiled_all":
    gpu.Engine.default_options = {"tiled_min": 1, "ray_large_min": 1}
gpu.capi.load()
R.load()
t0 = time.perf_counter()
out = {"script": "scripts/soak_oracle.py", "variant": "tiled_all" if gpu.Engine.default_options else "default", "scans_per_seed": per,
       "first_seed": seed, "seeds": 0, "scans": 0, "failure": None}
while time.perf_counter() - t0 < budget:
    T.N_SCANS[seed] = per
    base = {"tiled_min": 1, "ray_large_min": 1} if out["variant"] == "tiled_all" else {}
    gpu.Engine.default_options = dict(base, batch_max=32) if seed % 2 else base   # (every other seed: 32 scans per launch also with Kalman)
    try:
        T.test_thousands_of_scans_against_the_oracle(gpu, R, seed)
    except Exception as e:  # noqa: BLE001
        out["failure"] = {"seed": seed, "error": f"{type(e).__name__}: {str(e)[:400]}", "trace": traceback.format_exc()[-1500:]}
        break
    out["seeds"] += 1
    out["scans"] += per
    seed += 1
out["seconds"] = round(time.perf_counter() - t0, 1)
print(json.dumps(out))
