#!/usr/bin/env python3
"""Long soak of the engine against the ORACLE: the generator of tests/test_long_horizon_gpu.py, more seeds, more scans.
    python scripts/soak_oracle.py [seconds] [first seed] [scans per seed] [tiled_all] [extra]
`key=value`: engine options for every engine (e.g. ray_overlap=1 voxel_small=0: two raycasting stages in flight for every scan).
`extra`: between the calls, with a generator of its own, also GridMap::move() explicitly, ElevationMapping::update() directly
(no transforms, no crops; with and without a per-point variance channel), inpainting, median smoothing and uncertainty fusion
(they write layers the next scans and compares see).
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

import numpy as np
F32 = np.float32

def install_extra(seed):
    rng = np.random.default_rng(900000 + seed)
    LAST = {}

    def hook(call, eng, ref, scans, poses, Tbs):
        pick = int(rng.integers(0, 12))
        if T.TRACE is not None:
            T.TRACE.append({"extra_before_call": call, "pick": pick})
        if pick == 0:      # an explicit move (GridMap::move through the C ABI), a few cells or many
            g = ref.geometry()
            x, y = g.position_x + float(rng.uniform(-2.0, 2.0)), g.position_y + float(rng.uniform(-2.0, 2.0))
            eng.move(x, y)
            ref.move(x, y)
        elif pick == 1:    # ElevationMapping::update directly: map-frame points, no crops (elevation_mapping.cpp:110-125)
            g = ref.geometry()
            n = int(rng.integers(1, 3000))
            x = (g.position_x + rng.uniform(-9.0, 9.0, n)).astype(F32)
            y = (g.position_y + rng.uniform(-9.0, 9.0, n)).astype(F32)
            z = rng.uniform(-1.0, 1.0, n).astype(F32)
            var = rng.uniform(0.0, 0.02, n).astype(F32) if int(rng.integers(0, 2)) else None
            a = rng.uniform(0, 1, n).astype(F32)
            robot = (g.position_x + float(rng.uniform(-0.5, 0.5)), g.position_y + float(rng.uniform(-0.5, 0.5)))
            se = eng.update(x, y, z, robot, z_var=var, intensity=a)
            sr = ref.update(x, y, z, robot, z_var=var, intensity=a)
            assert se == sr, (se, sr)
        elif pick == 2 and ref.exists("elevation"):
            a = (int(rng.integers(1, 4)), int(rng.integers(1, 4)), bool(rng.integers(0, 2)))
            LAST["args"] = list(a)
            for o in (eng, ref):
                o.apply_inpainting(*a)
        elif pick == 3 and ref.exists("elevation"):
            k = int(rng.choice([3, 5]))
            a = ("elevation", k, int(rng.integers(1, 6)))
            LAST["args"] = list(a)
            for o in (eng, ref):
                o.apply_spatial_smoothing(*a)
        elif pick == 4 and ref.exists("upper_bound"):
            for o in (eng, ref):
                o.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3)
        if T.TRACE is not None and pick <= 4:   # (the replay, scripts/soak_oracle_repro.py: every layer right behind the operation)
            T.TRACE[-1]["args"] = LAST.get("args")
            mode = os.environ.get("EXTRA_CHECK", "compare")
            if mode == "compare":
                T.compare(eng, ref, f"behind extra operation {pick} before call {call}", None)
            elif mode == "sync":
                eng.sync()
            elif mode == "geometry":
                eng.geometry()
        return False
    T.HOOK = hook


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
    if "tiled_all" in sys.argv[4:]:
        gpu.Engine.default_options = {"tiled_min": 1, "ray_large_min": 1}
    EXTRA = "extra" in sys.argv[4:]
    gpu.capi.load()
    R.load()
    t0 = time.perf_counter()
    out = {"script": "scripts/soak_oracle.py" + (" extra" if "extra" in sys.argv[4:] else ""), "variant": "tiled_all" if gpu.Engine.default_options else "default", "scans_per_seed": per,
           "first_seed": seed, "seeds": 0, "scans": 0, "failure": None}
    while time.perf_counter() - t0 < budget:
        T.N_SCANS[seed] = per
        base = {"tiled_min": 1, "ray_large_min": 1} if out["variant"] == "tiled_all" else {}
        base.update({a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:] if "=" in a})   # (e.g. ray_overlap=1 voxel_small=0)
        gpu.Engine.default_options = dict(base, batch_max=32) if seed % 2 else base   # (every other seed: 32 scans per launch also with Kalman)
        if EXTRA:
            install_extra(seed)
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


if __name__ == "__main__":
    main()
