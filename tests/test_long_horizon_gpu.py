"""Long-horizon parity: thousands of scans, engine against the ORACLE, every layer bit for bit.

Why (round 5): a 240-second engine-against-engine soak found a stale obstacle cell after 352 K scans — a bug that had lived
under "bit-exact" for two rounds because every oracle comparison of the suite stopped at <= 60 scans, and the interleaving
it needs (a pipeline switch exactly on a scan that observes nothing) is rare.  This test walks that CLASS of interleavings
against the oracle: a seeded generator of >= 3 000 scans on a small LOCAL map mixing

  * scan sizes on both sides of every pipeline threshold (record-pool pipeline `tiled_min`, the sort-free voxel filter
    `voxel_small_max`, the sector-window ray walk `ray_large_min`), lowered so that the scans stay small;
  * scans filtered away as a whole (fastdem.cpp:138: no move, no update), scans that pass the crops and miss the map
    (elevation_mapping.cpp:118: move, no obstacle clear), exact duplicates, -0.0 heights, NaN intensities;
  * poses that shift the rolling window by fractions of a cell, by many cells and by more than the map;
  * host writes between the calls: `set_layer` of the obstacle / elevation layer, `clear(layer)`, `clearAll`, a user layer;
  * estimator switches at run time (fastdem.cpp:34-38), raycasting on and off (fastdem.cpp:152-159);
  * every entry point: batch call, enqueue-only scan by scan, the synchronous host call, the pageable host batch.

Behind every call: the obstacle and elevation layers bit for bit; every 50 scans and at the end: layer names, every layer
bit for bit (NaN pattern, signs of zeros), geometry, the last scan's statistics.  Through the C ABI.  `scripts/long_horizon_prefix.py` runs this file against a build of the tree before
f53d0fc: the generator's seeds reach the round-5 bug there (LABNOTES round 6).

Run on the GPU box:  python -m pytest tests -m gpu
"""
import ctypes as C
import time

import numpy as np
import pytest

from helpers import assert_layers_bit_identical, same_geometry
from test_batch_gpu import DeviceBatch, T

pytestmark = pytest.mark.gpu
F32 = np.float32
SIZE, RES = 16.0, 0.1          # 160 x 160 cells: the oracle keeps up
TILED_MIN, VOXEL_SMALL_MAX, RAY_LARGE_MIN = 1500, 1100, 2600


def base_cfg(c):
    c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 4.0, 0.2, 14.0
    c.rc_log_odds_ghost, c.rc_clear_threshold, c.rc_height_conflict_threshold = 0.9, -0.5, 0.02
    return c


class Gen:
    """The seeded scan / pose / event generator (numpy Generator: the stream is the test's definition)."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.px = self.py = 0.0
        self.k = 0

    def size(self):
        r = self.rng
        pick = int(r.integers(0, 10))
        if pick < 4:
            return int(r.integers(1, 400))
        if pick < 6:   # around the sort-free voxel filter's limit
            return int(r.integers(VOXEL_SMALL_MAX - 150, VOXEL_SMALL_MAX + 150))
        if pick < 8:   # around the record-pool pipeline's threshold
            return int(r.integers(TILED_MIN - 200, TILED_MIN + 200))
        if pick < 9:   # around the sector-window walk's threshold
            return int(r.integers(RAY_LARGE_MIN - 300, RAY_LARGE_MIN + 300))
        return int(r.integers(3000, 7000))

    def cloud(self, n, colour):
        r = self.rng
        half = SIZE / 2 + 1.0
        if int(r.integers(0, 3)) == 0:           # the whole map and a bit more
            x = r.uniform(-half, half, n).astype(F32)
            y = r.uniform(-half, half, n).astype(F32)
        else:                                    # a patch of it: most tiles see nothing of this scan (what the round-5 bug needed)
            cx, cy, w = r.uniform(-7.0, 7.0), r.uniform(-7.0, 7.0), r.uniform(0.6, 4.0)
            x = r.uniform(cx - w, cx + w, n).astype(F32)
            y = r.uniform(cy - w, cy + w, n).astype(F32)
        z = (r.uniform(-1.0, 0.4, n) - 1.2).astype(F32)
        a = r.uniform(0, 1, n).astype(F32)
        kind = int(r.integers(0, 14))
        if kind == 0:
            z += F32(40.0)                       # every point filtered (cropZ)
        elif kind == 1:
            x[:] = r.uniform(9.5, 12.0, n).astype(F32)   # passes the crops, misses the 16 m map
        elif kind == 2 and n > 30:               # a dense cluster with exact duplicates
            m = n // 2
            x[:m] = (1.0 + r.uniform(0, 0.5, m)).astype(F32)
            y[:m] = (-2.0 + r.uniform(0, 0.5, m)).astype(F32)
            x[3:m:5], y[3:m:5], z[3:m:5] = x[2], y[2], z[2]
        elif kind == 3:
            a[::7] = np.nan
            z[::5] = F32(-1.2)                   # map-frame z = +0.0 / -0.0 ties (sensor 1.2 m up)
            z[1::10] = F32(-1.2)
        elif kind == 4:                          # tall things: obstacle cells, rays above the ground
            z[::3] += F32(1.5)
        s = {"x": x, "y": y, "z": z, "intensity": a, "rgb": None}
        if colour:
            s["rgb"] = r.integers(0, 1 << 24, n, dtype=np.uint32)
        return s

    def pose(self):
        r = self.rng
        pick = int(r.integers(0, 40))
        if pick == 0:      # a jump beyond the map: every layer cleared by the move
            self.px += float(r.choice([-1.0, 1.0])) * float(r.uniform(17.0, 30.0))
        elif pick == 1:
            self.py += float(r.choice([-1.0, 1.0])) * float(r.uniform(16.0, 16.2))   # just about the map's size
        elif pick < 6:     # many cells
            self.px += float(r.uniform(-3.0, 3.0))
            self.py += float(r.uniform(-3.0, 3.0))
        else:              # fractions of a cell to a few cells
            self.px += float(r.uniform(-0.3, 0.4))
            self.py += float(r.uniform(-0.2, 0.2))
        self.k += 1
        return T(self.px, self.py, 0.0, yaw=0.01 * self.k)


def host_event(rng, eng, ref):
    """A host write between two calls, the same on both sides."""
    pick = int(rng.integers(0, 7))
    names = eng.layers()
    if pick == 0 and "obstacle" in names:
        a = ref.layer("obstacle").copy()
        a[int(rng.integers(0, a.shape[0])), :] = F32(0.7)
        for o in (eng, ref):
            o.set_layer("obstacle", a)
    elif pick == 1 and "elevation" in names:
        a = ref.layer("elevation").copy()
        r0, c0 = int(rng.integers(0, a.shape[0] - 8)), int(rng.integers(0, a.shape[1] - 8))
        a[r0:r0 + 8, c0:c0 + 8] = F32(1.25)    # a phantom block for the rays to clear
        for o in (eng, ref):
            o.set_layer("elevation", a)
    elif pick == 2 and "obstacle" in names:
        for o in (eng, ref):
            o.clear("obstacle")
    elif pick == 3:
        for o in (eng, ref):
            o.clear()                           # clearAll
    elif pick == 4:
        for o in (eng, ref):
            if not o.exists("user"):
                o.add("user", 3.0)
    elif pick == 5 and "variance" in names:
        for o in (eng, ref):
            o.clear("variance")
    # (6: nothing)


def compare(eng, ref, what, names=None):
    eng.sync()
    assert sorted(eng.layers()) == sorted(ref.layers()), (what, eng.layers(), ref.layers())
    assert_layers_bit_identical(eng, ref, names=names)
    assert same_geometry(eng.geometry(), ref.geometry()), what


N_SCANS = {2026: 12000, 7: 6000, 31: 6000}
TRACE = None   # scripts/soak_oracle_repro.py: a list that takes one record per call (what the call looked like)
HOOK = None    # ... and a callable(call number, eng, ref, scans, poses, Tbs) -> True if it integrated the call's scans itself


@pytest.mark.parametrize("seed", sorted(N_SCANS))
def test_thousands_of_scans_against_the_oracle(gpu, R, seed):
    n_scans = N_SCANS[seed]
    ce, cr = base_cfg(gpu.capi.default_config()), base_cfg(R.default_config())
    eng = gpu.Engine(SIZE, SIZE, RES, ce)
    ref = R.RefEngine(SIZE, SIZE, RES, cr)
    if "tiled_min" not in gpu.Engine.default_options:   # (the other fixture variants force every scan through the large-scan pipelines)
        eng.set_option("tiled_min", TILED_MIN)
        eng.set_option("ray_large_min", RAY_LARGE_MIN)
    eng.set_option("voxel_small_max", VOXEL_SMALL_MAX)
    if seed == 31:
        eng.set_option("batch_max", 32)   # (Kalman takes 16 scans per launch by default: the 32-scan layout of the update half here too)
    g = Gen(seed)
    rng = g.rng
    Tbs = T(0.0, 0.0, 1.2)
    done, next_check, calls = 0, 50, 0
    launches0 = sum(eng.batch_launches())
    keep = []
    t0 = time.perf_counter()
    colour = False
    while done < n_scans:
        # ---- what this call looks like ----
        ray = int(rng.integers(0, 3) == 0)
        if int(rng.integers(0, 60)) == 0:       # estimator switch at run time (the other estimator's layers stay)
            for o in (eng, ref):
                c = o.cfg
                c.estimation_type = 1 - c.estimation_type
                o.set_config(c)
        if int(rng.integers(0, 50)) == 0:
            colour = not colour
        for o in (eng, ref):
            c = o.cfg
            c.raycast_enabled = ray
            o.set_config(c)
        if int(rng.integers(0, 4)) == 0:
            host_event(rng, eng, ref)
            if TRACE is not None:
                TRACE.append({"host_event_before_call": calls + 1})
        count = int(rng.integers(1, 40))
        scans = [g.cloud(g.size(), colour) for _ in range(count)]
        poses = [g.pose() for _ in range(count)]
        how = int(rng.integers(0, 8))
        if HOOK is not None and HOOK(calls + 1, eng, ref, scans, poses, Tbs):
            done += count
            calls += 1
            continue
        # ---- the oracle, scan by scan ----
        rc_r = st_r = None
        for s, Twb in zip(scans, poses):
            kw = {"intensity": s["intensity"]}
            if s["rgb"] is not None:
                kw["rgb"] = s["rgb"]
            rc_r, st_r = ref.integrate(s["x"], s["y"], s["z"], Tbs, Twb, **kw)
        # ---- the engine, through one of its entry points ----
        if how == 0:      # enqueue-only, one scan per call
            b = DeviceBatch(gpu, scans, Tbs, poses)
            keep.append(b)
            for k in range(count):
                one = (gpu.capi.FdmDeviceScan * 1)(b.arr[k])
                assert eng.integrate_device_batch(one) == 0
        elif how == 1:    # the synchronous host call for the first scan, the rest as a batch
            s = scans[0]
            kw = {"intensity": s["intensity"]}
            if s["rgb"] is not None:
                kw["rgb"] = s["rgb"]
            eng.integrate(s["x"], s["y"], s["z"], Tbs, poses[0], **kw)
            if count > 1:
                b = DeviceBatch(gpu, scans[1:], Tbs, poses[1:])
                keep.append(b)
                assert eng.integrate_device_batch(b.arr) == 0
        elif how == 2:    # pageable HOST arrays in one call
            harr = (gpu.capi.FdmDeviceScan * count)()
            hk = []
            for k, (s, Twb) in enumerate(zip(scans, poses)):
                hs = [np.ascontiguousarray(s[c]) for c in ("x", "y", "z", "intensity")]
                hrgb = np.ascontiguousarray(s["rgb"]) if s["rgb"] is not None else None
                hk.append((hs, hrgb))
                harr[k].n = int(s["x"].size)
                harr[k].x, harr[k].y, harr[k].z, harr[k].intensity = (h.ctypes.data for h in hs)
                harr[k].rgb = hrgb.ctypes.data if hrgb is not None else None
                harr[k].sigma_z2 = None
                harr[k].T_base_sensor = (C.c_double * 16)(*np.ascontiguousarray(Tbs.T).reshape(16))
                harr[k].T_world_base = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(Twb).T).reshape(16))
            eng.integrate_host_batch(harr)
            keep.append(hk)
        else:             # the batch call on device arrays
            b = DeviceBatch(gpu, scans, Tbs, poses)
            keep.append(b)
            assert eng.integrate_device_batch(b.arr) == 0
        done += count
        calls += 1
        if TRACE is not None:
            TRACE.append({"call": calls, "done": done, "ray": ray, "how": how, "count": count, "estimator": int(eng.cfg.estimation_type),
                          "colour": bool(colour), "sizes": [int(s_["x"].size) for s_ in scans]})
        # behind EVERY call: the two layers a dropped clear or a missed strip shows in first (a stale cell lives only until
        # the next pipeline switch or host write wipes it: a compare every 50 scans walks past most of them)
        compare(eng, ref, f"after {done} scans ({calls} calls, seed {seed})",
                names=[n for n in ("obstacle", "elevation") if ref.exists(n)])
        if done >= next_check:
            compare(eng, ref, f"after {done} scans ({calls} calls, seed {seed})")
            assert eng.last_stats() == (rc_r, st_r), (done, eng.last_stats(), rc_r, st_r)
            next_check = done + 50
        keep.clear()   # (the stream has drained: the device arrays of this call are dead)
    compare(eng, ref, f"at the end ({done} scans, seed {seed})")
    # the stream did exercise what it is for
    if "tiled_min" not in gpu.Engine.default_options:
        assert sum(eng.batch_launches()) > launches0, "no batch launch in the whole stream"
    assert time.perf_counter() - t0 < 240.0
