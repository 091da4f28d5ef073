"""The batch pipeline (fastdem_amd/csrc/fdm_multi.hpp): fdm_engine_integrate_device_batch bins up to 16 small scans
in ONE launch and updates the map in ONE launch.  The map it leaves must be what the reference leaves after
integrating the same scans one by one (elevation_mapping.cpp:94-125 fixes only the per-cell order of the scans) —
every layer bit for bit, the geometry, the statistics of the last scan.  Checked against the CPU oracle run scan
by scan, through the C ABI.

Run on the GPU box:  python -m pytest tests -m gpu
"""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_arrays_close, assert_layers_bit_identical, assert_layers_equal, pair, same_geometry

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(autouse=True, params=["walk_auto", "walk_on"])
def chain_walk(request, gpu):
    """Every test of this module twice: the engine's own choice of who walks a batch's chain of LOCAL-mode moves
    (every bin block for the Kalman estimator, the walker block of the previous launch for the quantile estimator:
    option `batch_walk` -1), and the walker for every engine (`batch_walk` 1: scans that do not pass the crops in the
    middle of a batch then falsify what it assumed, and the blocks behind them — and the next walker — walk themselves)."""
    saved = dict(gpu.Engine.default_options)
    if request.param == "walk_on":
        gpu.Engine.default_options = dict(saved, batch_walk=1)
    yield request.param
    gpu.Engine.default_options = saved


def T(x=0.0, y=0.0, z=0.0, yaw=0.0, pitch=0.0):
    M = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    cp, sp = np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    M[:3, :3] = Rz @ Ry
    M[:3, 3] = (x, y, z)
    return M


class DeviceBatch:
    """`scans` (dicts of numpy channels) resident in HBM + the fdm_device_scan array describing them."""

    def __init__(self, gpu, scans, Tbs, poses):
        import torch
        self.keep = []
        self.arr = (gpu.capi.FdmDeviceScan * len(scans))()
        for k, (s, Twb) in enumerate(zip(scans, poses)):
            d = self.arr[k]
            d.n = int(s["x"].size)
            for name, field in (("x", "x"), ("y", "y"), ("z", "z"), ("intensity", "intensity"), ("rgb", "rgb"),
                                ("sigma_z2", "sigma_z2")):
                v = s.get(name)
                if v is None or v.size == 0:
                    setattr(d, field, None)
                    continue
                t = torch.from_numpy(np.ascontiguousarray(v)).cuda()
                self.keep.append(t)
                setattr(d, field, t.data_ptr())
            tb = Tbs[k] if isinstance(Tbs, list) else Tbs
            d.T_base_sensor = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(tb, dtype=np.float64).T).reshape(16))
            d.T_world_base = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(Twb, dtype=np.float64).T).reshape(16))
        torch.cuda.synchronize()


def oracle_scan_by_scan(ref, scans, Tbs, poses):
    rc = st = None
    for k, (s, Twb) in enumerate(zip(scans, poses)):
        kw = {c: s[c] for c in ("intensity", "rgb") if s.get(c) is not None}
        tb = Tbs[k] if isinstance(Tbs, list) else Tbs
        rc, st = ref.integrate(s["x"], s["y"], s["z"], tb, Twb, **kw)
    return rc, st


def check_batch(gpu, R, eng, ref, scans, Tbs, poses, exact=True, expect_batches=True):
    """One fdm_engine_integrate_device_batch call against the oracle run scan by scan.  `pair()` switches the per-point
    cell ids on, and an engine that owes its caller cell ids takes NO batch launch (fdm_engine_multi.inl:
    the batch bin halves do not write them) — so they are switched off here, and the call must have left in batch
    launches (round 3's version of this helper never did: its scans all took the one-scan path)."""
    eng.enable_cell_ids(False)
    before = sum(eng.batch_launches())
    b = DeviceBatch(gpu, scans, Tbs, poses)
    assert eng.integrate_device_batch(b.arr) == 0
    # (the first scan of a fresh engine goes alone: it creates the layers.  In the `tiled_all` fixture variant every scan is
    # forced through the record pools, which take one fused launch per scan: the call is then checked against the oracle
    # as a sequence of those)
    if expect_batches and len(scans) >= 3 and "tiled_min" not in gpu.Engine.default_options:
        assert sum(eng.batch_launches()) > before, "the call took no batch launch"
    rc_r, st_r = oracle_scan_by_scan(ref, scans, Tbs, poses)
    rc_e, st_e = eng.last_stats()
    assert (rc_e, st_e) == (rc_r, st_r), (rc_e, st_e, rc_r, st_r)
    if exact:
        assert_layers_bit_identical(eng, ref)
    else:
        assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())
    return b


def cloud(rng, n, spread, z0=0.0, intensity=False, rgb=False):
    s = {"x": (rng.uniform(-spread, spread, n)).astype(F32), "y": (rng.uniform(-spread, spread, n)).astype(F32),
         "z": (z0 + 0.2 * rng.standard_normal(n)).astype(F32), "intensity": None, "rgb": None}
    if intensity:
        s["intensity"] = rng.uniform(0, 1, n).astype(F32)
    if rgb:
        s["rgb"] = rng.integers(0, 1 << 24, n, dtype=np.uint32)
    return s


# ---------------------------------------------------------------------------------------------
def test_vlp16_stream_in_batches_equals_the_reference_scan_by_scan(gpu, R):
    """configs[1]: 37 VLP-16 scans (16 + 16 + 5) with the workload's pose sequence — a one-cell LOCAL shift every
    other scan, so most batches hold several move() strips.  Bit-identical to the oracle run scan by scan."""
    wl = gpu.synth.vlp16(n_scans=6)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    scans = [wl.scan(k) for k in range(37)]
    poses = [wl.pose(k) for k in range(37)]
    check_batch(gpu, R, eng, ref, scans, wl.T_base_sensor, poses)
    assert eng.geometry().start_row != 0
    assert "intensity" in eng.layers()
    # the stream goes on: single enqueue-only scans, then another batch, on the same map
    import torch
    keep = []  # (device arrays stay alive until the engine has read them)
    for k in range(37, 40):
        s = wl.scan(k)
        d = {c: torch.from_numpy(s[c]).cuda() for c in ("x", "y", "z", "intensity")}
        keep.append(d)
        eng.integrate_device(d["x"], d["y"], d["z"], wl.T_base_sensor, wl.pose(k), intensity=d["intensity"])
        ref.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
    check_batch(gpu, R, eng, ref, [wl.scan(k) for k in range(40, 51)], wl.T_base_sensor, [wl.pose(k) for k in range(40, 51)])


@pytest.mark.parametrize("batch_max", [2, 3, 16, 17, 32])
def test_moves_that_wrap_clear_everything_and_come_back(gpu, R, batch_max):
    """Ragged small clouds under a pose sequence built to stress GridMap::move inside a batch: multi-cell shifts in
    both directions and on both axes, a jump larger than the map (everything cleared), a return to the old place,
    and repeated wrap-arounds of the circular buffer."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 30.0

    eng, ref = pair(gpu, R, 8.0, 6.0, 0.1, fill)
    eng.set_option("batch_max", batch_max)
    rng = np.random.default_rng(5)
    steps = [(0, 0), (0.35, 0), (0.35, -0.4), (-1.2, 0.9), (-1.2, 0.9), (3.1, 3.3), (40.0, -25.0), (40.1, -25.0),
             (0.0, 0.0), (0.05, 0.04), (-7.9, 0.0), (-15.8, 0.0), (-23.7, 5.9), (-23.7, 11.8), (-23.65, 11.8),
             (2.0, 2.0), (2.0, 2.1), (2.1, 2.1), (2.1, 2.0), (2.0, 2.0), (10.0, 2.0), (10.0, -4.0), (4.0, -4.0)]
    scans, poses = [], []
    for k, (px, py) in enumerate(steps):
        n = int(rng.integers(200, 3000))
        scans.append(cloud(rng, n, 4.5, intensity=True))
        poses.append(T(px, py, 0.0, yaw=0.1 * k))
    check_batch(gpu, R, eng, ref, scans, T(z=0.5), poses)


@pytest.mark.parametrize("every_move_by_divide", [0, 1])
def test_moves_on_rounding_ties(gpu, R, every_move_by_divide):
    """The bin half rounds every scan's pose once and in parallel (total shift = round((pose - p0) / res)), which is
    only valid away from rounding ties: poses that sit exactly on half a cell, or within 1e-7 of it, must take the
    reference's subtract + divide at their place in the walk (move(): half away from zero).  A dyadic resolution
    makes the ties exact.  `every_move_by_divide` (dbg_batch 3) sends every move down that path: same map."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 30.0

    eng, ref = pair(gpu, R, 16.0, 12.0, 0.5, fill)
    if every_move_by_divide:
        eng.set_option("dbg_batch", 3)
    rng = np.random.default_rng(11)
    xs = [0.0, 0.25, 0.5, 0.75, 0.75 + 1e-7, 1.25 - 1e-7, -0.25, -0.75, -1.25, 2.25, 2.25, 3.0, 3.25 + 4e-5, 3.75 - 4e-5,
          -6.25, -6.25 + 1e-9, 40.25, 40.75, 0.25, -0.25, 0.24999999, 0.7500001]
    scans, poses = [], []
    for k, px in enumerate(xs):
        scans.append(cloud(rng, 1500 + 13 * k, 7.0, intensity=True))
        poses.append(T(px, -0.5 * px + 0.25 * (k % 3), 0.0, yaw=0.05 * k))
    check_batch(gpu, R, eng, ref, scans, T(z=0.5), poses)


def test_small_scans_whose_first_points_miss_the_map_still_clear_the_obstacle_layer(gpu, R):
    """Regression (found by scripts/soak_r03.py): 'some point of scan k landed in the map' was taken from lane 0 of
    each wavefront only, so a scan whose points 0, 64, 128, ... lie outside the map never counted as observing —
    the obstacle layer then kept the previous scan's values (elevation_mapping.cpp:144-146 clears it per observing
    scan).  Tiny scans behind big ones, their leading points outside the map."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 40.0

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.1, fill)
    rng = np.random.default_rng(41)
    scans, poses = [], []
    for k in range(14):
        n = 20000 if k % 3 == 0 else int(rng.integers(20, 300))
        s = cloud(rng, n, 4.5, intensity=True)
        if n < 20000:  # every 64th point (lane 0 of its wavefront) far outside the map, a few points in one cell
            s["x"][::64] = 30.0
            s["x"][1:9], s["y"][1:9] = 0.31, -0.72
            s["z"][1:9] = np.linspace(-0.3, 0.4, 8).astype(F32)
        scans.append(s)
        poses.append(T(0.11 * k, -0.06 * k, 0.0))
    check_batch(gpu, R, eng, ref, scans, T(z=0.5), poses)
    assert np.isfinite(eng.layer("obstacle")).sum() < 50  # only the last (small) scan's cells


def test_a_scan_with_every_point_filtered_does_not_move_the_map(gpu, R):
    """fastdem.cpp:138: a scan whose points all fail the crops returns before the move.  Inside a batch the scans
    BEHIND it must be binned against the geometry without that move — the chain of moves depends on device-side
    data (fdm_multi.hpp's in-launch wait).  Also: scans whose points all fall outside the map (true, map moved,
    nothing written, no obstacle clear)."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -1.0, 2.0, 0.5, 20.0

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.1, fill)
    rng = np.random.default_rng(9)
    scans, poses = [], []
    for k in range(14):
        s = cloud(rng, 1500 + 100 * k, 4.0, intensity=False)
        if k in (3, 4, 9):      # all filtered by cropZ
            s["z"] = (s["z"] + 50.0).astype(F32)
        if k in (6, 13):        # survive the crops, land outside the 10 x 10 m map
            s["x"] = (s["x"] + 12.0).astype(F32)
        scans.append(s)
        poses.append(T(0.33 * k, -0.21 * k, 0.0))
    check_batch(gpu, R, eng, ref, scans, T(z=0.4), poses)
    assert eng.last_stats()[1]["n_in_map"] == 0
    # first scan of a batch filtered / whole batch filtered
    scans2 = [dict(s, z=(s["z"] + 50.0).astype(F32)) for s in scans[:5]]
    check_batch(gpu, R, eng, ref, scans2, T(z=0.4), [T(9.0 + k, 1.0, 0.0) for k in range(5)])
    assert eng.last_stats()[0] == 2  # FDM_SKIP_ALL_FILTERED
    check_batch(gpu, R, eng, ref, scans[:7], T(z=0.4), [T(5.0 - k, 1.0, 0.0) for k in range(7)])


def test_filtered_scans_deep_inside_a_long_call(gpu, R):
    """60 scans in one call = four batches.  From the second batch on the crops were evaluated one launch AHEAD;
    scans 21, 22, 37 and 59 have no surviving point, so they must not move the map and the scans behind them are
    binned against the geometry that really resulted."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -1.0, 2.0, 0.5, 20.0

    eng, ref = pair(gpu, R, 12.0, 9.0, 0.1, fill)
    rng = np.random.default_rng(77)
    scans, poses = [], []
    for k in range(60):
        s = cloud(rng, 2000 + 37 * (k % 9), 4.2, intensity=True)
        if k in (21, 22, 37, 59):
            s["z"] = (s["z"] + 40.0).astype(F32)
        scans.append(s)
        poses.append(T(0.27 * k, 0.15 * np.sin(0.4 * k) * k, 0.0, yaw=0.03 * k))
    check_batch(gpu, R, eng, ref, scans, T(z=0.4), poses)
    assert eng.last_stats()[0] == 2  # the last scan was filtered
    # the same stream with the look-ahead switched off piece by piece leaves the same map
    for opts in ({"batch_crop": 0}, {"batch_fuse": 0}):
        e2, r2 = pair(gpu, R, 12.0, 9.0, 0.1, fill)
        for key, v in opts.items():
            e2.set_option(key, v)
        b = DeviceBatch(gpu, scans, T(z=0.4), poses)
        assert e2.integrate_device_batch(b.arr) == 0
        e2.sync()
        for n in eng.layers():
            assert_arrays_close(e2.layer(n), eng.layer(n), n, 0.0, 0.0)
        assert same_geometry(e2.geometry(), eng.geometry())


def test_global_mode_p2_colour_intensity(gpu, R):
    """GLOBAL map (no moves: no chain wait), P2 quantile estimator, colour + intensity channels, RGB-D model."""
    def fill(c):
        c.mode = 1
        c.estimation_type = 1
        c.sensor_type = 2
        c.z_min, c.z_max = -5.0, 5.0

    eng, ref = pair(gpu, R, 6.0, 6.0, 0.05, fill)
    rng = np.random.default_rng(21)
    scans = [cloud(rng, 4000 + 37 * k, 2.8, z0=1.0, intensity=True, rgb=True) for k in range(20)]
    poses = [T(0.01 * k, 0.02 * k, 0.0, yaw=0.02 * k) for k in range(20)]
    check_batch(gpu, R, eng, ref, scans, T(z=0.1), poses)
    assert "color" in eng.layers() and "intensity" in eng.layers()
    assert np.nanmax(eng.layer("n_points")) >= 5  # the P2 markers left their initialisation phase


def test_local_mode_p2_with_strips(gpu, R):
    def fill(c):
        c.estimation_type = 1
        c.sensor_type = 0

    eng, ref = pair(gpu, R, 5.0, 5.0, 0.1, fill)
    rng = np.random.default_rng(22)
    scans = [cloud(rng, 3000, 2.4, rgb=True) for k in range(19)]
    poses = [T(0.17 * k, 0.09 * k * (-1) ** k, 0.0) for k in range(19)]
    check_batch(gpu, R, eng, ref, scans, T(), poses)


def test_ties_signed_zeros_and_nan_intensity_across_a_batch(gpu, R):
    """First-point-wins ties, +-0 heights / intensities (the first zero's sign stays) and a NaN first intensity
    (sticks, elevation_mapping.cpp:73-79), in cells hit by several scans of one batch."""
    def fill(c):
        c.mode = 1
        c.sensor_type = 0

    eng, ref = pair(gpu, R, 4.0, 4.0, 0.5, fill)
    rng = np.random.default_rng(3)
    scans = []
    for k in range(9):
        n = 600
        x = rng.uniform(-1.9, 1.9, n).astype(F32)
        y = rng.uniform(-1.9, 1.9, n).astype(F32)
        z = rng.choice(np.array([0.0, -0.0, 0.25, -0.25, 0.5], dtype=F32), n)
        a = rng.choice(np.array([0.0, -0.0, np.nan, 0.5, 0.75], dtype=F32), n)
        scans.append({"x": x, "y": y, "z": z, "intensity": a, "rgb": None})
    check_batch(gpu, R, eng, ref, scans, T(), [T() for _ in range(9)])


def test_custom_sensor_model_variance_channel(gpu, R):
    """sigma_z2 override (user SensorModel subclass, fastdem.hpp:79-80): the batch reads it at the winning point.
    Engine against engine (the oracle has no such channel): batch launches vs one launch per scan."""
    cfg = gpu.capi.default_config()
    a = gpu.Engine(6.0, 6.0, 0.1, cfg)
    b = gpu.Engine(6.0, 6.0, 0.1, cfg)
    b.set_option("batch", 0)
    rng = np.random.default_rng(31)
    scans = []
    for k in range(6):
        s = cloud(rng, 2500, 2.9)
        s["sigma_z2"] = rng.uniform(1e-4, 5e-3, 2500).astype(F32)
        scans.append(s)
    db = DeviceBatch(gpu, scans, T(z=0.3), [T(0.1 * k, 0, 0) for k in range(6)])
    assert a.integrate_device_batch(db.arr) == 0
    assert b.integrate_device_batch(db.arr) == 0
    assert a.last_stats() == b.last_stats()
    for n in b.layers():
        assert_arrays_close(a.layer(n), b.layer(n), n, 0.0, 0.0)
    assert np.isfinite(a.layer("variance")).sum() > 500


def test_batch_option_off_is_the_same_map(gpu, R):
    """Engine against engine: the batch launches and one launch per scan leave identical maps; mixed channel sets
    split a call into several batches."""
    wl = gpu.synth.vlp16(n_scans=4)
    a = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    b = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    b.set_option("batch", 0)
    scans = []
    for k in range(21):
        s = dict(wl.scan(k))
        if 8 <= k < 13:
            s["intensity"] = None  # a different channel set: closes the running batch
        scans.append(s)
    poses = [wl.pose(k) for k in range(21)]
    db = DeviceBatch(gpu, scans, wl.T_base_sensor, poses)
    assert a.integrate_device_batch(db.arr) == 0
    assert b.integrate_device_batch(db.arr) == 0
    assert a.last_stats() == b.last_stats()
    for n in b.layers():
        assert_arrays_close(a.layer(n), b.layer(n), n, 0.0, 0.0)
    assert same_geometry(a.geometry(), b.geometry())


def test_first_batch_of_every_call_and_short_calls(gpu, R):
    """No block of a batch launch waits for another one: which scans of a batch move the map is decided before the launch
    starts, by scout blocks that ride in the previous launch — or, for the FIRST batch of a call, in a small launch of
    their own.  Many short calls (2 ... 5 scans: every batch is a first batch), filtered scans first / last / everywhere,
    against the oracle scan by scan."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -1.0, 2.0, 0.5, 20.0

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.1, fill)
    rng = np.random.default_rng(123)
    k = 0
    for call in range(14):
        n = 2 + call % 4
        dead = {(call * 3) % n} if call % 3 else ({0, n - 1} if call % 2 else set(range(n)))
        scans, poses = [], []
        for j in range(n):
            s = cloud(rng, 1800 + 50 * j, 4.0, intensity=True)
            if j in dead:
                s["z"] = (s["z"] + 50.0).astype(F32)
            scans.append(s)
            poses.append(T(0.23 * k, -0.17 * k, 0.0, yaw=0.02 * k))
            k += 1
        check_batch(gpu, R, eng, ref, scans, T(z=0.4), poses, expect_batches=False)
    if "tiled_min" not in gpu.Engine.default_options:
        assert sum(eng.batch_launches()) >= 14


def test_host_batch_entry_pinned_in_place_and_pageable(gpu, R):
    """fdm_engine_integrate_host_batch: N integrate() calls on HOST clouds as one call.  Clouds from the engine's pinned
    pool are read in place by the batch launches, pageable ones are staged — both against the oracle scan by scan;
    status and statistics are the last scan's."""
    import ctypes as C
    wl = gpu.synth.vlp16(n_scans=5)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.enable_cell_ids(False)
    keep = []

    def host_scans(ks, pinned):
        arr = (gpu.capi.FdmDeviceScan * len(ks))()
        for i, k in enumerate(ks):
            s = wl.scan(k % 5)
            d = arr[i]
            d.n = int(s["x"].size)
            for c in ("x", "y", "z", "intensity"):
                if pinned:
                    h = gpu.host_array(s[c], np.float32)
                    keep.append(h)
                    a = h.array
                else:
                    a = np.ascontiguousarray(s[c])
                    keep.append(a)
                setattr(d, c, a.ctypes.data)
            d.rgb, d.sigma_z2 = None, None
            d.T_base_sensor = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(wl.T_base_sensor, dtype=np.float64).T).reshape(16))
            d.T_world_base = (C.c_double * 16)(*np.ascontiguousarray(np.asarray(wl.pose(k), dtype=np.float64).T).reshape(16))
        return arr

    k0 = 0
    for pinned, n in ((True, 21), (False, 19), (True, 3)):
        ks = list(range(k0, k0 + n))
        before = sum(eng.batch_launches())
        rc_e, st_e = eng.integrate_host_batch(host_scans(ks, pinned))
        for k in ks:
            s = wl.scan(k % 5)
            rc_r, st_r = ref.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        assert (rc_e, st_e) == (rc_r, st_r)
        if "tiled_min" not in gpu.Engine.default_options:
            assert sum(eng.batch_launches()) > before
        assert_layers_bit_identical(eng, ref)
        assert same_geometry(eng.geometry(), ref.geometry())
        k0 += n
