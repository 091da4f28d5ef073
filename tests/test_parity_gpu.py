"""Parity of the HIP engine (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): cell indices bit-exact; fused height / variance within 1e-5
relative (helpers.RTOL).  In practice the engine reproduces the oracle's float operation order,
so most layers come out bit-identical; the tolerance is still the stated contract.
Run on the GPU box:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from helpers import (assert_arrays_close, assert_layers_bit_identical, assert_layers_equal, pair, run_both,
                     same_geometry)

pytestmark = pytest.mark.gpu
F32 = np.float32


def T(x=0.0, y=0.0, z=0.0, yaw=0.0):
    M = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    M[:2, :2] = [[c, -s], [s, c]]
    M[:3, 3] = (x, y, z)
    return M


def run_workload(gpu, R, wl, n_scans, check_every=1):
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    worst = 0.0
    for k in range(n_scans):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
        if (k + 1) % check_every == 0 or k == n_scans - 1:
            worst = max(worst, assert_layers_equal(eng, ref))
            assert same_geometry(eng.geometry(), ref.geometry())
    return eng, ref, worst


# ------------------------------------------------------------ BASELINE configs ----
@pytest.mark.parametrize("order", ["azimuth", "ring"])
def test_c2_vlp16_kalman_local(gpu, R, order):
    """configs[1]: VLP-16 28.8K pts, 15x15 m @ 0.1 m, Kalman, LOCAL; 10 scans incl. row shifts."""
    wl = gpu.synth.vlp16(n_scans=10, order=order)
    eng, ref, _ = run_workload(gpu, R, wl, 10, check_every=1)
    assert eng.geometry().start_row != 0  # the pose sequence did move the window
    assert "intensity" in eng.layers()


def test_c3_rgbd_p2_colour(gpu, R):
    """configs[2]: 640x480 RGB-D (~300K pts), 10x10 m @ 0.05 m, P2 quantile, colour channel."""
    wl = gpu.synth.rgbd(n_scans=8)
    eng, ref, _ = run_workload(gpu, R, wl, 8, check_every=2)
    assert "color" in eng.layers() and np.isfinite(eng.layer("elevation")).sum() > 1000


@pytest.mark.parametrize("order", ["azimuth", "ring"])
def test_c4_lidar128_rolling_reduced(gpu, R, order):
    """configs[3] at 1/8 azimuth density (262K pts): 60x60 m @ 0.05 m, 8-cell shift per scan.
    Ring-major order = long same-cell runs across the lanes of k_bin4 (cross-lane run merge)."""
    wl = gpu.synth.lidar128(n_scans=6, n_az=2048, order=order)
    eng, ref, _ = run_workload(gpu, R, wl, 6, check_every=3)
    assert eng.last_stats()[1]["shift_rows"] == -8


def test_c4_lidar128_full_size(gpu, R):
    """configs[3] at full size: 2 097 152 pts per scan, 1.44 M cells."""
    wl = gpu.synth.lidar128(n_scans=2)
    assert wl.n_points == 128 * 16384
    run_workload(gpu, R, wl, 2, check_every=2)


def test_c5_global_reduced_and_tiled(gpu, R):
    """configs[4] reduced to 100x100 m: GLOBAL fixed-origin map, and the same map stored as
    2x2 spatial tiles (owned windows only) reassembles bit-identically."""
    wl = gpu.synth.global_map(n_scans=3, size_m=100.0, n_az=2048, radius=30.0)
    eng, ref, _ = run_workload(gpu, R, wl, 3, check_every=3)
    rows, cols = eng.rows, eng.cols
    hr, hc = rows // 2, cols // 2
    full = {n: eng.layer(n) for n in eng.layers()}
    tiles = []
    for tr in range(2):
        for tc in range(2):
            r0, c0 = tr * hr, tc * hc
            tile = (r0, c0, hr, hc, r0, c0, hr, hc)
            t = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()),
                           tile=tile)
            for k in range(3):
                s = wl.scan(k)
                t.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k),
                            intensity=s["intensity"])
            tiles.append((r0, c0, t))
    for name, whole in full.items():
        for r0, c0, t in tiles:
            if not t.exists(name):  # a tile no point with the channel landed in
                assert np.isnan(whole[r0:r0 + hr, c0:c0 + hc]).all()
                continue
            assert_arrays_close(t.layer(name), whole[r0:r0 + hr, c0:c0 + hc], name, 0.0, 0.0)


def test_c5_full_size_properties(gpu):
    """configs[4] at full size (8000x8000 cells) — no oracle at this size, so size-independent
    properties: stats add up, min <= elevation <= max, n_points counts scans per cell."""
    wl = gpu.synth.global_map(n_scans=2, n_az=4096)
    eng = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    assert (eng.rows, eng.cols) == (8000, 8000)
    touched = 0
    for k in range(2):
        s = wl.scan(k)
        rc, st = eng.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k))
        assert rc == 0 and st["n_in_map"] == st["n_after_filter"] > 0
        touched += st["n_cells_touched"]
    n = eng.layer("n_points")
    assert int(n.sum()) == touched
    el, lo, hi = eng.layer("elevation"), eng.layer("elevation_min"), eng.layer("elevation_max")
    m = np.isfinite(el)
    assert m.sum() == (n > 0).sum()
    assert np.all(lo[m] <= el[m] + 1e-6) and np.all(el[m] <= hi[m] + 1e-6)


# ------------------------------------------------- sensor x estimator matrix ----
@pytest.mark.parametrize("sensor", [0, 1, 2])
@pytest.mark.parametrize("est", [0, 1])
@pytest.mark.parametrize("mode", [0, 1])
def test_sensor_estimator_mode_matrix(gpu, R, sensor, est, mode):
    rng = np.random.default_rng(100 * sensor + 10 * est + mode)

    def fill(c):
        c.sensor_type, c.estimation_type, c.mode = sensor, est, mode
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.3, 9.0

    eng, ref = pair(gpu, R, 12.0, 9.0, 0.25, fill)  # non-square map
    Tbs = T(0.1, -0.05, 0.7, yaw=0.3)
    Tbs[:3, :3] = Tbs[:3, :3] @ np.array([[np.cos(0.2), 0, np.sin(0.2)], [0, 1, 0],
                                          [-np.sin(0.2), 0, np.cos(0.2)]])
    for k in range(7):
        n = 5000
        p = rng.normal(0, 3.0, size=(n, 3)).astype(F32)
        p[:, 2] = (0.2 * np.sin(p[:, 0]) + rng.normal(0, 0.05, n)).astype(F32) + (1.0 if sensor == 2 else -0.7)
        s = {"x": p[:, 0].copy(), "y": p[:, 1].copy(), "z": p[:, 2].copy(),
             "intensity": rng.random(n, dtype=F32) if k % 2 == 0 else None, "rgb": None}
        run_both(eng, ref, s, Tbs, T(0.37 * k, -0.21 * k, 0.0, yaw=0.05 * k))
        assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


# ------------------------------------------------------------------ edge cases ----
def test_empty_cloud_and_all_filtered(gpu, R):
    def fill(c):
        c.z_min, c.z_max = 5.0, 6.0

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5, fill)
    e = np.zeros(0, dtype=F32)
    rc, _ = eng.integrate(e, e, e, T(), T(3.0, 0.0))
    assert rc == gpu.capi.FDM_SKIP_EMPTY_CLOUD == ref.integrate(e, e, e, T(), T(3.0, 0.0))[0]
    s = {"x": np.ones(100, F32), "y": np.ones(100, F32), "z": np.ones(100, F32)}
    rc, st = run_both(eng, ref, s, T(), T(3.0, 0.0))  # LOCAL mode, but nothing survives cropZ
    assert rc == gpu.capi.FDM_SKIP_ALL_FILTERED and st["n_after_filter"] == 0
    g = eng.geometry()
    assert (g.position_x, g.start_row) == (0.0, 0)  # fastdem.cpp:138 returns BEFORE the move
    assert np.isnan(eng.layer("elevation")).all()
    assert_layers_equal(eng, ref)


def test_nothing_lands_in_map_still_true_and_obstacle_kept(gpu, R):
    def fill(c):
        c.mode = 1

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5, fill)
    s = {"x": np.array([0.0, 0.0], F32), "y": np.zeros(2, F32), "z": np.array([0.0, 2.0], F32)}
    run_both(eng, ref, s, T(), T())
    far = {"x": np.array([0.0], F32), "y": np.zeros(1, F32), "z": np.zeros(1, F32)}
    rc, st = run_both(eng, ref, far, T(), T(500.0, 500.0))
    assert rc == 0 and st["n_in_map"] == 0
    # update() returns before updateObstacle when no cell was observed (elevation_mapping.cpp:118)
    assert np.isfinite(eng.layer("obstacle")).sum() == 1
    assert_layers_equal(eng, ref)


def test_non_finite_points_never_integrate(gpu, R):
    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5)
    x = np.array([0.0, np.nan, np.inf, 1.0, -np.inf, 2.0], F32)
    y = np.array([0.0, 0.0, 0.0, np.nan, 1.0, 2.0], F32)
    z = np.array([1.0, 1.0, 1.0, 1.0, 1.0, np.inf], F32)
    rc, st = run_both(eng, ref, {"x": x, "y": y, "z": z}, T(z=0.2), T())
    assert rc == 0 and st["n_in_map"] == 1
    assert_layers_equal(eng, ref)


def test_heavy_collisions_ties_and_signed_zero(gpu, R):
    """All points in a handful of cells; equal-z ties must take the FIRST point's variance
    (strict '<', elevation_mapping.cpp:65-68); -0.0 and +0.0 tie."""
    def fill(c):
        c.mode = 1
        c.kalman_max_variance = 1.0

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5, fill)
    rng = np.random.default_rng(7)
    n = 20000
    x = rng.uniform(-0.7, 0.7, n).astype(F32)
    y = rng.uniform(-0.7, 0.7, n).astype(F32)
    z = rng.choice(np.array([0.25, 0.25, 0.5, -0.0, 0.0, 1.5], F32), n).astype(F32)
    Tbs = T(0.0, 0.0, 0.0)
    for k in range(3):
        run_both(eng, ref, {"x": x, "y": y, "z": z}, Tbs, T())
        assert_layers_equal(eng, ref)
        x = x[::-1].copy()
    # one cell, many identical points
    one = {"x": np.zeros(4096, F32) + 2.1, "y": np.zeros(4096, F32) + 2.1, "z": np.zeros(4096, F32) + 0.3}
    run_both(eng, ref, one, Tbs, T())
    assert_layers_equal(eng, ref)


@pytest.mark.parametrize("order", ["ring", "azimuth"])
def test_bin_kernel_variants_agree(gpu, R, order):
    """k_bin4 (LDS-staged, default) == k_bin with wave merge == k_bin with plain atomics."""
    wl = gpu.synth.vlp16(n_scans=3, order=order)
    engs = [gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
            for _ in range(3)]
    engs[0].set_option("bin_variant", 4)  # (a 28.8 K-point scan would take k_bin by itself)
    engs[1].set_option("bin_variant", 1)
    engs[2].set_option("bin_variant", 1)
    engs[2].set_option("wave_merge", 0)
    for k in range(3):
        s = wl.scan(k)
        for e in engs:
            e.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
    for n in engs[0].layers():
        for e in engs[1:]:
            assert_arrays_close(engs[0].layer(n), e.layer(n), n, 0.0, 0.0)


def test_stamp_gated_update_equals_dense(gpu, R):
    """Very large maps gate update tiles by scan stamps; force that mode on a small map and check
    it against the oracle through moves, obstacle appear/disappear and a scan that lands nowhere."""
    eng, ref = pair(gpu, R, 12.0, 12.0, 0.1)
    eng.set_option("dense", 0)
    rng = np.random.default_rng(21)
    for k in range(12):
        n = 4000
        cx = -3.0 + 0.5 * k
        s = {"x": (rng.normal(cx, 0.8, n)).astype(F32), "y": rng.normal(0.3 * k, 0.8, n).astype(F32),
             "z": rng.normal(0, 0.3, n).astype(F32), "intensity": rng.random(n, dtype=F32)}
        if k == 6:
            s = {kk: (v + 500.0 if kk == "x" else v) for kk, v in s.items()}  # nothing lands
        run_both(eng, ref, s, T(z=0.4), T(0.13 * k, -0.07 * k, yaw=0.02 * k))
        assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_unaligned_device_pointers_fall_back_to_scalar_kernel(gpu, R):
    import torch
    wl = gpu.synth.vlp16(n_scans=1)
    s = wl.scan(0)
    a = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    b = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    n = wl.n_points - 3  # odd length + 4-byte offset views: not 16-byte aligned
    d = {k: torch.from_numpy(s[k]).cuda() for k in ("x", "y", "z", "intensity")}
    a.integrate(s["x"][1:1 + n], s["y"][1:1 + n], s["z"][1:1 + n], wl.T_base_sensor, wl.pose(0),
                intensity=s["intensity"][1:1 + n])
    b.integrate_device(d["x"][1:1 + n], d["y"][1:1 + n], d["z"][1:1 + n], wl.T_base_sensor, wl.pose(0),
                       intensity=d["intensity"][1:1 + n])
    b.sync()
    for name in a.layers():
        assert_arrays_close(a.layer(name), b.layer(name), name, 0.0, 0.0)


def test_update_direct_path_with_nan_z_and_variance(gpu, R):
    """ElevationMapping::update on a map-frame cloud (tests/test_dual_layer.cpp:71): NaN z creates
    a cell whose min_z stays FLT_MAX and max_z lowest() (SURVEY.md §8a edge semantics)."""
    def fill(c):
        c.mode = 1
        c.kalman_max_variance = 1.0

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5, fill)
    x = np.array([0, 0, 1.1, 1.1, 2.2, 3.3], F32)
    y = np.array([0, 0, 0, 0, 0, 0], F32)
    z = np.array([0.0, 3.0, np.nan, 1.0, np.nan, np.inf], F32)
    var = np.array([0.04, 0.01, 0.02, 0.0, 0.03, 0.05], F32)
    for zv in (None, var):
        se = eng.update(x, y, z, (0.0, 0.0), z_var=zv)
        sr = ref.update(x, y, z, (0.0, 0.0), z_var=zv)
        assert se == sr
        assert_layers_equal(eng, ref)
    el = eng.layer("elevation")
    assert (el > 3e38).sum() >= 1  # the all-NaN cell fed FLT_MAX to the estimator, like the reference


def test_update_empty_cloud_still_moves_in_local_mode(gpu, R):
    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5)
    e = np.zeros(0, F32)
    assert eng.update(e, e, e, (2.0, -1.0)) == ref.update(e, e, e, (2.0, -1.0))
    assert same_geometry(eng.geometry(), ref.geometry())
    assert eng.geometry().position_x == 2.0


def test_intensity_nan_first_rule_and_colour_last_wins(gpu, R):
    def fill(c):
        c.mode = 1

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5, fill)
    x = np.array([0, 0, 0, 1.1, 1.1, 1.1, 2.2, 2.2], F32)
    y = np.zeros(8, F32)
    z = np.array([1, 2, 3, 1, 2, 3, 1, 2], F32) * F32(0.1)
    inten = np.array([np.nan, 5.0, 7.0, 1.0, np.nan, 0.5, -3.0, -1.0], F32)
    rgb = np.array([0x010203, 0x7F0000, 0x00FF00, 0x800000, 0x0000FF, 0x000001, 0xFFFFFF, 0x123456],
                   np.uint32)
    s = {"x": x, "y": y, "z": z, "intensity": inten, "rgb": rgb}
    run_both(eng, ref, s, T(), T())
    assert_layers_equal(eng, ref)
    ok, (r, c) = ref.get_index(0.0, 0.0)
    assert np.isnan(eng.layer("intensity")[r, c])             # first point NaN -> NaN forever
    assert eng.layer("color").view(np.uint32)[r, c] == 0x00FF00  # last point of the cell
    # second scan: stored intensity only replaced by larger values
    s2 = dict(s, intensity=np.array([9, 9, 9, 0.2, 0.1, 0.3, np.nan, -2.0], F32))
    run_both(eng, ref, s2, T(), T())
    assert_layers_equal(eng, ref)


def test_lazy_layers_appear_only_after_a_landed_scan(gpu, R):
    def fill(c):
        c.mode = 1

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5, fill)
    far = {"x": np.array([400.0], F32), "y": np.zeros(1, F32), "z": np.zeros(1, F32),
           "intensity": np.ones(1, F32)}
    run_both(eng, ref, far, T(), T())
    assert not eng.exists("intensity") and not ref.exists("intensity")
    near = dict(far, x=np.array([0.4], F32))
    run_both(eng, ref, near, T(), T())
    assert eng.exists("intensity") and ref.exists("intensity")
    assert_layers_equal(eng, ref)


def test_move_large_jump_and_wraparound(gpu, R):
    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5)
    rng = np.random.default_rng(3)
    poses = [(0, 0), (1.3, -0.8), (1.3, -0.8), (-3.9, 4.2), (-3.9, 9.9), (100.0, 100.0), (100.2, 99.7),
             (95.1, 104.9)]
    for k, (px, py) in enumerate(poses):
        n = 3000
        s = {"x": rng.uniform(-6, 6, n).astype(F32), "y": rng.uniform(-6, 6, n).astype(F32),
             "z": rng.normal(0, 0.3, n).astype(F32), "intensity": rng.random(n, dtype=F32)}
        run_both(eng, ref, s, T(z=0.5), T(px, py, yaw=0.1 * k))
        assert_layers_equal(eng, ref)
        assert same_geometry(eng.geometry(), ref.geometry())
    # explicit GridMap::move
    eng.move(90.0, 101.0)
    ref.move(90.0, 101.0)
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())


def test_user_layers_are_cleared_by_move_and_reset(gpu, R):
    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5)
    tag = np.arange(400, dtype=F32).reshape(20, 20)
    for e in (eng, ref):
        e.add("traversability", 0.5)
        e.set_layer("custom", tag)
    s = {"x": np.zeros(10, F32), "y": np.zeros(10, F32), "z": np.zeros(10, F32)}
    run_both(eng, ref, s, T(), T(1.6, 0.0))
    assert_layers_equal(eng, ref)
    assert np.isnan(eng.layer("custom")).sum() == 3 * 20
    eng.clear()   # FastDEM::reset -> clearAll
    ref.clear()
    assert_layers_equal(eng, ref)
    run_both(eng, ref, s, T(), T(1.6, 0.0))  # estimators re-initialise from all-NaN state
    assert_layers_equal(eng, ref)


def test_switching_estimator_keeps_the_other_layers(gpu, R):
    eng, ref = pair(gpu, R, 10.0, 10.0, 0.5)
    rng = np.random.default_rng(5)

    def scan():
        n = 2000
        return {"x": rng.uniform(-4, 4, n).astype(F32), "y": rng.uniform(-4, 4, n).astype(F32),
                "z": rng.normal(0, 0.2, n).astype(F32)}

    run_both(eng, ref, scan(), T(z=0.3), T())
    for est in (1, 0, 1):
        ce, cr = eng.cfg, ref.cfg
        ce.estimation_type = cr.estimation_type = est
        eng.set_config(ce)
        ref.set_config(cr)
        for k in range(3):
            run_both(eng, ref, scan(), T(z=0.3), T(0.6 * k, 0.0))
        assert_layers_equal(eng, ref)
    assert "_kalman_p" in eng.layers() and "_p2_q0" in eng.layers()


def test_sigma_z2_override_for_custom_sensor_models(gpu, R):
    """A user SensorModel subclass (fastdem.hpp:79-80) is evaluated on the host and handed over
    as per-point sigma_z^2; feeding the oracle's own values must reproduce the built-in result."""
    wl = gpu.synth.vlp16(n_scans=1)
    s = wl.scan(0)
    cfg_r = wl.apply_to(R.default_config())
    sig = np.array([R.sigma_z2(cfg_r, [s["x"][i], s["y"][i], s["z"][i]], wl.T_base_sensor, wl.pose(0))
                    for i in range(0, wl.n_points)], dtype=F32)
    a = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    b = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    a.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(0))
    b.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(0), sigma_z2=sig)
    for n in a.layers():
        assert_arrays_close(a.layer(n), b.layer(n), n, 0.0, 0.0)


def test_device_resident_async_stream_matches_sync(gpu, R):
    """integrate_device (inputs in HBM, enqueue-only) over 20 scans == the synchronous path."""
    import torch
    wl = gpu.synth.vlp16(n_scans=4)
    a = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    b = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    dev = [{k: (torch.from_numpy(v).cuda() if v is not None else None) for k, v in s.items()}
           for s in wl.scans]
    for k in range(20):
        s, d = wl.scan(k), dev[k % 4]
        a.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        b.integrate_device(d["x"], d["y"], d["z"], wl.T_base_sensor, wl.pose(k), intensity=d["intensity"])
    rc, st = b.last_stats()
    assert rc == 0 and st == a.last_stats()[1]
    for n in a.layers():
        assert_arrays_close(a.layer(n), b.layer(n), n, 0.0, 0.0)
    assert same_geometry(a.geometry(), b.geometry())


def test_p2_fading_memory_and_marker_choice(gpu, R):
    def fill(c):
        c.estimation_type = 1
        c.mode = 1
        c.p2_max_sample_count = 8.0
        c.p2_elevation_marker = 2

    eng, ref = pair(gpu, R, 6.0, 6.0, 0.5, fill)
    rng = np.random.default_rng(11)
    for k in range(25):
        n = 600
        s = {"x": rng.uniform(-3, 3, n).astype(F32), "y": rng.uniform(-3, 3, n).astype(F32),
             "z": rng.normal(0.5, 0.3, n).astype(F32)}
        run_both(eng, ref, s, T(), T())
    assert_layers_equal(eng, ref)
    assert eng.layer("n_points").max() == 8.0


def test_tiled_engines_with_halo_exchange_match_single_map(gpu, R):
    """SURVEY.md §8e on one GPU: four tile engines (owned block + 6-cell halo ring) integrate the
    same scans, exchange halos through the HIP pack/unpack kernels, and every stored window then
    equals the untiled map — owned cells AND halo ring."""
    import torch
    from fastdem_amd import tiling
    wl = gpu.synth.global_map(n_scans=3, size_m=60.0, n_az=1024, radius=12.0)
    whole = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    world = 4
    plans = [tiling.make_plan(r, world, whole.rows, whole.cols, halo=6) for r in range(world)]
    engs = [gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()),
                       tile=p.fdm_tile()) for p in plans]
    tiles = [tiling.EngineTile(e, p, "cuda:0") for e, p in zip(engs, plans)]
    names = ["elevation", "variance", "elevation_min", "elevation_max", "upper_bound", "lower_bound",
             "n_points", "obstacle"]
    for k in range(3):
        s = wl.scan(k)
        for e in [whole] + engs:
            e.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k))
        # in-process stand-in for the RCCL isend/irecv pairs of tiling.exchange_halos
        for r, p in enumerate(plans):
            for other, rect in p.sends.items():
                buf = tiles[r].pack(rect, names)
                tiles[r].fence()
                assert plans[other].recvs[r] == rect
                tiles[other].unpack(rect, names, buf)
        torch.cuda.synchronize()
    for p, e in zip(plans, engs):
        st = p.stored
        for n in names:
            assert_arrays_close(e.layer(n), whole.layer(n)[st.r0:st.r1, st.c0:st.c1], n, 0.0, 0.0)


@pytest.mark.parametrize("big", [False, True])
def test_scan_callback_clouds_match_the_reference(gpu, R, big):
    """onScanPreprocessed / onScanRasterized payloads (fastdem.cpp:139-150, 200-214): the
    preprocessed cloud is bit-identical in input order incl. its 3x3 covariance channel; the rasterized cloud is the
    same SET of (cell centre, min_z) points (the reference's order is hash-map order)."""
    wl = gpu.synth.lidar128(n_scans=3, n_az=1024) if big else gpu.synth.vlp16(n_scans=3)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    eng.capture(2, True)  # 2: the preprocessed cloud with its 3x3 covariance channel, as the reference hands it over
    ref.capture(True)
    for k in range(3):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
        n = wl.n_points
        pe, pr = eng.last_preprocessed(n), ref.last_preprocessed(n)
        assert pe[0].size == pr[0].size > 0
        for a, b, name in zip(pe, pr, "xyzv"):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"preprocessed {name} differs"
        ce, cr = eng.last_preprocessed_cov(n), ref.last_preprocessed_cov(n)
        assert ce.shape == cr.shape == (pe[0].size, 3, 3)
        assert np.array_equal(ce.view(np.uint32), cr.view(np.uint32)), "R Sigma R^T differs"
        assert np.array_equal(ce[:, 2, 2].view(np.uint32), pe[3].view(np.uint32))
        re_, rr = eng.last_rasterized(eng.rows * eng.cols), ref.last_rasterized(eng.rows * eng.cols)
        assert re_[0].size == rr[0].size == eng.last_stats()[1]["n_cells_touched"]
        se = sorted(zip(*[v.tolist() for v in re_]))
        sr = sorted(zip(*[v.tolist() for v in rr]))
        assert se == sr
    assert_layers_equal(eng, ref)


def test_engine_on_a_caller_provided_stream(gpu, R):
    """fdm_engine_set_stream: the engine enqueues on the caller's HIP stream (here a torch stream),
    so torch.cuda events on that stream bracket its kernels and results are unchanged."""
    import torch
    wl = gpu.synth.vlp16(n_scans=2)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    stream = torch.cuda.Stream()
    eng.set_stream(stream.cuda_stream)
    dev = [{k: (torch.from_numpy(v).cuda() if v is not None else None) for k, v in s.items()} for s in wl.scans]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        for k in range(6):
            d = dev[k % 2]
            eng.integrate_device(d["x"], d["y"], d["z"], wl.T_base_sensor, wl.pose(k), intensity=d["intensity"])
        e1.record()
    stream.synchronize()
    assert 0.0 < e0.elapsed_time(e1) < 50.0
    for k in range(6):
        s = wl.scan(k)
        ref.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
    assert_layers_equal(eng, ref)
    eng.set_stream(0)  # back to the engine's own stream
    run_both(eng, ref, wl.scan(0), wl.T_base_sensor, wl.pose(6))
    assert_layers_equal(eng, ref)


@pytest.mark.parametrize("est", [0, 1])
def test_per_layer_layout_equals_cell_records(gpu, R, est):
    """The default packs the estimator state into 64 B / 128 B cell records; option records=0 keeps
    one array per layer.  Both must match the oracle through moves, uploads and clears."""
    def fill(c):
        c.estimation_type = est

    rng = np.random.default_rng(77 + est)
    a, ref = pair(gpu, R, 10.0, 8.0, 0.2, fill)
    b = gpu.Engine(10.0, 8.0, 0.2, a.cfg)
    b.set_option("records", 0)
    tag = rng.normal(0, 1, (a.rows, a.cols)).astype(F32)
    for e in (a, b, ref):
        e.set_layer("elevation", tag)           # host write into a record field
        e.add("extra", 0.25)
    for k in range(8):
        n = 3000
        s = {"x": rng.uniform(-5, 5, n).astype(F32), "y": rng.uniform(-4, 4, n).astype(F32),
             "z": rng.normal(0, 0.3, n).astype(F32), "intensity": rng.random(n, dtype=F32)}
        run_both(a, ref, s, T(z=0.4), T(0.31 * k, -0.17 * k, yaw=0.03 * k))
        b.integrate(s["x"], s["y"], s["z"], T(z=0.4), T(0.31 * k, -0.17 * k, yaw=0.03 * k),
                    intensity=s["intensity"])
        if k == 4:
            for e in (a, b, ref):
                e.clear("variance")
    assert_layers_equal(a, ref)
    for n in a.layers():
        assert_arrays_close(a.layer(n), b.layer(n), n, 0.0, 0.0)
    a.set_option("records", 0)   # unpack on the fly, keep going
    b.set_option("records", 1)   # pack on the fly
    s = {"x": rng.uniform(-5, 5, 500).astype(F32), "y": rng.uniform(-4, 4, 500).astype(F32),
         "z": rng.normal(0, 0.3, 500).astype(F32), "intensity": None}
    run_both(a, ref, s, T(z=0.4), T(2.9, -1.4))
    b.integrate(s["x"], s["y"], s["z"], T(z=0.4), T(2.9, -1.4))
    assert_layers_equal(a, ref)
    for n in a.layers():
        assert_arrays_close(a.layer(n), b.layer(n), n, 0.0, 0.0)


# ------------------------------------------------------------ large-scan (tiled) pipeline ----
@pytest.mark.parametrize("colour", [False, True])
def test_tiled_pipeline_sparse_cloud_overflows_the_block_table(gpu, R, colour):
    """A cloud whose consecutive points all fall into different cells: every block of the large-scan bin
    kernel fills its LDS cell table to the brim (one slot per point: the table has exactly as many slots as
    the block has points, so it cannot overflow) and writes one record per point, spread over many map
    tiles.  Ties, +-0 and NaN intensities ride along."""
    def fill(c):
        c.mode = 1
        c.kalman_max_variance = 1.0

    eng, ref = pair(gpu, R, 120.0, 120.0, 0.1, fill)
    rng = np.random.default_rng(11)
    n = 70000  # >= the engine's own threshold for the large-scan pipeline
    for k in range(3):
        s = {"x": rng.uniform(-55, 55, n).astype(F32), "y": rng.uniform(-55, 55, n).astype(F32),
             "z": rng.choice(np.array([0.25, -0.0, 0.0, 0.5, 1.5, -1.0], F32), n).astype(F32),
             "intensity": rng.choice(np.array([0.0, -0.0, 0.3, np.nan, 7.0], F32), n).astype(F32)}
        if colour:
            s["rgb"] = rng.integers(0, 1 << 24, n, dtype=np.uint32)
        if k == 2:  # a dense patch on top: blocks with few distinct cells next to overflowing ones
            s["x"][:20000] = rng.uniform(-1, 1, 20000).astype(F32)
            s["y"][:20000] = rng.uniform(-1, 1, 20000).astype(F32)
        run_both(eng, ref, s, T(z=0.3), T(0.4 * k, -0.2 * k, yaw=0.1 * k))
        assert_layers_equal(eng, ref, rtol=0.0)


def test_tiled_and_scratch_pipelines_interleave(gpu, R):
    """Small scans (per-cell scratch pipeline) and large ones (per-tile record pools) on the same map: the
    obstacle layer's whole-layer clear has to hold across the switch, in LOCAL mode with moves."""
    eng, ref = pair(gpu, R, 30.0, 30.0, 0.1)
    eng.set_option("tiled_min", 60000)  # (by itself the engine keeps a map of 88 tiles on the scratch pipeline)
    rng = np.random.default_rng(5)
    for k, n in enumerate([3000, 80000, 2000, 90000, 70000, 500]):
        s = {"x": rng.normal(0.5 * k, 4.0, n).astype(F32), "y": rng.normal(0, 4.0, n).astype(F32),
             "z": rng.normal(0, 0.5, n).astype(F32), "intensity": rng.random(n, dtype=F32)}
        run_both(eng, ref, s, T(z=0.5), T(0.37 * k, 0.11 * k, yaw=0.05 * k))
        assert_layers_equal(eng, ref, rtol=0.0)
    assert same_geometry(eng.geometry(), ref.geometry())


@pytest.mark.parametrize("n_pts", [28800, 100000])
def test_held_back_update_never_reads_the_callers_arrays(gpu, R, n_pts):
    """The streaming pattern of a real caller: ONE set of device buffers, refilled for every scan by
    asynchronous torch copies.  The update of scan t is still held back when the buffers are overwritten
    with scan t+1 (here: with garbage first) — it must gather from the engine's own copy."""
    import torch
    wl = gpu.synth.vlp16(n_scans=4) if n_pts < 50000 else gpu.synth.lidar128(n_scans=4, n_az=n_pts // 128)
    a = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    b = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()))
    n = wl.n_points
    buf = {c: torch.empty(n, dtype=torch.float32, device="cuda") for c in ("x", "y", "z", "intensity")}
    for k in range(6):
        s = wl.scan(k)
        a.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        for c in buf:
            buf[c].copy_(torch.from_numpy(s[c]), non_blocking=True)   # torch's stream ...
        b.integrate_device(buf["x"], buf["y"], buf["z"], wl.T_base_sensor, wl.pose(k),
                           intensity=buf["intensity"])               # ... the engine orders itself behind it
        torch.cuda.synchronize()  # the bin kernel has run; the map update of this scan is still held back
        for c in buf:
            buf[c].fill_(float("nan") if c != "intensity" else 1e9)
        torch.cuda.synchronize()
    rc, st = b.last_stats()
    assert rc == 0 and st == a.last_stats()[1]
    for name in a.layers():
        assert_arrays_close(a.layer(name), b.layer(name), name, 0.0, 0.0)
    assert same_geometry(a.geometry(), b.geometry())


# ------------------------------------------------------------ configs[4] as BASELINE.json states it ----
def test_c5_stated_size_engine_vs_oracle_and_2x4_tiles(gpu, R):
    """configs[4] at its stated size: GLOBAL 400x400 m @ 0.05 m map (8000x8000 cells), 2 097 152-point scans.
    (a) the untiled engine against the oracle — cell ids of every point bit-exact, every layer bit-identical;
    (b) the 2x4 spatial tiling of the 8-GPU configuration (tiling.make_plan: owned window + 6-cell halo per
        tile), all eight tile engines on this one GPU, each fed the whole scan: every OWNED window equals
        the untiled map bit for bit (the halo ring is filled by the exchange, tests/test_tiling_gpu.py)."""
    from fastdem_amd import tiling
    wl = gpu.synth.global_map(n_scans=2)
    assert wl.n_points == 128 * 16384
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    assert (eng.rows, eng.cols) == (8000, 8000)
    for k in range(2):
        rc, st = run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
        assert rc == 0 and st["n_in_map"] > 500000
    names = eng.layers()
    assert sorted(names) == sorted(ref.layers())
    whole = {}
    for name in names:
        whole[name] = eng.layer(name)
        assert_arrays_close(whole[name], ref.layer(name), name, 0.0, 0.0)
    del ref
    for rank in range(8):
        plan = tiling.make_plan(rank, 8, 8000, 8000, tiling.DEFAULT_HALO)
        t = gpu.Engine(wl.width, wl.height, wl.resolution, wl.apply_to(gpu.capi.default_config()),
                       tile=plan.fdm_tile())
        for k in range(2):
            s = wl.scan(k)
            t.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        o, st_ = plan.owned, plan.stored
        for name in names:
            want = whole[name][o.r0:o.r1, o.c0:o.c1]
            if not t.exists(name):  # a lazily created layer no point of this tile carried
                assert np.isnan(want).all(), (rank, name)
                continue
            got = t.layer(name)[o.r0 - st_.r0:o.r1 - st_.r0, o.c0 - st_.c0:o.c1 - st_.c0]
            assert_arrays_close(got, want, f"tile {rank} {name}", 0.0, 0.0)
        t.close()


@pytest.mark.parametrize("n", [6000, 90000])
def test_first_seen_signed_zero_is_kept_bit_for_bit(gpu, R, n):
    """-0.0 and +0.0 compare equal in the reference, so the FIRST zero a cell sees stays in min_z / max_z /
    max_intensity (elevation_mapping.cpp:65-79) and from there in elevation_min / _max / obstacle / intensity.
    Clouds of zeros and negatives in a few hundred cells, in both orders, through both pipelines: every layer
    bit-identical, the sign of every zero included."""
    def fill(c):
        c.mode = 1
        c.kalman_max_variance = 1.0

    eng, ref = pair(gpu, R, 40.0, 40.0, 0.1, fill)
    if n >= 65536:
        eng.set_option("tiled_min", 65536)  # (400x400 cells = 157 tiles: the engine would not pick the pipeline itself)
    rng = np.random.default_rng(3)
    for k in range(4):
        x = rng.uniform(-1.5, 1.5, n).astype(F32)
        y = rng.uniform(-1.5, 1.5, n).astype(F32)
        z = rng.choice(np.array([-0.0, 0.0, -0.5, -1.0, 0.0, -0.0], F32), n).astype(F32)
        v = rng.choice(np.array([-0.0, 0.0, -2.0, 0.0, -0.0, np.nan], F32), n).astype(F32)
        if k == 3:  # positive maxima overwrite zeros; zeros no longer decide
            z[::7] = 0.75
            v[::5] = 3.0
        run_both(eng, ref, {"x": x, "y": y, "z": z, "intensity": v}, T(), T(0.05 * k, 0.0))
        assert_layers_bit_identical(eng, ref)


def test_points4_host_entry_the_reference_cloud_layout(gpu, R):
    """fdm_engine_integrate_points4: FastDEM::integrate on nanopcl::PointCloud::points() — n contiguous {x, y, z, 1}
    records in HOST memory (pageable: copied; pinned: read in place), the optional channels as separate arrays —
    against the oracle's integrate of the same cloud (fastdem.cpp:122-190, point_cloud.hpp:126-134): status, statistics,
    cell ids, every layer bit for bit, geometry."""
    wl = gpu.synth.vlp16(n_scans=4)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    keep = []
    for k in range(4):
        s = wl.scan(k)
        n = int(s["x"].size)
        aos = np.empty((n, 4), dtype=np.float32)
        aos[:, 0], aos[:, 1], aos[:, 2], aos[:, 3] = s["x"], s["y"], s["z"], 1.0
        if k % 2:  # pinned: the records are read where they lie
            h = gpu.host_array(aos.reshape(-1), np.float32)
            keep.append(h)
            aos = h.array.reshape(n, 4)
        rc_e, st_e = eng.integrate_points4(aos, wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        rc_r, st_r = ref.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        assert (rc_e, st_e) == (rc_r, st_r)
        assert np.array_equal(eng.last_cell_ids(n), ref.last_cell_ids(n))
    assert_layers_bit_identical(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())
    # an empty cloud is skipped like the reference does (fastdem.cpp:125-128); a misaligned pointer is refused
    rc, _ = eng.integrate_points4(np.empty((0, 4), dtype=np.float32), wl.T_base_sensor, wl.pose(0))
    assert rc == gpu.capi.FDM_SKIP_EMPTY_CLOUD
    odd = np.zeros(4 * 8 + 1, dtype=np.float32)[1:].reshape(8, 4)
    with pytest.raises(gpu.EngineError):
        eng.integrate_points4(odd, wl.T_base_sensor, wl.pose(0))


@pytest.mark.parametrize("first", ["kalman", "p2"])
def test_p2_markers_sorted_as_libstdcxx_does_when_one_is_nan(gpu, R, first):
    """P2Quantile::updateP2 sorts its five markers with std::sort when the fifth sample arrives (quantile_estimation.hpp:150) —
    libstdc++'s insertion sort below 16 elements.  The estimators share `n_points` (fastdem.cpp:34-38: setEstimatorType at
    run time keeps the other estimator's layers): samples a Kalman estimator counted make P2 skip marker slots, and the
    fifth sample then sorts an array that holds NaN — where `NaN < x` is false both ways and the result is whatever the
    insertion sort's exact sequence of comparisons leaves.  Rounds 1-5 sorted with adjacent swaps carried all the way down:
    the same for finite values, another permutation with a NaN among them (every later quantile of the cell differed).
    Found by scripts/soak_oracle.py after 784 K scans (profiles/r06/soak_oracle.txt)."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -3.0, 3.0, 0.0, 30.0
        c.mode = 1                       # GLOBAL: the cells stay where they are
        c.estimation_type = 0 if first == "kalman" else 1
    eng, ref = pair(gpu, R, 8.0, 8.0, 0.1, fill)
    rng = np.random.default_rng(17)
    n = 1500
    x = rng.uniform(-3.5, 3.5, n).astype(F32)
    y = rng.uniform(-3.5, 3.5, n).astype(F32)
    Tbs = np.eye(4)

    def scan(z0):
        return {"x": x, "y": y, "z": (z0 + 0.3 * rng.standard_normal(n)).astype(F32), "intensity": None, "rgb": None}

    def switch(est):
        for o in (eng, ref):
            c = o.cfg
            c.estimation_type = est
            o.set_config(c)

    plan = ([("k", 2), ("p", 3), ("k", 1), ("p", 4)] if first == "kalman" else [("p", 1), ("k", 2), ("p", 4), ("k", 1), ("p", 3)])
    for est, scans in plan:
        switch(0 if est == "k" else 1)
        for _ in range(scans):
            run_both(eng, ref, scan(-0.2), Tbs, Tbs, check_ids=False)
    assert_layers_bit_identical(eng, ref)
    # the case is there: cells past their fifth sample that hold a NaN marker beside finite ones
    q = np.stack([eng.layer(f"_p2_q{k}") for k in range(5)])
    mixed = (eng.layer("n_points") >= 5) & np.isnan(q).any(axis=0) & np.isfinite(q).any(axis=0)
    assert mixed.sum() > 100, int(mixed.sum())
