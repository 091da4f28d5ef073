"""Raycasting INSIDE the batch pipeline (fastdem_amd/csrc/fdm_rbatch.hpp + the ray events of k_mbatch's update half):
an engine with raycast_enabled takes its small scans in batches of up to 16, the voxel filter + processScan of every
scan run in five launches per batch, and resolveGhostCells happens cell by cell, scan k's behind scan k's map update —
the order fastdem.cpp:125-159 fixes per scan.  Everything the reference leaves — every layer incl. raycasting /
ghost_removal / _visibility_logodds, the geometry, the statistics of the last scan — must be what the oracle leaves after
integrating the scans one by one.  Through the C ABI.

Run on the GPU box:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from helpers import assert_layers_bit_identical, pair, run_both, same_geometry
from test_batch_gpu import DeviceBatch, T, cloud

pytestmark = pytest.mark.gpu
F32 = np.float32


def ray_on(inner=None, **kw):
    def fill(cfg):
        if inner:
            inner(cfg)
        cfg.raycast_enabled = 1
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg
    return fill


def plant_ghost_block(o):
    e = o.layer("elevation")
    r, c = e.shape
    e[r // 2 + 8: r // 2 + 20, c // 2 - 6: c // 2 + 6] = 1.5  # phantom boxes the rays pass through,
    e[r // 2 - 20: r // 2 - 8, c // 2 - 6: c // 2 + 6] = 1.5  # behind and ahead of the robot
    o.set_layer("elevation", e)


def oracle_cleared(ref, scans, Tbs, poses):
    """The oracle scan by scan; how many cells its raycasting stages cleared."""
    cleared = 0
    rc = st = None
    for s, Twb in zip(scans, poses):
        kw = {c: s[c] for c in ("intensity", "rgb") if s.get(c) is not None}
        rc, st = ref.integrate(s["x"], s["y"], s["z"], Tbs, Twb, **kw)
        cleared += ref.last_ray_stats()["n_cleared"]
    return rc, st, cleared


def batch_vs_oracle(gpu, eng, ref, scans, Tbs, poses, expect_batches=True):
    eng.enable_cell_ids(False)
    before = sum(eng.batch_launches())
    b = DeviceBatch(gpu, scans, Tbs, poses)
    assert eng.integrate_device_batch(b.arr) == 0
    # (the `tiled_all` variant of the fixture pushes every scan through the record-pool pipeline, whose scans take the
    # raycasting stage one by one: the same comparison then covers that path)
    if expect_batches and not gpu.Engine.default_options.get("tiled_min"):
        assert sum(eng.batch_launches()) > before, "the call took no batch launch"
    rc_r, st_r, cleared = oracle_cleared(ref, scans, Tbs, poses)
    assert eng.last_stats() == (rc_r, st_r)
    assert_layers_bit_identical(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())
    return cleared


@pytest.mark.parametrize("walk", ["lds", "lds_parts=1", "lds_parts=7", "seg=4", "seg=1", "seg=16"])
def test_vlp16_stream_with_raycasting_in_batches(gpu, R, walk):
    """configs[1] under the shipped YAML's raycasting switch: 2 scans one by one (they create the layers), phantom
    obstacles planted, then 35 scans in one batch call (16 + 16 + 3) with a LOCAL shift every other scan.  Every variant
    of the ray walk: the per-quadrant LDS images (the default; 1 and 7 workgroups per quadrant) and the global-atomic
    walk with 4, 1 and 16 lanes per ray."""
    wl = gpu.synth.vlp16(n_scans=6)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, ray_on(wl.apply_to, rc_log_odds_ghost=1.2))
    if walk.startswith("lds_parts"):
        eng.set_option("batch_ray_parts", int(walk.split("=")[1]))
    elif walk.startswith("seg"):
        eng.set_option("batch_ray_lds", 0)
        eng.set_option("batch_ray_seg", int(walk.split("=")[1]))
    for k in range(2):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    for o in (eng, ref):
        plant_ghost_block(o)
    scans = [wl.scan(k) for k in range(2, 37)]
    poses = [wl.pose(k) for k in range(2, 37)]
    cleared = batch_vs_oracle(gpu, eng, ref, scans, wl.T_base_sensor, poses)
    assert cleared > 50, cleared
    assert np.nansum(eng.layer("ghost_removal")) > 0
    assert np.isfinite(eng.layer("raycasting")).sum() > 1000
    # the stream goes on scan by scan, then in another batch, on the same map
    for k in range(37, 39):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k), check_ids=False)
    batch_vs_oracle(gpu, eng, ref, [wl.scan(k) for k in range(39, 50)], wl.T_base_sensor, [wl.pose(k) for k in range(39, 50)])


def test_first_scans_of_a_fresh_engine_in_a_batch(gpu, R):
    """No scan by scan warm-up: the ray layers are born inside the batch call (visible from the first frame that runs)."""
    wl = gpu.synth.vlp16(n_scans=4)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, ray_on(wl.apply_to))
    batch_vs_oracle(gpu, eng, ref, [wl.scan(k) for k in range(20)], wl.T_base_sensor, [wl.pose(k) for k in range(20)])
    assert "raycasting" in eng.layers() and "_visibility_logodds" in eng.layers()


@pytest.mark.parametrize("batch_max", [2, 5, 16, 32])
def test_moves_strips_and_ghosts_inside_a_batch(gpu, R, batch_max):
    """Ragged small clouds under a pose sequence that stresses GridMap::move inside a batch (multi-cell shifts, a jump
    beyond the map, wrap-arounds) with aggressive ghost removal: cells are cleared by clearAt, vacated by strips and
    observed again inside one batch; frames whose sensor origin has left the map do not run."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 30.0

    eng, ref = pair(gpu, R, 8.0, 6.0, 0.1, ray_on(fill, rc_log_odds_ghost=0.9, rc_clear_threshold=-0.5,
                                                    rc_height_conflict_threshold=0.02))
    eng.set_option("batch_max", batch_max)
    rng = np.random.default_rng(5)
    steps = [(0, 0), (0.35, 0), (0.35, -0.4), (-1.2, 0.9), (-1.2, 0.9), (3.1, 3.3), (40.0, -25.0), (40.1, -25.0),
             (0.0, 0.0), (0.05, 0.04), (-7.9, 0.0), (-15.8, 0.0), (-23.7, 5.9), (-23.7, 11.8), (-23.65, 11.8),
             (2.0, 2.0), (2.0, 2.1), (2.1, 2.1), (2.1, 2.0), (2.0, 2.0), (10.0, 2.0), (10.0, -4.0), (4.0, -4.0),
             (4.0, -4.0), (4.05, -4.0), (4.05, -3.9), (4.0, -3.9), (4.0, -4.0), (4.0, -4.0), (4.0, -4.0)]
    scans, poses = [], []
    for k, (px, py) in enumerate(steps):
        n = int(rng.integers(200, 3000))
        s = cloud(rng, n, 4.5, intensity=True)
        s["z"] = (s["z"] + F32(0.6) * (rng.uniform(size=n) < 0.3)).astype(F32)  # bumps that later rays pass through
        scans.append(s)
        poses.append(T(px, py, 0.0, yaw=0.1 * k))
    cleared = batch_vs_oracle(gpu, eng, ref, scans, T(z=1.5), poses)
    assert cleared > 0


def test_p2_colour_scans_with_raycasting(gpu, R):
    """The quantile estimator + colour + intensity policy of the batch kernel with ray events (the elevation a ray event
    compares is the P2 marker, NaN until five samples)."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 30.0
        c.estimation_type = 1
        c.sensor_type = 2

    eng, ref = pair(gpu, R, 10.0, 10.0, 0.1, ray_on(fill, rc_log_odds_ghost=0.9, rc_clear_threshold=-0.5,
                                                      rc_height_conflict_threshold=0.02))
    rng = np.random.default_rng(9)
    scans, poses = [], []
    for k in range(26):
        n = int(rng.integers(3000, 20000))
        s = cloud(rng, n, 5.5, intensity=True, rgb=True)
        s["z"] = (s["z"] + F32(0.7) * (rng.uniform(size=n) < 0.25)).astype(F32)
        scans.append(s)
        poses.append(T(0.07 * k, -0.05 * k, 0.0, yaw=0.03 * k))
    cleared = batch_vs_oracle(gpu, eng, ref, scans, T(z=1.4), poses)
    assert cleared > 0


def test_frames_that_do_not_run_and_scans_that_leave_the_batch(gpu, R):
    """GLOBAL map: scans whose sensor origin is outside the map (stage skipped, raycasting.cpp:217-220), a scan with
    every point filtered (empty voxel cloud: skipped, the layer keeps the previous frame), and one scan beyond the
    sort-free filter's 64 K points in the middle of the call (it leaves the batch and takes the one-scan path)."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 40.0
        c.mode = 1

    eng, ref = pair(gpu, R, 12.0, 12.0, 0.1, ray_on(fill, rc_log_odds_ghost=0.9, rc_clear_threshold=-0.5))
    rng = np.random.default_rng(21)
    scans, poses = [], []
    for k in range(24):
        n = 70000 if k == 13 else int(rng.integers(500, 6000))
        s = cloud(rng, n, 7.0, intensity=True)
        s["z"] = (s["z"] + F32(0.5) * (rng.uniform(size=n) < 0.3)).astype(F32)
        px, py = (0.2 * k, -0.1 * k)
        if k in (4, 5, 17):
            px, py = 9.0 + k, 2.0   # the robot (and its sensor) outside the map; some points still land in it
            s["x"] = (s["x"] - F32(px)).astype(F32)
        if k == 8:
            s["z"] = (s["z"] + F32(50.0)).astype(F32)  # every point filtered by cropZ
        scans.append(s)
        poses.append(T(px, py, 0.0, yaw=0.02 * k))
    batch_vs_oracle(gpu, eng, ref, scans, T(z=1.2), poses)


def test_option_off_takes_the_one_scan_path_and_agrees(gpu, R):
    wl = gpu.synth.vlp16(n_scans=4)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, ray_on(wl.apply_to))
    eng.set_option("batch_ray", 0)
    eng.enable_cell_ids(False)
    before = sum(eng.batch_launches())
    scans, poses = [wl.scan(k) for k in range(6)], [wl.pose(k) for k in range(6)]
    b = DeviceBatch(gpu, scans, wl.T_base_sensor, poses)
    assert eng.integrate_device_batch(b.arr) == 0
    assert sum(eng.batch_launches()) == before
    oracle_cleared(ref, scans, wl.T_base_sensor, poses)
    assert_layers_bit_identical(eng, ref)


def test_large_map_and_a_sensor_off_the_centre(gpu, R):
    """40 x 40 m at 0.1 m (160 K cells): a quadrant of the map does not fit the LDS image of the default walk, the
    batch takes the global-atomic walk; then a 24 x 24 m GLOBAL map whose sensor wanders from the centre to a corner:
    the LDS walk's quadrant rectangles grow from a quarter of the map to all of it."""
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -2.0, 3.0, 0.0, 40.0
        c.mode = 1

    rng = np.random.default_rng(33)
    for size, spread in ((40.0, 22.0), (24.0, 13.0)):
        eng, ref = pair(gpu, R, size, size, 0.1, ray_on(fill, rc_log_odds_ghost=0.9, rc_clear_threshold=-0.5))
        scans, poses = [], []
        for k in range(18):
            n = int(rng.integers(2000, 9000))
            s = cloud(rng, n, spread, intensity=True)
            s["z"] = (s["z"] + F32(0.5) * (rng.uniform(size=n) < 0.3)).astype(F32)
            scans.append(s)
            d = (size / 2 - 0.3) * k / 17.0
            poses.append(T(d, -d, 0.0, yaw=0.05 * k))
            s["x"] = (s["x"] - F32(d) * F32(0.5)).astype(F32)
        batch_vs_oracle(gpu, eng, ref, scans, T(z=1.3), poses)
