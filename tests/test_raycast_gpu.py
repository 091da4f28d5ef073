"""Raycasting stage of the HIP engine, through the C ABI, against the CPU oracle.

* the reference's own raycasting tests (fastdem/tests/test_postprocess.cpp:73-190) re-run on the engine
* voxelGrid(ANY): selected point indices bit-exact vs the oracle's stable-tie variant, and a valid
  "ANY" outcome vs the reference-literal std::sort variant (same voxels, a member of each)
* applyRaycasting and integrate()+raycasting: every layer (incl. `raycasting`, `_visibility_logodds`,
  `ghost_removal`, and the clearAt NaNs in the estimator layers) identical to the oracle
"""
import numpy as np
import pytest

from helpers import assert_layers_equal, pair, run_both, same_geometry

pytestmark = pytest.mark.gpu
F32 = np.float32


def ray_cfg(**kw):
    def fill(cfg):
        cfg.raycast_enabled = 1
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg
    return fill


def post_pair(gpu, R, **kw):
    """PostprocessTest fixture: 10x10 m @ 0.5 (test_postprocess.cpp:24-36), engine + oracle."""
    return pair(gpu, R, 10.0, 10.0, 0.5, ray_cfg(**kw))


def set_cell(objs, name, rc, v):
    for o in objs:
        a = o.layer(name)
        a[rc] = v
        o.set_layer(name, a)


def both(objs, fn):
    return [fn(o) for o in objs]


ORIGIN = [0.0, 0.0, 5.0]


# ------------------------------------------------- the reference's own tests ----
class TestReferenceRaycastingTestsOnEngine:
    def test_creates_layers(self, gpu, R):  # test_postprocess.cpp:75-92
        eng, ref = post_pair(gpu, R)
        _, c = ref.get_index(0.0, 0.0)
        set_cell((eng, ref), "elevation", c, 1.0)
        both((eng, ref), lambda o: o.apply_raycasting([1.0], [0.0], [0.5], ORIGIN))
        for n in ("ghost_removal", "raycasting", "_visibility_logodds"):
            assert eng.exists(n)
        assert_layers_equal(eng, ref)

    def test_clears_ghost_cell(self, gpu, R):  # :94-117
        eng, ref = post_pair(gpu, R, rc_height_conflict_threshold=0.05, rc_log_odds_ghost=0.5,
                             rc_clear_threshold=-0.4)
        _, g = ref.get_index(2.0, 0.0)
        set_cell((eng, ref), "elevation", g, 10.0)
        both((eng, ref), lambda o: o.apply_raycasting([4.0], [0.0], [0.0], ORIGIN))
        assert np.isnan(eng.layer("elevation")[g]) and eng.layer("ghost_removal")[g] == F32(1.0)
        assert_layers_equal(eng, ref)

    def test_observed_cell_protected(self, gpu, R):  # :119-146
        eng, ref = post_pair(gpu, R, rc_log_odds_observed=0.8, rc_log_odds_ghost=0.5, rc_clear_threshold=-0.4)
        _, c = ref.get_index(2.0, 0.0)
        set_cell((eng, ref), "elevation", c, 2.0)
        both((eng, ref), lambda o: o.apply_raycasting([4.0, 2.0], [0.0, 0.0], [0.0, 0.3], ORIGIN))
        assert not np.isnan(eng.layer("elevation")[c])
        assert_layers_equal(eng, ref)

    def test_ghost_requires_accumulation(self, gpu, R):  # :148-175
        eng, ref = post_pair(gpu, R, rc_log_odds_ghost=0.2, rc_clear_threshold=-0.9)
        _, g = ref.get_index(2.0, 0.0)
        for _ in range(4):
            set_cell((eng, ref), "elevation", g, 10.0)
            both((eng, ref), lambda o: o.apply_raycasting([4.0], [0.0], [0.0], ORIGIN))
        assert not np.isnan(eng.layer("elevation")[g])
        assert_layers_equal(eng, ref)
        both((eng, ref), lambda o: o.apply_raycasting([4.0], [0.0], [0.0], ORIGIN))
        assert np.isnan(eng.layer("elevation")[g])
        assert_layers_equal(eng, ref)

    def test_disabled_is_noop(self, gpu, R):  # :177-190
        eng, ref = post_pair(gpu, R, raycast_enabled=0)
        both((eng, ref), lambda o: o.apply_raycasting([1.0], [0.0], [0.5], ORIGIN))
        for n in ("ghost_removal", "raycasting", "_visibility_logodds"):
            assert not eng.exists(n)


# --------------------------------------------------------------- voxelGrid ANY ----
def check_voxel(gpu, R, x, y, z, size):
    eng = gpu.Engine(4.0, 4.0, 0.5)
    sel = eng.voxel_any(x, y, z, size)
    want = R.voxel_any(x, y, z, size, stable=True)
    assert np.array_equal(sel, want), f"{(sel != want).sum() if sel.size == want.size else 'size'} differ"
    # vs the reference-literal std::sort variant: same voxels, one member of each
    lit = R.voxel_any(x, y, z, size, stable=False)
    assert lit.size == sel.size
    inv = F32(1.0) / F32(size)
    xs, ys, zs = (np.asarray(v, dtype=F32) for v in (x, y, z))
    for a in (xs, ys, zs):
        ka, kb = np.floor(a[sel] * inv), np.floor(a[lit] * inv)
        assert np.array_equal(ka, kb)
    return sel


class TestVoxelGridAny:
    def test_random_cloud(self, gpu, R):
        rng = np.random.default_rng(3)
        n = 200_000
        x, y, z = (rng.uniform(-6, 6, n).astype(F32) for _ in range(3))
        sel = check_voxel(gpu, R, x, y, z, 0.25)
        assert 1000 < sel.size < n

    @pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8193,
                                   65535, 65536, 65537, 262143, 262145, 599999, 600000, 600001, 1048577])
    def test_sizes_around_the_sort_s_tile_and_wavefront_edges(self, gpu, R, n):
        """The radix sort walks tiles of 4096 pairs (1024 up to 600 000 pairs), wavefronts of a quarter tile, rounds of
        64: sizes on either side of every edge, many duplicates of a few keys among unique ones (runs that straddle
        tiles)."""
        rng = np.random.default_rng(n)
        x, y, z = (rng.uniform(-30, 30, n).astype(F32) for _ in range(3))
        m = n // 3
        if m:
            x[:m] = rng.integers(0, 4, m).astype(F32) * 0.5
            y[:m] = 1.0
            z[:m] = -2.0
            rng.shuffle(x)
        check_voxel(gpu, R, x, y, z, 0.3)

    def test_five_million_points(self, gpu, R):
        """More than 1024 tiles of 4096 pairs: the radix sort's per-bin scan over the tiles carries across its steps
        (fdm_rsort.hpp k_rs_scan); 64-bit keys, eight passes."""
        rng = np.random.default_rng(5)
        n = 5_000_000
        x, y = (rng.uniform(-40, 40, n).astype(F32) for _ in range(2))
        z = rng.uniform(-1, 3, n).astype(F32)
        sel = check_voxel(gpu, R, x, y, z, 0.2)
        assert 100_000 < sel.size < n

    def test_dense_voxels_and_long_runs(self, gpu, R):
        rng = np.random.default_rng(4)
        n = 100_000  # ~100 voxels: runs of ~1000 points exercise the bisection of voxel_pick
        x, y = (rng.uniform(0, 1, n).astype(F32) for _ in range(2))
        z = np.zeros(n, dtype=F32)
        check_voxel(gpu, R, x, y, z, 0.1)

    def test_mixed_run_lengths_across_block_boundaries(self, gpu, R):
        # runs of 1 .. ~700 sorted positions: ends inside the head's block, in the 256 positions
        # after it (second boundary ballot) and beyond (bisection); NaNs make an invalid tail
        rng = np.random.default_rng(14)
        parts = []
        for v in range(400):
            m = int(rng.integers(1, 700))
            c = rng.uniform(-8, 8, 3)
            parts.append(c + rng.uniform(0, 0.09, (m, 3)))
        pts = np.concatenate(parts).astype(F32)
        pts = pts[rng.permutation(len(pts))]
        pts[rng.integers(0, len(pts), 500), 0] = np.nan
        check_voxel(gpu, R, pts[:, 0], pts[:, 1], pts[:, 2], 0.1)

    def test_everything_in_one_voxel(self, gpu, R):
        n = 70_001
        x = np.full(n, 0.01, dtype=F32)
        sel = check_voxel(gpu, R, x, x, x, 1.0)
        assert sel.size == 1 and sel[0] == (n * 7) % n

    def test_nonfinite_points_are_dropped(self, gpu, R):  # voxel_grid_impl.hpp:52-54
        x = np.array([1.0, np.nan, 4.0, np.inf, 1.01, -np.inf], dtype=F32)
        y = np.array([2.0, 0.0, 5.0, 0.0, 2.01, 0.0], dtype=F32)
        z = np.array([3.0, 0.0, 6.0, 0.0, 3.01, 0.0], dtype=F32)
        sel = check_voxel(gpu, R, x, y, z, 1.0)
        assert set(sel) <= {0, 2, 4} and sel.size == 2

    def test_single_empty_and_range(self, gpu, R):  # test_filters.cpp:786-798, voxel_grid_impl.hpp:31-33
        eng = gpu.Engine(4.0, 4.0, 0.5)
        assert eng.voxel_any([], [], [], 1.0).size == 0
        assert list(eng.voxel_any([1.0], [2.0], [3.0], 1.0)) == [0]
        for bad in (0.0005, 100.5):
            with pytest.raises(gpu.EngineError):
                eng.voxel_any([0.0], [0.0], [0.0], bad)

    def test_clamped_and_negative_coordinates(self, gpu, R):  # test_voxel.cpp:53-85
        x = np.array([-5.5, 2.0e6, -2.0e6, 3.0e9, -3.0e9, -0.05, 0.05], dtype=F32)
        y = np.array([-10.3, 0.0, 0.0, 0.0, 0.0, -0.05, 0.05], dtype=F32)
        z = np.array([-0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=F32)
        check_voxel(gpu, R, x, y, z, 1.0)

    def test_preprocessed_lidar_scan(self, gpu, R):
        wl = gpu.synth.vlp16(n_scans=1)
        s = wl.scan(0)
        check_voxel(gpu, R, s["x"], s["y"], s["z"], 0.1)


# ------------------------------------------------------------ applyRaycasting ----
def random_scene(rng, n, span, zlo, zhi):
    x, y = (rng.uniform(-span, span, n).astype(F32) for _ in range(2))
    z = rng.uniform(zlo, zhi, n).astype(F32)
    return x, y, z


class TestApplyRaycasting:
    def test_random_cloud_on_a_populated_map(self, gpu, R):
        rng = np.random.default_rng(11)
        eng, ref = pair(gpu, R, 20.0, 20.0, 0.1, ray_cfg(rc_log_odds_ghost=0.6, rc_clear_threshold=-1.0))
        elev = rng.uniform(-0.5, 2.5, (eng.rows, eng.cols)).astype(F32)
        elev[rng.uniform(size=elev.shape) < 0.3] = np.nan
        both((eng, ref), lambda o: o.set_layer("elevation", elev))
        for frame in range(3):  # logodds accumulates; some cells clear on frame 2
            x, y, z = random_scene(rng, 30_000, 12.0, -0.5, 3.0)  # some targets outside the map, some above the sensor
            both((eng, ref), lambda o: o.apply_raycasting(x, y, z, [0.3, -0.2, 2.0]))
            assert_layers_equal(eng, ref)
        st = ref.last_ray_stats()
        assert st["n_cleared"] > 0 and st["n_conflicts"] > st["n_cleared"]
        assert np.nansum(eng.layer("ghost_removal")) > 0

    def test_duplicate_points_fold_observed_evidence(self, gpu, R):
        eng, ref = post_pair(gpu, R, rc_log_odds_observed=0.7, rc_log_odds_max=2.0)
        x = np.full(5, 1.0, dtype=F32)
        both((eng, ref), lambda o: o.apply_raycasting(x, x * 0, x * 0 + 6.0, ORIGIN))  # 5 x +0.7, clamped at 2
        _, c = ref.get_index(1.0, 0.0)
        assert eng.layer("_visibility_logodds")[c] == F32(2.0)
        assert_layers_equal(eng, ref)

    def test_wrapped_buffer_and_moved_map(self, gpu, R):
        rng = np.random.default_rng(12)
        eng, ref = pair(gpu, R, 12.0, 12.0, 0.1, ray_cfg())
        both((eng, ref), lambda o: o.move(1.7, -2.3))
        assert same_geometry(eng.geometry(), ref.geometry())
        elev = rng.uniform(0.0, 2.0, (eng.rows, eng.cols)).astype(F32)
        both((eng, ref), lambda o: o.set_layer("elevation", elev))
        x, y, z = random_scene(rng, 20_000, 8.0, -0.2, 1.5)
        both((eng, ref), lambda o: o.apply_raycasting(x + 1.7, y - 2.3, z, [1.7, -2.3, 1.8]))
        assert np.isfinite(eng.layer("raycasting")).sum() > 1000
        assert_layers_equal(eng, ref)

    def test_sensor_outside_map_is_noop(self, gpu, R):  # raycasting.cpp:217-220
        eng, ref = post_pair(gpu, R)
        both((eng, ref), lambda o: o.apply_raycasting([1.0], [0.0], [0.5], [50.0, 0.0, 5.0]))
        assert not eng.exists("raycasting") and not ref.exists("raycasting")
        assert_layers_equal(eng, ref)

    def test_device_entry_point(self, gpu, R):
        import torch
        rng = np.random.default_rng(13)
        eng, ref = post_pair(gpu, R)
        x, y, z = random_scene(rng, 5000, 6.0, -1.0, 4.0)
        dx, dy, dz = (torch.from_numpy(v).cuda() for v in (x, y, z))
        eng.apply_raycasting_device(dx, dy, dz, ORIGIN)
        eng.sync()
        ref.apply_raycasting(x, y, z, ORIGIN)
        assert_layers_equal(eng, ref)


class TestSectorWindowWalk:
    """Large scans walk with an angular sector's minimum-height image in LDS (fdm_raywedge.hpp): option
    `ray_large_min` = 1 sends scans of any size that way.  The window decides only HOW a visit is stored; these cases
    put rays on both sides of every decision it takes."""

    def wedge_pair(self, gpu, R, w, h, res, **opts):
        eng, ref = pair(gpu, R, w, h, res, ray_cfg(rc_log_odds_ghost=0.6, rc_clear_threshold=-1.0))
        eng.set_option("ray_large_min", 1)
        for k, v in opts.items():
            eng.set_option(k, v)
        return eng, ref

    def test_dense_cloud_all_directions(self, gpu, R):
        rng = np.random.default_rng(41)
        eng, ref = self.wedge_pair(gpu, R, 40.0, 40.0, 0.1)
        elev = rng.uniform(-0.5, 2.5, (eng.rows, eng.cols)).astype(F32)
        elev[rng.uniform(size=elev.shape) < 0.3] = np.nan
        both((eng, ref), lambda o: o.set_layer("elevation", elev))
        for frame in range(2):
            x, y, z = random_scene(rng, 250_000, 22.0, -0.5, 3.0)  # targets inside and outside the map, some above the sensor
            both((eng, ref), lambda o: o.apply_raycasting(x, y, z, [0.31, -0.17, 2.0]))
            assert_layers_equal(eng, ref)
        assert np.isfinite(eng.layer("raycasting")).sum() > 100_000

    @pytest.mark.parametrize("origin", [[-28.5, 27.9, 2.5], [29.3, 0.04, 1.5], [0.0, -29.95, 3.0]])
    def test_sensor_at_the_edge_rays_longer_than_the_window(self, gpu, R, origin):
        """60 m map at 0.05 m (1200 cells): from a corner the rays run up to 1 600 cells — the window has 616 rows,
        the rest of such a ray goes on with memory-side atomics from the state the window walk left."""
        rng = np.random.default_rng(42)
        eng, ref = self.wedge_pair(gpu, R, 60.0, 60.0, 0.05)
        x, y, z = random_scene(rng, 60_000, 31.0, -1.0, 1.0)
        both((eng, ref), lambda o: o.apply_raycasting(x, y, z, origin))
        assert np.isfinite(eng.layer("raycasting")).sum() > 100_000
        assert_layers_equal(eng, ref)

    @pytest.mark.parametrize("n", [1, 7, 300, 5000])
    def test_sparse_scans_whose_workgroups_span_wide_angles(self, gpu, R, n):
        rng = np.random.default_rng(43 + n)
        eng, ref = self.wedge_pair(gpu, R, 30.0, 30.0, 0.1)
        x, y, z = random_scene(rng, n, 14.0, -1.0, 1.0)
        both((eng, ref), lambda o: o.apply_raycasting(x, y, z, [0.0, 0.0, 1.5]))
        assert_layers_equal(eng, ref)

    def test_one_narrow_beam_and_axis_aligned_rays(self, gpu, R):
        """Every ray in one direction (all lanes of a workgroup in the same cells), rays exactly along the axes and the
        diagonals (ties of the DDA at every step), zero-length rays."""
        rng = np.random.default_rng(44)
        eng, ref = self.wedge_pair(gpu, R, 30.0, 30.0, 0.1)
        n = 20_000
        rad = rng.uniform(0.5, 14.0, n).astype(F32)
        ang = F32(0.7) + rng.uniform(-0.002, 0.002, n).astype(F32)
        x, y = (rad * np.cos(ang)).astype(F32), (rad * np.sin(ang)).astype(F32)
        z = rng.uniform(-1.0, 0.5, n).astype(F32)
        ax = np.linspace(-14.0, 14.0, 400).astype(F32)
        zero = np.zeros_like(ax)
        x = np.concatenate([x, ax, zero, ax, ax, np.full(8, 0.05, F32)])
        y = np.concatenate([y, zero, ax, ax, -ax, np.full(8, 0.05, F32)])
        z = np.concatenate([z, zero - 1, zero - 1, zero - 1, zero - 1, np.full(8, 0.0, F32)])
        for origin in ([0.05, 0.05, 1.5], [0.0, 0.0, 1.5]):
            both((eng, ref), lambda o: o.apply_raycasting(x, y, z, origin))
            assert_layers_equal(eng, ref)

    def test_the_window_and_the_memory_side_walk_agree(self, gpu, R):
        """engine against engine at configs[3]'s density: `ray_wedge` 0 is the round-1..4 walk (one lane per ray on
        memory-side atomics), `dbg_ray` 64 keeps every ray of the new kernel out of the window."""
        wl = gpu.synth.lidar128(n_scans=1, n_az=4096)
        s = wl.scan(0)
        out = []
        for opts in ({}, {"ray_wedge": 0}, {"dbg_ray": 64}):
            def fill(cfg):
                wl.apply_to(cfg)
                return ray_cfg()(cfg)
            eng = gpu.Engine(wl.width, wl.height, wl.resolution, fill(gpu.capi.default_config()))
            eng.set_option("ray_large_min", 1)
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(0), intensity=s.get("intensity"))
            out.append({n: eng.layer(n) for n in ("raycasting", "_visibility_logodds", "elevation")})
        for other in out[1:]:
            for n, a in out[0].items():
                assert np.array_equal(a.view(np.uint32), other[n].view(np.uint32)), n
        assert np.isfinite(out[0]["raycasting"]).sum() > 10_000


# ------------------------------------------------- integrate() with raycasting ----
def run_ray_workload(gpu, R, wl, n_scans, ghosts=None, **rc):
    def fill(cfg):
        wl.apply_to(cfg)
        return ray_cfg(**rc)(cfg)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
    cleared = 0
    for k in range(n_scans):
        if ghosts is not None and k == 1:  # plant phantom obstacles after the first scan
            both((eng, ref), ghosts)
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
        assert_layers_equal(eng, ref)
        assert same_geometry(eng.geometry(), ref.geometry())
        cleared += ref.last_ray_stats()["n_cleared"]
    return eng, ref, cleared


def plant_ghost_block(o):
    e = o.layer("elevation")
    r, c = e.shape
    e[r // 2 + 8: r // 2 + 20, c // 2 - 6: c // 2 + 6] = 1.5  # phantom boxes the rays pass through,
    e[r // 2 - 20: r // 2 - 8, c // 2 - 6: c // 2 + 6] = 1.5  # behind and ahead of the robot
    o.set_layer("elevation", e)


class TestIntegrateWithRaycasting:
    def test_tilted_base_keeps_the_compact_voxel_key_exact(self, gpu, R):
        """The compact sort key bounds z by the cropZ slab pushed through T_world_base (roll / pitch
        widen it by hypot(R20, R21) * range_max): a tilted, lifted robot must give the oracle's voxels."""
        wl = gpu.synth.vlp16(n_scans=4)

        def fill(cfg):
            wl.apply_to(cfg)
            return ray_cfg()(cfg)
        eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
        for k in range(4):
            a, b = np.deg2rad(9.0 * (k + 1) / 4), np.deg2rad(-6.0)
            Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
            Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
            T = wl.pose(k).copy()
            T[:3, :3] = T[:3, :3] @ Rx @ Ry
            T[2, 3] += 0.35 * k
            run_both(eng, ref, wl.scan(k), wl.T_base_sensor, T)
            assert_layers_equal(eng, ref)
        assert ref.last_ray_stats()["n_rays"] > 1000

    def test_c2_vlp16_kalman_local(self, gpu, R):
        wl = gpu.synth.vlp16(n_scans=8)
        eng, ref, cleared = run_ray_workload(gpu, R, wl, 8, ghosts=plant_ghost_block,
                                             rc_log_odds_ghost=0.6, rc_clear_threshold=-1.0)
        assert cleared > 0 and np.nansum(eng.layer("ghost_removal")) > 0
        assert {"ghost_removal", "raycasting", "_visibility_logodds"} <= set(eng.layers())

    @pytest.mark.parametrize("voxel_small", [1, 0])
    def test_small_scans_without_and_with_the_library_sort(self, gpu, R, voxel_small):
        """Scans of <= 64 K points take the sort-free voxel filter (k_vs_count / k_vs_scatter / k_vs_mark: buckets of
        the compact key's top bits, every point ranks itself inside its bucket); option voxel_small 0 sends them
        through the stable radix sort instead.  Both must select the oracle's representatives: dense clusters
        (hundreds of points per voxel and thousands per bucket), duplicates, dropped points, ragged sizes."""
        def fill(cfg):
            cfg.z_min, cfg.z_max, cfg.range_min, cfg.range_max = -2.0, 4.0, 0.2, 12.0
            return ray_cfg(rc_log_odds_ghost=0.8, rc_clear_threshold=-1.0)(cfg)
        eng, ref = pair(gpu, R, 16.0, 16.0, 0.1, fill)
        eng.set_option("voxel_small", voxel_small)
        rng = np.random.default_rng(17)
        Tbs = np.eye(4)
        Tbs[2, 3] = 1.2
        for k, n in enumerate((1, 63, 257, 5000, 20000, 40000, 65536)):
            x = rng.uniform(-7.0, 7.0, n).astype(F32)
            y = rng.uniform(-7.0, 7.0, n).astype(F32)
            z = (rng.uniform(-1.0, 0.4, n) - 1.2).astype(F32)
            m = n // 3  # a third of the points in a 0.6 m cube: ~200 voxels share them; exact duplicates among them
            x[:m] = (2.0 + rng.uniform(0, 0.6, m)).astype(F32)
            y[:m] = (-1.0 + rng.uniform(0, 0.6, m)).astype(F32)
            z[:m] = (-1.5 + rng.uniform(0, 0.6, m)).astype(F32)
            if m > 10:
                x[5:m:7], y[5:m:7], z[5:m:7] = x[4], y[4], z[4]
            z[n // 2::11] += 30.0  # dropped by cropZ: x = NaN in the stage's input
            T = np.eye(4)
            T[0, 3], T[1, 3] = 0.13 * k, -0.07 * k
            run_both(eng, ref, {"x": x, "y": y, "z": z, "intensity": None, "rgb": None}, Tbs, T)
            assert_layers_equal(eng, ref)
        assert ref.last_ray_stats()["n_rays"] > 1000

    def test_sort_free_filter_beyond_64k_points(self, gpu, R):
        """Option voxel_small_max lifts the size limit of the sort-free filter (17-bit point indices, buckets far larger
        than the register window: the batched walk over memory)."""
        def fill(cfg):
            cfg.z_min, cfg.z_max, cfg.range_min, cfg.range_max = -2.0, 4.0, 0.2, 12.0
            return ray_cfg(rc_log_odds_ghost=0.8, rc_clear_threshold=-1.0)(cfg)
        eng, ref = pair(gpu, R, 16.0, 16.0, 0.1, fill)
        eng.set_option("voxel_small_max", 1 << 20)
        rng = np.random.default_rng(23)
        n = 100_000
        x = rng.uniform(-7.0, 7.0, n).astype(F32)
        y = (np.round(rng.uniform(-7.0, 7.0, n) * 2) / 2).astype(F32)   # 29 rows of voxels hold everything
        z = (np.round(rng.uniform(-1.0, 0.4, n) * 4) / 4 - 1.2).astype(F32)
        Tbs = np.eye(4)
        Tbs[2, 3] = 1.2
        run_both(eng, ref, {"x": x, "y": y, "z": z, "intensity": None, "rgb": None}, Tbs, np.eye(4))
        assert_layers_equal(eng, ref)

    def test_c3_rgbd_p2_records_cleared(self, gpu, R):
        """P2 cell records (128 B) wiped by clearAt.  The depth image observes every cell it can ray
        through except the strip between the camera and the nearest ground hit: ghosts go there."""
        def plant_under_camera(o):
            e = o.layer("elevation")
            r, c = e.shape
            e[r // 2 - 12: r // 2 + 3, c // 2 - 4: c // 2 + 5] = 1.5
            o.set_layer("elevation", e)
        wl = gpu.synth.rgbd(n_scans=4)
        eng, ref, cleared = run_ray_workload(gpu, R, wl, 4, ghosts=plant_under_camera,
                                             rc_log_odds_ghost=1.2, rc_clear_threshold=-1.0)
        assert cleared > 0
        g = eng.layer("ghost_removal") == 1.0
        assert g.any() and np.isnan(eng.layer("_p2_n0")[g]).all()

    def test_c4_lidar128_reduced(self, gpu, R):
        wl = gpu.synth.lidar128(n_scans=3, n_az=2048)
        run_ray_workload(gpu, R, wl, 3, ghosts=plant_ghost_block, rc_log_odds_ghost=1.2)

    @pytest.mark.parametrize("order", ["azimuth", "ring"])
    def test_c4_lidar128_full_size_sorted_ray_queue(self, gpu, R, order):
        """configs[3] at full size (2.1 M points): from 2^20 points up the ray queue is sorted by
        (direction wedge, length) before the walk — the result must not depend on the queue's order."""
        wl = gpu.synth.lidar128(n_scans=2, order=order)
        run_ray_workload(gpu, R, wl, 2, ghosts=plant_ghost_block, rc_log_odds_ghost=1.2)

    @pytest.mark.parametrize("overlap", [1, 0, -1])
    def test_two_stages_in_flight(self, gpu, R, overlap):
        """Option ray_overlap: voxel filter, queue and walk of a large scan leave on a stream of their own as soon as the
        scan's bin half has run — on the geometry its update WILL commit (moves, a scan that moves nothing because every
        point is filtered) — while k_ray_resolve stays behind the update; stages of consecutive scans overlap by scan
        parity.  Streams of enqueue-only calls, synchronous calls in between, a flush in the middle of the stream, scans
        whose sizes differ (the buffers of either parity grow at different times): every layer equals the oracle's."""
        def fill(cfg):
            cfg.z_min, cfg.z_max, cfg.range_min, cfg.range_max = -2.0, 4.0, 0.2, 14.0
            return ray_cfg(rc_log_odds_ghost=0.8, rc_clear_threshold=-1.0)(cfg)
        eng, ref = pair(gpu, R, 24.0, 24.0, 0.1, fill)
        eng.set_option("ray_overlap", overlap)
        eng.set_option("ray_large_min", 1)      # every scan takes the large path ...
        eng.set_option("voxel_small", 0)        # ... and the sort (the sort-free filter never overlaps)
        rng = np.random.default_rng(101 + overlap)
        Tbs = np.eye(4)
        Tbs[2, 3] = 1.1
        sizes = (30000, 1200, 52000, 52000, 7, 40000, 90000, 90000, 3000, 61000, 61000, 20000)
        scans, poses = [], []
        for k, n in enumerate(sizes):
            x = rng.uniform(-11.0, 11.0, n).astype(F32)
            y = rng.uniform(-11.0, 11.0, n).astype(F32)
            z = (rng.uniform(-1.0, 0.5, n) - 1.1).astype(F32)
            if k == 4:
                z[:] = 50.0   # everything filtered: no move, no stage
            if k % 3 == 1:    # a ghost block for the stage to find
                x[: n // 10] = F32(3.0) + rng.uniform(0, 0.5, n // 10).astype(F32)
                y[: n // 10] = F32(-2.0) + rng.uniform(0, 0.5, n // 10).astype(F32)
                z[: n // 10] = F32(0.9 - 1.1)
            T = np.eye(4)
            T[0, 3], T[1, 3] = 0.37 * k, -0.23 * k   # LOCAL map: a move of a few cells with every scan
            scans.append({"x": x, "y": y, "z": z, "intensity": rng.uniform(0, 1, n).astype(F32), "rgb": None})
            poses.append(T)
        for s_, T in zip(scans, poses):
            ref.integrate(s_["x"], s_["y"], s_["z"], Tbs, T, intensity=s_["intensity"])
        from test_batch_gpu import DeviceBatch
        b = DeviceBatch(gpu, scans[:5], Tbs, poses[:5])
        assert eng.integrate_device_batch(b.arr) == 0                      # a stream of enqueue-only scans
        eng.integrate(scans[5]["x"], scans[5]["y"], scans[5]["z"], Tbs, poses[5], intensity=scans[5]["intensity"])  # synchronous
        b2 = DeviceBatch(gpu, scans[6:9], Tbs, poses[6:9])
        assert eng.integrate_device_batch(b2.arr) == 0
        eng.sync()                                                          # a flush with a stage in flight
        b3 = DeviceBatch(gpu, scans[9:], Tbs, poses[9:])
        for k in range(len(scans) - 9):                                     # one enqueue-only call per scan
            one = (gpu.capi.FdmDeviceScan * 1)(b3.arr[k])
            assert eng.integrate_device_batch(one) == 0
        assert_layers_equal(eng, ref)
        assert ref.last_ray_stats()["n_rays"] > 1000

    def test_per_layer_storage(self, gpu, R):
        wl = gpu.synth.vlp16(n_scans=4)

        def fill(cfg):
            wl.apply_to(cfg)
            return ray_cfg(rc_log_odds_ghost=1.2)(cfg)
        ce, cr = gpu.capi.default_config(), R.default_config()
        fill(ce), fill(cr)
        eng = gpu.Engine(wl.width, wl.height, wl.resolution, ce)
        eng.set_option("records", 0)
        eng.set_config(ce)
        ref = R.RefEngine(wl.width, wl.height, wl.resolution, cr)
        for k in range(4):
            if k == 1:
                both((eng, ref), plant_ghost_block)
            run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k), check_ids=False)
            assert_layers_equal(eng, ref)

    def test_global_mode_sensor_leaves_the_map(self, gpu, R):
        """GLOBAL map at the origin; the robot drives out of it: while the sensor is outside the
        stage must not run (raycasting.cpp:217-220) — layers appear only once it has been inside."""
        wl = gpu.synth.vlp16(n_scans=1)

        def fill(cfg):
            wl.apply_to(cfg)
            cfg.mode = 1
            return ray_cfg()(cfg)
        eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
        s = wl.scan(0)
        far = np.eye(4)
        far[0, 3] = 12.0  # sensor 12 m from the centre of a 15 m map: outside
        run_both(eng, ref, s, wl.T_base_sensor, far)
        assert not eng.exists("raycasting")
        assert_layers_equal(eng, ref)
        run_both(eng, ref, s, wl.T_base_sensor, np.eye(4))
        assert eng.exists("raycasting")
        assert_layers_equal(eng, ref)
        run_both(eng, ref, s, wl.T_base_sensor, far)  # outside again: the frame layer keeps its values
        assert_layers_equal(eng, ref)

    def test_everything_filtered_skips_the_stage(self, gpu, R):
        wl = gpu.synth.vlp16(n_scans=1)

        def fill(cfg):
            wl.apply_to(cfg)
            cfg.z_min, cfg.z_max = 100.0, 101.0
            return ray_cfg()(cfg)
        eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
        rc, _ = run_both(eng, ref, wl.scan(0), wl.T_base_sensor, wl.pose(0))
        assert rc == 2 and not eng.exists("raycasting")
        assert_layers_equal(eng, ref)

    def test_tiled_engines_reassemble(self, gpu, R):
        """2x2 spatial tiles of a GLOBAL map, every tile sees the whole scan: the stage needs no
        exchange (rays are traced in the global grid, each tile keeps its own cells)."""
        wl = gpu.synth.global_map(n_scans=3, size_m=60.0, n_az=1024, radius=10.0)

        def fill(cfg):
            wl.apply_to(cfg)
            return ray_cfg(rc_log_odds_ghost=1.2)(cfg)
        eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, fill)
        rows, cols = eng.rows, eng.cols
        hr, hc = rows // 2, cols // 2
        tiles = []
        for tr in range(2):
            for tc in range(2):
                r0, c0 = tr * hr, tc * hc
                t = gpu.Engine(wl.width, wl.height, wl.resolution, fill(gpu.capi.default_config()),
                               tile=(r0, c0, hr, hc, r0, c0, hr, hc))
                tiles.append((r0, c0, t))
        for k in range(3):
            if k == 1:
                both((eng, ref), plant_ghost_block)
                for r0, c0, t in tiles:
                    e = eng.layer("elevation")
                    t.set_layer("elevation", e[r0:r0 + hr, c0:c0 + hc])
            s = wl.scan(k)
            run_both(eng, ref, s, wl.T_base_sensor, wl.pose(k))
            for _, _, t in tiles:
                t.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k),
                            intensity=s.get("intensity"))
        assert_layers_equal(eng, ref)
        for name in eng.layers():
            full = eng.layer(name)
            for r0, c0, t in tiles:
                a, b = t.layer(name), full[r0:r0 + hr, c0:c0 + hc]
                assert np.array_equal(np.isnan(a), np.isnan(b)), name
                assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), name
