"""Pins the raycasting part of the CPU oracle (oracle/fdm_ref_raycast.hpp) against
the known-answer tests the reference holds for it.  Each test cites what it re-expresses.  CPU only."""
import numpy as np
import pytest

F32 = np.float32


def post_map(R, **kw):
    """PostprocessTest fixture (fastdem/tests/test_postprocess.cpp:24-36): 10x10 m @ 0.5 -> 20x20."""
    cfg = R.default_config()
    cfg.raycast_enabled = 1
    for k, v in kw.items():
        setattr(cfg, k, v)
    return R.RefEngine(10.0, 10.0, 0.5, cfg), cfg


def set_cell(e, name, rc, v):
    a = e.layer(name)
    a[rc] = v
    e.set_layer(name, a)


# ------------------------------------------------------------------- raycasting ----
class TestRaycastingReference:
    def test_creates_layers(self, R):  # test_postprocess.cpp:75-92
        e, _ = post_map(R)
        _, c = e.get_index(0.0, 0.0)
        set_cell(e, "elevation", c, 1.0)
        e.apply_raycasting([1.0], [0.0], [0.5], [0.0, 0.0, 5.0])
        for n in ("ghost_removal", "raycasting", "_visibility_logodds"):
            assert e.exists(n)

    def test_clears_ghost_cell(self, R):  # :94-117
        e, _ = post_map(R, rc_height_conflict_threshold=0.05, rc_log_odds_ghost=0.5, rc_clear_threshold=-0.4)
        ok, g = e.get_index(2.0, 0.0)
        assert ok
        set_cell(e, "elevation", g, 10.0)
        e.apply_raycasting([4.0], [0.0], [0.0], [0.0, 0.0, 5.0])
        assert np.isnan(e.layer("elevation")[g])
        assert e.layer("ghost_removal")[g] == F32(1.0)
        # ElevationMap::clearAt (elevation_map.hpp:131-135): every other layer is NaN there too
        for n in e.layers():
            if n != "ghost_removal":
                assert np.isnan(e.layer(n)[g]), n

    def test_observed_cell_protected(self, R):  # :119-146
        e, _ = post_map(R, rc_height_conflict_threshold=0.05, rc_log_odds_observed=0.8,
                        rc_log_odds_ghost=0.5, rc_clear_threshold=-0.4)
        _, c = e.get_index(2.0, 0.0)
        set_cell(e, "elevation", c, 2.0)
        e.apply_raycasting([4.0, 2.0], [0.0, 0.0], [0.0, 0.3], [0.0, 0.0, 5.0])
        assert not np.isnan(e.layer("elevation")[c])
        assert abs(e.layer("_visibility_logodds")[c] - 0.3) < 1e-6  # +0.8 - 0.5

    def test_ghost_requires_accumulation(self, R):  # :148-175
        e, _ = post_map(R, rc_height_conflict_threshold=0.05, rc_log_odds_ghost=0.2, rc_clear_threshold=-0.9)
        _, g = e.get_index(2.0, 0.0)
        set_cell(e, "elevation", g, 10.0)
        for _ in range(4):
            set_cell(e, "elevation", g, 10.0)
            e.apply_raycasting([4.0], [0.0], [0.0], [0.0, 0.0, 5.0])
        assert not np.isnan(e.layer("elevation")[g])
        e.apply_raycasting([4.0], [0.0], [0.0], [0.0, 0.0, 5.0])
        assert np.isnan(e.layer("elevation")[g])

    def test_disabled_is_noop(self, R):  # :177-190
        e, _ = post_map(R, raycast_enabled=0)
        e.apply_raycasting([1.0], [0.0], [0.5], [0.0, 0.0, 5.0])
        for n in ("ghost_removal", "raycasting", "_visibility_logodds"):
            assert not e.exists(n)


class TestRaycastingSemantics:
    """Behaviour stated by raycasting.cpp itself (no gtest pins it); hand-derived values."""

    def test_sensor_outside_map_is_noop(self, R):  # raycasting.cpp:217-220
        e, _ = post_map(R)
        st = e.apply_raycasting([1.0], [0.0], [0.5], [50.0, 0.0, 5.0])
        assert st["n_rays"] == 0 and not e.exists("raycasting")

    def test_upward_ray_only_counts_as_observation(self, R):  # raycasting.cpp:163-170
        e, _ = post_map(R)
        st = e.apply_raycasting([1.0], [0.0], [6.0], [0.0, 0.0, 5.0])
        assert st["n_observed"] == 1 and st["n_rays"] == 0
        _, c = e.get_index(1.0, 0.0)
        assert e.layer("_visibility_logodds")[c] == F32(0.4)
        assert np.isnan(e.layer("raycasting")).all()

    def test_logodds_clamped_at_max(self, R):  # raycasting.cpp:166-167
        e, _ = post_map(R, rc_log_odds_observed=0.9, rc_log_odds_max=2.0)
        for _ in range(4):
            e.apply_raycasting([1.0], [0.0], [6.0], [0.0, 0.0, 5.0])
        _, c = e.get_index(1.0, 0.0)
        assert e.layer("_visibility_logodds")[c] == F32(2.0)

    def test_min_height_is_height_at_cell_exit(self, R):  # raycasting.cpp:113-117
        # sensor (0.25,0.25,5) -> target (4.25,0.25,1): straight along -row; dz=-4 over 8 cells:
        # cell k (k=0 sensor cell) exits at t=(k+0.5)/8 -> z = 5 - 4*(k+0.5)/8, last cell t=1 -> 1.0
        e, _ = post_map(R)
        e.apply_raycasting([4.25], [0.25], [1.0], [0.25, 0.25, 5.0])
        ray = e.layer("raycasting")
        _, (r0, c0) = e.get_index(0.25, 0.25)
        for k in range(9):
            t = min((k + 0.5) / 8.0, 1.0)
            assert abs(ray[r0 - k, c0] - (5.0 - 4.0 * t)) < 1e-5, k
        assert np.isfinite(ray).sum() == 9

    def test_raycasting_layer_reset_every_frame(self, R):  # raycasting.cpp:229
        e, _ = post_map(R)
        e.apply_raycasting([4.25], [0.25], [1.0], [0.25, 0.25, 5.0])
        e.apply_raycasting([0.25], [4.25], [1.0], [0.25, 0.25, 5.0])
        ray = e.layer("raycasting")
        _, (r0, c0) = e.get_index(0.25, 0.25)
        assert np.isfinite(ray).sum() == 9 and np.isfinite(ray[r0, c0 - 8])

    def test_short_ray_skipped(self, R):  # raycasting.cpp:52-54 (kMinRayLength)
        e, _ = post_map(R)
        st = e.apply_raycasting([0.25], [0.25], [1.0], [0.25, 0.25, 5.0])
        assert st["n_rays"] == 1 and st["n_ray_cells"] == 0

    def test_wrapped_buffer(self, R):  # raycasting.cpp:112-113 (start index)
        e, _ = post_map(R)
        e.move(1.0, -1.5)
        g = e.geometry()
        assert (g.start_row, g.start_col) != (0, 0)
        e.apply_raycasting([4.25], [-1.25], [1.0], [1.25, -1.25, 5.0])
        ray = e.layer("raycasting")
        fin = np.argwhere(np.isfinite(ray))
        assert len(fin) == 7
        for r, c in fin:
            ok, (x, y) = e.get_position(int(r), int(c))
            assert ok and abs(y + 1.25) < 1e-9 and 1.0 < x < 4.5


# ------------------------------------------------------------------- voxel grid ----
class TestVoxelKey:  # fastdem/lib/nanoPCL/tests/test_voxel.cpp
    def test_pack_unpack_roundtrip(self, R):  # :24-51
        inv = F32(1.0) / F32(0.1)
        for x, y, z in [(0, 0, 0), (1, 2, 3), (0.05, 0.15, 0.25), (10.5, 20.3, 30.7)]:
            _, (ix, iy, iz) = R.voxel_pack(x, y, z, inv)
            assert (ix, iy, iz) == tuple(int(np.floor(F32(v) * inv)) for v in (x, y, z))

    def test_negative_coordinates(self, R):  # :53-68
        assert R.voxel_pack(-5.5, -10.3, -0.1, 1.0)[1] == (-6, -11, -1)

    def test_clamping(self, R):  # :70-85
        _, (ix, iy, _) = R.voxel_pack(2000000.0, -2000000.0, 0.0, 1.0)
        assert ix == (1 << 20) - 1 and iy == -(1 << 20)

    def test_key_layout(self, R):  # voxel.hpp:28-43
        k, _ = R.voxel_pack(1.5, 2.5, 3.5, 1.0)
        off = 1 << 20
        assert k == ((3 + off) << 42) | ((2 + off) << 21) | (1 + off)


class TestVoxelGridAny:
    def test_empty_and_single(self, R):  # test_filters.cpp:786-798
        assert R.voxel_any([], [], [], 1.0).size == 0
        assert list(R.voxel_any([1.0], [2.0], [3.0], 1.0)) == [0]

    def test_nan_dropped(self, R):  # test_filters.cpp:804-815
        sel = R.voxel_any([1.0, np.nan, 4.0], [2.0, 0.0, 5.0], [3.0, 0.0, 6.0], 10.0)
        assert 1 not in sel and len(sel) == 1

    def test_size_range_check(self, R):  # voxel_grid_impl.hpp:31-33
        for bad in (0.0005, 100.5):
            with pytest.raises(ValueError):
                R.voxel_any([0.0], [0.0], [0.0], bad)

    def test_selection_formula(self, R):  # voxel_grid_impl.hpp:171-173
        # voxel A (x in [0,0.1)) holds points 0,1,3 ; voxel B holds 2,4.  Sorted: A(0,1,3) B(2,4).
        x = [0.01, 0.02, 0.15, 0.03, 0.16]
        sel = R.voxel_any(x, [0.0] * 5, [0.0] * 5, 0.1, stable=True)
        # A: start 0 count 3 -> (21+0)%3 = 0 -> idx 0 ; B: start 3 count 2 -> (14+39)%2 = 1 -> idx 4
        assert list(sel) == [0, 4]

    def test_one_point_per_voxel_and_same_voxels_in_both_orders(self, R):
        rng = np.random.default_rng(7)
        n = 20000
        x, y, z = (rng.uniform(-3, 3, n).astype(F32) for _ in range(3))
        inv = F32(1.0) / F32(0.25)
        key = np.array([R.voxel_pack(a, b, c, inv)[0] for a, b, c in zip(x[:2000], y[:2000], z[:2000])],
                       dtype=np.uint64)
        for stable in (True, False):
            sel = R.voxel_any(x[:2000], y[:2000], z[:2000], 0.25, stable)
            ks = key[sel]
            assert len(np.unique(ks)) == len(ks) == len(np.unique(key))
            assert (np.diff(ks.astype(np.int64)) > 0).all()  # output is in key order
        a = R.voxel_any(x, y, z, 0.25, True)
        b = R.voxel_any(x, y, z, 0.25, False)
        assert len(a) == len(b)


class TestIntegrateWithRaycasting:
    def test_layers_and_ghost_clear_through_integrate(self, R):  # fastdem.cpp:152-159
        cfg = R.default_config()
        cfg.raycast_enabled = 1
        cfg.rc_log_odds_ghost = 0.5
        cfg.rc_clear_threshold = -0.4
        e = R.RefEngine(10.0, 10.0, 0.5, cfg)
        T = np.eye(4)
        Ts = np.eye(4)
        Ts[2, 3] = 5.0  # sensor 5 m above base
        _, g = e.get_index(2.0, 0.0)
        # a ground point far away, in the sensor frame (z = -5 -> map z = 0)
        rc, _ = e.integrate([4.0], [0.0], [-5.0], Ts, T)
        assert rc == 0 and e.exists("raycasting")
        set_cell(e, "elevation", g, 10.0)  # plant a ghost on the ray
        e.integrate([4.0], [0.0], [-5.0], Ts, T)
        assert np.isnan(e.layer("elevation")[g]) and e.layer("ghost_removal")[g] == F32(1.0)
        assert e.last_ray_stats()["n_cleared"] == 1

    def test_sensor_origin(self, R):  # fastdem.cpp:153-154
        Twb = np.eye(4)
        Twb[:3, :3] = [[0, -1, 0], [1, 0, 0], [0, 0, 1]]
        Twb[:3, 3] = [1.0, 2.0, 0.5]
        Tbs = np.eye(4)
        Tbs[:3, 3] = [0.3, 0.0, 0.6]
        o = R.sensor_origin(Tbs, Twb)
        assert np.allclose(o, [1.0, 2.3, 1.1], atol=1e-6)
