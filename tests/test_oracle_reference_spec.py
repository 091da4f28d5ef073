"""Pins the CPU oracle against the known-answer values the reference's OWN tests hold for
the integrate() path.  Each test cites the reference test it re-expresses.
CPU only — this is what makes the oracle a trustworthy checker for the HIP engine."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32 = np.float32


def feq(a, b):
    """gtest EXPECT_FLOAT_EQ: within 4 ULPs."""
    a, b = F32(a), F32(b)
    if a == b:
        return True
    ia, ib = int(a.view(np.int32)), int(b.view(np.int32))
    return abs(ia - ib) <= 4


# ---------------------------------------------------------------- Kalman ----
# fastdem/tests/test_kalman_estimation.cpp
class TestKalman:
    def test_first_measurement_initializes(self, R):  # :18-28
        c = R.KalmanCell(0.0001, 0.01, 0.0)
        c.update(5.0, 0.04)
        assert feq(c.x, 5.0) and feq(c.P, 0.04) and feq(c.count, 1.0)

    def test_repeated_low_variance_reduces_p(self, R):  # :30-44
        c = R.KalmanCell(0.0001, 1.0, 0.0)
        c.update(5.0, 0.5)
        p0 = c.P
        for _ in range(20):
            c.update(5.0, 0.01)
        assert c.P < p0

    def test_p_clamping(self, R):  # :46-62
        c = R.KalmanCell(0.001, 0.1, 0.0)
        c.update(5.0, 0.05)
        for _ in range(100):
            c.update(5.0, 0.0001)
        assert 0.001 <= c.P <= 0.1

    def test_bounds_from_sample_variance(self, R):  # :64-80
        c = R.KalmanCell(0.0001, 1.0, 0.0)
        c.update(3.0, 0.04)
        c.update(7.0, 0.04, bounds=True)
        sigma = np.sqrt(F32(c.variance))
        assert abs(c.upper - (c.x + 2.0 * sigma)) < 1e-5
        assert abs(c.lower - (c.x - 2.0 * sigma)) < 1e-5

    def test_zero_variance_falls_back_to_max(self, R):  # :82-90
        c = R.KalmanCell(0.0001, 0.5, 0.0)
        c.update(5.0, 0.0)
        assert feq(c.P, 0.5)

    def test_converges(self, R):  # :92-104
        c = R.KalmanCell(0.0001, 1.0, 0.0)
        c.update(10.0, 1.0)
        for _ in range(50):
            c.update(5.0, 0.01)
        assert abs(c.x - 5.0) < 0.1

    def test_sample_variance_3_7_is_8(self, R):  # :106-119
        c = R.KalmanCell(0.0001, 1.0, 0.0)
        c.update(3.0, 0.01)
        c.update(7.0, 0.01)
        assert feq(c.variance, 8.0)

    def test_variance_is_sample_variance_not_p(self, R):  # :121-139
        c = R.KalmanCell(0.0001, 1.0, 0.0)
        for _ in range(50):
            c.update(5.0, 0.01)
        assert abs(c.P - 0.0001) < 0.001
        assert abs(c.variance) < 1e-6


# -------------------------------------------------------------------- P2 ----
# fastdem/tests/test_quantile_estimation.cpp
class TestP2:
    def test_less_than_five_counted(self, R):  # :40-46
        c = R.P2Cell()
        for v in (3.0, 1.0, 4.0):
            c.update(v)
        assert feq(c.count, 3.0)

    def test_five_observations_sorted(self, R):  # :48-67
        c = R.P2Cell()
        for v in (5.0, 3.0, 1.0, 4.0, 2.0):
            c.update(v)
        assert feq(c.count, 5.0)
        assert list(c.q) == [1.0, 2.0, 3.0, 4.0, 5.0]
        assert list(c.n) == [0.0, 1.0, 2.0, 3.0, 4.0]

    def test_marker_monotonicity_mt19937_uniform(self, R):  # :69-87
        seq = np.fromfile(os.path.join(GOLD, "mt19937_42_uniform_0_10_x100.f32"), dtype=F32)
        assert seq.size == 100
        c = R.P2Cell()
        for v in seq:
            c.update(float(v))
        q = c.q
        assert np.all(q[:-1] <= q[1:])

    def test_normal_median_near_mean(self, R):  # :89-103
        seq = np.fromfile(os.path.join(GOLD, "mt19937_42_normal_5_1_x1000.f32"), dtype=F32)
        c = R.P2Cell()
        for v in seq:
            c.update(float(v))
        assert abs(c.q[2] - 5.0) < 0.2

    def test_bounds_ordered(self, R):  # :105-119
        seq = np.fromfile(os.path.join(GOLD, "mt19937_42_normal_5_1_x1000.f32"), dtype=F32)[:500]
        c = R.P2Cell()
        for i, v in enumerate(seq):
            c.update(float(v), bounds=(i == len(seq) - 1))
        assert c.lower < c.upper

    def test_elevation_before_p2_is_last_sample(self, R):  # :121-128
        c = R.P2Cell()
        c.update(3.0)
        assert feq(c.elevation, 3.0)
        c.update(7.0)
        assert feq(c.elevation, 7.0)

    def test_elevation_after_p2_tracks_marker(self, R):  # :130-148
        c = R.P2Cell()
        for v in (1.0, 2.0, 3.0, 4.0, 5.0):
            c.update(v)
        assert feq(c.elevation, c.q[3])
        c.update(6.0)
        assert feq(c.elevation, c.q[3])

    def test_estimate_keeps_nan_until_marker_filled(self, R):
        # SURVEY.md §7 "P2 NaN semantics": computeBounds overwrites elevation with q[marker]
        # (quantile_estimation.hpp:161-162 then :171-172) -> NaN until the 4th scan of a cell.
        c = R.P2Cell()
        for k, v in enumerate((1.0, 2.0, 3.0)):
            c.update(v, bounds=True)
            assert np.isnan(c.elevation), k
        c.update(4.0, bounds=True)
        assert feq(c.elevation, 4.0)


# ---------------------------------------------------------- sensor models ----
# fastdem/tests/test_sensor_models.cpp
class TestSensorModels:
    def cfg(self, R, **kw):
        c = R.default_config()
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def test_constant_scaled_identity(self, R):  # :56-79
        c = self.cfg(R, sensor_type=0, constant_uncertainty=0.1)
        cov = R.sensor_covariance(c, [1, 2, 3])
        assert np.allclose(cov, np.eye(3) * F32(0.1) * F32(0.1), rtol=1e-6)
        c2 = self.cfg(R, sensor_type=0, constant_uncertainty=0.05)
        assert np.array_equal(R.sensor_covariance(c2, [0, 0, 0]), R.sensor_covariance(c2, [10, 20, 30]))
        c0 = self.cfg(R, sensor_type=0, constant_uncertainty=0.0)
        assert np.all(R.sensor_covariance(c0, [1, 2, 3]) == 0)

    def test_lidar_symmetric_psd(self, R):  # :91-103, :168-175
        c = self.cfg(R, sensor_type=1)
        for p in ([5, 3, 2], [50, 0, 0]):
            cov = R.sensor_covariance(c, p).astype(np.float64)
            assert np.allclose(cov, cov.T, rtol=1e-6)
            assert np.linalg.eigvalsh(0.5 * (cov + cov.T)).min() >= -1e-8

    def test_lidar_zero_distance_fallback(self, R):  # :105-112
        cov = R.sensor_covariance(self.cfg(R, sensor_type=1), [0, 0, 0])
        assert np.allclose(np.diag(cov), 0.01, atol=1e-6)

    def test_lidar_on_axis_variances(self, R):  # :114-129, :131-166
        c = self.cfg(R, sensor_type=1)
        d = 10.0
        vr = 0.02 * 0.02
        vl = (d * 0.001) ** 2
        for axis in range(3):
            p = [0.0, 0.0, 0.0]
            p[axis] = d
            cov = R.sensor_covariance(c, p)
            for k in range(3):
                assert abs(cov[k, k] - (vr if k == axis else vl)) < 1e-6
        dd = d / np.sqrt(3.0)
        ev = np.sort(np.linalg.eigvalsh(R.sensor_covariance(c, [dd, dd, dd]).astype(np.float64)))
        assert abs(ev[0] - vl) < 1e-5 and abs(ev[1] - vl) < 1e-5 and abs(ev[2] - vr) < 1e-5

    def test_rgbd(self, R):  # :190-262
        c = self.cfg(R, sensor_type=2)
        cov = R.sensor_covariance(c, [0.1, 0.2, 1.0])
        assert cov[0, 1] == 0 and cov[0, 2] == 0 and cov[1, 2] == 0
        opt = R.sensor_covariance(c, [0, 0, 0.4])
        far = R.sensor_covariance(c, [0, 0, 2.4])
        assert opt[2, 2] < far[2, 2]
        assert abs(opt[2, 2] - 0.001 * 0.001) < 1e-10
        for p in ([0, 0, 0], [1, 2, -0.5]):
            assert np.allclose(np.diag(R.sensor_covariance(c, p)), 0.01, atol=1e-6)
        c1 = R.sensor_covariance(c, [0, 0, 1.0])
        c2 = R.sensor_covariance(c, [0, 0, 2.0])
        assert abs(c2[0, 0] / c1[0, 0] - 4.0) < 1e-4
        assert np.array_equal(R.sensor_covariance(c, [0, 0, 1.5]), R.sensor_covariance(c, [3, 4, 1.5]))


# -------------------------------------------------------------- dual layer ----
# fastdem/tests/test_dual_layer.cpp — ElevationMapping::update driven directly, GLOBAL mode
def dual_cfg(R, est=0):
    c = R.default_config()
    c.mode = 1
    c.estimation_type = est
    c.kalman_min_variance, c.kalman_max_variance, c.kalman_process_noise = 0.0001, 1.0, 0.0
    return c


def cloud(pts):
    a = np.array(pts, dtype=F32).reshape(-1, 3)
    return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()


class TestDualLayer:
    def make(self, R, est=0):
        e = R.RefEngine(10.0, 10.0, 0.5, dual_cfg(R, est))
        ok, idx = e.get_index(0.0, 0.0)
        assert ok
        return e, idx

    def test_ground_obstacle_separation(self, R):  # :66-83
        e, (r, c) = self.make(R)
        e.update(*cloud([[0, 0, 0.0], [0, 0, 3.0]]))
        assert abs(e.layer("elevation")[r, c] - 0.0) < 0.1
        assert abs(e.layer("obstacle")[r, c] - 3.0) < 0.1

    def test_single_point_only_ground(self, R):  # :106-119
        e, (r, c) = self.make(R)
        e.update(*cloud([[0, 0, 2.0]]))
        assert abs(e.layer("elevation")[r, c] - 2.0) < 0.1
        assert np.isnan(e.layer("obstacle")[r, c])

    def test_kalman_obstacle_overwritten_per_frame(self, R):  # :121-143
        e, (r, c) = self.make(R)
        e.update(*cloud([[0, 0, 0.0], [0, 0, 3.0]]))
        e.update(*cloud([[0, 0, 0.1], [0, 0, 3.1]]))
        assert -0.05 < e.layer("elevation")[r, c] < 0.15
        assert feq(e.layer("obstacle")[r, c], 3.1)

    def test_quantile_with_dual_layer(self, R):  # :145-165
        e, (r, c) = self.make(R, est=1)
        for i in range(10):
            noise = 0.05 if i % 2 == 0 else -0.05
            e.update(*cloud([[0, 0, 0.0 + noise], [0, 0, 5.0 + noise]]))
        assert abs(e.layer("elevation")[r, c]) < 0.5
        assert abs(e.layer("obstacle")[r, c] - 5.0) < 0.1

    def test_elevation_max_monotone(self, R):  # :167-186
        e, (r, c) = self.make(R)
        e.update(*cloud([[0, 0, 0.0], [0, 0, 3.0]]))
        assert feq(e.layer("elevation_max")[r, c], 3.0)
        e.update(*cloud([[0, 0, 0.0], [0, 0, 5.0]]))
        assert feq(e.layer("elevation_max")[r, c], 5.0)

    def test_obstacle_clears_when_flat(self, R):  # :188-203
        e, (r, c) = self.make(R)
        e.update(*cloud([[0, 0, 0.0], [0, 0, 2.0]]))
        assert feq(e.layer("obstacle")[r, c], 2.0)
        e.update(*cloud([[0, 0, 0.0]]))
        assert np.isnan(e.layer("obstacle")[r, c])

    def test_no_covariance_channel_uses_max_variance(self, R):
        # elevation_mapping.cpp:57-60 -> pt_z_var = 0 -> Kalman R = max_variance (kalman_estimation.hpp:112-113)
        e, (r, c) = self.make(R)
        e.update(*cloud([[0, 0, 1.0]]))
        assert feq(e.layer("_kalman_p")[r, c], 1.0)


# ------------------------------------------------------- end-to-end facade ----
# fastdem/tests/test_fastdem_integration.cpp (explicit transforms), test_online_mode.cpp:221-241
def grid_cloud(z=1.0, n=7, spacing=0.3):
    # 7x7 grid at 0.3 m spacing (test_fastdem_integration.cpp:32-41)
    g = (np.arange(n) - (n - 1) / 2.0) * spacing
    X, Y = np.meshgrid(g, g, indexing="ij")
    return X.ravel().astype(F32), Y.ravel().astype(F32), np.full(n * n, z, dtype=F32)


def T(x=0.0, y=0.0, z=0.0, yaw=0.0):
    M = np.eye(4)
    c, s = np.cos(yaw), np.sin(yaw)
    M[:2, :2] = [[c, -s], [s, c]]
    M[:3, 3] = (x, y, z)
    return M


class TestIntegration:
    def make(self, R, **kw):
        c = R.default_config()
        for k, v in kw.items():
            setattr(c, k, v)
        return R.RefEngine(10.0, 10.0, 0.5, c)  # 20x20 map (:26-29)

    def test_basic_integration_elevation_near_one(self, R):  # :46-60
        e = self.make(R)
        x, y, z = grid_cloud(1.0)
        rc, st = e.integrate(x, y, z, T(), T())
        assert rc == 0
        el = e.layer("elevation")
        assert np.isfinite(el).sum() > 0
        assert np.all(np.abs(el[np.isfinite(el)] - 1.0) < 0.1)

    def test_empty_cloud_returns_false(self, R):  # :62-70
        e = self.make(R)
        z = np.zeros(0, dtype=F32)
        rc, _ = e.integrate(z, z, z, T(), T())
        assert rc == 1
        assert np.isnan(e.layer("elevation")).all()

    def test_all_filtered_returns_false_and_map_empty(self, R):  # :72-80, :357-378
        e = self.make(R, z_min=5.0, z_max=6.0)
        x, y, z = grid_cloud(1.0)
        rc, st = e.integrate(x, y, z, T(), T())
        assert rc == 2 and st["n_after_filter"] == 0
        assert np.isnan(e.layer("elevation")).all()

    def test_height_and_range_filters(self, R):  # :237-249, :287-316
        e = self.make(R, z_min=-0.5, z_max=0.5)
        x, y, _ = grid_cloud()
        z = np.where(np.arange(x.size) % 2 == 0, 0.0, 3.0).astype(F32)
        rc, st = e.integrate(x, y, z, T(), T())
        assert rc == 0 and st["n_after_filter"] == int((z == 0).sum())
        el = e.layer("elevation")
        assert np.nanmax(el) < 0.5
        e2 = self.make(R, range_min=0.5, range_max=0.8)
        rc, st = e2.integrate(*grid_cloud(0.0), T(), T())
        d = np.hypot(*grid_cloud(0.0)[:2])
        assert st["n_after_filter"] == int(((d >= 0.5) & (d <= 0.8)).sum())

    def test_local_follows_robot_global_does_not(self, R):  # :179-215
        e = self.make(R, mode=0)
        e.integrate(*grid_cloud(1.0), T(), T(100.0, 100.0))
        g = e.geometry()
        assert abs(g.position_x - 100.0) < 0.5 and abs(g.position_y - 100.0) < 0.5
        assert not e.get_index(0.0, 0.0)[0]
        eg = self.make(R, mode=1)
        eg.integrate(*grid_cloud(1.0), T(), T(1.0, 0.0))
        g = eg.geometry()
        assert g.position_x == 0.0 and g.position_y == 0.0

    def test_sensor_offset_and_yaw(self, R):  # :253-283
        e = self.make(R)
        x, y, z = grid_cloud(0.0)
        e.integrate(x, y, z, T(z=1.5), T())
        el = e.layer("elevation")
        assert np.all(np.abs(el[np.isfinite(el)] - 1.5) < 0.1)
        e2 = self.make(R, mode=1)
        px = np.array([2.0], dtype=F32)
        e2.integrate(px, np.zeros(1, F32), np.ones(1, F32), T(), T(yaw=np.pi / 2))
        ok, (r, c) = e2.get_index(0.0, 2.0)
        assert ok and abs(e2.layer("elevation")[r, c] - 1.0) < 0.1

    def test_pose_offset_lands_data_at_2_0(self, R):  # test_online_mode.cpp:221-241
        e = self.make(R, mode=0)
        e.integrate(*grid_cloud(1.0), T(), T(2.0, 0.0))
        ok, (r, c) = e.get_index(2.0, 0.0)
        assert ok and np.isfinite(e.layer("elevation")[r, c])

    @pytest.mark.parametrize("sensor", [0, 1, 2])
    @pytest.mark.parametrize("est", [0, 1])
    def test_sensor_estimator_matrix_smoke(self, R, sensor, est):  # :129-175
        e = self.make(R, sensor_type=sensor, estimation_type=est)
        for _ in range(6):
            rc, _ = e.integrate(*grid_cloud(1.0), T(z=0.5), T())
            assert rc == 0
        assert np.isfinite(e.layer("elevation")).sum() > 0

    def test_returns_true_even_if_nothing_lands_in_map(self, R):  # fastdem.cpp:145-161
        e = self.make(R, mode=1)
        rc, st = e.integrate(*grid_cloud(1.0), T(), T(500.0, 500.0))
        assert rc == 0 and st["n_in_map"] == 0 and st["n_cells_touched"] == 0


# ------------------------------------------------------- ElevationMap / grid ----
# fastdem/tests/test_elevation_map.cpp (relative pins at the nanoGrid boundary)
class TestElevationMapSurface:
    def test_inside_outside(self, R):  # :30-33
        e = R.RefEngine(10.0, 10.0, 0.5)
        assert e.get_index(0.0, 0.0)[0]
        assert not e.get_index(100.0, 100.0)[0]

    def test_default_layers_nan(self, R):  # :17-28
        e = R.RefEngine(10.0, 10.0, 0.5)
        for name in ("elevation", "elevation_min", "elevation_max"):
            assert np.isnan(e.layer(name)).all()
        assert e.rows == 20 and e.cols == 20

    def test_at_position_roundtrip(self, R):  # :40-48, :63-71, :142-151
        e = R.RefEngine(10.0, 10.0, 0.5)
        el = e.layer("elevation")
        ok, (r, c) = e.get_index(1.2, -3.4)
        assert ok
        el[r, c] = 2.5
        e.set_layer("elevation", el)
        ok2, (x, y) = e.get_position(r, c)
        assert ok2 and e.get_index(x, y) == (True, (r, c))
        assert abs(x - 1.2) <= 0.25 + 1e-9 and abs(y + 3.4) <= 0.25 + 1e-9

    def test_estimator_layer_constants(self, R):  # kalman_estimation.hpp:64-82, quantile :97-115
        e = R.RefEngine(10.0, 10.0, 0.5)
        assert (e.layer("variance") == 0).all() and (e.layer("n_points") == 0).all()
        assert (e.layer("_kalman_p") == 0).all() and np.isnan(e.layer("_sample_mean")).all()
        assert np.isnan(e.layer("obstacle")).all()
        c = R.default_config()
        c.estimation_type = 1
        p = R.RefEngine(10.0, 10.0, 0.5, c)
        assert np.isnan(p.layer("variance")).all()
        for k in range(5):
            assert (p.layer(f"_p2_n{k}") == k).all() and np.isnan(p.layer(f"_p2_q{k}")).all()
        assert len(p.layers()) == 3 + 15 - 1 + 1  # 3 base + 15 P2 (elevation shared) + obstacle
