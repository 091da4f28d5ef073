"""Pins the stencil post-processing part of the oracle (oracle/fdm_ref_post.hpp)
against the reference's own tests (fastdem/tests/test_postprocess.cpp:37-72, 192-420).  CPU only."""
import numpy as np
import pytest

F32 = np.float32


@pytest.fixture()
def m(R):
    """PostprocessTest fixture (test_postprocess.cpp:24-36): 10x10 m @ 0.5 -> 20x20."""
    return R.RefEngine(10.0, 10.0, 0.5)


def center(e):
    return e.get_index(0.0, 0.0)[1]


def put(e, name, arr):
    e.set_layer(name, np.asarray(arr, dtype=F32))


class TestInpainting:
    def test_fills_simple_hole(self, m):  # :39-57
        r, c = center(m)
        el = m.layer("elevation")
        el[r - 1:r + 2, c - 1:c + 2] = 1.0
        el[r, c] = np.nan
        put(m, "elevation", el)
        m.apply_inpainting(3, 2)
        assert m.exists("elevation_inpainted")
        assert abs(m.layer("elevation_inpainted")[r, c] - 1.0) < 0.01
        assert np.isnan(m.layer("elevation")[r, c])  # original untouched (inplace = false)

    def test_preserves_existing_values(self, m):  # :59-69
        put(m, "elevation", np.full((20, 20), 2.0))
        m.apply_inpainting(3, 2)
        assert m.layer("elevation_inpainted")[center(m)] == F32(2.0)

    def test_inplace_and_iteration_growth(self, m):  # inpainting.cpp:23-31,41-64
        el = np.full((20, 20), np.nan, dtype=F32)
        el[10, 10], el[10, 11] = 1.0, 3.0
        put(m, "elevation", el)
        m.apply_inpainting(1, 2, inplace=True)
        out = m.layer("elevation")
        assert not m.exists("elevation_inpainted")
        assert out[9, 10] == F32(2.0) and out[11, 11] == F32(2.0)  # both seeds are neighbours
        assert np.isnan(out[9, 9]) and out[9, 12] != out[9, 12]    # only one seed in reach: < min_valid
        n1 = np.isfinite(out).sum()
        m.apply_inpainting(2, 2, inplace=True)
        assert np.isfinite(m.layer("elevation")).sum() > n1       # the front grows one ring per pass


class TestSpatialSmoothing:
    def test_removes_spike(self, m):  # :243-259
        r, c = center(m)
        el = m.layer("elevation")
        el[r - 2:r + 3, c - 2:c + 3] = 1.0
        el[r, c] = 100.0
        put(m, "elevation", el)
        m.apply_spatial_smoothing("elevation", 3, 5)
        assert abs(m.layer("elevation")[r, c] - 1.0) < 0.01

    def test_missing_layer_and_sparse_cells(self, m):  # :261-264, spatial_smoothing.hpp:53,61
        m.apply_spatial_smoothing("nonexistent_layer")
        el = m.layer("elevation")
        el[3, 3], el[3, 4] = 5.0, 7.0
        put(m, "elevation", el)
        m.apply_spatial_smoothing("elevation", 3, 5)
        out = m.layer("elevation")
        assert out[3, 3] == F32(5.0) and out[3, 4] == F32(7.0)  # fewer than 5 valid: untouched
        assert np.isfinite(out).sum() == 2                     # NaN cells are never filled

    def test_median_of_even_window_is_upper_middle(self, m):  # nth_element(size/2)
        el = m.layer("elevation")
        el[0, 0], el[0, 1], el[1, 0], el[1, 1] = 1.0, 2.0, 3.0, 4.0  # corner: window of 4
        put(m, "elevation", el)
        m.apply_spatial_smoothing("elevation", 3, 4)
        assert m.layer("elevation")[0, 0] == F32(3.0)


class TestUncertaintyFusion:
    def fill_block(self, m):  # :194-208
        r, c = center(m)
        up, lo = (np.full((20, 20), np.nan, dtype=F32) for _ in range(2))
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                h = F32(1.0) + F32(0.1) * dr
                up[r + dr, c + dc], lo[r + dr, c + dc] = h + F32(0.2), h - F32(0.2)
        put(m, "upper_bound", up)
        put(m, "lower_bound", lo)
        return r, c

    def test_computes_bounds(self, m):  # :192-226
        r, c = self.fill_block(m)
        m.apply_uncertainty_fusion(True, 0.6, 0.3, 0.01, 0.99, 1)
        u, l = m.layer("upper_bound")[r, c], m.layer("lower_bound")[r, c]
        assert np.isfinite(u) and np.isfinite(l) and u > l
        # 0.6 m at 0.5 m: centre + 4-neighbours ("slightly more than 1 cell"); lower q0.01 = the
        # smallest lower bound among them, upper q0.99 = the largest upper bound
        assert abs(l - (0.9 - 0.2)) < 1e-6 and abs(u - (1.1 + 0.2)) < 1e-6

    def test_missing_bounds_and_disabled(self, R, m):  # :228-240
        bare = R.RefEngine(10.0, 10.0, 0.5)
        bare.clear()  # no crash when the layers are absent / all NaN
        m.apply_uncertainty_fusion(False)
        self.fill_block(m)
        before = m.layer("upper_bound").copy()
        m.apply_uncertainty_fusion(False, 0.6, 0.3)
        assert np.array_equal(m.layer("upper_bound"), before, equal_nan=True)

    def test_min_valid_neighbors_keeps_old_value(self, m):
        r, c = self.fill_block(m)
        before = m.layer("upper_bound").copy()
        m.apply_uncertainty_fusion(True, 0.6, 0.3, 0.01, 0.99, 6)  # the disc holds at most 5 cells
        assert np.array_equal(m.layer("upper_bound"), before, equal_nan=True)


class TestFeatureExtraction:
    NAMES = ("step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z")

    def test_creates_all_layers_and_flat_plane(self, m):  # :268-297
        put(m, "elevation", np.full((20, 20), 1.0))
        m.apply_feature_extraction(0.6, 4)
        for n in self.NAMES:
            assert m.exists(n)
        rc = center(m)
        assert abs(m.layer("slope")[rc]) < 1.0 and abs(m.layer("roughness")[rc]) < 1e-3
        assert abs(m.layer("step")[rc]) < 1e-3 and abs(m.layer("_normal_z")[rc] - 1.0) < 0.01

    def test_tilted_plane(self, m):  # :299-315 (cell.row * res * 0.5)
        rows = np.arange(20, dtype=F32)[:, None] * F32(0.5) * F32(0.5)
        put(m, "elevation", np.broadcast_to(rows, (20, 20)).copy())
        m.apply_feature_extraction(0.6, 4)
        s = m.layer("slope")[center(m)]
        assert 10.0 < s < 45.0 and abs(s - np.degrees(np.arctan(0.5))) < 0.05

    def test_step_detection(self, m):  # :317-331
        el = np.zeros((20, 20), dtype=F32)
        el[:, 10:] = 1.0
        put(m, "elevation", el)
        m.apply_feature_extraction(0.6, 4)
        assert m.layer("step")[center(m)] > 0.5

    def test_nan_cells_and_insufficient_neighbors(self, m):  # :339-361
        m.apply_feature_extraction(0.6, 4)
        assert m.exists("slope") and not np.isfinite(m.layer("slope")[center(m)])
        el = m.layer("elevation")
        el[center(m)] = 1.0
        put(m, "elevation", el)
        m.apply_feature_extraction(0.6, 4)
        assert not np.isfinite(m.layer("slope")[center(m)])

    def test_normal_points_up_and_curvature_bounded(self, m):  # :363-400
        rows = np.arange(20, dtype=F32)[:, None] * F32(0.5)
        put(m, "elevation", np.broadcast_to(rows, (20, 20)).copy())
        m.apply_feature_extraction(0.6, 4)
        nz = m.layer("_normal_z")
        assert (nz[np.isfinite(nz)] > 0).all()
        el = np.full((20, 20), 1.0, dtype=F32)
        el[center(m)] = 2.0
        put(m, "elevation", el)
        m.apply_feature_extraction(0.6, 4)
        cv = m.layer("curvature")[center(m)]
        assert not np.isfinite(cv) or 0.0 <= cv <= 1.0


def test_trig_modes_agree_within_the_conditioning_bound(R):
    """trig_mode 1 (correctly rounded atan2 / cos / sin / acos, what the device computes) against trig_mode 0
    (this machine's float libm): same NaN pattern, eigen layers within 2e-6 absolute, `step` identical —
    the bound tests/test_post_gpu.py asserts for the engine against mode 0."""
    rng = np.random.default_rng(5)
    r, c = np.meshgrid(np.arange(160), np.arange(160), indexing="ij")
    el = (0.4 * np.sin(r * 0.11) * np.cos(c * 0.07) + 0.002 * r + rng.normal(0, 0.01, r.shape)).astype(F32)
    el[rng.uniform(size=el.shape) < 0.15] = np.nan
    out = []
    try:
        for mode in (0, 1):
            R.set_trig_mode(mode)
            m = R.RefEngine(8.0, 8.0, 0.05)
            m.set_layer("elevation", el)
            m.apply_feature_extraction(0.3, 4, 0.05, 0.95)
            out.append({n: m.layer(n) for n in TestFeatureExtraction.NAMES})
    finally:
        R.set_trig_mode(0)
    a, b = out
    assert np.array_equal(a["step"], b["step"], equal_nan=True)
    differ = 0
    for n in ("roughness", "curvature", "_normal_x", "_normal_y", "_normal_z"):
        assert np.array_equal(np.isnan(a[n]), np.isnan(b[n])), n
        ok = np.isfinite(a[n])
        assert ok.sum() > 10000
        assert np.abs(a[n][ok].astype(np.float64) - b[n][ok]).max() <= 2e-6, n
        differ += int((a[n][ok] != b[n][ok]).sum())
    assert differ > 0  # the two libm behaviours are really different on this machine (glibc 2.35)


def test_eig3_against_lapack(R):
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.normal(size=(3, 3)).astype(F32) * F32(rng.uniform(1e-3, 10))
        cov = (a @ a.T).astype(F32)
        val, vec = R.eig3(cov)
        w, v = np.linalg.eigh(cov.astype(np.float64))
        assert np.abs(val - w).max() <= 2e-5 * max(1.0, np.abs(w).max())
        if min(w[1] - w[0], w[2] - w[1]) > 1e-3 * np.abs(w).max():
            assert np.abs(np.abs(vec.T @ v) - np.eye(3)).max() < 5e-3
