"""Stencil post-processing of the HIP engine through the C ABI vs the oracle.

Tolerances (stated per stage):
  inpainting / median smoothing / uncertainty fusion: bit-exact (same float operation order; the
    fusion's exp() is evaluated in double on the device and agrees with libm's expf);
  feature extraction: bit-exact in all seven layers once both sides use correctly rounded atan2 / cos / sin /
    acos (the device does; the oracle's trig_mode 1); against this machine's float libm the eigen-derived
    layers are bounded at 2e-6 absolute and slope by its acos conditioning (see the two tests);
    `step` (order statistics) is bit-exact either way.
"""
import numpy as np
import pytest

from helpers import assert_arrays_close, assert_layers_equal, pair, run_both

pytestmark = pytest.mark.gpu
F32 = np.float32


def both(objs, fn):
    return [fn(o) for o in objs]


def terrain(rng, shape, holes=0.3, noise=0.02):
    r, c = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij")
    z = 0.4 * np.sin(r * 0.11) * np.cos(c * 0.07) + 0.002 * r + rng.normal(0, noise, shape)
    z = z.astype(F32)
    z[rng.uniform(size=shape) < holes] = np.nan
    return z


def exact(eng, ref, names=None):
    assert sorted(eng.layers()) == sorted(ref.layers())
    for n in (names or ref.layers()):
        assert_arrays_close(eng.layer(n), ref.layer(n), n, 0.0, 0.0)


# ------------------------------------------------- the reference's own tests on the engine ----
class TestReferencePostprocessTestsOnEngine:  # fastdem/tests/test_postprocess.cpp
    def fixture(self, gpu, R):
        return pair(gpu, R, 10.0, 10.0, 0.5)

    def test_inpainting_fills_simple_hole(self, gpu, R):  # :39-57
        eng, ref = self.fixture(gpu, R)
        _, (r, c) = ref.get_index(0.0, 0.0)
        el = np.full((20, 20), np.nan, dtype=F32)
        el[r - 1:r + 2, c - 1:c + 2] = 1.0
        el[r, c] = np.nan
        both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_inpainting(3, 2)))
        assert abs(eng.layer("elevation_inpainted")[r, c] - 1.0) < 0.01
        exact(eng, ref)

    def test_smoothing_removes_spike(self, gpu, R):  # :243-259
        eng, ref = self.fixture(gpu, R)
        _, (r, c) = ref.get_index(0.0, 0.0)
        el = np.full((20, 20), np.nan, dtype=F32)
        el[r - 2:r + 3, c - 2:c + 3] = 1.0
        el[r, c] = 100.0
        both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_spatial_smoothing("elevation", 3, 5)))
        assert abs(eng.layer("elevation")[r, c] - 1.0) < 0.01
        both((eng, ref), lambda o: o.apply_spatial_smoothing("nonexistent_layer"))  # :261-264
        exact(eng, ref)

    def test_fusion_computes_bounds(self, gpu, R):  # :192-226
        eng, ref = self.fixture(gpu, R)
        _, (r, c) = ref.get_index(0.0, 0.0)
        up, lo = (np.full((20, 20), np.nan, dtype=F32) for _ in range(2))
        for dr in (-1, 0, 1):
            for dc in (-1, 0, 1):
                h = F32(1.0) + F32(0.1) * dr
                up[r + dr, c + dc], lo[r + dr, c + dc] = h + F32(0.2), h - F32(0.2)
        both((eng, ref), lambda o: (o.set_layer("upper_bound", up), o.set_layer("lower_bound", lo),
                                    o.apply_uncertainty_fusion(True, 0.6, 0.3, 0.01, 0.99, 1)))
        assert eng.layer("upper_bound")[r, c] > eng.layer("lower_bound")[r, c]
        exact(eng, ref)

    def test_feature_extraction_planes(self, gpu, R):  # :268-315
        eng, ref = self.fixture(gpu, R)
        rows = np.arange(20, dtype=F32)[:, None] * F32(0.25)
        both((eng, ref), lambda o: (o.set_layer("elevation", np.broadcast_to(rows, (20, 20)).copy()),
                                    o.apply_feature_extraction(0.6, 4)))
        _, rc = ref.get_index(0.0, 0.0)
        s = eng.layer("slope")[rc]
        assert abs(s - np.degrees(np.arctan(0.5))) < 0.05 and eng.layer("_normal_z")[rc] > 0
        for n in ("step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z"):
            assert eng.exists(n)


# --------------------------------------------------------------------- parity on terrain ----
def rolled_pair(gpu, R, rng, size=24.0, res=0.1):
    eng, ref = pair(gpu, R, size, size, res)
    both((eng, ref), lambda o: o.move(3.7, -5.2))  # the circular buffer seam crosses the map
    shape = (eng.rows, eng.cols)
    return eng, ref, shape


def test_inpainting_parity(gpu, R):
    rng = np.random.default_rng(21)
    eng, ref, shape = rolled_pair(gpu, R, rng)
    el = terrain(rng, shape, holes=0.55)
    for iters, mv, inplace in ((3, 2, False), (1, 3, False), (4, 1, True), (0, 2, False)):
        both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_inpainting(iters, mv, inplace)))
        exact(eng, ref)
    both((eng, ref), lambda o: o.apply_inpainting(3, 2))
    assert np.isfinite(eng.layer("elevation_inpainted")).sum() > np.isfinite(el).sum()


def test_median_smoothing_parity(gpu, R):
    rng = np.random.default_rng(22)
    eng, ref, shape = rolled_pair(gpu, R, rng)
    el = terrain(rng, shape, holes=0.2)
    el[rng.uniform(size=shape) < 0.02] = 50.0  # spikes
    # odd kernels, EVEN kernels (region(Size(k, k)) spans [-k/2, k/2]: the (k + 1)-wide box) and windows beyond 256
    # cells (17 x 17 = 289, 22 -> 23 x 23 = 529: the pooled kernel) — every size the reference accepts
    # (17 and up: k_median_sel, selection by bisection on an LDS tile; 41 x 41 = 1 681 cells needs 48 KB of it)
    for k, mv in ((3, 5), (5, 9), (7, 1), (4, 5), (15, 30), (17, 40), (22, 3), (41, 200)):
        both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_spatial_smoothing("elevation", k, mv)))
        exact(eng, ref, ["elevation"])
    with pytest.raises(gpu.EngineError):
        eng.apply_spatial_smoothing("elevation", 0, 5)


def test_uncertainty_fusion_parity(gpu, R):
    rng = np.random.default_rng(23)
    eng, ref, shape = rolled_pair(gpu, R, rng, size=20.0, res=0.05)
    el = terrain(rng, shape, holes=0.25)
    half = np.abs(rng.normal(0.05, 0.03, shape)).astype(F32) + F32(0.005)
    for cfgv in ((True, 0.15, 0.05, 0.01, 0.99, 3), (True, 0.3, 0.1, 0.25, 0.75, 5)):
        both((eng, ref), lambda o: (o.set_layer("upper_bound", el + half), o.set_layer("lower_bound", el - half),
                                    o.apply_uncertainty_fusion(*cfgv)))
        exact(eng, ref, ["upper_bound", "lower_bound"])
    assert not np.array_equal(eng.layer("upper_bound"), el + half, equal_nan=True)


def test_uncertainty_fusion_small_discs_ties_and_quantile_edges(gpu, R):
    """The double-sample kernel (k_fusion_f64_tiled, discs of up to 29 cells) at every disc size below the default, with
    what its branch-free walk must get right: entries without data or with a weight <= 1e-6 (they stay in the sorted list
    with the weight 0.0), ties in value (entry order), -0.0 / +0.0, min_valid gating, and quantiles at and beyond the
    ends of the range it takes (0 and > 1 go to the integer-sample kernel) — all bit for bit against the oracle."""
    rng = np.random.default_rng(41)
    eng, ref, shape = rolled_pair(gpu, R, rng, size=12.0, res=0.05)
    el = terrain(rng, shape, holes=0.35)
    half = np.abs(rng.normal(0.05, 0.03, shape)).astype(F32) + F32(0.005)
    up, lo = el + half, el - half
    # plateaus of equal bounds (ties keep the entry order), signed zeros, and pairs so far apart that the weight is <= 1e-6
    up[20:60, 20:60] = np.where(np.isfinite(up[20:60, 20:60]), F32(0.25), up[20:60, 20:60])
    lo[20:60, 20:60] = np.where(np.isfinite(lo[20:60, 20:60]), F32(-0.125), lo[20:60, 20:60])
    z = rng.uniform(size=shape) < 0.05
    lo[z & np.isfinite(lo)] = F32(-0.0)
    z2 = rng.uniform(size=shape) < 0.05
    up[z2 & np.isfinite(up) & (lo <= 0)] = F32(0.0)
    far = rng.uniform(size=shape) < 0.05
    up[far & np.isfinite(up)] = F32(3.0e6)   # 1 / (range + 1e-4) * w_spatial <= 1e-6: counted as valid, never sampled
    inv = rng.uniform(size=shape) < 0.02
    up[inv & np.isfinite(up)] = F32(-5.0)    # upper below lower: a negative range, a negative (or huge) weight
    cases = []
    for radius in (0.049, 0.05, 0.0708, 0.1, 0.112, 0.1415, 0.15):   # 1, 5, 9, 13, 21, 25, 29 cells
        cases.append((True, radius, 0.05, 0.01, 0.99, 3))
    cases += [(True, 0.15, 0.05, 1e-6, 1.0, 1), (True, 0.15, 0.05, 0.0, 1.0, 3), (True, 0.15, 0.05, 0.5, 0.5, 29),
              (True, 0.15, 0.05, 0.25, 1.5, 3), (True, 0.1, 0.02, 0.01, 0.99, 14), (True, 0.15, 0.3, 0.999, 0.001, 2)]
    for cfgv in cases:
        both((eng, ref), lambda o: (o.set_layer("upper_bound", up), o.set_layer("lower_bound", lo),
                                    o.apply_uncertainty_fusion(*cfgv)))
        exact(eng, ref, ["upper_bound", "lower_bound"])
    # the same call twice in a row: the second one finds its region table on the device
    both((eng, ref), lambda o: (o.apply_uncertainty_fusion(*cases[6]), o.apply_uncertainty_fusion(*cases[6])))
    exact(eng, ref, ["upper_bound", "lower_bound"])


def test_feature_extraction_order_statistic_slots(gpu, R):
    """k_features_tiled keeps exactly the order statistics the percentiles can ask for: 2 / 3, 4 / 4, 6 / 7 (the defaults on
    discs of 29 / 49 / 113 cells), 8 / 8 and 16 / 16 slots — one v_med3_f32 per slot; `step` and the PCA layers bit for bit."""
    rng = np.random.default_rng(43)
    eng, ref, shape = rolled_pair(gpu, R, rng, size=15.0, res=0.05)
    el = terrain(rng, shape, holes=0.3, noise=0.01)
    el[rng.uniform(size=shape) < 0.03] = F32(0.0)
    el[rng.uniform(size=shape) < 0.02] = F32(-0.0)
    R.set_trig_mode(1)
    try:
        # slots: 6 / 7, 8 / 8, 16 / 16, 4 / 4 (49-cell disc), 6 / 7, 2 / 3 (29-cell disc: the defaults on a 0.1 m map), 2 / 3 (13 cells)
        for radius, lo, hi in ((0.3, 0.05, 0.95), (0.3, 0.06, 0.94), (0.3, 0.13, 0.87), (0.2, 0.05, 0.95), (0.3, 0.0, 0.95),
                               (0.15, 0.05, 0.95), (0.1, 0.05, 0.95), (0.3, 0.05, 0.95)):
            both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_feature_extraction(radius, 4, lo, hi)))
            exact(eng, ref, ["step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z"])
        # a publish cycle runs fusion and feature extraction one after the other, again and again: each keeps its table on
        # the device, neither disturbs the other's (and a changed parameter still uploads)
        half = np.abs(rng.normal(0.05, 0.03, shape)).astype(F32) + F32(0.005)
        both((eng, ref), lambda o: (o.set_layer("upper_bound", el + half), o.set_layer("lower_bound", el - half)))
        for cycle in range(3):
            radius = 0.3 if cycle < 2 else 0.25
            both((eng, ref), lambda o: (o.apply_uncertainty_fusion(True, 0.15, 0.05, 0.01, 0.99, 3),
                                        o.apply_feature_extraction(radius, 4, 0.05, 0.95)))
            exact(eng, ref, ["upper_bound", "lower_bound", "step", "slope", "roughness", "curvature", "_normal_z"])
    finally:
        R.set_trig_mode(0)


def test_discs_of_more_than_256_cells(gpu, R):
    """config/postprocess.hpp:35,45 put no bound on the radii: 0.3 m on a 0.02 m map is a disc of 709 cells, 0.15 m
    one of 177.  The big-neighbourhood kernels (per-cell lists in a global pool) against the oracle: fusion bit-exact;
    feature extraction bit-exact with correctly rounded trig (as the tiled kernel's test), `step` included."""
    rng = np.random.default_rng(29)
    eng, ref, shape = rolled_pair(gpu, R, rng, size=3.0, res=0.02)  # 150 x 150 cells
    el = terrain(rng, shape, holes=0.2, noise=0.005)
    half = np.abs(rng.normal(0.03, 0.02, shape)).astype(F32) + F32(0.004)
    both((eng, ref), lambda o: (o.set_layer("upper_bound", el + half), o.set_layer("lower_bound", el - half),
                                o.apply_uncertainty_fusion(True, 0.3, 0.1, 0.05, 0.95, 5)))
    exact(eng, ref, ["upper_bound", "lower_bound"])
    assert not np.array_equal(eng.layer("upper_bound"), el + half, equal_nan=True)
    R.set_trig_mode(1)
    try:
        both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_feature_extraction(0.3, 4, 0.05, 0.95)))
        assert np.isfinite(eng.layer("slope")).sum() > 0.5 * el.size
        exact(eng, ref, ["step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z"])
    finally:
        R.set_trig_mode(0)


def test_feature_extraction_any_percentile_pair(gpu, R):
    """Percentiles whose order statistics are not among the 16 smallest / largest heights of the disc take
    k_features_sel (bisection on the keys of an LDS tile) also for the default radius: `step` bit-exact."""
    rng = np.random.default_rng(31)
    eng, ref, shape = rolled_pair(gpu, R, rng, size=20.0, res=0.05)
    el = terrain(rng, shape, holes=0.2, noise=0.01)
    R.set_trig_mode(1)
    try:
        for lo, hi in ((0.3, 0.7), (0.0, 1.0), (0.5, 0.5)):
            both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_feature_extraction(0.3, 4, lo, hi)))
            exact(eng, ref, ["step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z"])
    finally:
        R.set_trig_mode(0)


EIGEN_LAYERS = ("roughness", "curvature", "_normal_x", "_normal_y", "_normal_z", "slope")


def _feature_pair(gpu, R, seed, size, radius):
    rng = np.random.default_rng(seed)
    eng, ref, shape = rolled_pair(gpu, R, rng, size=size, res=0.05)
    el = terrain(rng, shape, holes=0.15, noise=0.01)
    el[:, shape[1] // 2:] += F32(0.3)  # a step edge
    both((eng, ref), lambda o: (o.set_layer("elevation", el), o.apply_feature_extraction(radius, 4, 0.05, 0.95)))
    assert sorted(eng.layers()) == sorted(ref.layers())
    assert np.isfinite(eng.layer("slope")).sum() > 0.5 * el.size
    return eng, ref


@pytest.mark.parametrize("seed,size,radius", [(24, 20.0, 0.3), (7, 30.0, 0.35)])
def test_feature_extraction_bit_exact_with_correctly_rounded_trig(gpu, R, seed, size, radius):
    """The device evaluates atan2 / cos / sin / acos in double and rounds once, i.e. the correctly rounded float.
    With the oracle doing the same (trig_mode 1; glibc >= 2.41's CORE-MATH float functions return exactly
    that) every other float operation of the stage is pinned: all seven layers bit for bit."""
    R.set_trig_mode(1)
    try:
        eng, ref = _feature_pair(gpu, R, seed, size, radius)
        for n in ("step",) + EIGEN_LAYERS:
            assert_arrays_close(eng.layer(n), ref.layer(n), n, 0.0, 0.0)
    finally:
        R.set_trig_mode(0)


def test_feature_extraction_against_platform_libm(gpu, R):
    """Oracle on this machine's float libm (glibc 2.35: fdlibm atan2f, <= 1 ulp off the rounded value in ~3 % of the
    calls).  One ulp in theta moves the scaled roots by <= ~2 eps; the eigenvalues carry that as an ABSOLUTE error of
    a few eps * scale (scale = max |cov - mean| <= ~1e-1 here), so the bound is absolute, not relative:
      roughness, curvature, normals: 2e-6 (measured max 4.3e-7 over 1.7 M cells; the bar was 2e-4);
      slope = acos(|nz|) in degrees: (180/pi) * 3 ulp(1) / sqrt(1 - nz^2), floor 1e-5 — acos is ill-conditioned
      towards nz = 1 (0.03 deg there, 7e-5 deg at nz = 0.99; measured max 4.8e-4 deg);
      step (order statistics): bit-exact."""
    eng, ref = _feature_pair(gpu, R, 24, 20.0, 0.3)
    assert_arrays_close(eng.layer("step"), ref.layer("step"), "step", 0.0, 0.0)
    nz = ref.layer("_normal_z").astype(np.float64)
    for n in EIGEN_LAYERS:
        a, b = eng.layer(n), ref.layer(n)
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{n}: NaN pattern differs in {(np.isnan(a) != np.isnan(b)).sum()}"
        ok = np.isfinite(b)
        err = np.abs(a[ok].astype(np.float64) - b[ok])
        if n == "slope":
            tol = np.degrees(3.0 * 2.0 ** -24 / np.sqrt(np.maximum(1.0 - nz[ok] ** 2, 2.0 ** -23))) + 1e-5
        else:
            tol = 2e-6
        assert (err <= tol).all(), f"{n}: max abs err {err.max():.3e}, {(err > tol).sum()} cells over the bound"
        same = a.view(np.uint32)[ok] == b.view(np.uint32)[ok]
        assert same.mean() > 0.9, f"{n}: only {same.mean():.3f} of the cells bit-identical"


def test_after_real_scans_with_records(gpu, R):
    """The stages read/write record fields (elevation, upper/lower bounds) of a mapped scene."""
    wl = gpu.synth.vlp16(n_scans=6)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    for k in range(6):
        run_both(eng, ref, wl.scan(k), wl.T_base_sensor, wl.pose(k))
    both((eng, ref), lambda o: o.apply_uncertainty_fusion(True, 0.25, 0.1, 0.01, 0.99, 3))
    both((eng, ref), lambda o: o.apply_inpainting(3, 2))
    both((eng, ref), lambda o: o.apply_spatial_smoothing("elevation_inpainted", 3, 5))
    exact(eng, ref)
    run_both(eng, ref, wl.scan(0), wl.T_base_sensor, wl.pose(6))  # mapping goes on afterwards
    exact(eng, ref)


def test_tiled_engines_with_halo(gpu, R):
    """2x2 spatial tiles of a GLOBAL map, 6-cell halo: every stage is exact on the owned cells (the
    halo is as wide as the stencils reach) and refused when the halo is too narrow."""
    rng = np.random.default_rng(31)
    W, RES, HALO = 20.0, 0.1, 6
    gcfg = gpu.capi.default_config()
    gcfg.mode = 1
    whole = gpu.Engine(W, W, RES, gcfg)
    rows, cols = whole.rows, whole.cols
    el = terrain(rng, (rows, cols), holes=0.3)
    half = np.abs(rng.normal(0.05, 0.02, (rows, cols))).astype(F32) + F32(0.005)
    hr, hc = rows // 2, cols // 2
    tiles = []
    for tr in range(2):
        for tc in range(2):
            o_r0, o_c0 = tr * hr, tc * hc
            s_r0, s_c0 = max(0, o_r0 - HALO), max(0, o_c0 - HALO)
            s_r1, s_c1 = min(rows, o_r0 + hr + HALO), min(cols, o_c0 + hc + HALO)
            t = gpu.Engine(W, W, RES, gcfg, tile=(s_r0, s_c0, s_r1 - s_r0, s_c1 - s_c0, o_r0, o_c0, hr, hc))
            tiles.append((t, (s_r0, s_r1, s_c0, s_c1), (o_r0, o_r0 + hr, o_c0, o_c0 + hc)))
    for name, arr in (("elevation", el), ("upper_bound", el + half), ("lower_bound", el - half)):
        whole.set_layer(name, arr)
        for t, (a, b, c, d), _ in tiles:
            t.set_layer(name, arr[a:b, c:d])

    def run(o):
        o.apply_uncertainty_fusion(True, 0.6, 0.2, 0.05, 0.95, 3)   # 6 cells
        o.apply_inpainting(3, 2)                                    # 3 cells
        o.apply_spatial_smoothing("elevation_inpainted", 5, 5)      # 2 cells
        o.apply_feature_extraction(0.6, 4, 0.05, 0.95)              # 6 cells
    run(whole)
    for t, _, _ in tiles:
        run(t)
    for name in ("upper_bound", "lower_bound", "elevation_inpainted", "step", "slope", "_normal_z", "roughness"):
        full = whole.layer(name)
        for t, (a, b, c, d), (oa, ob, oc, od) in tiles:
            got = t.layer(name)[oa - a:ob - a, oc - c:od - c]
            want = full[oa:ob, oc:od]
            assert np.array_equal(np.isnan(got), np.isnan(want)), name
            assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]), name
    t0 = tiles[0][0]
    with pytest.raises(gpu.EngineError):
        t0.apply_feature_extraction(0.75, 4)          # 7 cells > halo
    with pytest.raises(gpu.EngineError):
        t0.apply_inpainting(7, 2)
