"""Shared helpers for the parity tests (engine vs oracle on identical inputs)."""
import numpy as np

RTOL = 1e-5   # north_star: fused height/variance within 1e-5 relative
ATOL = 1e-7   # abs floor (SURVEY.md §8d parity gate)


def assert_layers_equal(a_eng, b_ref, names=None, rtol=RTOL, atol=ATOL):
    """NaN pattern identical, finite values within rtol relative (abs floor atol).
    Returns the worst relative error seen."""
    na, nb = a_eng.layers(), b_ref.layers()
    assert sorted(na) == sorted(nb), (na, nb)
    worst = 0.0
    for name in (names or nb):
        a, b = a_eng.layer(name), b_ref.layer(name)
        worst = max(worst, assert_arrays_close(a, b, name, rtol, atol))
    return worst


def assert_arrays_close(a, b, name="", rtol=RTOL, atol=ATOL):
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if name == "color":  # packed 0x00RRGGBB bit patterns are denormal floats: compare bits
        ia, ib = a.view(np.uint32), b.view(np.uint32)
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        assert np.array_equal(nan_a, nan_b), f"{name}: NaN pattern differs"
        assert np.array_equal(ia[~nan_a], ib[~nan_b]), f"{name}: packed colours differ"
        return 0.0
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{name}: NaN pattern differs in {(na != nb).sum()} cells"
    fa, fb = a[~na].astype(np.float64), b[~nb].astype(np.float64)
    inf = np.isinf(fb)
    assert np.array_equal(fa[inf], fb[inf]), f"{name}: inf mismatch"
    fa, fb = fa[~inf], fb[~inf]
    if fa.size == 0:
        return 0.0
    if rtol == 0.0:
        assert np.array_equal(fa, fb), f"{name}: {(fa != fb).sum()} finite values differ (exact compare)"
        return 0.0
    err = np.abs(fa - fb) / np.maximum(np.abs(fb), atol / rtol)
    assert err.max() <= rtol, f"{name}: max rel err {err.max():.3e} at {err.argmax()}"
    return float(err.max())


def assert_layers_bit_identical(a_eng, b_ref, names=None):
    """Every non-NaN value bit for bit (the SIGN OF A ZERO included), NaN pattern identical."""
    na, nb = a_eng.layers(), b_ref.layers()
    assert sorted(na) == sorted(nb), (na, nb)
    for name in (names or nb):
        a, b = a_eng.layer(name), b_ref.layer(name)
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        if name == "color":
            nan_a, nan_b = np.zeros_like(nan_a), np.zeros_like(nan_b)
        assert np.array_equal(nan_a, nan_b), f"{name}: NaN pattern differs"
        ia, ib = a.view(np.uint32)[~nan_a], b.view(np.uint32)[~nan_b]
        bad = ia != ib
        assert not bad.any(), f"{name}: {int(bad.sum())} values differ in their bits, e.g. {ia[bad][:3]} vs {ib[bad][:3]}"


def same_geometry(ge, gr):
    return (ge.position_x, ge.position_y, ge.start_row, ge.start_col, ge.rows, ge.cols,
            ge.resolution, ge.length_x, ge.length_y) == \
           (gr.position_x, gr.position_y, gr.start_row, gr.start_col, gr.rows, gr.cols,
            gr.resolution, gr.length_x, gr.length_y)


def pair(fastdem_amd, R, width, height, res, fill_cfg=None, position=(0.0, 0.0)):
    """Engine + oracle built from the same configuration."""
    ce, cr = fastdem_amd.capi.default_config(), R.default_config()
    if fill_cfg:
        fill_cfg(ce)
        fill_cfg(cr)
    eng = fastdem_amd.Engine(width, height, res, ce, position=position)
    ref = R.RefEngine(width, height, res, cr, position=position)
    eng.enable_cell_ids()
    ref.enable_cell_ids()
    return eng, ref


def run_both(eng, ref, s, Tbs, Twb, check_ids=True):
    kw = {}
    if s.get("intensity") is not None:
        kw["intensity"] = s["intensity"]
    if s.get("rgb") is not None:
        kw["rgb"] = s["rgb"]
    rc_e, st_e = eng.integrate(s["x"], s["y"], s["z"], Tbs, Twb, **kw)
    rc_r, st_r = ref.integrate(s["x"], s["y"], s["z"], Tbs, Twb, **kw)
    assert rc_e == rc_r, (rc_e, rc_r)
    assert st_e == st_r, (st_e, st_r)
    if check_ids and s["x"].size:
        n = int(s["x"].size)
        ie, ir = eng.last_cell_ids(n), ref.last_cell_ids(n)
        assert np.array_equal(ie, ir), f"cell ids differ at {(ie != ir).sum()} of {n} points"
    return rc_e, st_e
