"""Which layers GridMap::move() clears in the strips it vacates — the switch for the reading this repo does NOT assume.

The oracle (and the engine) clear EVERY layer there: "basicLayers is empty for ElevationMap" (oracle/fdm_grid.hpp
clearStrip, ASSUMED — nanoGrid is not on disk).  The reference hints the other way: it constructs
`nanogrid::GridMap({elevation, elevation_min, elevation_max})` (elevation_map.hpp:101-103) and its tests speak of
"basicLayers = {elevation}" (tests/test_elevation_map.cpp:91); in grid_map_core clearRows / clearCols take the basic layers
when that list is not empty.  Until scripts/conformance/probe.cpp has been run against the real library, the other
reading is one option away on both sides — engine option `move_clear_basic`, oracle `set_move_clear_basic` — and held to
the same bar: every layer bit for bit, both estimators, both pipelines, strips that wrap, a touched cell inside a strip
(Welford goes on from the stale mean instead of starting over: the observable difference), a jump beyond the map
(clearAll either way), explicit move().  Through the C ABI.

Run on the GPU box:  python -m pytest tests -m gpu
"""
import numpy as np
import pytest

from helpers import assert_layers_bit_identical, pair, run_both, same_geometry
from test_batch_gpu import T

pytestmark = pytest.mark.gpu
F32 = np.float32


def cloud(rng, n, spread=6.0):
    return {"x": rng.uniform(-spread, spread, n).astype(F32), "y": rng.uniform(-spread, spread, n).astype(F32),
            "z": (rng.uniform(-0.3, 0.3, n) - 1.0).astype(F32), "intensity": rng.uniform(0, 1, n).astype(F32), "rgb": None}


@pytest.mark.parametrize("estimator", [0, 1])
@pytest.mark.parametrize("n", [900, 5000])   # (the default fixture variant: per-cell scratch / record pools; tiled_min is 2 048)
def test_strips_clear_the_basic_layers_only(gpu, R, estimator, n):
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -3.0, 3.0, 0.0, 30.0
        c.estimation_type = estimator
    eng, ref = pair(gpu, R, 40.0, 40.0, 0.1, fill)            # 400 x 400: enough tiles for the record pools
    allm, allr = pair(gpu, R, 40.0, 40.0, 0.1, fill)          # ... and the default reading beside it
    eng.set_option("move_clear_basic", 1)
    ref.set_move_clear_basic(1)
    for o in (eng, ref, allm, allr):
        o.add("user", 2.5)
    rng = np.random.default_rng(11 + n + estimator)
    Tbs = T(0.0, 0.0, 1.0)
    poses = [T(0, 0), T(0.35, -0.2), T(0.35, -0.2), T(3.1, 2.7), T(-16.0, 9.0), T(-16.3, 9.4), T(-15.9, 30.0),
             T(40.0, 30.0), T(40.2, 30.1), T(41.0, 29.0), T(41.05, 29.0)]
    for k, Twb in enumerate(poses):
        s = cloud(rng, n)
        run_both(eng, ref, s, Tbs, Twb)
        run_both(allm, allr, s, Tbs, Twb)
        if k == 5:   # an explicit move() between two scans
            for o in (eng, ref, allm, allr):
                o.move(-16.0, 9.9)
            # the switch does something: so far every move was shorter than the map — the user layer has survived all
            # their strips in the one reading and lost them in the other; so has the estimator's hidden state
            u_b, u_a = eng.layer("user"), allm.layer("user")
            assert np.isfinite(u_b).all() and np.isfinite(u_a).sum() < u_a.size
            assert np.array_equal(np.isnan(eng.layer("elevation")), np.isnan(allm.layer("elevation")))
            assert not np.array_equal(eng.layer("n_points"), allm.layer("n_points"), equal_nan=True)
    assert_layers_bit_identical(eng, ref)
    assert_layers_bit_identical(allm, allr)
    assert same_geometry(eng.geometry(), ref.geometry()) and same_geometry(eng.geometry(), allm.geometry())
    # (the jumps beyond the map at the end are clearAll() in both readings)
    assert not np.isfinite(eng.layer("user")).any() and not np.isfinite(allm.layer("user")).any()


def test_a_batch_call_on_such_an_engine_takes_the_scan_by_scan_path(gpu, R):
    """The batch kernels (fdm_multi.hpp) implement the default reading only: an engine with the switch on must not take them."""
    from test_batch_gpu import DeviceBatch, oracle_scan_by_scan
    def fill(c):
        c.z_min, c.z_max, c.range_min, c.range_max = -3.0, 3.0, 0.0, 30.0
    eng, ref = pair(gpu, R, 15.0, 15.0, 0.1, fill)
    eng.set_option("move_clear_basic", 1)
    ref.set_move_clear_basic(1)
    eng.enable_cell_ids(False)
    rng = np.random.default_rng(3)
    scans = [cloud(rng, 700, 5.0) for _ in range(20)]
    poses = [T(0.17 * k, -0.11 * k) for k in range(20)]
    before = sum(eng.batch_launches())
    b = DeviceBatch(gpu, scans, T(0, 0, 1.0), poses)
    assert eng.integrate_device_batch(b.arr) == 0
    oracle_scan_by_scan(ref, scans, T(0, 0, 1.0), poses)
    assert sum(eng.batch_launches()) == before
    assert_layers_bit_identical(eng, ref)
