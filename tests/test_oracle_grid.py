"""Index arithmetic of the oracle grid ("parity unpinned" vs upstream nanoGrid, which is not
on disk) checked against the three in-tree restatements of the same geometry in the
reference, and the move() semantics of the grid_map_core lineage.  CPU only."""
import numpy as np

F32 = np.float32


def test_index_matches_raycasting_formula(R):
    # fastdem/src/raycasting.cpp:63-80,112-113: row = floor((center.x + length.x/2 - x)/res),
    # buffer index = (logical + startIndex) mod size
    rng = np.random.default_rng(0)
    e = R.RefEngine(15.0, 15.0, 0.1)
    e.move(3.33, -1.27)  # non-trivial start index + position
    g = e.geometry()
    for _ in range(2000):
        x = g.position_x + rng.uniform(-7.4, 7.4)
        y = g.position_y + rng.uniform(-7.4, 7.4)
        fr = (g.position_x + g.length_x / 2 - x) / g.resolution
        fc = (g.position_y + g.length_y / 2 - y) / g.resolution
        if min(abs(fr - round(fr)), abs(fc - round(fc))) < 1e-6:
            continue  # cell-boundary rounding is exactly what is unpinned
        ok, (r, c) = e.get_index(x, y)
        assert ok
        assert r == (int(np.floor(fr)) + g.start_row) % g.rows
        assert c == (int(np.floor(fc)) + g.start_col) % g.cols


def test_cell_centre_matches_bridge_formula(R):
    # fastdem/include/fastdem/bridge/ros/impl.hpp:43-63 and src/pcd_convert.cpp:335-348:
    # x = center.x + length.x/2 - res/2 - unwrapped_row*res
    e = R.RefEngine(10.0, 10.0, 0.05)
    e.move(-0.73, 2.41)
    g = e.geometry()
    ox = g.position_x + g.length_x / 2.0 - g.resolution / 2.0
    oy = g.position_y + g.length_y / 2.0 - g.resolution / 2.0
    for r, c in [(0, 0), (5, 7), (199, 199), (g.start_row, g.start_col), (17, 123)]:
        ok, (x, y) = e.get_position(r, c)
        assert ok
        ur = (r - g.start_row + g.rows) % g.rows
        uc = (c - g.start_col + g.cols) % g.cols
        assert abs(x - (ox - ur * g.resolution)) < 1e-9
        assert abs(y - (oy - uc * g.resolution)) < 1e-9
        assert e.get_index(x, y) == (True, (r, c))


def test_float_resolution_promotion(R):
    # ElevationMap::setGeometry(float...) promotes float -> double (elevation_map.hpp:112-116)
    e = R.RefEngine(15.0, 15.0, 0.1)
    g = e.geometry()
    assert g.resolution == float(F32(0.1)) and g.rows == 150 and g.cols == 150
    assert g.length_x == 150 * float(F32(0.1))


def test_map_edges(R):
    e = R.RefEngine(10.0, 10.0, 0.5)
    # upper map edge inclusive, lower exclusive (checkIfPositionWithinMap)
    assert e.get_index(5.0, 5.0) == (True, (0, 0))
    assert not e.get_index(-5.0, 0.0)[0]
    assert e.get_index(-4.999, -4.999) == (True, (19, 19))
    assert not e.get_index(float("nan"), 0.0)[0]
    assert not e.get_index(float("inf"), 0.0)[0]


def test_move_preserves_world_data_and_clears_strips(R):
    e = R.RefEngine(10.0, 10.0, 0.5)
    rows, cols = e.rows, e.cols
    # tag every cell with a unique value in every layer
    tag = np.arange(rows * cols, dtype=F32).reshape(rows, cols)
    for name in e.layers():
        e.set_layer(name, tag)
    world = {}
    for r in range(rows):
        for c in range(cols):
            world[(r, c)] = e.get_position(r, c)[1]
    sh = e.move(1.26, -2.0)  # 1.26/0.5 -> 3 cells (round half away), -2.0 -> -4 cells
    assert sh == (-3, 4)
    g = e.geometry()
    assert (g.start_row, g.start_col) == ((0 - 3) % rows, 4)
    assert abs(g.position_x - 1.5) < 1e-12 and abs(g.position_y + 2.0) < 1e-12
    kept = cleared = 0
    for name in e.layers():
        lay = e.layer(name)
        for (r, c), (x, y) in world.items():
            ok, (r2, c2) = e.get_index(x, y)
            if ok:
                assert (r2, c2) == (r, c)  # data never moves in the buffer
                assert lay[r, c] == tag[r, c]
                kept += 1
            else:
                assert np.isnan(lay[r, c])  # vacated strip is NaN in EVERY layer
                cleared += 1
    assert kept > 0 and cleared > 0


def test_move_beyond_map_clears_everything(R):
    e = R.RefEngine(10.0, 10.0, 0.5)
    e.set_layer("elevation", np.ones((20, 20), dtype=F32))
    e.move(100.0, 0.0)
    assert np.isnan(e.layer("elevation")).all()
    assert not e.get_index(0.0, 0.0)[0]  # test_fastdem_integration.cpp:198-215


def test_move_wraparound_two_regions(R):
    e = R.RefEngine(10.0, 10.0, 0.5)
    e.move(-4.0, 0.0)      # shift +8 rows: start_row = 8
    e.set_layer("elevation", np.ones((20, 20), dtype=F32))
    e.move(-4.0 + 7.0, 0.0)  # shift -14 rows from start 8 -> rows [14..19] and [0..7] cleared
    lay = e.layer("elevation")
    cleared = np.isnan(lay).all(axis=1)
    assert set(np.nonzero(cleared)[0]) == set(range(14, 20)) | set(range(0, 8))
    assert e.geometry().start_row == (8 - 14) % 20


def test_move_clear_scope_switch():
    """fdm_grid.hpp clearStrip: the default clears every layer in the strips move() vacates; with the switch only the
    basic layers {elevation, elevation_min, elevation_max} (grid_map_core's clearRows / clearCols with a non-empty
    basicLayers list; elevation_map.hpp:101-103 passes exactly these three to the GridMap constructor).  A move of >= the
    map's size is clearAll() in both readings."""
    import fdm_ref_py as R
    for basic in (0, 1):
        e = R.RefEngine(2.0, 2.0, 0.1)   # 20 x 20
        e.set_move_clear_basic(basic)
        for name in ("elevation", "elevation_min", "elevation_max", "variance", "user"):
            if not e.exists(name):
                e.add(name, 1.0)
            e.set_layer(name, np.full((20, 20), 1.0, dtype=np.float32))
        e.move(0.3, 0.0)                 # three rows vacated
        for name in ("elevation", "elevation_min", "elevation_max"):
            assert np.isnan(e.layer(name)).sum() == 3 * 20, (basic, name)
        for name in ("variance", "user"):
            assert np.isnan(e.layer(name)).sum() == (0 if basic else 3 * 20), (basic, name)
        e.move(5.0, 0.0)                 # beyond the map: clearAll()
        for name in ("elevation", "variance", "user"):
            assert np.isnan(e.layer(name)).all(), (basic, name)
