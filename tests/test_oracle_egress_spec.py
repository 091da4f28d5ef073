"""Pins the egress part of the oracle (oracle/fdm_ref_egress.hpp) with hand-derived
values of fastdem/include/fastdem/bridge/ros/impl.hpp:28-166 (the reference has no test for it)."""
import numpy as np

F32 = np.float32


def small_map(R):
    e = R.RefEngine(2.0, 1.5, 0.5)  # 4 x 3 cells, centre (0,0)
    el = e.layer("elevation")
    el[0, 0], el[1, 2], el[3, 1] = 1.0, 3.0, -2.0
    e.set_layer("elevation", el)
    return e


def test_fields_and_step(R):
    e = small_map(R)
    fields, step, data = e.pack_cloud()
    # impl.hpp:66-77: elevation -> z, internal ('_' prefix) layers skipped, getLayers() order kept
    assert fields == ["x", "y", "z", "elevation_min", "elevation_max", "variance", "n_points",
                      "upper_bound", "lower_bound", "obstacle"]
    assert step == 4 * len(fields) and data.shape == (3, len(fields))


def test_positions_and_order(R):
    e = small_map(R)
    _, _, d = e.pack_cloud()
    # column-major visiting order (impl.hpp:132-137): (0,0), (3,1), (1,2)
    # x = 0 + 2/2 - 0.25 - row*0.5 ; y = 0 + 1.5/2 - 0.25 - col*0.5  (impl.hpp:43-63)
    assert np.array_equal(d[:, :3], np.array([[0.75, 0.5, 1.0], [-0.75, 0.0, -2.0], [0.25, -0.5, 3.0]], dtype=F32))
    assert np.array_equal(d[:, 5], np.zeros(3, dtype=F32))  # Kalman variance layer starts at 0


def test_wrapped_buffer_starts_at_start_index(R):
    e = small_map(R)
    e.move(0.5, -0.5)  # one row and one col of shift: strips cleared, start index moves
    g = e.geometry()
    assert (g.start_row, g.start_col) != (0, 0)
    el = np.full((4, 3), np.nan, dtype=F32)
    el[g.start_row, g.start_col] = 7.0  # logical (0,0): the +x/+y corner of the moved map
    e.set_layer("elevation", el)
    _, _, d = e.pack_cloud()
    assert d.shape[0] == 1
    assert np.array_equal(d[0, :3], np.array([g.position_x + 0.75, g.position_y + 0.5, 7.0], dtype=F32))
    ok, (x, y) = e.get_position(g.start_row, g.start_col)
    assert ok and F32(x) == d[0, 0] and F32(y) == d[0, 1]


def test_submap_region(R):
    e = small_map(R)
    _, _, d = e.pack_cloud(sub=(1, 1, 3, 2))  # rows 1..3, cols 1..2
    assert np.array_equal(d[:, :3], np.array([[-0.75, 0.0, -2.0], [0.25, -0.5, 3.0]], dtype=F32))


def test_other_elevation_layer_and_colour(R):
    e = small_map(R)
    col = np.full((4, 3), np.nan, dtype=F32)
    col[0, 0] = np.array([0x00FF8040], dtype=np.uint32).view(F32)[0]
    e.set_layer("color", col)  # adds the layer
    fields, _, d = e.pack_cloud("elevation_max")
    assert fields[-1] == "rgb" and "elevation" in fields and "elevation_max" not in fields
    assert d.shape[0] == 0  # elevation_max is all NaN
    mx = np.full((4, 3), np.nan, dtype=F32)
    mx[0, 0] = 2.0
    e.set_layer("elevation_max", mx)
    fields, _, d = e.pack_cloud("elevation_max")
    assert d.shape[0] == 1 and d[0, 2] == 2.0
    assert d[0, -1:].view(np.uint32)[0] == 0x00FF8040


def test_inf_elevation_is_not_a_point(R):  # std::isfinite (impl.hpp:113,139)
    e = small_map(R)
    el = e.layer("elevation")
    el[2, 2] = np.inf
    e.set_layer("elevation", el)
    assert e.pack_cloud()[2].shape[0] == 3
