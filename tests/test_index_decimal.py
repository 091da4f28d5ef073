"""An independent pin of the index oracle: getIndex / getPosition / move() of oracle/fdm_grid.hpp checked against
arbitrary-precision arithmetic written here from the definition (Python `decimal` and integers) — NOT against
anything derived from fdm_grid.hpp.  ("the index oracle is checked by something that is not itself".)

Definition being pinned (grid_map_core lineage; SURVEY.md §8c "semantics the restatement adopts"):
  row = floor((center_x + length_x / 2 - x) / res)   col = floor((center_y + length_y / 2 - y) / res)
  inside  <=>  0 <= center + length / 2 - pos < length     (the upper map edge belongs to the map, the lower does not)
  buffer index = (logical index + start index) mod size;   cell centre x = center_x + length_x / 2 - res / 2 - row * res
  move(p): shift = round-half-away((p - center) / res) cells per axis; start index -= shift (mod size); center += shift * res;
           the rows / columns that scroll out are NaN in every layer (everything if |shift| >= size)
Cases are built from dyadic numbers (resolution 0.5 / 0.25 / 0.125, positions k / 64), for which the reference's fp64
arithmetic is EXACT — so points sitting precisely on cell edges, on the map border and on shift-rounding ties have one
right answer, and the decimal model gives it.  A second set uses the float-promoted resolution double(0.1f) with random
positions kept 1e-9 cells away from any edge, where fp64 and exact arithmetic agree.
"""
from decimal import Decimal, getcontext, ROUND_FLOOR

import numpy as np
import pytest

getcontext().prec = 60
F32 = np.float32


def exact_index(x, y, center, length, res, size, start):
    """(row, col) buffer index or None, in exact arithmetic (all arguments Decimal / int)."""
    out = []
    for pos, c, ln, n, s0 in ((x, center[0], length[0], size[0], start[0]), (y, center[1], length[1], size[1], start[1])):
        t = c + ln / 2 - pos
        if not (Decimal(0) <= t < ln):
            return None
        k = int((t / res).to_integral_value(rounding=ROUND_FLOOR))
        k = min(k, n - 1)  # t == ln cannot happen (t < ln); floor(t / res) <= n - 1 exactly
        out.append((k + s0) % n)
    return tuple(out)


def linear(rc, rows):
    return rc[1] * rows + rc[0]


@pytest.mark.parametrize("res,w,h", [(0.5, 10.0, 7.0), (0.25, 6.0, 6.0), (0.125, 4.0, 2.5)])
def test_get_index_on_dyadic_grids_matches_exact_arithmetic(R, res, w, h):
    cfg = R.default_config()
    cfg.mode = 1
    ref = R.RefEngine(w, h, res, cfg, position=(0.75, -1.5))
    ref.enable_cell_ids()
    g = ref.geometry()
    rows, cols = g.rows, g.cols
    assert (rows, cols) == (round(w / res), round(h / res))
    D = Decimal
    center, length = (D(g.position_x), D(g.position_y)), (D(g.length_x), D(g.length_y))
    # every multiple of res / 4 across the map and one cell beyond: cell edges, centres, the map border, outside
    xs = np.arange(-w / 2 - res, w / 2 + res + res / 8, res / 4) + 0.75
    ys = np.arange(-h / 2 - res, h / 2 + res + res / 8, res / 4) - 1.5
    for sr, sc in ((0, 0), (3, 0), (0, 5), (rows - 1, cols - 2)):
        ref.set_start_index(sr, sc)
        X, Y = np.meshgrid(xs, ys, indexing="ij")
        x, y = X.ravel().astype(F32), Y.ravel().astype(F32)
        assert np.array_equal(x.astype(np.float64), X.ravel()) and np.array_equal(y.astype(np.float64), Y.ravel())  # dyadic
        ref.update(x, y, np.zeros_like(x))
        ids = ref.last_cell_ids(x.size)
        on_edges = 0
        for i in range(x.size):
            want = exact_index(D(float(x[i])), D(float(y[i])), center, length, D(res), (rows, cols), (sr, sc))
            got = int(ids[i])
            assert got == (linear(want, rows) if want is not None else got if got < 0 else -99), \
                (float(x[i]), float(y[i]), want, got, (sr, sc))
            assert (want is None) == (got < 0)
            on_edges += 1 if want is not None and (D(float(x[i])) - center[0] - length[0] / 2) % D(res) == 0 else 0
            # getPosition of that cell is its centre, in exact arithmetic
            if want is not None and i % 37 == 0:
                ok, (px, py) = ref.get_position(*want)
                assert ok
                ur, uc = (want[0] - sr) % rows, (want[1] - sc) % cols
                assert D(px) == center[0] + length[0] / 2 - D(res) / 2 - ur * D(res)
                assert D(py) == center[1] + length[1] / 2 - D(res) / 2 - uc * D(res)
        assert on_edges > 50  # the sweep did sit on cell edges
        ref.clear()


def test_get_index_with_the_float_promoted_resolution_away_from_edges(R):
    """ElevationMap::setGeometry(float ...) promotes 0.1f to double: res = 0.100000001490116...  Random positions at
    least 1e-9 cell away from every edge: exact arithmetic on the DOUBLE inputs and fp64 agree."""
    cfg = R.default_config()
    cfg.mode = 1
    ref = R.RefEngine(15.0, 15.0, 0.1, cfg, position=(3.2, -7.7))
    ref.enable_cell_ids()
    g = ref.geometry()
    D = Decimal
    res = D(g.resolution)
    assert res == D(float(F32(0.1))) and (g.rows, g.cols) == (150, 150)
    center, length = (D(g.position_x), D(g.position_y)), (D(g.length_x), D(g.length_y))
    assert length[0] == 150 * res
    rng = np.random.default_rng(12)
    ref.set_start_index(149, 17)
    x = (rng.uniform(-9, 9, 4000) + 3.2).astype(F32)
    y = (rng.uniform(-9, 9, 4000) - 7.7).astype(F32)
    ref.update(x, y, np.zeros_like(x))
    ids = ref.last_cell_ids(x.size)
    checked = 0
    for i in range(x.size):
        tx = center[0] + length[0] / 2 - D(float(x[i]))
        ty = center[1] + length[1] / 2 - D(float(y[i]))
        if min(abs((tx / res) - (tx / res).to_integral_value()), abs((ty / res) - (ty / res).to_integral_value())) < D("1e-9"):
            continue
        want = exact_index(D(float(x[i])), D(float(y[i])), center, length, res, (150, 150), (149, 17))
        assert (int(ids[i]) < 0) == (want is None)
        if want is not None:
            assert int(ids[i]) == linear(want, 150)
        checked += 1
    assert checked > 3900


def test_move_rounding_strips_and_wrap_match_exact_arithmetic(R):
    """GridMap::move on a dyadic grid: shift rounded half away from zero (ties at k + 0.5 cells are exact here), the
    start index and the position after the move, and exactly which cells come back NaN."""
    res, rows, cols = 0.25, 24, 16
    cfg = R.default_config()
    ref = R.RefEngine(rows * res, cols * res, res, cfg, position=(0.0, 0.0))
    D = Decimal
    cx, cy, sr, sc = D(0), D(0), 0, 0
    # a finite value everywhere, so that NaN == "vacated by the move"
    ref.set_layer("elevation", np.full((rows, cols), 1.0, dtype=F32, order="F"))
    alive = np.ones((rows, cols), dtype=bool)  # in BUFFER coordinates

    def round_half_away(v):
        a = int((abs(v) + D("0.5")).to_integral_value(rounding=ROUND_FLOOR))
        return a if v >= 0 else -a

    targets = [(0.125, 0.0), (0.375, -0.125), (0.625, 0.875), (-1.0, 0.875), (-1.0, -2.625), (2.125, 1.0),
               (2.0, 1.0), (2.0 + 30 * res, 1.0), (-3.375, 1.0 - 40 * res), (-3.375, 1.0 - 40 * res)]
    for tx, ty in targets:
        ref.move(tx, ty)
        nx, ny = round_half_away((D(tx) - cx) / D(res)), round_half_away((D(ty) - cy) / D(res))
        # logical row r shows x = cx + L/2 - res/2 - r res: moving the centre by +n cells in x scrolls n rows out at the
        # high-row end and in at row 0; in buffer coordinates the vacated rows are [sr - n, sr) (mod rows)
        for n, size, axis, s0 in ((nx, rows, 0, sr), (ny, cols, 1, sc)):
            if abs(n) >= size:
                alive[:] = False
            elif n > 0:
                idx = [(s0 - 1 - k) % size for k in range(n)]
                if axis == 0:
                    alive[idx, :] = False
                else:
                    alive[:, idx] = False
            elif n < 0:
                idx = [(s0 + k) % size for k in range(-n)]
                if axis == 0:
                    alive[idx, :] = False
                else:
                    alive[:, idx] = False
        sr, sc = (sr - nx) % rows, (sc - ny) % cols
        cx, cy = cx + nx * D(res), cy + ny * D(res)
        g = ref.geometry()
        assert (D(g.position_x), D(g.position_y), g.start_row, g.start_col) == (cx, cy, sr, sc), (tx, ty)
        got = np.isfinite(ref.layer("elevation"))
        assert np.array_equal(got, alive), (tx, ty, int((got != alive).sum()))
        # refill, so that every move is checked on its own
        ref.set_layer("elevation", np.full((rows, cols), 1.0, dtype=F32, order="F"))
        alive[:] = True
