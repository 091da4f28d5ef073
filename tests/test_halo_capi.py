"""CPU-side checks of include/fdm_halo.h (libfdm_halo.so): the library loads, exports every declared
symbol, and its tile plan is the plan fastdem_amd.tiling computes (the torch.distributed path and a C++
host with RCCL cut the map the same way).  No compute calls, no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("FDM_HALO_LIB") or os.path.join(ROOT, "fastdem_amd", "lib", "libfdm_halo.so")  # (scripts/asan_cpu.sh: the sanitizer build)


class Rect(C.Structure):
    _fields_ = [("r0", C.c_int32), ("c0", C.c_int32), ("nr", C.c_int32), ("nc", C.c_int32)]

    def t(self):
        return (self.r0, self.c0, self.nr, self.nc)


class Plan(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("halo", C.c_int32), ("grid_rows", C.c_int32), ("grid_cols", C.c_int32),
                ("owned", Rect), ("stored", Rect), ("n_sends", C.c_int32), ("n_recvs", C.c_int32),
                ("send_rank", C.c_int32 * 8), ("send_rect", Rect * 8),
                ("recv_rank", C.c_int32 * 8), ("recv_rect", Rect * 8)]


class Tile(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("row0", "col0", "rows", "cols", "own_row0", "own_col0", "own_rows", "own_cols")]


def lib():
    from fastdem_amd import capi
    capi.load()  # libfdm_engine.so first (RTLD_GLOBAL), as a host application links both
    return C.CDLL(LIB)


def test_library_exports_every_declared_symbol():
    src = open(os.path.join(ROOT, "include", "fdm_halo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(fdm_[a-z0-9_]+)\s*\(", src)))
    assert "fdm_halo_exchange" in syms and "fdm_tile_plan_make" in syms
    h = lib()
    assert not [s for s in syms if not hasattr(h, s)]


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8])
@pytest.mark.parametrize("shape", [(8000, 8000), (1201, 37), (64, 4000), (2000, 2000)])
@pytest.mark.parametrize("halo", [0, 1, 6])
def test_plan_equals_the_torch_distributed_plan(world, shape, halo):
    from fastdem_amd import tiling
    h = lib()
    rows, cols = shape
    for rank in range(world):
        want = tiling.make_plan(rank, world, rows, cols, halo)
        p = Plan()
        rc = h.fdm_tile_plan_make(rank, world, rows, cols, halo, C.byref(p))
        assert rc == 0, h.fdm_halo_last_error()
        assert (p.grid_rows, p.grid_cols) == tiling.grid_for(world)
        o, s = want.owned, want.stored
        assert p.owned.t() == (o.r0, o.c0, o.nr, o.nc) and p.stored.t() == (s.r0, s.c0, s.nr, s.nc)
        sends = {p.send_rank[k]: p.send_rect[k].t() for k in range(p.n_sends)}
        recvs = {p.recv_rank[k]: p.recv_rect[k].t() for k in range(p.n_recvs)}
        assert sends == {k: (r.r0, r.c0, r.nr, r.nc) for k, r in want.sends.items()}
        assert recvs == {k: (r.r0, r.c0, r.nr, r.nc) for k, r in want.recvs.items()}
        t = Tile()
        h.fdm_tile_plan_tile(C.byref(p), C.byref(t))
        assert tuple(getattr(t, n) for n, _ in Tile._fields_) == want.fdm_tile()
        h.fdm_halo_workspace_bytes.restype = C.c_uint64
        cells = sum(r.nr * r.nc for r in want.sends.values()) + sum(r.nr * r.nc for r in want.recvs.values())
        assert h.fdm_halo_workspace_bytes(C.byref(p), 9) == cells * 9 * 4


def test_bad_arguments_fail_loudly():
    h = lib()
    p = Plan()
    assert h.fdm_tile_plan_make(2, 2, 100, 100, 6, C.byref(p)) < 0      # rank outside the world
    assert h.fdm_tile_plan_make(5, 16, 100, 100, 80, C.byref(p)) < 0    # halo wider than a tile: > 8 neighbours
    h.fdm_halo_exchange.restype = C.c_int64
    assert h.fdm_halo_exchange(None, None, C.byref(p), None, 0, None, C.c_uint64(0)) < 0


def test_route_plan_edges_match_the_python_plan():
    """fdm_tile_plan_route (C++ hosts) and tiling.route_plan (torch.distributed) cut the owners the same way."""
    from fastdem_amd import capi, tiling
    L = lib()
    for world, rows, cols in ((1, 400, 400), (2, 8000, 8000), (4, 2000, 2000), (8, 8000, 8000), (6, 1001, 777), (16, 4096, 333)):
        for rank in (0, world - 1):
            p = Plan()
            assert L.fdm_tile_plan_make(rank, world, rows, cols, 6, C.byref(p)) == 0
            rp = capi.FdmRoutePlan()
            L.fdm_tile_plan_route(C.byref(p), C.byref(rp))
            py = tiling.route_plan(tiling.make_plan(rank, world, rows, cols, 6))
            assert (rp.world, rp.grid_rows, rp.grid_cols) == (py.world, py.grid_rows, py.grid_cols)
            assert list(rp.row_edge)[:rp.grid_rows + 1] == list(py.row_edge)[:py.grid_rows + 1]
            assert list(rp.col_edge)[:rp.grid_cols + 1] == list(py.col_edge)[:py.grid_cols + 1]
            # the edges are the owned rects of the plan
            pr, pc = tiling.grid_for(world)
            o = tiling.owned_rect(rank, world, rows, cols)
            i, j = divmod(rank, pc)
            assert (rp.row_edge[i], rp.row_edge[i + 1], rp.col_edge[j], rp.col_edge[j + 1]) == (o.r0, o.r1, o.c0, o.c1)
