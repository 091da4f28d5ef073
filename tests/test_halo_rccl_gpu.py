"""libfdm_halo.so (include/fdm_halo.h) driving RCCL for real: a ONE-rank ncclComm on the MI355X — the scan
broadcast, a halo exchange whose plan names the rank itself as its neighbour (pack -> ncclSend / ncclRecv to self in
one group -> unpack), and the routed-scan sequence (route -> all-gather of the counters -> exchange -> integrate).
These are the calls an N-GPU C++ host makes; with one rank every collective still goes through librccl on the
engine's stream.  Run on the GPU box:  python -m pytest tests -m gpu
"""
import ctypes as C
import os

import numpy as np
import pytest

from test_halo_capi import LIB, Plan, Rect

pytestmark = pytest.mark.gpu
F32 = np.float32


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.fixture(scope="module")
def rccl():
    import torch
    assert torch.cuda.is_available()
    from fastdem_amd import capi
    capi.load()
    halo = C.CDLL(LIB)
    nccl = C.CDLL("librccl.so.1") if os.path.exists("/opt/rocm/lib/librccl.so.1") else C.CDLL("librccl.so")
    uid = UniqueId()
    assert nccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    nccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert nccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    halo.fdm_halo_last_error.restype = C.c_char_p
    halo.fdm_halo_exchange.restype = C.c_int64
    halo.fdm_halo_exchange.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Plan), C.POINTER(C.c_char_p), C.c_int32,
                                       C.c_void_p, C.c_uint64]
    halo.fdm_halo_broadcast_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32]
    halo.fdm_halo_gather_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    halo.fdm_halo_route_exchange.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Plan), C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    halo.fdm_tile_plan_make.argtypes = [C.c_int32] * 5 + [C.POINTER(Plan)]
    yield halo, comm
    nccl.ncclCommDestroy.argtypes = [C.c_void_p]
    nccl.ncclCommDestroy(comm)


def cloud(rng, n, spread):
    return {"x": rng.uniform(-spread, spread, n).astype(F32), "y": rng.uniform(-spread, spread, n).astype(F32),
            "z": (0.3 * rng.standard_normal(n)).astype(F32), "intensity": rng.uniform(0, 1, n).astype(F32)}


def global_cfg(gpu_mod):
    cfg = gpu_mod.capi.default_config()
    cfg.mode = 1
    return cfg


def test_broadcast_and_self_halo_exchange_through_rccl(rccl):
    import torch
    import fastdem_amd as fa
    halo, comm = rccl
    eng = fa.Engine(12.8, 12.8, 0.2, global_cfg(fa))  # 64 x 64 cells
    rng = np.random.default_rng(1)
    s = cloud(rng, 20000, 6.0)
    packed = torch.from_numpy(np.stack([s[c] for c in ("x", "y", "z", "intensity")])).cuda()
    before = packed.clone()
    # scan distribution: ncclBroadcast from root 0 on the engine's stream (a 1-rank communicator: in place)
    assert halo.fdm_halo_broadcast_scan(eng._h, comm, C.c_void_p(packed.data_ptr()), packed.numel(), 0) == 0, \
        halo.fdm_halo_last_error()
    T = np.eye(4)
    eng.integrate_device(packed[0], packed[1], packed[2], T, T, intensity=packed[3])
    eng.sync()
    assert torch.equal(packed, before)
    # a plan whose only neighbour is the rank itself: rows 0..5 travel to rows 58..63 through ncclSend / ncclRecv
    p = Plan()
    p.rank, p.world, p.rows, p.cols, p.halo, p.grid_rows, p.grid_cols = 0, 1, 64, 64, 6, 1, 1
    p.owned = Rect(0, 0, 64, 64)
    p.stored = Rect(0, 0, 64, 64)
    p.n_sends = p.n_recvs = 1
    p.send_rank[0] = p.recv_rank[0] = 0
    p.send_rect[0] = Rect(0, 0, 6, 64)
    p.recv_rect[0] = Rect(58, 0, 6, 64)
    names = [n for n in eng.layers() if not n.startswith("_")]
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    ws = torch.empty(2 * 6 * 64 * len(names), dtype=torch.float32, device="cuda")
    want = {n: eng.layer(n) for n in names}
    sent = halo.fdm_halo_exchange(eng._h, comm, C.byref(p), arr, len(names), C.c_void_p(ws.data_ptr()), ws.numel() * 4)
    assert sent == 6 * 64 * len(names) * 4, halo.fdm_halo_last_error()
    eng.sync()
    moved = 0
    for n in names:
        got = eng.layer(n)
        assert np.array_equal(got[58:64].view(np.uint32), want[n][0:6].view(np.uint32)), n
        assert np.array_equal(got[:58].view(np.uint32), want[n][:58].view(np.uint32)), n
        moved += int(np.isfinite(want[n][0:6]).sum())
    assert moved > 100  # the strip carried data


def test_routed_scan_sequence_through_rccl(rccl):
    """route -> ncclAllGather of the counters -> fdm_halo_route_exchange (the rank's own share) -> integrate of the
    received records: the map equals the plain engine's, bit for bit."""
    import torch
    import fastdem_amd as fa
    halo, comm = rccl
    eng = fa.Engine(40.0, 40.0, 0.1, global_cfg(fa))
    ref = fa.Engine(40.0, 40.0, 0.1, global_cfg(fa))
    plan = Plan()
    assert halo.fdm_tile_plan_make(0, 1, 400, 400, 6, C.byref(plan)) == 0
    rp = fa.capi.FdmRoutePlan()
    halo.fdm_tile_plan_route(C.byref(plan), C.byref(rp))
    assert (rp.world, rp.grid_rows, rp.grid_cols, rp.row_edge[0], rp.row_edge[1], rp.col_edge[1]) == (1, 1, 1, 0, 400, 400)
    rng = np.random.default_rng(2)
    T = np.eye(4)
    T[:3, 3] = (0.5, -0.25, 0.4)
    for k in range(3):
        s = cloud(rng, 150000, 25.0)  # a good part outside the 40 x 40 m map
        d = {c: torch.from_numpy(s[c]).cuda() for c in s}
        n = s["x"].size
        send = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        counts = torch.zeros(3, dtype=torch.int32, device="cuda")
        eng.route_scan(rp, d["x"], d["y"], d["z"], T, T, send, counts, intensity=d["intensity"])
        dmat = torch.zeros(3, dtype=torch.int32, device="cuda")
        hmat = (C.c_uint32 * 3)()
        assert halo.fdm_halo_gather_counts(eng._h, comm, C.c_void_p(counts.data_ptr()), C.c_void_p(dmat.data_ptr()),
                                           hmat, 1) == 0, halo.fdm_halo_last_error()
        recv = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        n_recv, any_in = C.c_uint64(0), C.c_int32(0)
        assert halo.fdm_halo_route_exchange(eng._h, comm, C.byref(plan), C.c_void_p(send.data_ptr()), hmat,
                                            C.c_void_p(recv.data_ptr()), n, C.byref(n_recv), C.byref(any_in)) == 0, \
            halo.fdm_halo_last_error()
        eng.integrate_points4_device(recv, n_recv.value, T, T, has_intensity=True, any_in_map=bool(any_in.value))
        rc, st = ref.integrate(s["x"], s["y"], s["z"], T, T, intensity=s["intensity"])
        assert (hmat[0], hmat[1], hmat[2]) == (st["n_in_map"], st["n_after_filter"], st["n_in_map"])
        assert n_recv.value == st["n_in_map"] and any_in.value == 1
    for name in ref.layers():
        assert np.array_equal(eng.layer(name).view(np.uint32), ref.layer(name).view(np.uint32)), name


@pytest.mark.parametrize("sensors", [False, True])
def test_routed_step_as_one_c_call(rccl, sensors):
    """fdm_halo_routed_step (fastdem_amd/halo.py::NativeRoutedScan): route -> table all-gather + one read-back -> point
    exchange -> integrate, inside ONE call — with a 1-rank communicator and without one (world == 1 needs none).  The
    map equals the plain engine's bit for bit; the counter matrix is the plain engine's statistics."""
    import torch
    import fastdem_amd as fa
    from fastdem_amd import halo as H
    _, comm = rccl
    ref = fa.Engine(40.0, 40.0, 0.1, global_cfg(fa))
    engs = [fa.Engine(40.0, 40.0, 0.1, global_cfg(fa)) for _ in range(2)]
    routed = [H.NativeRoutedScan(engs[0], 0, 1, 400, 400, 6, 200000, comm=comm),
              H.NativeRoutedScan(engs[1], 0, 1, 400, 400, 6, 200000, comm=None)]
    rng = np.random.default_rng(7)
    for k in range(4):
        n = 150000 if k != 2 else 777
        s = cloud(rng, n, 25.0)
        if k == 3:
            s["z"] = (s["z"] + 50.0).astype(F32)  # every point filtered: no move, no obstacle clear
        T = np.eye(4)
        T[:3, 3] = (0.5 + 0.3 * k, -0.25, 0.4)
        d = {c: torch.from_numpy(s[c]).cuda() for c in s}
        rc, st = ref.integrate(s["x"], s["y"], s["z"], np.eye(4), T, intensity=s["intensity"])
        for r in routed:
            m = r.integrate(d["x"], d["y"], d["z"], np.eye(4), T, intensity=d["intensity"], sensors=sensors)
            assert m.shape == (1, 3) and (m[0, 0], m[0, 1], m[0, 2]) == (st["n_in_map"], st["n_after_filter"], st["n_in_map"])
    for e in engs:
        e.sync()
        for name in ref.layers():
            assert np.array_equal(e.layer(name).view(np.uint32), ref.layer(name).view(np.uint32)), name
    for r in routed:
        r.close()


def test_routed_step_pipelined_over_consecutive_scans(rccl):
    """fdm_halo_routed_submit: scan k+1 is routed before scan k is exchanged and integrated; after flush() the map equals
    the plain engine's, and the matrix a call returns is the previous scan's."""
    import torch
    import fastdem_amd as fa
    from fastdem_amd import halo as H
    _, comm = rccl
    ref = fa.Engine(40.0, 40.0, 0.1, global_cfg(fa))
    eng = fa.Engine(40.0, 40.0, 0.1, global_cfg(fa))
    r = H.NativeRoutedScan(eng, 0, 1, 400, 400, 6, 200000, comm=comm)
    rng = np.random.default_rng(8)
    stats, keep = [], []
    for k in range(6):
        n = 120000 if k % 2 == 0 else 3000
        s = cloud(rng, n, 25.0)
        T = np.eye(4)
        T[:3, 3] = (0.2 * k, -0.1 * k, 0.4)
        d = {c: torch.from_numpy(s[c]).cuda() for c in s}
        keep.append(d)
        rc, st = ref.integrate(s["x"], s["y"], s["z"], np.eye(4), T, intensity=s["intensity"])
        stats.append(st)
        m = r.integrate(d["x"], d["y"], d["z"], np.eye(4), T, intensity=d["intensity"], sensors=(k >= 3), pipelined=True)
        if k:
            assert (m[0, 0], m[0, 1]) == (stats[k - 1]["n_in_map"], stats[k - 1]["n_after_filter"])
    m = r.flush()
    assert (m[0, 0], m[0, 1]) == (stats[-1]["n_in_map"], stats[-1]["n_after_filter"])
    eng.sync()
    for name in ref.layers():
        assert np.array_equal(eng.layer(name).view(np.uint32), ref.layer(name).view(np.uint32)), name
    r.close()
