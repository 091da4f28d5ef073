"""The N>1 path of the spatial tiling on ONE GPU: 2 and 4 processes share device 0, each owns a tile
engine (fdm_tile: owned window + halo ring), the scan is broadcast, every rank integrates it and the halo
exchange runs through the real plan / HIP pack / HIP unpack code of fastdem_amd.tiling — over gloo with
host-staged buffers, because RCCL refuses two ranks on one device.  Every STORED window (owned cells and
the exchanged halo ring) must equal the untiled engine's map bit for bit.
Run on the GPU box:  python -m pytest tests -m gpu
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        from fastdem_amd import Engine, capi, synth, tiling
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        wl = synth.global_map(n_scans=3, size_m=100.0, n_az=2048, radius=30.0)
        rows = cols = 2000
        plan = tiling.make_plan(rank, world, rows, cols, tiling.DEFAULT_HALO)
        eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()),
                     tile=plan.fdm_tile(), device=0)
        assert (eng.rows, eng.cols) == (rows, cols)
        tile = tiling.HostStagedTile(tiling.EngineTile(eng, plan, "cuda:0"))
        whole = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()), device=0)
        names = None
        for k in range(3):
            # scan distribution as tiling.bench_global does it: one packed [4, N] tensor from rank 0
            s = wl.scan(k)
            n = s["x"].size
            packed = torch.empty((4, n), dtype=torch.float32)
            if rank == 0:
                for i, c in enumerate(("x", "y", "z", "intensity")):
                    packed[i] = torch.from_numpy(s[c])
            dist.broadcast(packed, 0)
            d = packed.cuda()
            eng.integrate_device(d[0], d[1], d[2], wl.T_base_sensor, wl.pose(k), intensity=d[3])
            names = [nm for nm in tiling.visible_layers(eng.layers())]
            # (lazily created layers exist on every rank or on none only if every tile saw the channel;
            # exchange the intersection all ranks agree on)
            have = [None] * world
            dist.all_gather_object(have, names)
            names = [nm for nm in names if all(nm in h for h in have)]
            tiling.exchange_halos(tile, plan, names, dist)
            whole.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
        st = plan.stored
        bad = []
        for nm in names:
            got = eng.layer(nm)
            want = whole.layer(nm)[st.r0:st.r1, st.c0:st.c1]
            same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
            if not same:
                nan_ok = np.array_equal(np.isnan(got), np.isnan(want))
                bad.append((nm, int((got.view(np.uint32) != want.view(np.uint32)).sum()), nan_ok))
        q.put((rank, bad, len(names)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, [("exception", traceback.format_exc(), str(e))], 0))


@pytest.mark.parametrize("world", [2, 4])
def test_tile_engines_with_halo_exchange_equal_the_untiled_map(world):
    import torch
    assert torch.cuda.is_available()
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, bad, n_names in sorted(results):
        assert not bad, f"rank {rank}: {bad}"
        assert n_names >= 8


# ---------------------------------------------------------------------------------------------
# Scan routing (fastdem_amd/csrc/fdm_route.hpp, tiling.RoutedScan): every rank holds a SLICE of the scan; the points
# travel to the owner of their cell; the owners integrate what they receive.
def _routed_worker(rank, world, port, q, backend):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        from fastdem_amd import Engine, capi, synth, tiling
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        wl = synth.global_map(n_scans=3, size_m=100.0, n_az=2048, radius=30.0)
        rows = cols = 2000
        plan = tiling.make_plan(rank, world, rows, cols, tiling.DEFAULT_HALO)
        eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()),
                     tile=plan.fdm_tile() if world > 1 else None, device=0)
        inner = tiling.EngineTile(eng, plan, "cuda:0")
        tile = tiling.HostStagedTile(inner) if backend == "gloo" else inner
        whole = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()), device=0)
        n = wl.n_points
        bounds = tiling.slice_bounds(n, world, align=4)
        lo, hi = bounds[rank]
        router = tiling.RoutedScan(eng, plan, "cuda:0", max_points=hi - lo + 8, staged=backend == "gloo")
        names = None
        totals = []
        for k in range(3):
            s = wl.scan(k)
            sl = {c: torch.from_numpy(np.ascontiguousarray(s[c][lo:hi])).cuda() for c in ("x", "y", "z", "intensity")}
            m = router.integrate(sl["x"], sl["y"], sl["z"], wl.T_base_sensor, wl.pose(k), dist, intensity=sl["intensity"])
            totals.append((int(m[:, :world].sum()), int(m[:, world].sum()), int(m[:, world + 1].sum())))
            names = [nm for nm in tiling.visible_layers(eng.layers())]
            have = [None] * world
            dist.all_gather_object(have, names)
            names = [nm for nm in names if all(nm in h for h in have)]
            tiling.exchange_halos(tile, plan, names, dist)
            rc, st = whole.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
            # the routing counters add up to the single map's statistics
            assert totals[-1] == (st["n_in_map"], st["n_after_filter"], st["n_in_map"]), (totals[-1], st)
        st_rect = plan.stored if world > 1 else tiling.Rect(0, 0, rows, cols)
        bad = []
        for nm in names:
            got = eng.layer(nm)
            want = whole.layer(nm)[st_rect.r0:st_rect.r1, st_rect.c0:st_rect.c1]
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad.append((nm, int((got.view(np.uint32) != want.view(np.uint32)).sum())))
        q.put((rank, bad, len(names)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, [("exception", traceback.format_exc(), str(e))], 0))


def _sensors_worker(rank, world, port, q, backend):
    """N sensors mode: every rank contributes ITS OWN scan (own pose) to every step; the step is N integrate() calls
    in rank order on the global map."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        import torch
        import torch.distributed as dist
        from fastdem_amd import Engine, capi, synth, tiling
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world,
                                **({"device_id": torch.device("cuda:0")} if backend == "nccl" else {}))
        wl = synth.global_map(n_scans=4, size_m=100.0, n_az=1024, radius=30.0)
        rows = cols = 2000
        plan = tiling.make_plan(rank, world, rows, cols, tiling.DEFAULT_HALO)
        eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()),
                     tile=plan.fdm_tile() if world > 1 else None, device=0)
        inner = tiling.EngineTile(eng, plan, "cuda:0")
        tile = tiling.HostStagedTile(inner) if backend == "gloo" else inner
        whole = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()), device=0)
        router = tiling.RoutedScan(eng, plan, "cuda:0", max_points=wl.n_points, staged=backend == "gloo")
        pose_of = lambda k, r: wl.pose(11 * k + 37 * r)  # noqa: E731  (robots far apart on the circle; some overlap over time)
        names = None
        for k in range(4):
            s = wl.scan((k + rank) % 4)
            d = {c: torch.from_numpy(s[c]).cuda() for c in ("x", "y", "z", "intensity")}
            router.integrate(d["x"], d["y"], d["z"], wl.T_base_sensor, pose_of(k, rank), dist, intensity=d["intensity"],
                             sensors=True)
            names = [nm for nm in tiling.visible_layers(eng.layers())]
            have = [None] * world
            dist.all_gather_object(have, names)
            names = [nm for nm in names if all(nm in h for h in have)]
            tiling.exchange_halos(tile, plan, names, dist)
            for r in range(world):  # the reference order: one integrate() per sensor, rank order
                sr = wl.scan((k + r) % 4)
                whole.integrate(sr["x"], sr["y"], sr["z"], wl.T_base_sensor, pose_of(k, r), intensity=sr["intensity"])
        st_rect = plan.stored if world > 1 else tiling.Rect(0, 0, rows, cols)
        bad = []
        for nm in names:
            got = eng.layer(nm)
            want = whole.layer(nm)[st_rect.r0:st_rect.r1, st_rect.c0:st_rect.c1]
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad.append((nm, int((got.view(np.uint32) != want.view(np.uint32)).sum())))
        q.put((rank, bad, len(names)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, [("exception", traceback.format_exc(), str(e))], 0))


def _native_worker(rank, world, port, q, mode):
    """libfdm_halo's routed step (fdm_halo_routed_step / _submit + fdm_halo_exchange: the C code bench.py's c5 runs over
    RCCL) with REAL peers: the library's transport table is pointed at a host-staged one over gloo
    (fastdem_amd.halo.HostStagedTransport), everything else — routing kernels, per-peer offsets, one group per
    exchange, source order, pack / unpack — is the production path.  mode: "slices" (one scan cut into slices),
    "sensors" (one scan per rank), "pipelined" (sensors, software-pipelined over consecutive scans)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        import ctypes as C
        import torch
        import torch.distributed as dist
        from fastdem_amd import Engine, capi, synth, tiling
        from fastdem_amd import halo as H
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        wl = synth.global_map(n_scans=4, size_m=100.0, n_az=1024, radius=30.0)
        rows = cols = 2000
        plan = tiling.make_plan(rank, world, rows, cols, tiling.DEFAULT_HALO)
        eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()),
                     tile=plan.fdm_tile() if world > 1 else None, device=0)
        whole = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()), device=0)
        transport = H.HostStagedTransport(dist)
        transport.install()
        native = H.NativeRoutedScan(eng, rank, world, rows, cols, tiling.DEFAULT_HALO, wl.n_points, comm=None)
        lib = H.load()
        sensors = mode != "slices"
        pose_of = lambda k, r: wl.pose(11 * k + 37 * r)  # noqa: E731
        bounds = tiling.slice_bounds(wl.n_points, world, align=4)
        ws = None
        names = None
        for k in range(4):
            if sensors:
                s = wl.scan((k + rank) % 4)
                lo, hi = (0, 0) if (k == 0 and rank == 1) else (0, wl.n_points)  # rank 1's FIRST scan is empty
                Twb = pose_of(k, rank)
            else:
                s = wl.scan(k)
                lo, hi = bounds[rank]
                if k == 0 and rank == 1:
                    hi = lo  # (an empty slice on the first step)
                Twb = wl.pose(k)
            d = {c: torch.from_numpy(np.ascontiguousarray(s[c][lo:hi])).cuda() for c in ("x", "y", "z", "intensity")}
            if hi == lo:  # (data_ptr() of an empty tensor is null; the C call takes n = 0)
                d = {c: torch.zeros(4, device="cuda") for c in d}
                class _Empty:  # noqa: E306
                    def __init__(self, t): self.t = t
                    def numel(self): return 0
                    def data_ptr(self): return self.t.data_ptr()
                d = {c: _Empty(t) for c, t in d.items()}
            native.integrate(d["x"], d["y"], d["z"], wl.T_base_sensor, Twb, intensity=d["intensity"], sensors=sensors,
                             pipelined=mode == "pipelined")
            if mode == "pipelined" and k == 3:
                native.flush()
            if mode != "pipelined":  # (halo rings through the library's own exchange, on the same transport)
                names = [nm for nm in tiling.visible_layers(eng.layers())]
                have = [None] * world
                dist.all_gather_object(have, names)
                names = [nm for nm in names if all(nm in h for h in have)]
                arr = (C.c_char_p * len(names))(*[nm.encode() for nm in names])
                need = lib.fdm_halo_workspace_bytes(C.byref(native.plan), len(names))
                if ws is None or ws.numel() * 4 < need:
                    ws = torch.empty(max(int(need) // 4, 4), dtype=torch.float32, device="cuda")
                sent = lib.fdm_halo_exchange(eng._h, None, C.byref(native.plan), arr, len(names), C.c_void_p(ws.data_ptr()),
                                             ws.numel() * 4)
                assert sent >= 0, lib.fdm_halo_last_error()
            # the reference order on the untiled engine
            if sensors:
                for r in range(world):
                    sr = wl.scan((k + r) % 4)
                    if k == 0 and r == 1:
                        continue  # (its scan was empty: integrate() of an empty cloud touches nothing)
                    whole.integrate(sr["x"], sr["y"], sr["z"], wl.T_base_sensor, pose_of(k, r), intensity=sr["intensity"])
            else:
                keep = np.ones(wl.n_points, dtype=bool)
                if k == 0 and world > 1:
                    keep[bounds[1][0]:bounds[1][1]] = False
                whole.integrate(s["x"][keep], s["y"][keep], s["z"][keep], wl.T_base_sensor, wl.pose(k),
                                intensity=s["intensity"][keep])
        eng.sync()
        names = [nm for nm in tiling.visible_layers(eng.layers())]
        have = [None] * world
        dist.all_gather_object(have, names)
        names = [nm for nm in names if all(nm in h for h in have)]
        rect = (plan.stored if mode != "pipelined" else plan.owned) if world > 1 else tiling.Rect(0, 0, rows, cols)
        st = plan.stored if world > 1 else tiling.Rect(0, 0, rows, cols)
        bad = []
        for nm in names:
            got = eng.layer(nm)[rect.r0 - st.r0:rect.r1 - st.r0, rect.c0 - st.c0:rect.c1 - st.c0]
            want = whole.layer(nm)[rect.r0:rect.r1, rect.c0:rect.c1]
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad.append((nm, int((got.view(np.uint32) != want.view(np.uint32)).sum())))
        if world > 1 and not (transport.calls["all_gather"] >= 4 and transport.calls["send"] > 0 and transport.calls["recv"] > 0):
            bad.append(("transport", dict(transport.calls)))
        H.HostStagedTransport.uninstall()
        q.put((rank, bad, len(names)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, [("exception", traceback.format_exc(), str(e))], 0))


def _run(target, world, *extra):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    return sorted(results)


@pytest.mark.parametrize("world", [2, 4])
def test_routed_slices_equal_the_untiled_map(world):
    """2 / 4 processes on GPU 0, each holding a quarter / half of every scan: route -> exchange (host-staged over gloo)
    -> integrate on the owners -> halo exchange; every stored window equals the untiled engine integrating the WHOLE
    scan, bit for bit, and the routing counters add up to its statistics."""
    for rank, bad, n_names in _run(_routed_worker, world, "gloo"):
        assert not bad, f"rank {rank}: {bad}"
        assert n_names >= 8


def test_routed_scan_over_rccl_with_one_rank():
    """The same driver over backend "nccl" (RCCL) with a 1-rank communicator: all-gather of the counters, the self
    share of the exchange, integrate of the received records — the device-to-device path an N-GPU node runs."""
    for rank, bad, n_names in _run(_routed_worker, 1, "nccl"):
        assert not bad, f"rank {rank}: {bad}"
        assert n_names >= 8


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (4, "gloo"), (1, "nccl")])
def test_n_sensors_feeding_one_global_map(world, backend):
    """Every rank is a sensor with its own scan stream and pose; a step = N integrate() calls in rank order.  Owners
    integrate each source's records with that source's transforms.  Bit-identical to the untiled engine fed the N
    scans one after the other."""
    for rank, bad, n_names in _run(_sensors_worker, world, backend):
        assert not bad, f"rank {rank}: {bad}"
        assert n_names >= 8


@pytest.mark.parametrize("world,mode", [(2, "sensors"), (4, "sensors"), (2, "slices"), (4, "slices"), (2, "pipelined")])
def test_native_routed_step_with_real_peers(world, mode):
    """fdm_halo_routed_step had only ever run with a 1-rank communicator, where its send / receive loop is the
    self copy.  Two and four processes on GPU 0, libfdm_halo's transport table pointed at a host-staged one over gloo:
    per-peer offsets, grouping, source order and the halo exchange with real peers — every stored window bit-identical to
    the untiled engine; rank 1's first scan / slice is EMPTY (fdm_engine_route_scan used to write through a null table)."""
    for rank, bad, n_names in _run(_native_worker, world, mode):
        assert not bad, f"rank {rank}: {bad}"
        assert n_names >= 8
