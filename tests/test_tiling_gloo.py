"""The N>1 path on CPU: world_size-2 and -4 gloo process groups drive the SAME tiling plan and
halo-exchange code the GPU ranks use (fastdem_amd/tiling.py), with a numpy-backed tile whose
contents come from the CPU oracle.  After one exchange every rank's stored window (owned cells +
halo ring) must equal the single-map result cell for cell."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastdem_amd import synth, tiling


def test_grid_and_plan_cover_the_map_exactly():
    for world in (1, 2, 4, 8):
        rows, cols = 1000, 730
        cover = np.zeros((rows, cols), dtype=np.int32)
        for r in range(world):
            p = tiling.make_plan(r, world, rows, cols, halo=6)
            o = p.owned
            cover[o.r0:o.r1, o.c0:o.c1] += 1
            assert p.stored.intersect(o) == o
            # what I send to a neighbour is exactly what it expects to receive from me
            for other, rect in p.sends.items():
                assert tiling.make_plan(other, world, rows, cols, 6).recvs[r] == rect
            assert len(p.sends) <= 8
        assert (cover == 1).all()
    assert tiling.grid_for(8) == (2, 4) and tiling.grid_for(4) == (2, 2) and tiling.grid_for(2) == (1, 2)


class NumpyTile:
    """CPU stand-in for tiling.EngineTile: layers as (rows, cols) arrays of the stored window."""

    def __init__(self, plan, layers):
        self.plan, self.layers = plan, layers

    def pack(self, rect, names):
        loc = rect.local_to(self.plan.stored)
        parts = [np.asfortranarray(self.layers[n][loc.r0:loc.r0 + loc.nr, loc.c0:loc.c0 + loc.nc]).ravel(order="F")
                 for n in names]
        return torch.from_numpy(np.concatenate(parts).astype(np.float32))

    def recv_buffer(self, rect, names):
        return torch.empty(len(names) * rect.nr * rect.nc, dtype=torch.float32)

    def unpack(self, rect, names, buf):
        loc = rect.local_to(self.plan.stored)
        a = buf.numpy().reshape(len(names), -1)
        for k, n in enumerate(names):
            self.layers[n][loc.r0:loc.r0 + loc.nr, loc.c0:loc.c0 + loc.nc] = \
                a[k].reshape((loc.nr, loc.nc), order="F")

    def fence(self):
        pass


def _truth():
    """Single-map result from the oracle (every rank recomputes it deterministically)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import fdm_ref_py as R
    wl = synth.global_map(n_scans=2, size_m=40.0, n_az=512, radius=8.0)
    ref = R.RefEngine(wl.width, wl.height, wl.resolution, wl.apply_to(R.default_config()))
    for k in range(2):
        s = wl.scan(k)
        ref.integrate(s["x"], s["y"], s["z"], wl.T_base_sensor, wl.pose(k), intensity=s["intensity"])
    names = tiling.visible_layers(ref.layers())
    return ref.rows, ref.cols, {n: ref.layer(n) for n in names}, names


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, cols, full, names = _truth()
        plan = tiling.make_plan(rank, world, rows, cols, halo=6)
        st, ow = plan.stored, plan.owned
        layers = {}
        for n in names:
            a = np.full((st.nr, st.nc), -777.0, dtype=np.float32)  # poisoned halo ring
            lo = ow.local_to(st)
            a[lo.r0:lo.r0 + lo.nr, lo.c0:lo.c0 + lo.nc] = full[n][ow.r0:ow.r1, ow.c0:ow.c1]
            layers[n] = a
        tile = NumpyTile(plan, layers)
        sent = tiling.exchange_halos(tile, plan, names, dist)
        ok = True
        for n in names:
            exp = full[n][st.r0:st.r1, st.c0:st.c1]
            got = layers[n]
            same = np.array_equal(np.isnan(exp), np.isnan(got)) and \
                np.array_equal(exp[~np.isnan(exp)], got[~np.isnan(got)])
            ok = ok and same
        q.put((rank, ok, sent, len(plan.sends)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_halo_exchange_reassembles_the_single_map(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=10) for _ in range(world))
    for rank, ok, sent, n_nb in results:
        assert ok, f"rank {rank}: stored window differs from the single map after the exchange"
        assert sent > 0 and n_nb >= 1
