"""Spatial tiling of ONE fixed-origin (GLOBAL) elevation map across the ranks of a node
 — one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Partitioning
  The logical grid is cut into pr x pc blocks (1x2, 2x2, 2x4 for 2/4/8 ranks).  Rank k OWNS one
  block and STORES it plus a read-only halo ring of `halo` cells (clamped at the map border).
  Cell indices are always computed against the global geometry inside the engine (fdm_tile), so
  every owned cell is bit-identical to the single-GPU map.

Per scan
  1. scan distribution: broadcast of the SoA channels from the ingest rank (skipped when every
     rank already holds the scan);
  2. every rank integrates the whole scan; its bin kernel keeps only points that land in its
     owned window — the per-cell update has no neighbour coupling, so no interior state is ever
     exchanged;
  3. halo exchange: each rank sends the owned cells its neighbours keep in their halo ring
     (<= 8 neighbours, point-to-point isend/irecv, one packed buffer per neighbour).  The ring
     width is set by the downstream stencils of the reference (inpainting 1 cell,
     uncertainty fusion ceil(0.15/res), feature extraction ceil(0.3/res) — 6 cells at 0.05 m:
     config/postprocess.hpp:35,45).  No all-reduce anywhere: xGMI is point-to-point, and a halo
     strip is 10^2 KB, latency- not bandwidth-bound.

The exchange logic is backend-agnostic: `EngineTile` packs/unpacks with HIP kernels straight
from the engine's layers; tests drive the same plan over gloo with a numpy-backed tile.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

DEFAULT_HALO = 6


def grid_for(world: int) -> Tuple[int, int]:
    """pr x pc with pr*pc == world, as square as possible, pr <= pc."""
    pr = int(np.floor(np.sqrt(world)))
    while world % pr:
        pr -= 1
    return pr, world // pr


@dataclass(frozen=True)
class Rect:
    r0: int
    c0: int
    nr: int
    nc: int

    @property
    def r1(self):
        return self.r0 + self.nr

    @property
    def c1(self):
        return self.c0 + self.nc

    @property
    def empty(self):
        return self.nr <= 0 or self.nc <= 0

    def intersect(self, o: "Rect") -> "Rect":
        r0, c0 = max(self.r0, o.r0), max(self.c0, o.c0)
        r1, c1 = min(self.r1, o.r1), min(self.c1, o.c1)
        return Rect(r0, c0, max(0, r1 - r0), max(0, c1 - c0))

    def local_to(self, origin: "Rect") -> "Rect":
        return Rect(self.r0 - origin.r0, self.c0 - origin.c0, self.nr, self.nc)


@dataclass(frozen=True)
class TilePlan:
    rank: int
    world: int
    rows: int
    cols: int
    halo: int
    owned: Rect
    stored: Rect
    # neighbour rank -> (rect I send, rect I receive), both in GLOBAL cell coordinates
    sends: Dict[int, Rect]
    recvs: Dict[int, Rect]

    def fdm_tile(self) -> Tuple[int, ...]:
        """Argument tuple for fdm_tile (include/fdm_engine.h)."""
        s, o = self.stored, self.owned
        return (s.r0, s.c0, s.nr, s.nc, o.r0, o.c0, o.nr, o.nc)


def _split(n: int, parts: int) -> List[Tuple[int, int]]:
    edges = [round(i * n / parts) for i in range(parts + 1)]
    return [(edges[i], edges[i + 1] - edges[i]) for i in range(parts)]


def owned_rect(rank: int, world: int, rows: int, cols: int) -> Rect:
    pr, pc = grid_for(world)
    rs, cs = _split(rows, pr), _split(cols, pc)
    i, j = divmod(rank, pc)
    return Rect(rs[i][0], cs[j][0], rs[i][1], cs[j][1])


def stored_rect(owned: Rect, rows: int, cols: int, halo: int) -> Rect:
    r0, c0 = max(0, owned.r0 - halo), max(0, owned.c0 - halo)
    r1, c1 = min(rows, owned.r1 + halo), min(cols, owned.c1 + halo)
    return Rect(r0, c0, r1 - r0, c1 - c0)


def make_plan(rank: int, world: int, rows: int, cols: int, halo: int = DEFAULT_HALO) -> TilePlan:
    mine = owned_rect(rank, world, rows, cols)
    mine_st = stored_rect(mine, rows, cols, halo)
    sends, recvs = {}, {}
    for other in range(world):
        if other == rank:
            continue
        theirs = owned_rect(other, world, rows, cols)
        theirs_st = stored_rect(theirs, rows, cols, halo)
        s = mine.intersect(theirs_st)       # my owned cells inside their halo ring
        r = theirs.intersect(mine_st)       # their owned cells inside my halo ring
        if not s.empty:
            sends[other] = s
        if not r.empty:
            recvs[other] = r
    return TilePlan(rank, world, rows, cols, halo, mine, mine_st, sends, recvs)


def visible_layers(names: Sequence[str]) -> List[str]:
    """Layers a consumer sees: internal ones start with '_' (elevation_map.hpp:42-45)."""
    return [n for n in names if not n.startswith("_")]


class EngineTile:
    """Tile backend over a live engine: packs / unpacks with HIP kernels into torch buffers that live as
    long as the tile (one pair per neighbour, reused by every exchange: nothing is allocated per scan and
    no buffer can be recycled by torch's allocator while a kernel on the ENGINE's stream still uses it).
    Ordering between the engine's stream and torch's (the collective's) is by events, never by the host."""

    def __init__(self, engine, plan: TilePlan, device):
        import torch
        self.eng, self.plan, self.device, self.torch = engine, plan, device, torch
        self._bufs = {}

    def _buf(self, kind, rect: Rect, names: Sequence[str]):
        key = (kind, rect, len(names))
        if key not in self._bufs:
            self._bufs[key] = self.torch.empty(len(names) * rect.nr * rect.nc, dtype=self.torch.float32,
                                               device=self.device)
        return self._bufs[key]

    def pack(self, rect: Rect, names: Sequence[str]):
        loc = rect.local_to(self.plan.stored)
        buf = self._buf("send", rect, names)
        self.eng.region_pack(loc.r0, loc.c0, loc.nr, loc.nc, list(names), buf.data_ptr())
        return buf

    def recv_buffer(self, rect: Rect, names: Sequence[str]):
        return self._buf("recv", rect, names)

    def unpack(self, rect: Rect, names: Sequence[str], buf):
        loc = rect.local_to(self.plan.stored)
        self.eng.region_unpack(loc.r0, loc.c0, loc.nr, loc.nc, list(names), buf.data_ptr())

    def fence(self):
        """The pack kernels run on the engine's stream, the collective on torch's: torch's stream waits for
        the engine (an event; the host does not).  Also orders this exchange's receives behind the previous
        exchange's unpack kernels, which read the same buffers."""
        self.eng.torch_wait()

    def after_comm(self):
        """`req.wait()` has ordered torch's current stream behind the transfers; the unpack kernels go to the
        engine's stream, which waits for torch's stream here (again an event)."""
        self.eng.wait_torch()


class HostStagedTile:
    """The same tile over a backend that only moves HOST memory (gloo): buffers are staged through the host
    around the real HIP pack / unpack kernels.  Used by the one-GPU multi-process test (RCCL refuses two ranks
    on one device) — the plan, the pack / unpack code and the exchange loop are the ones the RCCL path runs."""

    def __init__(self, inner: "EngineTile"):
        self.inner = inner
        self._recv = {}

    def pack(self, rect, names):
        buf = self.inner.pack(rect, names)
        self.inner.eng.sync()
        return buf.cpu()

    def recv_buffer(self, rect, names):
        t = self.inner.torch.empty(len(names) * rect.nr * rect.nc, dtype=self.inner.torch.float32)
        return t

    def unpack(self, rect, names, buf):
        dev = self.inner.recv_buffer(rect, names)
        dev.copy_(buf)
        self.inner.torch.cuda.current_stream().synchronize()
        self.inner.unpack(rect, names, dev)

    def fence(self):
        pass


def exchange_halos(tile, plan: TilePlan, names: Sequence[str], dist, group=None):
    """One halo exchange: post every irecv/isend of the plan, wait, unpack."""
    if plan.world == 1 or not names:
        return 0
    ops, recv_bufs, keep = [], {}, []
    for other in sorted(plan.recvs):
        buf = tile.recv_buffer(plan.recvs[other], names)
        recv_bufs[other] = buf
        ops.append(dist.P2POp(dist.irecv, buf, other, group))
    for other in sorted(plan.sends):
        buf = tile.pack(plan.sends[other], names)
        keep.append(buf)
        ops.append(dist.P2POp(dist.isend, buf, other, group))
    tile.fence()
    nbytes = sum(int(b.numel()) * 4 for b in keep)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if hasattr(tile, "after_comm"):
        tile.after_comm()
    for other, buf in recv_bufs.items():
        tile.unpack(plan.recvs[other], names, buf)
    return nbytes


# --------------------------------------------------------------------------- scan routing ----
def route_plan(plan: TilePlan):
    """capi.FdmRoutePlan of the tiling: the row / column edges of the owned rects (fdm_route.hpp)."""
    from . import capi
    pr, pc = grid_for(plan.world)
    rp = capi.FdmRoutePlan()
    rp.world, rp.grid_rows, rp.grid_cols = plan.world, pr, pc
    rs, cs = _split(plan.rows, pr), _split(plan.cols, pc)
    for i in range(pr):
        rp.row_edge[i] = rs[i][0]
    rp.row_edge[pr] = plan.rows
    for j in range(pc):
        rp.col_edge[j] = cs[j][0]
    rp.col_edge[pc] = plan.cols
    return rp


def slice_bounds(n: int, world: int, align: int = 1) -> List[Tuple[int, int]]:
    """Contiguous slices of an n-point scan, one per rank, in rank order (starts aligned to `align` points)."""
    edges = [min(n, ((round(i * n / world) + align - 1) // align) * align) for i in range(world)] + [n]
    return [(edges[i], edges[i + 1]) for i in range(world)]


class RoutedScan:
    """One logical scan whose points are spread over the ranks (rank r holds slice r; the scan is the
    concatenation in rank order): every rank routes its slice to the owners of the cells, the owners integrate
    what they receive — in rank order, i.e. scan order, so every owned cell equals the single map's bit for bit.

    Per scan and rank: fdm_engine_route_scan (3 small kernels) -> all-gather of the world + 2 counters (one host
    read-back: the sizes of the exchange are host-side arguments) -> point-to-point exchange of 16-byte point
    records (the rank's own share is a device copy) -> fdm_engine_integrate_points4_device.  `staged`: move the
    records through host memory (gloo, several ranks on one GPU) instead of the device-to-device path (nccl)."""

    def __init__(self, engine, plan: TilePlan, device, max_points: int, staged: bool = False):
        import torch
        self.eng, self.plan, self.torch, self.staged = engine, plan, torch, staged
        self.rp = route_plan(plan)
        self.device = device
        # (+ 3 rows per owner: N-sensor mode pads every owner's share to a multiple of four points)
        self.send = torch.empty((max(max_points, 1) + 3 * plan.world + 4, 4), dtype=torch.float32, device=device)
        # this rank's row of the per-step table: world + 2 counters (written by the routing kernels) | the rank's two
        # transforms as raw float64 bits (64 words, copied from pinned memory without a host sync)
        W = plan.world
        self.row = torch.zeros(W + 2 + 64, dtype=torch.int32, device=device)
        self.counts = self.row[:W + 2]
        self.pose_pinned = torch.zeros(32, dtype=torch.float64).pin_memory()
        self.table_dev = torch.zeros((W, W + 2 + 64), dtype=torch.int32, device=device)
        self.table_pinned = torch.zeros((W, W + 2 + 64), dtype=torch.int32).pin_memory()
        self.recv = None
        self.matrix = None
        self._obstacle_dirty = True  # (sensors mode) the tile's obstacle layer may hold non-NaN cells

    def _recv_buffer(self, n):
        if self.recv is None or self.recv.shape[0] < n:
            self.recv = self.torch.empty((max(n + n // 4, 1024), 4), dtype=self.torch.float32, device=self.device)
        return self.recv

    def integrate(self, x, y, z, T_base_sensor, T_world_base, dist, intensity=None, group=None, sensors=False):
        """x, y, z[, intensity]: this rank's part of the step (device tensors).
        sensors=False — ONE logical scan cut into slices: every rank passes the same transforms, the owners integrate
          everything they received as one scan (= FastDEM::integrate of the concatenated cloud).
        sensors=True  — N scans, one per rank (N sensors / robots feeding one global map), each with its OWN
          transforms: the step is N FastDEM::integrate calls in rank order; an owner integrates the records of each
          source with that source's transforms, in that order.
        Returns the counter matrix (numpy, world x (world + 2)): [src, dst] points sent, [src, world]
        n_after_filter, [src, world + 1] n_in_map."""
        torch, W, me = self.torch, self.plan.world, self.plan.rank
        n = int(x.numel())
        assert n + 3 * W <= self.send.shape[0]
        # N scans, one per rank: a share is integrated as a scan of its own, so it travels as four channel blocks the bin
        # kernels read in place (no de-interleave pass, the rank's own share is never copied).  ONE scan cut into slices:
        # 16-byte records, contiguous across sources (the owner integrates them as one scan).
        soa = bool(sensors)
        self.eng.route_scan(self.rp, x, y, z, T_base_sensor, T_world_base, self.send, self.counts, intensity=intensity,
                            soa=soa)
        self.eng.torch_wait()  # torch's stream (the collective's) behind the routing kernels
        # one small all-gather: the world + 2 counters and the two transforms of every rank; ONE host read-back
        self.pose_pinned[:16] = torch.from_numpy(np.asarray(T_base_sensor, dtype=np.float64).reshape(16))
        self.pose_pinned[16:] = torch.from_numpy(np.asarray(T_world_base, dtype=np.float64).reshape(16))
        self.row[W + 2:].copy_(self.pose_pinned.view(torch.int32), non_blocking=True)
        if W > 1 and self.staged:
            gathered = [torch.empty(W + 66, dtype=torch.int32) for _ in range(W)]
            dist.all_gather(gathered, self.row.cpu(), group=group)
            raw = torch.stack(gathered).numpy()
        else:
            if W > 1:
                dist.all_gather_into_tensor(self.table_dev, self.row, group=group)
                self.table_pinned.copy_(self.table_dev, non_blocking=True)
            else:
                self.table_pinned[0].copy_(self.row, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            raw = self.table_pinned.numpy()
        table = np.concatenate([raw[:, :W + 2].astype(np.float64),
                                np.ascontiguousarray(raw[:, W + 2:]).view(np.float64)], axis=1)
        matrix = table[:, :W + 2].astype(np.int64)
        self.matrix = matrix
        pad = (lambda v: (v + 3) // 4 * 4) if soa else (lambda v: v)  # rows a share of v points takes
        n_recv = int(pad(matrix[:, me]).sum()) - (int(pad(matrix[me, me])) if soa else 0)
        any_in_map = bool(matrix[:, W + 1].sum() > 0)
        recv = self._recv_buffer(n_recv)
        send_off = np.concatenate([[0], np.cumsum(pad(matrix[me, :W]))])
        rsizes = pad(matrix[:, me]).copy()
        if soa:
            rsizes[me] = 0  # (the own share stays where the routing kernels put it)
        recv_off = np.concatenate([[0], np.cumsum(rsizes)])
        ops, keep = [], []
        for peer in range(W):
            ns, nr = int(pad(matrix[me, peer])), int(pad(matrix[peer, me]))
            if peer == me:
                if ns and not soa:
                    recv[recv_off[peer]:recv_off[peer] + ns].copy_(self.send[send_off[peer]:send_off[peer] + ns])
                continue
            if nr:
                rb = torch.empty((nr, 4), dtype=torch.float32) if self.staged else recv[recv_off[peer]:recv_off[peer] + nr]
                keep.append((peer, rb, nr))
                ops.append(dist.P2POp(dist.irecv, rb, peer, group))
            if ns:
                sb = self.send[send_off[peer]:send_off[peer] + ns]
                sb = sb.cpu() if self.staged else sb
                keep.append((None, sb, ns))
                ops.append(dist.P2POp(dist.isend, sb, peer, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.staged:
            for peer, rb, nr in keep:
                if peer is not None:
                    recv[recv_off[peer]:recv_off[peer] + nr].copy_(rb)
            torch.cuda.current_stream().synchronize()
        if not sensors:
            self.eng.integrate_points4_device(recv, n_recv, T_base_sensor, T_world_base,
                                              has_intensity=intensity is not None, any_in_map=any_in_map)
            return matrix
        for src in range(W):  # N scans, in rank order
            ns = int(matrix[src, me])
            seen = bool(matrix[src, W + 1] > 0)  # that scan observed a cell somewhere: update() ran (obstacle clear)
            if ns == 0 and not (seen and self._obstacle_dirty):
                continue  # nothing for this tile, and its obstacle layer is clear already: the clear would be a no-op
            Tbs = table[src, W + 2:W + 18].reshape(4, 4)
            Twb = table[src, W + 18:W + 34].reshape(4, 4)
            share = (self.send[send_off[me]:] if src == me else recv[recv_off[src]:]).reshape(-1)
            self.eng.integrate_soa4_device(share, ns, Tbs, Twb, has_intensity=intensity is not None, any_in_map=seen)
            self._obstacle_dirty = ns > 0
        return matrix


# --------------------------------------------------------------------------- bench (C5) ----
class Watchdog:
    """A rank that stalls (a peer died inside a collective, a hung kernel) must not hang the node's benchmark: if
    `kick()` is not called for `seconds`, the process prints what it was doing and leaves with exit code 3 — plain
    os._exit from a helper thread; nothing is exec'ed, the GPU context dies with the process."""

    def __init__(self, seconds, what="bench", on_stall=None):
        import threading
        import time
        self._t, self._time, self.what, self.seconds, self.stage = time.monotonic(), time, what, seconds, "start"
        self._stop = False
        self.on_stall = on_stall  # (called with the message instead of leaving with exit code 3)
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def kick(self, stage=None):
        self._t = self._time.monotonic()
        if stage is not None:
            self.stage = stage

    def stop(self):
        self._stop = True

    def _run(self):
        import os
        import sys
        while not self._stop:
            self._time.sleep(1.0)
            if self._time.monotonic() - self._t > self.seconds:
                msg = f"[{self.what}] no progress for {self.seconds} s in stage '{self.stage}': giving up"
                sys.stderr.write(msg + "\n")
                sys.stderr.flush()
                if self.on_stall is not None:
                    self.on_stall(msg)
                os._exit(3)


def bench_global(args, rank: int, local_rank: int, world: int) -> Optional[dict]:
    """configs[4]: one 400x400 m @ 0.05 m GLOBAL map tiled over `world` GPUs.  N-sensor mode (weak scaling): every
    rank is a robot with its own 2 M-point scan stream, on the same 150 m circle a world-th of a turn apart; a step =
    `world` FastDEM::integrate calls (rank order) into the one global map = route the own scan by owner -> gather the
    counters -> exchange the point records -> integrate what arrived, source by source -> halo exchange.  Runs over
    RCCL (backend nccl) also with ONE rank: the collectives are then 1-rank collectives, the code path is the N-rank one."""
    import time

    import torch
    import torch.distributed as dist

    from . import Engine, capi, synth

    # `global_map_fatal` False (the leg rides behind the replicas of the default workload): a failure or a stall of THIS
    # leg must not take the replicas' line with it — rank 0 prints the line it has, with the error in `global_map`, and
    # every rank leaves with code 0 (peers blocked in a collective leave through their own watchdogs the same way).
    fatal = bool(getattr(args, "global_map_fatal", True))

    def give_up(msg, code):
        import json
        import os
        import sys
        if fatal:
            os._exit(code)
        if rank == 0 and getattr(args, "partial_result", None) is not None:
            line = dict(args.partial_result)
            line["global_map"] = {"error": msg}
            line["global_map_ok"] = False  # (top level: a driver sees the dead leg without looking inside the object)
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        os._exit(0)

    dog = Watchdog(float(getattr(args, "stall_timeout", 180.0)), what=f"bench c5 rank {rank}",
                   on_stall=lambda msg: give_up(msg, 3))
    try:
        if int(getattr(args, "fail_global_rank", -1)) == rank:  # (tests: this leg failing on one rank — an ARGUMENT of the
            raise RuntimeError("fail_global_rank")               # bench command, the product reads no test switch from the environment)
        size_m = float(getattr(args, "global_size_m", 400.0))
        # (a reduced map for the N-process tests on one GPU: the circle and the scan shrink with it)
        wl = synth.global_map(n_scans=2) if size_m >= 400.0 else \
            synth.global_map(n_scans=2, size_m=size_m, n_az=int(getattr(args, "global_n_az", 2048)), radius=max(4.0, size_m / 2.0 - 30.0))
        dev = f"cuda:{local_rank}"
        staged = dist.get_backend() == "gloo"  # host-staged exchange (several ranks on one device: RCCL refuses that)
        rows = cols = int(round(float(np.float32(wl.width)) / float(np.float32(wl.resolution))))
        plan = make_plan(rank, world, rows, cols, DEFAULT_HALO)
        dog.kick("engine")
        eng = Engine(wl.width, wl.height, wl.resolution, wl.apply_to(capi.default_config()),
                     tile=plan.fdm_tile() if world > 1 else None, device=local_rank)
        tile = EngineTile(eng, plan, dev)
        if staged:
            tile = HostStagedTile(tile)
        names = ["elevation", "variance", "elevation_min", "elevation_max", "upper_bound",
                 "lower_bound", "n_points", "obstacle", "intensity"]
        n_pts = wl.n_points
        router = RoutedScan(eng, plan, dev, max_points=n_pts, staged=staged)
        native = None
        pipelined = int(getattr(args, "native_routed", 1)) == 2 and world == 1  # (with halo rings to refresh per step: not pipelined)
        if int(getattr(args, "native_routed", 1)) and not staged:  # the routed step as ONE C call (libfdm_halo: fdm_halo_routed_step)
            from . import halo as halo_c
            comm = halo_c.make_comm(rank, world, dist)
            native = halo_c.NativeRoutedScan(eng, rank, world, rows, cols, DEFAULT_HALO, n_pts, comm=comm)
        mine = [{c: torch.from_numpy(s[c]).to(dev) for c in ("x", "y", "z", "intensity")} for s in wl.scans]
        steps, warm = min(args.steps, 200), min(args.warmup, 400)
        # robot `rank` of `world`: the workload's 150 m circle (0.4 m per pose: 2356 poses per turn), a world-th of a
        # turn ahead per rank
        turn = int(round(2.0 * np.pi * (150.0 if size_m >= 400.0 else max(4.0, size_m / 2.0 - 30.0)) / 0.4))
        pose = lambda k: wl.pose(k + (turn * rank) // world)  # noqa: E731

        trace = [] if getattr(args, "trace_steps", False) else None  # (measurement, `bench.py --trace-steps`: host time of every step's parts)

        def step(k):
            d = mine[k % len(mine)]
            t_a = time.perf_counter()
            if native is not None:
                native.integrate(d["x"], d["y"], d["z"], wl.T_base_sensor, pose(k), intensity=d["intensity"],
                                 sensors=True, pipelined=pipelined, want_matrix=False)
                if trace is not None:
                    trace.append(time.perf_counter() - t_a)
            else:
                router.integrate(d["x"], d["y"], d["z"], wl.T_base_sensor, pose(k), dist, intensity=d["intensity"],
                                 sensors=True)
            exchange_halos(tile, plan, names, dist)
            dog.kick(f"step {k}")
            return n_pts

        def barrier():
            eng.sync()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()  # (the barrier's own device work is done before the timed region starts)

        k = 0
        for _ in range(warm):
            step(k)
            k += 1
        barrier()
        t0 = time.perf_counter()
        pts = 0
        for _ in range(steps):
            pts += step(k)
            k += 1
        if native is not None:
            native.flush()
        eng.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if trace:
            import sys
            tr = np.asarray(trace[-steps:]) * 1e6
            sys.stderr.write(f"[trace] native.integrate host us per step: median {np.median(tr):.1f} p90 {np.percentile(tr, 90):.1f} "
                             f"max {tr.max():.1f} sum_ms {tr.sum() / 1e3:.2f} of {dt * 1e3:.2f}; first 12: {np.round(tr[:12], 1).tolist()}\n")
        if native is not None:  # (the routing matrix of one more step, outside the timed region)
            d = mine[k % len(mine)]
            router.matrix = native.integrate(d["x"], d["y"], d["z"], wl.T_base_sensor, pose(k), intensity=d["intensity"],
                                             sensors=True)
            exchange_halos(tile, plan, names, dist)
            eng.sync()
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if staged else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rc, st = eng.last_stats()
        routed = router.matrix
        # SURVEY.md §8d algorithmic bytes of a step, summed over the ranks: every source's points are read once where
        # they are routed (16 B: x, y, z, intensity) and once where they are integrated; every touched cell is one
        # record read-modify-write (72 B Kalman + 8 B intensity).  GLOBAL map: no map-sized term.  Touched cells: this
        # rank's last integrate, taken for every (source, owner) pair that exchanged points in the last step.
        touched = torch.tensor([float(st["n_cells_touched"])], dtype=torch.float64, device="cpu" if staged else dev)
        dist.all_reduce(touched, op=dist.ReduceOp.SUM)
        touched = float(touched.item())
    except BaseException as exc:  # a failing rank leaves at once with a non-zero code: its peers' watchdogs / the
        import os                 # collective timeout then end them, the launcher sees the failure
        import sys
        import traceback
        traceback.print_exc()
        sys.stderr.write(f"[bench c5 rank {rank}] failed: {exc!r}\n")
        sys.stderr.flush()
        give_up(f"rank {rank}: {exc!r}", 2)
    dog.stop()
    if rank != 0:
        return None
    pr, pc = grid_for(world)
    return {
        "metric": "M points/s integrated into ElevationMap",
        "value": pts * world / dt / 1e6, "unit": "Mpts/s", "n_gpus": world, "steps": steps, "warmup": warm,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl.name, "points_per_scan": wl.n_points, "scans_per_step": world, "map_cells": rows * cols,
                   "tile_plan": f"{pr}x{pc}", "halo_cells": DEFAULT_HALO,
                   "parallelism": f"spatial tiles {pr}x{pc}, halo {DEFAULT_HALO} cells; N-sensor mode: every rank routes its own "
                                  "2 M-point scan to the owners of the cells (shares as four channel blocks the owner's bin "
                                  "kernel reads in place, point-to-point), the owners integrate source by source, p2p halo "
                                  "exchange per step (one pack + one unpack launch); ONE rank: nothing is routed, the step is "
                                  "one integrate()",
                   "inputs": "SoA float32 (x, y, z, intensity) resident in HBM on every rank",
                   "routed_step": ("one C call per step, pipelined over consecutive scans (fdm_halo_routed_submit)" if pipelined else
                                   "one C call per step (fdm_halo_routed_step)") if native is not None else "Python loop (tiling.RoutedScan)"},
        "roofline": _global_roofline(world, n_pts, routed, touched, dt / steps),
        "rank0_last_scan": st,
        "rank0_routing_matrix_last_step": routed.tolist() if routed is not None else None,
    }


def _global_roofline(world, n_pts, matrix, touched_last, step_s):
    """The routed step against the HBM roofline of the `world` GPUs it ran on (8 TB/s each): bytes the algorithm has
    to move per step / step time.  A step is launch- and synchronisation-bound (one host read-back per step), so
    the fraction is small by construction — it is there so that every bench line carries the same object."""
    if world == 1:  # nothing is routed: the step is one integrate() of the scan
        alg = n_pts * 16.0 + touched_last * 80.0
    else:           # slice read where it is routed + share written + share read where it is integrated + the cells
        routed_pts = float(np.asarray(matrix)[:, :world].sum()) if matrix is not None else float(n_pts) * world
        alg = world * n_pts * 16.0 + 2.0 * routed_pts * 16.0 + touched_last * 80.0
    gbps = alg / step_s / 1e9
    peak = 8000.0 * world
    return {"bound": "hbm", "limited_by": "issue", "kernel": "k_tupdate_tbin (one rank: the step is one integrate())" if world == 1 else
            "routed step (k_route_* + exchange + k_tupdate_tbin per source)", "achieved": gbps,
            "peak": peak, "unit": "GB/s", "frac": gbps / peak, "traffic": None, "alg_bytes_per_step": alg,
            "avg_step_us": step_s * 1e6,
            "note": "algorithmic bytes of all ranks per step / step time; peak = n_gpus x 8 TB/s"}
