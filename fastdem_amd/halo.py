"""ctypes plumbing for libfdm_halo.so (include/fdm_halo.h): the tile plan and the routed step of a tiled GLOBAL map as
ONE C call per scan (fdm_halo_routed_step) — the host loop of tiling.RoutedScan.integrate without an interpreter between
its launches.  The RCCL communicator is the library's own (ncclCommInitRank through ctypes on the librccl that torch
loaded); the unique id travels over whatever torch.distributed group the ranks already share."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfdm_halo.so")


class FdmRect(C.Structure):
    _fields_ = [("r0", C.c_int32), ("c0", C.c_int32), ("nr", C.c_int32), ("nc", C.c_int32)]


class FdmTilePlan(C.Structure):  # fdm_tile_plan
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("halo", C.c_int32), ("grid_rows", C.c_int32), ("grid_cols", C.c_int32),
                ("owned", FdmRect), ("stored", FdmRect), ("n_sends", C.c_int32), ("n_recvs", C.c_int32),
                ("send_rank", C.c_int32 * 8), ("send_rect", FdmRect * 8),
                ("recv_rank", C.c_int32 * 8), ("recv_rect", FdmRect * 8)]


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_lib = None
_nccl = None


def load():
    """dlopen libfdm_halo.so (after libfdm_engine.so, as a host application links both)."""
    global _lib
    if _lib is not None:
        return _lib
    capi.load()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C fastdem_amd/csrc`")
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    lib.fdm_halo_last_error.restype = C.c_char_p
    lib.fdm_tile_plan_make.argtypes = [C.c_int32] * 5 + [C.POINTER(FdmTilePlan)]
    lib.fdm_tile_plan_route.argtypes = [C.POINTER(FdmTilePlan), C.POINTER(capi.FdmRoutePlan)]
    lib.fdm_tile_plan_route.restype = None
    lib.fdm_halo_routed_ws_create.argtypes = [C.POINTER(FdmTilePlan), C.c_uint64, C.POINTER(P)]
    lib.fdm_halo_routed_ws_destroy.argtypes = [P]
    lib.fdm_halo_routed_ws_destroy.restype = None
    lib.fdm_halo_routed_step.argtypes = [P, P, C.POINTER(FdmTilePlan), C.POINTER(capi.FdmRoutePlan), P, C.c_uint64,
                                         P, P, P, P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, P]
    lib.fdm_halo_routed_submit.argtypes = lib.fdm_halo_routed_step.argtypes
    lib.fdm_halo_routed_flush.argtypes = [P, P, C.POINTER(FdmTilePlan), P, P]
    lib.fdm_halo_workspace_bytes.argtypes = [C.POINTER(FdmTilePlan), C.c_int32]
    lib.fdm_halo_workspace_bytes.restype = C.c_uint64
    lib.fdm_halo_exchange.argtypes = [P, P, C.POINTER(FdmTilePlan), C.POINTER(C.c_char_p), C.c_int32, P, C.c_uint64]
    lib.fdm_halo_exchange.restype = C.c_int64
    lib.fdm_halo_set_transport.argtypes = [C.POINTER(FdmHaloTransport)]
    lib.fdm_halo_set_transport.restype = None
    _lib = lib
    return lib


# ---- transport (include/fdm_halo.h): RCCL by default; tests install a host-staged one ----
_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
_GS = C.CFUNCTYPE(C.c_int, C.c_void_p)
_SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p)
_GE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class FdmHaloTransport(C.Structure):  # fdm_halo_transport
    _fields_ = [("all_gather", _AG), ("group_start", _GS), ("send", _SR), ("recv", _SR), ("group_end", _GE)]


class HostStagedTransport:
    """libfdm_halo's collectives over a torch.distributed group that only moves HOST memory (gloo): device buffers are
    staged through the host around every operation.  For the tests: it lets the library's multi-rank code (per-peer
    offsets, one group per exchange, source order) run with real peers on a box with ONE GPU, where RCCL refuses two ranks
    on one device.  Every operation synchronises the engine's stream — correct, slow, not for production."""

    def __init__(self, dist, group=None):
        import torch
        self.dist, self.group, self.torch = dist, group, torch
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.ops = None
        self.calls = {"all_gather": 0, "send": 0, "recv": 0, "groups": 0}
        self._cb = FdmHaloTransport(_AG(self._all_gather), _GS(self._group_start), _SR(self._send), _SR(self._recv),
                                    _GE(self._group_end))

    def install(self):
        load().fdm_halo_set_transport(C.byref(self._cb))

    @staticmethod
    def uninstall():
        load().fdm_halo_set_transport(None)

    def _to_host(self, d_ptr, nbytes, stream):
        self.hip.hipStreamSynchronize(stream)
        t = self.torch.empty(nbytes, dtype=self.torch.uint8)
        assert self.hip.hipMemcpy(t.data_ptr(), d_ptr, nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return t

    def _to_device(self, d_ptr, t):
        assert self.hip.hipMemcpy(d_ptr, t.data_ptr(), t.numel(), 1) == 0  # hipMemcpyHostToDevice

    def _all_gather(self, comm, d_send, d_recv, nbytes, stream):
        try:
            self.calls["all_gather"] += 1
            mine = self._to_host(d_send, nbytes, stream)
            W = self.dist.get_world_size(self.group)
            parts = [self.torch.empty(nbytes, dtype=self.torch.uint8) for _ in range(W)]
            self.dist.all_gather(parts, mine, group=self.group)
            self._to_device(d_recv, self.torch.cat(parts))
            return 0
        except Exception:  # (no exception may cross the C boundary)
            import traceback
            traceback.print_exc()
            return 1

    def _group_start(self, comm):
        self.ops = []
        self.calls["groups"] += 1
        return 0

    def _send(self, comm, d_buf, nbytes, peer, stream):
        try:
            self.calls["send"] += 1
            self.ops.append(("send", self._to_host(d_buf, nbytes, stream), peer, None))
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _recv(self, comm, d_buf, nbytes, peer, stream):
        self.calls["recv"] += 1
        self.ops.append(("recv", self.torch.empty(nbytes, dtype=self.torch.uint8), peer, d_buf))
        return 0

    def _group_end(self, comm, stream):
        try:
            p2p = [self.dist.P2POp(self.dist.isend if kind == "send" else self.dist.irecv, t, peer, self.group)
                   for kind, t, peer, _ in self.ops]
            if p2p:
                for req in self.dist.batch_isend_irecv(p2p):
                    req.wait()
            self.hip.hipStreamSynchronize(stream)
            for kind, t, _, d_buf in self.ops:
                if kind == "recv":
                    self._to_device(d_buf, t)
            self.ops = None
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1


def _rccl():
    global _nccl
    if _nccl is None:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _nccl = C.CDLL(cand if os.path.exists(cand) else "librccl.so")
        _nccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
        _nccl.ncclCommDestroy.argtypes = [C.c_void_p]
    return _nccl


def make_comm(rank, world, dist=None, group=None):
    """An ncclComm of `world` ranks (None for world == 1 when no collective is wanted).  Rank 0 draws the unique id;
    it reaches the others through torch.distributed's object broadcast (any backend)."""
    nccl = _rccl()
    uid = NcclUniqueId()
    if rank == 0:
        assert nccl.ncclGetUniqueId(C.byref(uid)) == 0
    if world > 1:
        box = [bytes(uid.internal)] if rank == 0 else [None]
        dist.broadcast_object_list(box, src=0, group=group)
        C.memmove(C.byref(uid), box[0].ljust(128, b"\0"), 128)
    comm = C.c_void_p()
    rc = nccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError(f"ncclCommInitRank failed: {rc}")
    return comm


def destroy_comm(comm):
    if comm:
        _rccl().ncclCommDestroy(comm)


def _col16(T):
    a = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).reshape(16)
    return (C.c_double * 16)(*a.tolist())


class NativeRoutedScan:
    """tiling.RoutedScan.integrate as one C call per step (fdm_halo_routed_step).  `comm`: an ncclComm of the plan's
    world (make_comm); may be None when world == 1."""

    def __init__(self, engine, rank, world, rows, cols, halo, max_points, comm=None):
        self.lib = load()
        self.eng, self.comm = engine, comm
        self.plan = FdmTilePlan()
        rc = self.lib.fdm_tile_plan_make(rank, world, rows, cols, halo, C.byref(self.plan))
        if rc != 0:
            raise RuntimeError(self.lib.fdm_halo_last_error().decode())
        self.route = capi.FdmRoutePlan()
        self.lib.fdm_tile_plan_route(C.byref(self.plan), C.byref(self.route))
        self.ws = C.c_void_p()
        rc = self.lib.fdm_halo_routed_ws_create(C.byref(self.plan), int(max_points), C.byref(self.ws))
        if rc != 0:
            raise RuntimeError(self.lib.fdm_halo_last_error().decode())
        self.matrix = np.zeros((world, world + 2), dtype=np.uint32)

    def close(self):
        if self.ws:
            self.lib.fdm_halo_routed_ws_destroy(self.ws)
            self.ws = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate(self, x, y, z, T_base_sensor, T_world_base, intensity=None, sensors=False, pipelined=False,
                  want_matrix=True):
        """x, y, z[, intensity]: this rank's part of the step (torch device tensors).  Returns the counter matrix of the
        scan the call integrated — with `pipelined` that is the PREVIOUS scan (fdm_halo_routed_submit: this scan is
        routed now and integrated by the next call or by flush()); the tensors may be reused once the routing kernels
        have run (the engine's stream).  `want_matrix` False: no matrix is asked for (with ONE rank nothing is routed and
        the matrix would be a statistics read-back that waits for the scan)."""
        self.eng.wait_torch()
        n = int(x.numel())
        fn = self.lib.fdm_halo_routed_submit if pipelined else self.lib.fdm_halo_routed_step
        rc = fn(
            self.eng._h, self.comm, C.byref(self.plan), C.byref(self.route), self.ws, n,
            C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(z.data_ptr()),
            C.c_void_p(intensity.data_ptr()) if intensity is not None else None,
            _col16(T_base_sensor), _col16(T_world_base), 1 if sensors else 0,
            self.matrix.ctypes.data_as(C.c_void_p) if want_matrix else None)
        if rc != 0:
            raise RuntimeError(f"fdm_halo_routed_step: {rc} {self.lib.fdm_halo_last_error().decode()}")
        return self.matrix.astype(np.int64) if want_matrix else None

    def flush(self):
        """Integrate the scan a pipelined call left pending (no-op otherwise)."""
        rc = self.lib.fdm_halo_routed_flush(self.eng._h, self.comm, C.byref(self.plan), self.ws,
                                            self.matrix.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"fdm_halo_routed_flush: {rc} {self.lib.fdm_halo_last_error().decode()}")
        return self.matrix.astype(np.int64)
