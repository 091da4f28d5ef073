"""Thin Python handle over the C ABI (include/fdm_engine.h) — used by tests and bench.py.

Mirrors the reference call shapes:
  ElevationMap(width, height, resolution, frame)  -> Engine(width, height, resolution, cfg)
  FastDEM::integrate(cloud, T_base_sensor, T_world_base) -> Engine.integrate(...)
  ElevationMapping::update(cloud, robot_position)        -> Engine.update(...)
Transforms are 4x4 row-major numpy arrays here and are handed to the ABI column-major
(Eigen::Isometry3d::matrix().data()).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import FdmCloud2Layout, FdmConfig, FdmGeometry, FdmScanStats, FdmTile


class EngineError(RuntimeError):
    pass


def _ck(rc):
    if rc < 0:
        raise EngineError(f"fdm_engine error {rc}: {capi.load().fdm_last_error().decode()}")
    return rc


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _colmajor16(T):
    T = np.asarray(T, dtype=np.float64).reshape(4, 4)
    return np.ascontiguousarray(T.T).reshape(16)  # column-major flattening


def _dptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


class HostArray:
    """A 1-D numpy array over pinned memory from the engine's pool (fdm_host_alloc): what the host
    entry points read IN PLACE instead of copying.  Returned to the pool when collected."""

    def __init__(self, n, dtype=np.float32):
        self._lib = capi.load()
        dt = np.dtype(dtype)
        self._p = self._lib.fdm_host_alloc(max(int(n) * dt.itemsize, 1))
        if not self._p:
            raise MemoryError("fdm_host_alloc failed")
        buf = (C.c_char * (int(n) * dt.itemsize)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=dt, count=int(n))
        self.pinned = bool(self._lib.fdm_host_is_pinned(self._p))

    def __del__(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.fdm_host_free(self._p)
            self._p = None


def host_array(values, dtype=np.float32):
    """Copy `values` into a pooled pinned array; keep the returned HostArray alive while `.array` is used."""
    v = np.ascontiguousarray(values, dtype=dtype).reshape(-1)
    h = HostArray(v.size, dtype)
    h.array[:] = v
    return h


class Engine:
    # engine options (fdm_engine_set_option) every new Engine receives right after creation; the GPU test
    # suite uses it to run each test on both scan pipelines (tests/conftest.py)
    default_options = {}

    def __init__(self, width, height, resolution, cfg=None, position=(0.0, 0.0), tile=None,
                 device=0):
        self._lib = capi.load()
        self.cfg = cfg if cfg is not None else capi.default_config()
        g = FdmGeometry()
        # ElevationMap::setGeometry(float,float,float) promotes float -> double
        # (elevation_map.hpp:112-116): 0.1f becomes 0.100000001490116...
        g.length_x = float(np.float32(width))
        g.length_y = float(np.float32(height))
        g.resolution = float(np.float32(resolution))
        g.position_x, g.position_y = float(position[0]), float(position[1])
        self._tile = None
        tp = None
        if tile is not None:
            self._tile = FdmTile(*[int(v) for v in tile])
            tp = C.byref(self._tile)
        h = C.c_void_p()
        _ck(self._lib.fdm_engine_create(C.byref(g), C.byref(self.cfg), tp, int(device), C.byref(h)))
        self._h = h
        geo = self.geometry()
        self.rows, self.cols = geo.rows, geo.cols
        self.s_rows = self._tile.rows if self._tile else self.rows
        self.s_cols = self._tile.cols if self._tile else self.cols
        for key, value in type(self).default_options.items():
            self.set_option(key, value)

    # -- lifetime --
    def close(self):
        if getattr(self, "_h", None):
            self._lib.fdm_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- configuration --
    def set_config(self, cfg):
        self.cfg = cfg
        _ck(self._lib.fdm_engine_set_config(self._h, C.byref(cfg)))

    def set_stream(self, stream_handle):
        _ck(self._lib.fdm_engine_set_stream(self._h, C.c_void_p(stream_handle)))

    def set_option(self, key, value):
        _ck(self._lib.fdm_engine_set_option(self._h, key.encode(), int(value)))

    # -- the hot path --
    def integrate(self, x, y, z, T_base_sensor, T_world_base, intensity=None, rgb=None,
                  sigma_z2=None):
        """Host arrays in, synchronous.  Returns (status, stats dict); status 0 == `true`."""
        x, y, z = _f32(x), _f32(y), _f32(z)
        a, c, v = _f32(intensity), _u32(rgb), _f32(sigma_z2)
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        st = FdmScanStats()
        rc = _ck(self._lib.fdm_engine_integrate(
            self._h, x.size, _ptr(x), _ptr(y), _ptr(z), _ptr(a), _ptr(c), _ptr(v),
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double)),
            C.byref(st)))
        return rc, st.as_dict()

    def integrate_points4(self, xyz1, T_base_sensor, T_world_base, intensity=None, rgb=None, sigma_z2=None):
        """The reference's own cloud layout: `xyz1` = n x 4 float32 records {x, y, z, 1} (nanopcl::PointCloud::points()),
        a C-contiguous HOST array (pinned memory is read in place).  Synchronous; returns (status, stats dict)."""
        import numpy as np
        p = xyz1 if isinstance(xyz1, np.ndarray) and xyz1.dtype == np.float32 and xyz1.flags["C_CONTIGUOUS"] \
            else np.ascontiguousarray(xyz1, dtype=np.float32)
        assert p.ndim == 2 and p.shape[1] == 4, p.shape
        a, c, v = _f32(intensity), _u32(rgb), _f32(sigma_z2)
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        st = FdmScanStats()
        rc = _ck(self._lib.fdm_engine_integrate_points4(
            self._h, p.shape[0], p.ctypes.data if p.shape[0] else None, _ptr(a), _ptr(c), _ptr(v),
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)))
        return rc, st.as_dict()

    def wait_torch(self):
        """Order the engine's stream behind torch's current stream (an event + hipStreamWaitEvent): tensors
        produced by asynchronous torch work (copy_, kernels, NCCL results) are complete before the engine
        reads them.  The tensor-taking enqueue-only entry points call this themselves."""
        import torch
        # torch's stream idle: everything it produced is complete, nothing to order behind (a stream query is ~1 us; an
        # event record + a cross-stream wait are two more commands per call)
        if torch.cuda.current_stream().query():
            return
        if getattr(self, "_ev_in", None) is None:
            self._ev_in = torch.cuda.Event()
        self._ev_in.record()
        _ck(self._lib.fdm_engine_wait_event(self._h, C.c_void_p(self._ev_in.cuda_event)))

    def torch_wait(self):
        """Order torch's current stream behind the engine: launches a held-back map update, then makes
        torch's stream wait for everything the engine has enqueued (the map is current, the input arrays
        of every enqueued scan are free)."""
        import torch
        if getattr(self, "_ev_out", None) is None:
            self._ev_out = torch.cuda.Event()
            self._ev_out.record()  # (creates the HIP event)
        _ck(self._lib.fdm_engine_record_event(self._h, C.c_void_p(self._ev_out.cuda_event)))
        torch.cuda.current_stream().wait_event(self._ev_out)

    def integrate_device(self, x, y, z, T_base_sensor, T_world_base, intensity=None, rgb=None,
                         sigma_z2=None):
        """torch device tensors in, enqueue only (no sync).  The engine's stream is ordered behind torch's
        current stream first; the tensors may be reused once the enqueued work has run (torch_wait(), sync())
        — the held-back map update does not read them."""
        self.wait_torch()
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        _ck(self._lib.fdm_engine_integrate_device(
            self._h, x.numel(), _dptr(x), _dptr(y), _dptr(z), _dptr(intensity), _dptr(rgb),
            _dptr(sigma_z2), tbs.ctypes.data_as(C.POINTER(C.c_double)),
            twb.ctypes.data_as(C.POINTER(C.c_double))))

    def integrate_async(self, x, y, z, T_base_sensor, T_world_base, intensity=None, rgb=None):
        """Host arrays (ideally pinned), enqueue-only: H2D copy + scan on the engine stream."""
        x, y, z = _f32(x), _f32(y), _f32(z)
        a, c = _f32(intensity), _u32(rgb)
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        _ck(self._lib.fdm_engine_integrate_async(
            self._h, x.size, _ptr(x), _ptr(y), _ptr(z), _ptr(a), _ptr(c), None,
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double))))

    def integrate_async_raw(self, n, hx, hy, hz, tbs16, twb16, hint=None, hrgb=None):
        """Raw host pointers + pre-flattened transforms: the minimum-overhead streaming call."""
        return self._lib.fdm_engine_integrate_async(self._h, n, hx, hy, hz, hint, hrgb, None, tbs16, twb16)

    def integrate_device_raw(self, n, dx, dy, dz, tbs16, twb16, dint=None, drgb=None, dvar=None):
        """Pre-flattened column-major transforms (ctypes double arrays) + raw device pointers:
        the minimum-overhead call used inside bench.py's timed loop."""
        return self._lib.fdm_engine_integrate_device(self._h, n, dx, dy, dz, dint, drgb, dvar,
                                                     tbs16, twb16)

    def integrate_device_batch(self, scans, count=None):
        """`scans`: a ctypes array of capi.FdmDeviceScan — `count` device-resident scans enqueued by ONE call
        across the language boundary."""
        return self._lib.fdm_engine_integrate_device_batch(self._h, len(scans) if count is None else count, scans)

    def integrate_host_batch(self, scans, count=None, wait=True):
        """`scans`: a ctypes array of capi.FdmDeviceScan whose pointers are HOST pointers (pinned: read in place; pageable:
        staged) — N consecutive integrate() calls in batch launches.  wait: (status, statistics) of the last scan."""
        st = FdmScanStats()
        rc = _ck(self._lib.fdm_engine_integrate_host_batch(self._h, len(scans) if count is None else count, scans,
                                                           C.byref(st) if wait else None))
        return (rc, st.as_dict()) if wait else rc

    def integrate_device_batch_timed(self, scans, count=None):
        """The same between timer_start() and timer_stop(): timer_ms() afterwards is the batch's device time."""
        return self._lib.fdm_engine_integrate_device_batch_timed(self._h, len(scans) if count is None else count, scans)

    # -- scan routing for spatially tiled global maps (fdm_route.hpp) --
    def route_scan(self, plan, x, y, z, T_base_sensor, T_world_base, send, counts, intensity=None, soa=False):
        """Partition this rank's slice (torch device tensors) by owner rank into `send` ([n, 4] float32 device
        tensor: x, y, z, intensity records; `soa`: per owner four channel blocks of pad4(count) floats, the buffer
        then needs n + 3 * world rows) and leave the per-owner counts + n_after_filter + n_in_map in
        `counts` (device int32 tensor of world + 2).  Enqueue-only."""
        self.wait_torch()
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        fn = self._lib.fdm_engine_route_scan_soa if soa else self._lib.fdm_engine_route_scan
        _ck(fn(
            self._h, C.byref(plan), x.numel(), _dptr(x), _dptr(y), _dptr(z), _dptr(intensity),
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double)),
            _dptr(send), _dptr(counts)))

    def integrate_points4_device(self, points4, n, T_base_sensor, T_world_base, has_intensity=True, any_in_map=True):
        """FastDEM::integrate of `n` received {x, y, z, intensity} records (device tensor).  Enqueue-only."""
        self.wait_torch()
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        _ck(self._lib.fdm_engine_integrate_points4_device(
            self._h, int(n), _dptr(points4) if n else None, int(bool(has_intensity)), int(bool(any_in_map)),
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double))))

    def integrate_soa4_device(self, share, n, T_base_sensor, T_world_base, has_intensity=True, any_in_map=True):
        """FastDEM::integrate of one routed share (route_scan(soa=True)): `share` = flat float32 device tensor holding
        x | y | z | intensity, pad4(n) floats each, read in place.  Enqueue-only."""
        self.wait_torch()
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        _ck(self._lib.fdm_engine_integrate_soa4_device(
            self._h, int(n), _dptr(share) if n else None, int(bool(has_intensity)), int(bool(any_in_map)),
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double))))

    def update(self, x, y, z, robot_xy=(0.0, 0.0), z_var=None, intensity=None, rgb=None):
        x, y, z = _f32(x), _f32(y), _f32(z)
        v, a, c = _f32(z_var), _f32(intensity), _u32(rgb)
        st = FdmScanStats()
        n = 0 if x is None else x.size
        _ck(self._lib.fdm_engine_update(self._h, n, _ptr(x), _ptr(y), _ptr(z), _ptr(v), _ptr(a),
                                        _ptr(c), float(robot_xy[0]), float(robot_xy[1]),
                                        C.byref(st)))
        return st.as_dict()

    def last_pipeline(self):
        """0 = per-cell scratch pipeline, 1 = per-tile record pools (large scans), -1 = no scan yet."""
        return self._lib.fdm_engine_last_pipeline(self._h)

    def last_batch(self):
        """Scans of the batch launch the last scan left in (0: it took the single-scan path)."""
        return self._lib.fdm_engine_last_batch(self._h)

    def timer_start(self):
        """Mark the start of a timed run of enqueue-only calls on the engine's stream."""
        _ck(self._lib.fdm_engine_timer_start(self._h))

    def timer_stop(self):
        """Launch the last scan's held-back update and mark the end."""
        _ck(self._lib.fdm_engine_timer_stop(self._h))

    def timer_ms(self):
        ms = C.c_float(0.0)
        _ck(self._lib.fdm_engine_timer_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def debug_batch_dirty(self):
        """(keys, aux words, zero-sign words) of the batch scratch that are not clean after a flush: (0, 0, 0)."""
        out = (C.c_uint64 * 3)()
        _ck(self._lib.fdm_engine_debug_batch_dirty(self._h, out))
        return tuple(int(v) for v in out)

    def batch_launches(self):
        """(small-scan batch launches, tile-batch launches) enqueued since the engine was created."""
        out = (C.c_uint64 * 2)()
        _ck(self._lib.fdm_engine_debug_batch_launches(self._h, out))
        return int(out[0]), int(out[1])

    def debug_timeline(self, cap_blocks=1 << 16):
        """(option dbg_timeline=1) -> (ticks[n_blocks, 2] uint64 of the 100 MHz clock, n_update_blocks) of the
        last fused large-scan launch."""
        buf = np.zeros((cap_blocks, 2), dtype=np.uint64)
        nb, nu = C.c_uint32(0), C.c_uint32(0)
        _ck(self._lib.fdm_engine_debug_timeline(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), cap_blocks,
                                                C.byref(nb), C.byref(nu)))
        return buf[:nb.value], nu.value

    def flush(self):
        """Launch a held-back map update (no wait)."""
        _ck(self._lib.fdm_engine_flush(self._h))

    def stream(self):
        """hipStream_t of the engine as an int."""
        return self._lib.fdm_engine_stream(self._h)

    def update_device(self, x, y, z, robot_xy=(0.0, 0.0), z_var=None, intensity=None, rgb=None):
        """ElevationMapping::update on torch device tensors (map-frame cloud), enqueue only."""
        self.wait_torch()
        _ck(self._lib.fdm_engine_update_device(self._h, x.numel(), _dptr(x), _dptr(y), _dptr(z), _dptr(z_var),
                                               _dptr(intensity), _dptr(rgb), float(robot_xy[0]),
                                               float(robot_xy[1])))

    def sync(self):
        _ck(self._lib.fdm_engine_sync(self._h))

    def last_stats(self):
        st = FdmScanStats()
        rc = _ck(self._lib.fdm_engine_last_stats(self._h, C.byref(st)))
        return rc, st.as_dict()

    # -- grid --
    def move(self, x, y):
        _ck(self._lib.fdm_engine_move(self._h, float(x), float(y)))

    def geometry(self):
        g = FdmGeometry()
        _ck(self._lib.fdm_engine_get_geometry(self._h, C.byref(g)))
        return g

    def set_position(self, x, y):
        _ck(self._lib.fdm_engine_set_position(self._h, float(x), float(y)))

    def set_start_index(self, r, c):
        _ck(self._lib.fdm_engine_set_start_index(self._h, int(r), int(c)))

    # -- layers --
    def layers(self):
        n = _ck(self._lib.fdm_engine_num_layers(self._h))
        return [self._lib.fdm_engine_layer_name(self._h, i).decode() for i in range(n)]

    def exists(self, name):
        return bool(_ck(self._lib.fdm_engine_layer_exists(self._h, name.encode())))

    def add(self, name, value=float("nan")):
        _ck(self._lib.fdm_engine_layer_add(self._h, name.encode(), float(value)))

    def layer(self, name):
        """Download one layer as a (rows, cols) float32 array (Fortran order like MatrixXf)."""
        out = np.empty((self.s_rows, self.s_cols), dtype=np.float32, order="F")
        _ck(self._lib.fdm_engine_layer_download(self._h, name.encode(), _ptr(out), self.s_rows,
                                                self.s_cols))
        return out

    def set_layer(self, name, arr):
        a = np.asfortranarray(arr, dtype=np.float32)
        assert a.shape == (self.s_rows, self.s_cols)
        _ck(self._lib.fdm_engine_layer_upload(self._h, name.encode(), _ptr(a), self.s_rows,
                                              self.s_cols))

    def copy_layer_from(self, other, name):
        """Layer `name` of another engine (same device, same window) into this one, device to device."""
        _ck(self._lib.fdm_engine_layer_copy(self._h, other._h, name.encode()))

    def layer_device_ptr(self, name):
        return self._lib.fdm_engine_layer_device_ptr(self._h, name.encode())

    def clear(self, name=None):
        _ck(self._lib.fdm_engine_clear(self._h, None if name is None else name.encode()))

    def region_pack(self, r0, c0, nr, nc, names, dbuf_ptr):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        _ck(self._lib.fdm_engine_region_pack(self._h, r0, c0, nr, nc, arr, len(names),
                                             C.c_void_p(dbuf_ptr)))

    def region_unpack(self, r0, c0, nr, nc, names, dbuf_ptr):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        _ck(self._lib.fdm_engine_region_unpack(self._h, r0, c0, nr, nc, arr, len(names),
                                               C.c_void_p(dbuf_ptr)))

    # -- stencil post-processing --
    def apply_inpainting(self, max_iterations=3, min_valid_neighbors=2, inplace=False):
        _ck(self._lib.fdm_engine_apply_inpainting(self._h, int(max_iterations), int(min_valid_neighbors),
                                                  int(inplace)))

    def apply_spatial_smoothing(self, layer, kernel_size=3, min_valid_neighbors=5):
        _ck(self._lib.fdm_engine_apply_spatial_smoothing(self._h, layer.encode(), int(kernel_size),
                                                         int(min_valid_neighbors)))

    def apply_uncertainty_fusion(self, enabled=True, search_radius=0.15, spatial_sigma=0.05,
                                 quantile_lower=0.01, quantile_upper=0.99, min_valid_neighbors=3):
        cfg = capi.FdmFusionConfig(int(enabled), search_radius, spatial_sigma, quantile_lower,
                                   quantile_upper, int(min_valid_neighbors))
        _ck(self._lib.fdm_engine_apply_uncertainty_fusion(self._h, C.byref(cfg)))

    def apply_feature_extraction(self, analysis_radius=0.3, min_valid_neighbors=4,
                                 step_lower_percentile=0.05, step_upper_percentile=0.95):
        _ck(self._lib.fdm_engine_apply_feature_extraction(self._h, analysis_radius, int(min_valid_neighbors),
                                                          step_lower_percentile, step_upper_percentile))

    # -- ingest --
    @staticmethod
    def cloud2_layout(point_step, x, y, z, intensity=-1, intensity_type=7, rgb=-1):
        return FdmCloud2Layout(int(point_step), int(x), int(y), int(z), int(intensity),
                               int(intensity_type), int(rgb))

    def ingest_cloud2(self, blob, n_points, layout):
        """nanopcl::from(PointCloud2): decode a host byte blob into SoA channels in HBM.
        Returns dict of numpy copies of the kept points' channels (for tests)."""
        b = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
        n = C.c_uint64(0)
        _ck(self._lib.fdm_engine_ingest_cloud2(self._h, _ptr(b), 0, int(n_points), C.byref(layout),
                                               C.byref(n)))
        ptrs = [C.c_void_p() for _ in range(5)]
        m = C.c_uint64(0)
        _ck(self._lib.fdm_engine_ingested(self._h, *[C.byref(p) for p in ptrs], C.byref(m)))
        assert m.value == n.value
        import torch
        out = {}
        for name, p, dt in zip(("x", "y", "z", "intensity", "rgb"), ptrs,
                               (np.float32,) * 4 + (np.uint32,)):
            if not p.value or n.value == 0:
                out[name] = None if name in ("intensity", "rgb") and not p.value else np.empty(0, dt)
                continue
            h = np.empty(n.value, dtype=dt)
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(_ptr(h), p, h.nbytes, 2) == 0
            out[name] = h
        del torch
        return out

    def integrate_cloud2(self, blob, n_points, layout, T_base_sensor, T_world_base, on_device_ptr=None):
        """from_impl + integrate in one call; blob = bytes-like (host) or a device pointer."""
        tbs, twb = _colmajor16(T_base_sensor), _colmajor16(T_world_base)
        st = FdmScanStats()
        if on_device_ptr is not None:
            data, dev = C.c_void_p(on_device_ptr), 1
        else:
            b = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
            data, dev = _ptr(b), 0
        rc = _ck(self._lib.fdm_engine_integrate_cloud2(
            self._h, data, dev, int(n_points), C.byref(layout),
            tbs.ctypes.data_as(C.POINTER(C.c_double)), twb.ctypes.data_as(C.POINTER(C.c_double)),
            C.byref(st)))
        return rc, st.as_dict()

    # -- egress --
    def pack_cloud(self, elevation_layer="elevation", sub=None):
        """toPointCloud2Impl on the device: (fields, point_step, data[n_points, n_fields] float32)."""
        r0, c0, nr, nc = sub if sub is not None else (0, 0, -1, -1)
        n, step = C.c_uint64(0), C.c_uint32(0)
        names = C.create_string_buffer(4096)
        name = elevation_layer.encode()
        _ck(self._lib.fdm_engine_pack_cloud(self._h, name, r0, c0, nr, nc, None, 0, C.byref(n),
                                            C.byref(step), names, 4096))
        fields = names.value.decode().split("\n")
        data = np.empty((n.value, len(fields)), dtype=np.float32)
        if n.value:
            _ck(self._lib.fdm_engine_pack_cloud(self._h, name, r0, c0, nr, nc, _ptr(data), data.nbytes,
                                                C.byref(n), C.byref(step), None, 0))
        return fields, step.value, data

    def pack_cloud_device(self, elevation_layer="elevation", sub=None):
        """Records stay in HBM: (device pointer, n_points, point_step)."""
        r0, c0, nr, nc = sub if sub is not None else (0, 0, -1, -1)
        n, step, d = C.c_uint64(0), C.c_uint32(0), C.c_void_p()
        _ck(self._lib.fdm_engine_pack_cloud_device(self._h, elevation_layer.encode(), r0, c0, nr, nc,
                                                   C.byref(d), C.byref(n), C.byref(step)))
        return d.value, n.value, step.value

    # -- raycasting stage --
    def apply_raycasting(self, x, y, z, sensor_origin, rc=None):
        """fastdem::applyRaycasting(map, scan, sensor_origin, config); host arrays, sync.
        rc: capi.FdmRaycastConfig, None = the raycasting fields of the engine config."""
        x, y, z = _f32(x), _f32(y), _f32(z)
        o = (C.c_float * 3)(*[float(v) for v in sensor_origin])
        _ck(self._lib.fdm_engine_apply_raycasting(self._h, x.size, _ptr(x), _ptr(y), _ptr(z), o,
                                                  None if rc is None else C.byref(rc)))

    def apply_raycasting_device(self, x, y, z, sensor_origin, rc=None):
        o = (C.c_float * 3)(*[float(v) for v in sensor_origin])
        _ck(self._lib.fdm_engine_apply_raycasting_device(self._h, x.numel(), _dptr(x), _dptr(y),
                                                         _dptr(z), o,
                                                         None if rc is None else C.byref(rc)))

    def voxel_any(self, x, y, z, voxel_size):
        """filters::voxelGrid(cloud, voxel_size, VoxelMode::ANY): original indices, output order."""
        x, y, z = _f32(x), _f32(y), _f32(z)
        out = np.empty(max(x.size, 1), dtype=np.uint32)
        n = C.c_uint64(0)
        _ck(self._lib.fdm_engine_voxel_any(self._h, x.size, _ptr(x), _ptr(y), _ptr(z),
                                           float(voxel_size), _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def last_ray_ms(self):
        ms = C.c_float(0)
        _ck(self._lib.fdm_engine_last_ray_ms(self._h, C.byref(ms)))
        return float(ms.value)

    # -- scan callbacks --
    def capture(self, preprocessed=True, rasterized=True):
        _ck(self._lib.fdm_engine_capture(self._h, int(preprocessed), int(rasterized)))

    def last_preprocessed(self, cap):
        a = [np.empty(cap, dtype=np.float32) for _ in range(4)]
        n = C.c_uint64(0)
        _ck(self._lib.fdm_engine_last_preprocessed(self._h, cap, *[_ptr(v) for v in a], C.byref(n)))
        return [v[:n.value] for v in a]

    def last_preprocessed_cov(self, cap):
        """(n, 3, 3) covariance channel of the preprocessed cloud (capture(preprocessed=2))."""
        a = np.empty((cap, 9), dtype=np.float32)
        n = C.c_uint64(0)
        _ck(self._lib.fdm_engine_last_preprocessed_cov(self._h, cap, _ptr(a), C.byref(n)))
        return a[:n.value].reshape(-1, 3, 3).transpose(0, 2, 1)  # column-major 3x3 per point

    def last_rasterized(self, cap):
        a = [np.empty(cap, dtype=np.float32) for _ in range(3)]
        n = C.c_uint64(0)
        _ck(self._lib.fdm_engine_last_rasterized(self._h, cap, *[_ptr(v) for v in a], C.byref(n)))
        return [v[:n.value] for v in a]

    # -- instrumentation --
    def enable_cell_ids(self, on=True):
        _ck(self._lib.fdm_engine_enable_cell_ids(self._h, int(on)))

    def last_cell_ids(self, n):
        out = np.empty(n, dtype=np.int32)
        _ck(self._lib.fdm_engine_last_cell_ids(self._h, _ptr(out), n))
        return out

    def enable_profile(self, on=True):
        _ck(self._lib.fdm_engine_enable_profile(self._h, int(on)))

    def last_kernel_ms(self):
        ms = (C.c_float * 2)()
        _ck(self._lib.fdm_engine_last_kernel_ms(self._h, ms))
        return float(ms[0]), float(ms[1])
