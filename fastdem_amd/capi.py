"""ctypes declarations for include/fdm_engine.h (libfdm_engine.so).

Plumbing only: the product is the HIP library.  Loading fails loudly when the library
has not been built — there is no Python or CPU fallback for any entry point.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfdm_engine.so")  # (measurement scripts assign another build of the same ABI here before load())


class FdmConfig(C.Structure):
    """fdm_config == fastdem::Config for this path (config/fastdem.hpp:23-38)."""

    _fields_ = [
        ("z_min", C.c_float), ("z_max", C.c_float),
        ("range_min", C.c_float), ("range_max", C.c_float),
        ("sensor_type", C.c_int32),
        ("lidar_range_noise", C.c_float), ("lidar_angular_noise", C.c_float),
        ("rgbd_normal_a", C.c_float), ("rgbd_normal_b", C.c_float),
        ("rgbd_normal_c", C.c_float), ("rgbd_lateral_factor", C.c_float),
        ("constant_uncertainty", C.c_float),
        ("mode", C.c_int32), ("estimation_type", C.c_int32),
        ("kalman_min_variance", C.c_float), ("kalman_max_variance", C.c_float),
        ("kalman_process_noise", C.c_float),
        ("p2_dn", C.c_float * 5),
        ("p2_elevation_marker", C.c_int32),
        ("p2_max_sample_count", C.c_float),
        # config::Raycasting (config/postprocess.hpp:16-23)
        ("raycast_enabled", C.c_int32),
        ("rc_height_conflict_threshold", C.c_float), ("rc_log_odds_observed", C.c_float),
        ("rc_log_odds_ghost", C.c_float), ("rc_log_odds_max", C.c_float),
        ("rc_clear_threshold", C.c_float),
    ]


class FdmRaycastConfig(C.Structure):
    """fdm_raycast_config == config::Raycasting (config/postprocess.hpp:16-23)."""

    _fields_ = [
        ("enabled", C.c_int32),
        ("height_conflict_threshold", C.c_float), ("log_odds_observed", C.c_float),
        ("log_odds_ghost", C.c_float), ("log_odds_max", C.c_float), ("clear_threshold", C.c_float),
    ]


class FdmCloud2Layout(C.Structure):
    """fdm_cloud2_layout: byte offsets of the PointCloud2 fields integrate() consumes (-1 = absent)."""

    _fields_ = [
        ("point_step", C.c_uint32),
        ("off_x", C.c_int32), ("off_y", C.c_int32), ("off_z", C.c_int32),
        ("off_intensity", C.c_int32), ("intensity_type", C.c_int32),
        ("off_rgb", C.c_int32),
    ]


class FdmFusionConfig(C.Structure):
    """fdm_fusion_config == config::UncertaintyFusion (config/postprocess.hpp:32-39)."""

    _fields_ = [
        ("enabled", C.c_int32),
        ("search_radius", C.c_float), ("spatial_sigma", C.c_float),
        ("quantile_lower", C.c_float), ("quantile_upper", C.c_float),
        ("min_valid_neighbors", C.c_int32),
    ]


class FdmGeometry(C.Structure):
    _fields_ = [
        ("length_x", C.c_double), ("length_y", C.c_double), ("resolution", C.c_double),
        ("position_x", C.c_double), ("position_y", C.c_double),
        ("rows", C.c_int32), ("cols", C.c_int32),
        ("start_row", C.c_int32), ("start_col", C.c_int32),
    ]


class FdmTile(C.Structure):
    _fields_ = [
        ("row0", C.c_int32), ("col0", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
        ("own_row0", C.c_int32), ("own_col0", C.c_int32),
        ("own_rows", C.c_int32), ("own_cols", C.c_int32),
    ]


class FdmScanStats(C.Structure):
    _fields_ = [
        ("n_input", C.c_uint32), ("n_after_filter", C.c_uint32),
        ("n_in_map", C.c_uint32), ("n_cells_touched", C.c_uint32),
        ("shift_rows", C.c_int32), ("shift_cols", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class FdmRoutePlan(C.Structure):  # fdm_route_plan (include/fdm_engine.h)
    _fields_ = [("world", C.c_int32), ("grid_rows", C.c_int32), ("grid_cols", C.c_int32), ("pad", C.c_int32),
                ("row_edge", C.c_int32 * 17), ("col_edge", C.c_int32 * 17)]


class FdmDeviceScan(C.Structure):  # fdm_device_scan (include/fdm_engine.h)
    _fields_ = [
        ("n", C.c_uint64),
        ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("intensity", C.c_void_p),
        ("rgb", C.c_void_p), ("sigma_z2", C.c_void_p),
        ("T_base_sensor", C.c_double * 16), ("T_world_base", C.c_double * 16),
    ]


SENSOR_CONSTANT, SENSOR_LIDAR, SENSOR_RGBD = 0, 1, 2
MODE_LOCAL, MODE_GLOBAL = 0, 1
EST_KALMAN, EST_P2 = 0, 1
FDM_OK, FDM_SKIP_EMPTY_CLOUD, FDM_SKIP_ALL_FILTERED = 0, 1, 2

_P = C.c_void_p
_F = C.POINTER(C.c_float)
_U = C.POINTER(C.c_uint32)
_D = C.POINTER(C.c_double)

# name -> (restype, argtypes): every symbol include/fdm_engine.h declares
PROTOTYPES = {
    "fdm_default_config": (None, [C.POINTER(FdmConfig)]),
    "fdm_last_error": (C.c_char_p, []),
    "fdm_engine_create": (C.c_int, [C.POINTER(FdmGeometry), C.POINTER(FdmConfig),
                                    C.POINTER(FdmTile), C.c_int, C.POINTER(_P)]),
    "fdm_engine_create_map": (C.c_int, [C.POINTER(FdmGeometry), C.POINTER(FdmTile), C.c_int,
                                        C.POINTER(_P)]),
    "fdm_engine_destroy": (None, [_P]),
    "fdm_engine_set_config": (C.c_int, [_P, C.POINTER(FdmConfig)]),
    "fdm_engine_set_stream": (C.c_int, [_P, _P]),
    "fdm_engine_integrate": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, _D, _D,
                                       C.POINTER(FdmScanStats)]),
    "fdm_engine_integrate_points4": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _D, _D, C.POINTER(FdmScanStats)]),
    "fdm_engine_integrate_device": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, _D, _D]),
    "fdm_engine_integrate_device_batch": (C.c_int, [_P, C.c_uint32, _P]),
    "fdm_engine_integrate_device_batch_timed": (C.c_int, [_P, C.c_uint32, _P]),
    "fdm_engine_integrate_host_batch": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "fdm_engine_integrate_async": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double)]),
    "fdm_engine_route_scan": (C.c_int, [_P, C.POINTER(FdmRoutePlan), C.c_uint64, _P, _P, _P, _P, _D, _D, _P, _P]),
    "fdm_engine_integrate_points4_device": (C.c_int, [_P, C.c_uint64, _P, C.c_int, C.c_int, _D, _D]),
    "fdm_engine_route_scan_soa": (C.c_int, [_P, C.POINTER(FdmRoutePlan), C.c_uint64, _P, _P, _P, _P, _D, _D, _P, _P]),
    "fdm_engine_integrate_soa4_device": (C.c_int, [_P, C.c_uint64, _P, C.c_int, C.c_int, _D, _D]),
    "fdm_engine_update": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, C.c_double,
                                    C.c_double, C.POINTER(FdmScanStats)]),
    "fdm_engine_update_device": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, _P, C.c_double,
                                           C.c_double]),
    "fdm_host_alloc": (_P, [C.c_uint64]),
    "fdm_host_free": (None, [_P]),
    "fdm_host_trim": (None, []),
    "fdm_host_is_pinned": (C.c_int, [_P]),
    "fdm_engine_flush": (C.c_int, [_P]),
    "fdm_engine_stream": (_P, [_P]),
    "fdm_engine_last_pipeline": (C.c_int, [_P]),
    "fdm_engine_last_batch": (C.c_int, [_P]),
    "fdm_engine_timer_start": (C.c_int, [_P]),
    "fdm_engine_timer_stop": (C.c_int, [_P]),
    "fdm_engine_timer_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "fdm_engine_debug_batch_dirty": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "fdm_engine_debug_batch_launches": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "fdm_engine_debug_timeline": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fdm_engine_record_event": (C.c_int, [_P, _P]),
    "fdm_engine_wait_event": (C.c_int, [_P, _P]),
    "fdm_engine_sync": (C.c_int, [_P]),
    "fdm_engine_last_stats": (C.c_int, [_P, C.POINTER(FdmScanStats)]),
    "fdm_engine_move": (C.c_int, [_P, C.c_double, C.c_double]),
    "fdm_engine_get_geometry": (C.c_int, [_P, C.POINTER(FdmGeometry)]),
    "fdm_engine_set_position": (C.c_int, [_P, C.c_double, C.c_double]),
    "fdm_engine_set_start_index": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "fdm_engine_num_layers": (C.c_int, [_P]),
    "fdm_engine_layer_name": (C.c_char_p, [_P, C.c_int]),
    "fdm_engine_layer_exists": (C.c_int, [_P, C.c_char_p]),
    "fdm_engine_layer_add": (C.c_int, [_P, C.c_char_p, C.c_float]),
    "fdm_engine_layer_download": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int32]),
    "fdm_engine_layer_upload": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int32]),
    "fdm_engine_layer_device_ptr": (_P, [_P, C.c_char_p]),
    "fdm_engine_clear": (C.c_int, [_P, C.c_char_p]),
    "fdm_engine_layer_copy": (C.c_int, [_P, _P, C.c_char_p]),
    "fdm_engine_region_pack": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_char_p), C.c_int, _P]),
    "fdm_engine_region_unpack": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.POINTER(C.c_char_p), C.c_int, _P]),
    "fdm_engine_regions_pack": (C.c_int, [_P, C.c_int32, _P, C.POINTER(C.c_char_p), C.c_int, _P]),
    "fdm_engine_regions_unpack": (C.c_int, [_P, C.c_int32, _P, C.POINTER(C.c_char_p), C.c_int, _P]),
    "fdm_engine_capture": (C.c_int, [_P, C.c_int, C.c_int]),
    "fdm_engine_last_preprocessed": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, C.POINTER(C.c_uint64)]),
    "fdm_engine_last_preprocessed_cov": (C.c_int, [_P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "fdm_engine_last_rasterized": (C.c_int, [_P, C.c_uint64, _P, _P, _P, C.POINTER(C.c_uint64)]),
    "fdm_engine_apply_raycasting": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _F, C.POINTER(FdmRaycastConfig)]),
    "fdm_engine_apply_raycasting_device": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _F,
                                                     C.POINTER(FdmRaycastConfig)]),
    "fdm_engine_voxel_any": (C.c_int, [_P, C.c_uint64, _P, _P, _P, C.c_float, _P, C.POINTER(C.c_uint64)]),
    "fdm_engine_last_ray_ms": (C.c_int, [_P, _F]),
    "fdm_engine_apply_inpainting": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "fdm_engine_apply_spatial_smoothing": (C.c_int, [_P, C.c_char_p, C.c_int, C.c_int]),
    "fdm_engine_apply_uncertainty_fusion": (C.c_int, [_P, C.POINTER(FdmFusionConfig)]),
    "fdm_engine_apply_feature_extraction": (C.c_int, [_P, C.c_float, C.c_int, C.c_float, C.c_float]),
    "fdm_engine_ingest_cloud2": (C.c_int, [_P, _P, C.c_int, C.c_uint64, C.POINTER(FdmCloud2Layout),
                                           C.POINTER(C.c_uint64)]),
    "fdm_engine_ingested": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                                      C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "fdm_engine_integrate_cloud2": (C.c_int, [_P, _P, C.c_int, C.c_uint64, C.POINTER(FdmCloud2Layout),
                                              C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.POINTER(FdmScanStats)]),
    "fdm_engine_pack_cloud": (C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P,
                                        C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                        C.c_char_p, C.c_uint64]),
    "fdm_engine_pack_cloud_device": (C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.POINTER(_P), C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint32)]),
    "fdm_engine_enable_cell_ids": (C.c_int, [_P, C.c_int]),
    "fdm_engine_last_cell_ids": (C.c_int, [_P, _P, C.c_uint64]),
    "fdm_engine_enable_profile": (C.c_int, [_P, C.c_int]),
    "fdm_engine_last_kernel_ms": (C.c_int, [_P, _F]),
    "fdm_engine_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
}

_lib = None


def load():
    """dlopen libfdm_engine.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  fastdem_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first; ours resolves to the same soname)
    except Exception:  # pragma: no cover - torch is plumbing, the library works without it
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_config():
    cfg = FdmConfig()
    load().fdm_default_config(C.byref(cfg))
    return cfg
