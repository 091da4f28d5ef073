/* fdm_engine.h — C ABI of the MI355X elevation-map update engine (libfdm_engine.so).
 *
 * This is the drop-in boundary for ONE path of Ikhyeon-Cho/FastDEM:
 *   fastdem::FastDEM::integrate(cloud, T_base_sensor, T_world_base)
 * The reference has no FFI; its boundary is the C++ class surface
 * (fastdem/include/fastdem/fastdem.hpp:54-158).  A maintainer swaps the body of
 * FastDEM::integrateImpl (fastdem/src/fastdem.cpp:133-162) for the calls below —
 * INTEGRATION.md shows the binding.  Every entry point cites what it replaces.
 *
 * Conventions
 *   - plain pointers and sizes; no C++/torch types; no exceptions cross the boundary
 *   - int status: 0 ok, >0 "skipped" (the reference's `return false` cases), <0 error
 *   - transforms are column-major double[16] == Eigen::Isometry3d::matrix().data()
 *   - layers are float32, column-major rows x cols == Eigen::MatrixXf storage
 *     (fastdem/include/fastdem/bridge/ros/impl.hpp:117-119)
 *   - one engine = one device + one HIP stream; not thread-safe, caller serialises
 *     (same contract as fastdem.hpp:48-53)
 *   - host pointers are borrowed for the duration of the call only (pinned arrays handed to the
 *     enqueue-only fdm_engine_integrate_async: until the work that call enqueued has run); device
 *     pointers passed to the *_device entry points follow the ordinary stream contract — free to
 *     reuse once the work the call enqueued has run (fdm_engine_record_event / fdm_engine_sync);
 *     a held-back map update never reads them (see fdm_engine_integrate_device)
 */
#ifndef FDM_ENGINE_H
#define FDM_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdm_engine fdm_engine;

/* Mirrors fastdem::Config for this path, field for field
 * (config/fastdem.hpp:23-38, config/sensor_model.hpp:10-37, config/mapping.hpp:10-48).
 * The last six fields are config::Raycasting (config/postprocess.hpp:16-23). */
typedef struct fdm_config {
  float z_min, z_max, range_min, range_max;     /* config::PointFilter */
  int32_t sensor_type;                          /* SensorType: 0 Constant, 1 LiDAR, 2 RGBD */
  float lidar_range_noise, lidar_angular_noise;
  float rgbd_normal_a, rgbd_normal_b, rgbd_normal_c, rgbd_lateral_factor;
  float constant_uncertainty;
  int32_t mode;                                 /* MappingMode: 0 LOCAL, 1 GLOBAL */
  int32_t estimation_type;                      /* EstimationType: 0 Kalman, 1 P2Quantile */
  float kalman_min_variance, kalman_max_variance, kalman_process_noise;
  float p2_dn[5];
  int32_t p2_elevation_marker;
  float p2_max_sample_count;
  int32_t raycast_enabled;                      /* config::Raycasting::enabled (default 0) */
  float rc_height_conflict_threshold, rc_log_odds_observed, rc_log_odds_ghost, rc_log_odds_max,
      rc_clear_threshold;
} fdm_config;

/* config::Raycasting on its own (config/postprocess.hpp:16-23), for applyRaycasting called directly. */
typedef struct fdm_raycast_config {
  int32_t enabled;
  float height_conflict_threshold, log_odds_observed, log_odds_ghost, log_odds_max, clear_threshold;
} fdm_raycast_config;

/* nanogrid::GridMap geometry (length, resolution, position, circular-buffer start).
 * rows/cols are outputs of create (size = round(length / resolution)). */
typedef struct fdm_geometry {
  double length_x, length_y, resolution, position_x, position_y;
  int32_t rows, cols, start_row, start_col;
} fdm_geometry;

/* Multi-GPU spatial tiling, GLOBAL mode only.  This engine STORES the
 * window [row0,row0+rows) x [col0,col0+cols) of the global buffer (owned cells plus a
 * read-only halo ring) and UPDATES only the owned window [own_row0,own_row0+own_rows) x
 * [own_col0,own_col0+own_cols); points landing elsewhere are ignored by this engine.
 * Cell indices are always computed against the GLOBAL geometry, so they are bit-identical
 * to the single-GPU map. */
typedef struct fdm_tile {
  int32_t row0, col0, rows, cols;
  int32_t own_row0, own_col0, own_rows, own_cols;
} fdm_tile;

typedef struct fdm_scan_stats {
  uint32_t n_input;         /* points handed in */
  uint32_t n_after_filter;  /* survived cropRange + cropZ (fastdem.cpp:175-176) */
  uint32_t n_in_map;        /* getIndex succeeded (elevation_mapping.cpp:55) */
  uint32_t n_cells_touched; /* |CellObservations| */
  int32_t shift_rows, shift_cols; /* index shift of the LOCAL-mode move applied */
} fdm_scan_stats;

enum {
  FDM_OK = 0,
  FDM_SKIP_EMPTY_CLOUD = 1,   /* fastdem.cpp:125-128 -> false */
  FDM_SKIP_ALL_FILTERED = 2,  /* fastdem.cpp:138     -> false */
  FDM_ERR_INVALID = -1,
  FDM_ERR_HIP = -2,
  FDM_ERR_NO_LAYER = -3,
  FDM_ERR_NO_DEVICE = -4
};

void fdm_default_config(fdm_config* cfg);      /* Config{} defaults */
const char* fdm_last_error(void);              /* text of the last <0 status on this thread */

/* ElevationMap(width,height,resolution,frame) + FastDEM(map,cfg) construction
 * (elevation_map.hpp:101-116, fastdem.cpp:19-22, elevation_mapping.cpp:11-39):
 * allocates the layers in HBM with the reference's initial constants.
 * g->length_*, resolution, position_* are read; tile may be NULL (whole map). */
int fdm_engine_create(const fdm_geometry* g, const fdm_config* cfg, const fdm_tile* tile,
                      int device, fdm_engine** out);
/* ElevationMap(width,height,resolution,frame) ALONE (elevation_map.hpp:101-116): the three
 * default layers, no estimator layers yet.  fdm_engine_set_config() then plays the role of
 * constructing FastDEM / ElevationMapping on that map (ensureLayers + obstacle layer). */
int fdm_engine_create_map(const fdm_geometry* g, const fdm_tile* tile, int device, fdm_engine** out);
void fdm_engine_destroy(fdm_engine* e);

/* FastDEM fluent setters (fastdem.cpp:28-62): filters/sensor params take effect on
 * the next scan; a new estimator type adds its layers, the old ones persist. */
int fdm_engine_set_config(fdm_engine* e, const fdm_config* cfg);

/* Run on the caller's HIP stream (hipStream_t as void*; NULL = engine's own stream). */
int fdm_engine_set_stream(fdm_engine* e, void* hip_stream);

/* FastDEM::integrate(const PointCloud&, T_base_sensor, T_world_base) — fastdem.cpp:122-162.
 * SoA host arrays (x,y,z required; intensity / rgb = 0x00RRGGBB / sigma_z2 nullable).
 * sigma_z2, when given, replaces the built-in sensor model: it is the (2,2) element of
 * R*Sigma*R^T per point, for user SensorModel subclasses (fastdem.hpp:79-80).
 * Synchronous: returns after the map update; status as above. */
int fdm_engine_integrate(fdm_engine* e, uint64_t n, const float* x, const float* y,
                         const float* z, const float* intensity, const uint32_t* rgb,
                         const float* sigma_z2, const double T_base_sensor[16],
                         const double T_world_base[16], fdm_scan_stats* out);

/* Same, inputs already resident in HBM; enqueue-only (no host sync).  The skip
 * decisions of fastdem.cpp:125-138 are taken on the device; read them back with
 * fdm_engine_last_stats().
 * The map update of the scan is HELD BACK and leaves in the same launch as the next scan's bin kernel
 * (one launch per scan instead of two); every other entry point — fdm_engine_sync() included — first
 * launches a held-back update, so the map is always current when it is read through this API.
 * A caller that reads layers through raw device pointers on its own stream orders itself behind
 * fdm_engine_record_event() (which launches the held-back update first).
 * Input arrays follow the ordinary stream contract: they may be reused as soon as the work this call
 * enqueued has run (an event from fdm_engine_record_event, or fdm_engine_sync) — the held-back update
 * never reads them (the bin kernel leaves what it needs in engine-owned memory).  Producers on another
 * stream hand the arrays over with fdm_engine_wait_event().
 * fdm_engine_set_option(e, "overlap", 0) turns the hold-back off. */
int fdm_engine_integrate_device(fdm_engine* e, uint64_t n, const float* d_x, const float* d_y,
                                const float* d_z, const float* d_intensity, const uint32_t* d_rgb,
                                const float* d_sigma_z2, const double T_base_sensor[16],
                                const double T_world_base[16]);

/* A batch of device-resident scans in ONE call (a bag replay, a driver that queues ahead): the map it leaves is
 * exactly what `count` consecutive fdm_engine_integrate_device calls leave (every layer bit for bit), without the
 * per-call crossing of the language boundary — and, because the engine sees the scans up front, without one launch
 * per scan: runs of small plain scans of one sensor (same T_base_sensor, same optional channels, no captures / cell
 * ids, a map of at most 2^18 cells) are binned sixteen to a launch and applied cell by cell in scan order by the
 * same launch's update half (fastdem_amd/csrc/fdm_multi.hpp; configs[1]: 1.05 us instead of 4-6 us per scan).  With
 * raycast_enabled (fastdem.cpp:152-159) the stage of every scan rides in the batch as well — voxel filter, ray walks,
 * ghost resolution, each scan's behind that scan's map update (fdm_rbatch.hpp; scans of <= 64 K points on an untiled
 * engine; 9.3 us instead of 60 us per VLP-16 scan).  Scans that do not qualify take the single-scan path in place.
 * Options "batch" 0/1, "batch_max" 0 (automatic: 32 with the quantile estimator or with raycasting on, 16 with Kalman alone) or 2..32, "batch_ray" 0/1. */
typedef struct fdm_device_scan {
  uint64_t n;
  const float *x, *y, *z, *intensity; /* device pointers; intensity nullable */
  const uint32_t* rgb;                /* nullable */
  const float* sigma_z2;              /* nullable */
  double T_base_sensor[16];           /* column-major */
  double T_world_base[16];
} fdm_device_scan;
int fdm_engine_integrate_device_batch(fdm_engine* e, uint32_t count, const fdm_device_scan* scans);
/* Returns FDM_OK or the first error (< 0); an empty cloud in the batch is skipped and the batch goes on. */
/* The same for HOST clouds (the pointers of `host_scans` are host pointers): N consecutive FastDEM::integrate calls —
 * a bag replay, a driver that hands over the scans of the last 100 ms — in batch launches.  Pinned channels
 * (fdm_host_alloc: every nanopcl::PointCloud of the C++ mirror) are read in place over PCIe, once; pageable ones are
 * staged.  out_last != NULL: waits and returns the LAST scan's status and statistics like fdm_engine_integrate;
 * NULL: enqueue-only — pinned clouds must then stay untouched until fdm_engine_sync(). */
int fdm_engine_integrate_host_batch(fdm_engine* e, uint32_t count, const fdm_device_scan* host_scans,
                                    fdm_scan_stats* out_last);

/* FastDEM::integrate(cloud, T_base_sensor, T_world_base) on the reference's own point layout — replaces
 * fastdem/src/fastdem.cpp:122-190 for a caller that holds a nanopcl::PointCloud: `xyz1` = cloud.points().data(), n
 * contiguous 16-byte {x, y, z, 1} records (nanopcl/core/point_cloud.hpp:126-134, core/types.hpp:19-22), HOST memory,
 * 16-byte aligned.  Pinned memory (fdm_host_alloc / hipHostMalloc) is read in place, once, over PCIe; pageable memory
 * is copied first.  intensity / rgb / sigma_z2: the optional channels, separate host arrays of n entries (nullable), as
 * in fdm_engine_integrate.  Synchronous; status and statistics as fdm_engine_integrate. */
int fdm_engine_integrate_points4(fdm_engine* e, uint64_t n, const float* xyz1, const float* intensity,
                                 const uint32_t* rgb, const float* sigma_z2, const double T_base_sensor[16],
                                 const double T_world_base[16], fdm_scan_stats* out);

/* Same, HOST arrays, enqueue-only; nothing waits.  For a stream of scans from host memory (bag replay,
 * a ROS callback).
 *   PINNED arrays (fdm_host_alloc / hipHostMalloc / hipHostRegister): no copy is queued — the bin kernel
 *     reads the arrays in place over PCIe (once) and leaves a copy in an HBM staging block for the
 *     update kernel; one launch per scan.  Keep a scan's arrays untouched until its launch has run
 *     (fdm_engine_sync(), or an event recorded on fdm_engine_stream()).
 *   pageable arrays: copied into a rotating staging block with hipMemcpyAsync (which the runtime
 *     stages, i.e. the call blocks for the copy), scan enqueued behind the copies.
 * The synchronous host entry points (fdm_engine_integrate, fdm_engine_update) make the same choice.
 * fdm_engine_set_option(e, "zero_copy", max_points) bounds the in-place path (0 = always copy). */
int fdm_engine_integrate_async(fdm_engine* e, uint64_t n, const float* x, const float* y,
                               const float* z, const float* intensity, const uint32_t* rgb,
                               const float* sigma_z2, const double T_base_sensor[16],
                               const double T_world_base[16]);

/* ElevationMapping::update(cloud, robot_position) — elevation_mapping.cpp:110-125 —
 * for a cloud already in the map frame (tests/test_dual_layer.cpp:71).  z_var NULL
 * == cloud without covariance channel (pt_z_var = 0, elevation_mapping.cpp:57-60). */
int fdm_engine_update(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                      const float* z_var, const float* intensity, const uint32_t* rgb,
                      double robot_x, double robot_y, fdm_scan_stats* out);
int fdm_engine_update_device(fdm_engine* e, uint64_t n, const float* d_x, const float* d_y,
                             const float* d_z, const float* d_z_var, const float* d_intensity,
                             const uint32_t* d_rgb, double robot_x, double robot_y);

/* Launch a held-back map update now, without waiting for it (see fdm_engine_integrate_device). */
int fdm_engine_flush(fdm_engine* e);
/* The HIP stream (hipStream_t) the engine launches on: for callers that bracket engine work with
 * their own HIP events or order their own kernels after it. */
void* fdm_engine_stream(fdm_engine* e);
/* Which pipeline the last scan took: 0 = per-cell scratch (k_bin / k_update), 1 = per-tile record pools
 * (k_tbin / k_tupdate: maps of >= 240 tiles of 32 x 32 cells, scans of >= 2 K points), -1 = no scan yet.  Diagnostic. */
int fdm_engine_last_pipeline(fdm_engine* e);
/* Scans of the batch launch the last scan left in (fdm_engine_integrate_device_batch groups up to 16 small scans
 * per launch, see below), 0 when it took the single-scan path.  Diagnostic. */
int fdm_engine_last_batch(fdm_engine* e);
/* Device-side stopwatch on the engine's stream (two engine-owned HIP events): _start marks "everything enqueued so
 * far" (a held-back update is launched first), _stop launches the last scan's held-back update and marks its end,
 * _ms waits for the stop mark and returns the time between the two.  For callers without HIP of their own that
 * want the GPU's view of a run of enqueue-only calls (bench.py's `device_value`). */
int fdm_engine_timer_start(fdm_engine* e);
int fdm_engine_timer_stop(fdm_engine* e);
int fdm_engine_timer_ms(fdm_engine* e, float* ms);
/* Order a consumer behind the engine: launches a held-back update, then records `hip_event`
 * (hipEvent_t) on the engine's stream — everything enqueued so far, the map update of the last scan
 * included, is complete when the event fires (the reference's callers hold a shared_mutex around
 * integrate() for this, fastdem.hpp:48-53). */
int fdm_engine_record_event(fdm_engine* e, void* hip_event);
/* Order the engine behind a producer: work enqueued after this call waits for `hip_event`
 * (hipStreamWaitEvent), e.g. the kernel or copy that fills the next scan's device arrays. */
int fdm_engine_wait_event(fdm_engine* e, void* hip_event);
int fdm_engine_sync(fdm_engine* e);
/* Waits for the stream, returns the status (0/1/2) and stats of the last enqueued scan. */
int fdm_engine_last_stats(fdm_engine* e, fdm_scan_stats* out);

/* nanogrid::GridMap::move(Position) (elevation_mapping.cpp:113): rolling-window shift. */
int fdm_engine_move(fdm_engine* e, double x, double y);
int fdm_engine_get_geometry(fdm_engine* e, fdm_geometry* out);       /* getPosition/getStartIndex/... */
int fdm_engine_set_position(fdm_engine* e, double x, double y);      /* GridMap::setPosition */
int fdm_engine_set_start_index(fdm_engine* e, int32_t row, int32_t col);

/* GridMap::getLayers / exists / add / get / clear / clearAll (FastDEM::reset = clear(NULL)). */
int fdm_engine_num_layers(fdm_engine* e);
const char* fdm_engine_layer_name(fdm_engine* e, int i);
int fdm_engine_layer_exists(fdm_engine* e, const char* name);
int fdm_engine_layer_add(fdm_engine* e, const char* name, float value);
int fdm_engine_layer_download(fdm_engine* e, const char* name, float* host, int32_t rows, int32_t cols);
int fdm_engine_layer_upload(fdm_engine* e, const char* name, const float* host, int32_t rows, int32_t cols);
float* fdm_engine_layer_device_ptr(fdm_engine* e, const char* name); /* NULL if absent or if the layer is
                                                                      a field of the packed cell records */
int fdm_engine_clear(fdm_engine* e, const char* name /* NULL = clearAll */);
/* Copy of a map, layer by layer, device to device (ElevationMap's copy constructor / snapshot(), elevation_map.hpp:95-99;
 * the ROS node's copy under a shared lock, ros1/src/fastdem_ros_node.cpp:192-199): layer `name` of `src` into `dst`
 * (created there if missing).  Both engines on one device with the same stored window. */
int fdm_engine_layer_copy(fdm_engine* dst, fdm_engine* src, const char* name);

/* Halo exchange support for spatial tiling: pack the rectangle [r0,r0+nr) x [c0,c0+nc)
 * (storage-local indices) of `n_layers` named layers into a contiguous device buffer
 * (layer-major, each rectangle column-major), or write such a buffer back.  The RCCL
 * send/recv between neighbouring ranks is done by the caller (fastdem_amd/tiling.py). */
int fdm_engine_region_pack(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                           const char* const* names, int n_layers, float* d_buf);
int fdm_engine_region_unpack(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                             const char* const* names, int n_layers, const float* d_buf);
/* The same for several rectangles at once — ONE launch for all the strips and layers of a halo exchange (up to 8
 * rectangles x 24 layers per launch): rectangle q's block starts at float `offset` of d_buf and is laid out as
 * above (layer-major over the n_layers, column-major inside). */
typedef struct fdm_region {
  int32_t r0, c0, nr, nc;
  uint64_t offset;
} fdm_region;
int fdm_engine_regions_pack(fdm_engine* e, int32_t n_rects, const fdm_region* rects, const char* const* names,
                            int n_layers, float* d_buf);
int fdm_engine_regions_unpack(fdm_engine* e, int32_t n_rects, const fdm_region* rects, const char* const* names,
                              int n_layers, const float* d_buf);

/* ---- Scan routing for a spatially tiled global map (fastdem_amd/tiling.py, include/fdm_halo.h) ----
 * One logical scan = the concatenation, in rank order, of per-rank slices.  fdm_engine_route_scan runs a slice through
 * T_base_sensor, cropRange, cropZ, T_world_base and getIndex against the GLOBAL geometry — the operations the owner's
 * bin kernel repeats — and partitions the slice's RAW points by the rank that owns the cell they fall into, order
 * preserved: d_send receives {x, y, z, intensity} records (16 B), the points for rank 0 first, then rank 1, ...;
 * d_counts (device, world + 2 words) receives the points per owner, then the slice's n_after_filter and n_in_map.
 * Enqueue-only on the engine's stream; the host reads d_counts to size the exchange (fdm_halo_route_exchange).
 * The plan: owned rect of rank i * grid_cols + j = rows [row_edge[i], row_edge[i+1]) x cols [col_edge[j], col_edge[j+1])
 * (fdm_tile_plan_route).  GLOBAL mode only.
 *
 * fdm_engine_integrate_points4_device: FastDEM::integrate of the points an owner received ({x, y, z, intensity}
 * records in HBM, rank order = scan order).  any_in_map: some point of the LOGICAL scan landed in the map (the OR over
 * all slices' n_in_map) — it gates the obstacle-layer clear on every tile, also one that received no point. */
typedef struct fdm_route_plan {
  int32_t world, grid_rows, grid_cols, pad;
  int32_t row_edge[17], col_edge[17];
} fdm_route_plan;
int fdm_engine_route_scan(fdm_engine* e, const fdm_route_plan* plan, uint64_t n, const float* d_x, const float* d_y,
                          const float* d_z, const float* d_intensity, const double T_base_sensor[16],
                          const double T_world_base[16], float* d_send, uint32_t* d_counts);
int fdm_engine_integrate_points4_device(fdm_engine* e, uint64_t n, const float* d_points4, int has_intensity,
                                        int any_in_map, const double T_base_sensor[16],
                                        const double T_world_base[16]);
/* The same two halves with the shares laid out as CHANNEL BLOCKS (N scans, one per rank: every source's share is
 * integrated as a scan of its own, so nothing has to be contiguous across sources): the share of owner d starts at
 * float 4 * base_d of d_send, base_d = sum of pad4(count) over the lower owners, and holds x | y | z | intensity, each
 * pad4(count_d) floats.  A share is what the bin kernels read in place — no de-interleave pass on the receiving side,
 * the rank's own share is never copied.  d_send must hold 4 * (n + 3 * world) floats. */
int fdm_engine_route_scan_soa(fdm_engine* e, const fdm_route_plan* plan, uint64_t n, const float* d_x, const float* d_y,
                              const float* d_z, const float* d_intensity, const double T_base_sensor[16],
                              const double T_world_base[16], float* d_send, uint32_t* d_counts);
int fdm_engine_integrate_soa4_device(fdm_engine* e, uint64_t n, const float* d_share, int has_intensity,
                                     int any_in_map, const double T_base_sensor[16], const double T_world_base[16]);

/* Scan callbacks of the reference (fastdem.hpp:129-136, fastdem.cpp:139-150): when enabled the
 * kernels also keep (a) every point in the map frame + whether it survived the crops and (b) the
 * min-z observation of every observed cell.
 *   last_preprocessed: the cloud handed to onScanPreprocessed — points that survived cropRange and
 *     cropZ, in input order, map frame, with sigma_z2 = cov(2,2) (nullable).
 *   last_rasterized:   the cloud handed to onScanRasterized — one point per observed cell at the
 *     cell centre (GridMap::getPosition) with z = min_z (fastdem.cpp:200-214); order unspecified,
 *     as in the reference (hash-map iteration order).
 * Both return the number of points through n_out; `cap` is the capacity of the output arrays.
 * preprocessed == 2 also keeps the cloud's covariance channel (the reference's preprocessed cloud carries the
 * rotated 3x3 covariance of every point, nanopcl/core/point_cloud.hpp:126-147). */
int fdm_engine_capture(fdm_engine* e, int preprocessed, int rasterized);
/* The covariance channel of the preprocessed cloud: R * Sigma_sensor * R^T per surviving point
 * (fastdem.cpp:182-187), 9 floats per point, column-major like Eigen::Matrix3f, same order as
 * fdm_engine_last_preprocessed.  Needs fdm_engine_capture(e, 2, ..). */
int fdm_engine_last_preprocessed_cov(fdm_engine* e, uint64_t cap, float* cov9, uint64_t* n_out);
int fdm_engine_last_preprocessed(fdm_engine* e, uint64_t cap, float* x, float* y, float* z,
                                 float* sigma_z2, uint64_t* n_out);
int fdm_engine_last_rasterized(fdm_engine* e, uint64_t cap, float* x, float* y, float* z,
                               uint64_t* n_out);

/* ---- Raycasting stage ----
 * With cfg.raycast_enabled the integrate entry points also run step 3 of integrateImpl
 * (fastdem.cpp:152-159) on the device: sensor origin = (T_world_base*T_base_sensor).translation(),
 * voxelGrid(points, resolution, VoxelMode::ANY), applyRaycasting.  Layers `ghost_removal`,
 * `raycasting`, `_visibility_logodds` appear as in raycasting.cpp:223-226.  A resolution outside
 * voxelGrid's [0.001, 100] range makes integrate return FDM_ERR_INVALID (the reference throws,
 * voxel_grid_impl.hpp:31-33).
 *
 * fastdem::applyRaycasting(map, scan, sensor_origin, config) called directly
 * (postprocess/raycasting.hpp:47-49; tests/test_postprocess.cpp:73-190): `scan` is used as given,
 * no voxel filter.  rc == NULL takes the raycasting fields of the engine's fdm_config.  A no-op
 * when raycasting is disabled in that config, n == 0, or the sensor origin lies outside the map.
 * Host arrays, synchronous / device arrays, enqueue-only. */
int fdm_engine_apply_raycasting(fdm_engine* e, uint64_t n, const float* x, const float* y,
                                const float* z, const float sensor_origin[3],
                                const fdm_raycast_config* rc);
int fdm_engine_apply_raycasting_device(fdm_engine* e, uint64_t n, const float* d_x, const float* d_y,
                                       const float* d_z, const float sensor_origin[3],
                                       const fdm_raycast_config* rc);
/* nanopcl::filters::voxelGrid(cloud, voxel_size, VoxelMode::ANY) (voxel_grid_impl.hpp:30-60,171-189)
 * on the device: writes the ORIGINAL indices of the kept points, in the filter's output order
 * (ascending voxel key), to out_idx (capacity n) and their count to n_out.  Ties inside a voxel are
 * in original point order (the reference: whatever std::sort leaves — see DESIGN.md).
 * FDM_ERR_INVALID for a voxel_size outside [0.001, 100]. */
int fdm_engine_voxel_any(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                         float voxel_size, uint32_t* out_idx, uint64_t* n_out);
/* HIP-event duration of the last scan's raycasting stage (keys + sort + rays + resolve), ms;
 * needs fdm_engine_enable_profile. */
int fdm_engine_last_ray_ms(fdm_engine* e, float* ms);

/* ---- Map egress ----
 * fastdem::detail::toPointCloud2Impl (bridge/ros/impl.hpp:28-166) on the device: one record per cell
 * with a finite `elevation_layer` value, in the reference's visiting order (column by column through
 * the submap, starting at sub_start), each record = x, y, z, every non-internal layer (name not
 * starting with '_', elevation_map.hpp:42-45) in getLayers() order, then the packed colour as `rgb`.
 * All fields are 4 bytes (FLOAT32): point_step = 4 * n_fields.
 *   sub_rows < 0 : the whole map (sub_start = start index, sub_size = size), impl.hpp:160-166
 *   fields_buf   : receives the field names separated by '\n' (nullable)
 *   host_out     : receives n_points * point_step bytes when cap_bytes allows; pass NULL (or a
 *                  too small cap) to learn n_points / point_step first
 * Compaction and packing run in HBM; the host sees one contiguous copy. */
int fdm_engine_pack_cloud(fdm_engine* e, const char* elevation_layer, int32_t sub_r0, int32_t sub_c0,
                          int32_t sub_rows, int32_t sub_cols, void* host_out, uint64_t cap_bytes,
                          uint64_t* n_points, uint32_t* point_step, char* fields_buf,
                          uint64_t fields_cap);
/* Same, the packed records stay in an engine-owned HBM buffer (valid until the next pack call). */
int fdm_engine_pack_cloud_device(fdm_engine* e, const char* elevation_layer, int32_t sub_r0,
                                 int32_t sub_c0, int32_t sub_rows, int32_t sub_cols, void** d_out,
                                 uint64_t* n_points, uint32_t* point_step);

/* ---- Ingest ----
 * nanopcl::from(sensor_msgs::PointCloud2) (nanopcl/bridge/ros/impl.hpp:174-246) on the device.
 * The layout carries the byte offsets the reference parses from msg.fields (impl.hpp:65-99):
 * x, y, z are FLOAT32 and required; intensity (datatype UINT8=2, UINT16=4, FLOAT32=7, FLOAT64=8,
 * anything else reads as 0, impl.hpp:104-118) and rgb / rgba (packed 0x00RRGGBB, impl.hpp:163-171)
 * are optional: offset -1 = absent.  ring / time / label / normals are not consumed by integrate(). */
typedef struct fdm_cloud2_layout {
  uint32_t point_step;           /* msg.point_step */
  int32_t off_x, off_y, off_z;
  int32_t off_intensity, intensity_type;
  int32_t off_rgb;
} fdm_cloud2_layout;

/* from_impl: decode n_points records (msg.data, width*height of them) into the engine's SoA
 * channels in HBM, dropping points with a non-finite coordinate and keeping the order.
 * `data` is a host pointer (copied once) or, with data_on_device != 0, already in HBM.
 * n_valid receives the number of points kept (host sync). */
int fdm_engine_ingest_cloud2(fdm_engine* e, const void* data, int data_on_device, uint64_t n_points,
                             const fdm_cloud2_layout* layout, uint64_t* n_valid);
/* The channels the last ingest produced (device pointers, engine-owned, valid until the next
 * ingest): intensity / rgb are NULL when the message has no such field. */
int fdm_engine_ingested(fdm_engine* e, const float** d_x, const float** d_y, const float** d_z,
                        const float** d_intensity, const uint32_t** d_rgb, uint64_t* n);
/* from_impl + FastDEM::integrate(cloud, T_base_sensor, T_world_base) in one call (what the ROS
 * callback does, ros1/src/fastdem_ros_node.cpp:171-182).  Status as fdm_engine_integrate; a message
 * without x/y/z or whose points are all non-finite is an empty cloud (FDM_SKIP_EMPTY_CLOUD).
 * One decode kernel + the scan, no host round trip in between: the channels are written at the
 * message's indices, the bin kernel drops the non-finite points (the compaction from_impl does
 * changes indices, not decisions) and cloud.size() comes back with the scan statistics.  A message in
 * pinned memory (fdm_host_alloc) is decoded in place over PCIe; a pageable one is copied first. */
int fdm_engine_integrate_cloud2(fdm_engine* e, const void* data, int data_on_device, uint64_t n_points,
                                const fdm_cloud2_layout* layout, const double T_base_sensor[16],
                                const double T_world_base[16], fdm_scan_stats* out);

/* ---- Stencil post-processing ----
 * The reference's post-processing functions (called by the ROS timers on a snapshot of the map) on
 * the device-resident layers; each call only enqueues kernels.  Neighbourhoods are taken in logical
 * (unwrapped) coordinates and clipped at the map border; see DESIGN.md §7 f2 for the exact
 * neighbourhood definition (nanoGrid's region()/neighbors() are not on disk).  On a spatial tile the
 * stage is exact for the OWNED cells provided the halo ring is at least as wide as the stencil
 * reaches (inpainting: one cell per pass; median: kernel/2; the discs: radius/resolution) —
 * otherwise FDM_ERR_INVALID; halo cells must be refreshed from their owners afterwards
 * (fdm_engine_region_pack/_unpack).
 *   applyInpainting(map, max_iterations, min_valid_neighbors, inplace)        src/inpainting.cpp:21-67
 *   applySpatialSmoothing(map, layer, kernel_size, min_valid_neighbors)       postprocess/spatial_smoothing.hpp:38-67
 *   applyUncertaintyFusion(map, config::UncertaintyFusion)                    src/uncertainty_fusion.cpp:103-186
 *   applyFeatureExtraction(map, radius, min_valid, lower_pct, upper_pct)      src/feature_extraction.cpp:28-118
 * Any kernel size / radius the reference accepts is accepted: region(Size(k, k)) spans dr, dc in [-k/2, k/2] (for
 * an even k the (k + 1)-wide box); neighbourhoods beyond 256 cells (a 17 x 17 median, a 0.3 m disc
 * on a 0.02 m map) run through slower kernels whose per-cell lists live in a global pool and are insertion-sorted:
 * ~ entries^2 / 4 moves per cell.  A call whose neighbourhood would need more than ~1e13 moves over the map (minutes of
 * device time: 5 000 entries per cell at 1.44 M cells, 790 at 64 M) returns FDM_ERR_INVALID instead of hanging. */
typedef struct fdm_fusion_config {          /* config::UncertaintyFusion (config/postprocess.hpp:32-39) */
  int32_t enabled;
  float search_radius, spatial_sigma, quantile_lower, quantile_upper;
  int32_t min_valid_neighbors;
} fdm_fusion_config;
int fdm_engine_apply_inpainting(fdm_engine* e, int max_iterations, int min_valid_neighbors, int inplace);
int fdm_engine_apply_spatial_smoothing(fdm_engine* e, const char* layer, int kernel_size,
                                       int min_valid_neighbors);
int fdm_engine_apply_uncertainty_fusion(fdm_engine* e, const fdm_fusion_config* cfg);
int fdm_engine_apply_feature_extraction(fdm_engine* e, float analysis_radius, int min_valid_neighbors,
                                        float step_lower_percentile, float step_upper_percentile);

/* Pinned host memory for input clouds (the arrays fdm_engine_integrate* read in place, see
 * fdm_engine_integrate_async).  Blocks come from a process-wide pool of hipHostMalloc'ed memory in
 * power-of-two size classes — a cloud allocated per sensor message costs a free-list pop, not a
 * driver call — and go back to the pool on fdm_host_free; fdm_host_trim() returns the pool's idle
 * blocks to the system.  Thread-safe.  Without a usable GPU the block is ordinary (pageable) memory,
 * which the engine then copies instead of reading in place; fdm_host_is_pinned tells which.
 * (Replaces the std::vector storage behind nanopcl::PointCloud, point_cloud.hpp:126-134.) */
void* fdm_host_alloc(uint64_t bytes);
void fdm_host_free(void* p);
void fdm_host_trim(void);
int fdm_host_is_pinned(const void* p);

/* Parity / measurement instrumentation (not in the reference). */
int fdm_engine_enable_cell_ids(fdm_engine* e, int on);
/* per input point of the last scan: linear cell id (col*rows+row), -1 cropped, -2 outside map */
int fdm_engine_last_cell_ids(fdm_engine* e, int32_t* host_out, uint64_t n);
int fdm_engine_enable_profile(fdm_engine* e, int on);
/* HIP-event durations of the last scan's launches: ms[0] = bin kernel, ms[1] = update kernel.
 * When the update was held back (see fdm_engine_integrate_device) ms[0] is the fused launch
 * (previous scan's update + this scan's bin) and ms[1] ~ 0. */
int fdm_engine_last_kernel_ms(fdm_engine* e, float* ms2);

/* Tuning knobs for A/B measurements (bench.py); unknown keys are an error.
 *   "wave_merge"  0/1   : k_bin merges same-cell runs inside the wavefront before the atomics
 *   "bin_variant" 0/1/4 : bin kernel by scan size (0), one point per thread (1), LDS-staged (4)
 *   "dense"       0/1   : update sweep visits every tile (1) or only stamped tiles (0)
 *   "records"     0/1   : estimator state packed into per-cell records (1) or one array per layer (0)
 *   "bin_table"   0/1   : k_bin folds poorly merged waves into a per-block LDS cell table before the atomics
 *   "zero_copy"   n     : host entry points read PINNED input arrays of up to n points in place (0 = always copy)
 *   "overlap"     0/1   : hold the update of a small scan back and fuse it with the next scan's bin launch
 *   "voxel_small" 0/1, "voxel_small_max" n : raycasting's voxel filter without a sort for scans of up to n points
 *                         (default on, 65536; fdm_raycast.hpp k_vs_*); 0 = every scan through the stable radix sort
 *   "batch" 0/1, "batch_max" n, "batch_crop" 0/1, "batch_fuse" 0/1 : fdm_engine_integrate_device_batch groups small
 *                         scans into batch launches (default on, 16 per launch); evaluate the next batch's crops
 *                         one launch ahead; hold a batch's update back for the next batch's launch
 *   "sync_spin_us" n     : a synchronous call polls the pinned statistics block for up to n microseconds before it
 *                         falls back to a stream wait (default 150; 0 = wait at once)
 *   "tiled" 0/1, "tiled_min" n : large-scan pipeline (per-tile record pools) on/off, its point-count threshold
 *   "upd_blocks" n, "upd_blocks_alone" n : update blocks (four tile wavefronts each) of a fused large-scan launch
 *                         (default 768) / of an update launch of its own (2048); every wavefront walks its share of the
 *                         16 x 16-cell tiles
 *   "upd_prio" 0/1      : the update wavefronts raise their issue priority (default 1)
 *   "bin_stagger" n     : fused large-scan launch: start stagger of the first-round bin blocks, n x 512 cycles per
 *                         resident slot (default 0)
 *   "tiled_lds_pad" n   : extra dynamic LDS per block of the large-scan launches, bytes (default 4096: six blocks
 *                         per CU instead of seven)
 *   "cnt_shift" 0..5    : one tile counter per 2^n words of the counter array (default 5 = one per 128 bytes:
 *                         memory-side atomics on one line queue up); before the first large scan only
 *   "batch_walk" -1/0/1 : small-scan batches: the chain of moves walked one launch ahead (-1 = for the quantile
 *                         estimator only)
 *   "ray_large_min" n   : raycasting of scans from n points up takes the large-scan path (default 196608): ray queue
 *                         ordered by (angular sector, length class)
 *   "ray_wedge" 0/1     : ... and walks it with the sector's minimum-height window in LDS (fdm_raywedge.hpp, default 1;
 *                         0 = one lane per ray on memory-side atomics)
 *   "ray_overlap" -1/0/1: the stage's map-independent part (voxel filter, queue, walk) of a large scan leaves on a stream of its
 *                         own as soon as the scan's bin half has run, its resolve stays behind the scan's update (0, default:
 *                         never; 1 whenever possible; -1 for synchronous calls and scans of >= 1 M points).  Worth 10-14 % in a
 *                         process with few active streams, a loss in one with many
 *   "ray_wedge_parts" n : workgroups per sector of that walk (0 = by the scan's size, the default; 1 .. 16: measurement)
 *   "ray_hold" 0/1      : a scan's raycasting stage is held back together with its map update and runs right behind it —
 *                         in the next scan's launch sequence (the update then shares a launch with that scan's bin half) or
 *                         at the next flush (default 1; 0 = update and stage at once, the round-1..4 order)
 *   "batch_ray" 0/1     : raycasting inside the small-scan batches (1); "batch_ray_lds" 0/1: its ray walk on LDS images
 *                         (1) or memory-side atomics with "batch_ray_seg" 1/4/8/16 lanes per ray
 *   "dbg_*"             : measurement-only switches used by scripts/ab_kernels.py */
int fdm_engine_set_option(fdm_engine* e, const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* FDM_ENGINE_H */
