/* fdm_ref.h — C entry points of the CPU oracle (libfdm_ref.so).
 *
 * *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***
 * Callers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 * PARITY STATUS: algorithm pinned by the reference's known-answer tests
 * (tests/test_oracle_reference_spec.py); nanoGrid index arithmetic
 * "parity unpinned" (library absent from /root/reference — see fdm_grid.hpp).
 *
 * The struct layouts are deliberately identical to include/fdm_engine.h so the
 * same ctypes definitions drive both the checker and the engine.
 */
#ifndef FDM_REF_H
#define FDM_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fdmref_config {
  float z_min, z_max, range_min, range_max;            /* config/fastdem.hpp:23-28 */
  int32_t sensor_type;                                 /* 0 Constant, 1 LiDAR, 2 RGBD */
  float lidar_range_noise, lidar_angular_noise;        /* config/sensor_model.hpp:22-25 */
  float rgbd_normal_a, rgbd_normal_b, rgbd_normal_c, rgbd_lateral_factor;
  float constant_uncertainty;
  int32_t mode;                                        /* 0 LOCAL, 1 GLOBAL */
  int32_t estimation_type;                             /* 0 Kalman, 1 P2Quantile */
  float kalman_min_variance, kalman_max_variance, kalman_process_noise;
  float p2_dn[5];
  int32_t p2_elevation_marker;
  float p2_max_sample_count;
  int32_t raycast_enabled;                             /* config/postprocess.hpp:16-23 */
  float rc_height_conflict_threshold, rc_log_odds_observed, rc_log_odds_ghost, rc_log_odds_max,
      rc_clear_threshold;
} fdmref_config;

typedef struct fdmref_stats {
  uint32_t n_input, n_after_filter, n_in_map, n_cells_touched;
  int32_t shift_rows, shift_cols;
} fdmref_stats;

typedef struct fdmref_geometry {
  double length_x, length_y, resolution, position_x, position_y;
  int32_t rows, cols, start_row, start_col;
} fdmref_geometry;

void fdmref_default_config(fdmref_config* cfg);

void* fdmref_create(float width, float height, float resolution, const fdmref_config* cfg);
void fdmref_destroy(void* e);
void fdmref_set_config(void* e, const fdmref_config* cfg);
void fdmref_reset(void* e);                   /* FastDEM::reset -> clearAll */
void fdmref_track_ids(void* e, int on);       /* record per-input-point cell ids */

/* FastDEM::integrate(cloud, T_base_sensor, T_world_base); T are column-major double[16]
 * (Eigen Isometry3d::matrix().data()).  rgb = 0x00RRGGBB.  Returns 0 ok, 1 empty cloud,
 * 2 everything filtered (the reference's `false` cases). */
int fdmref_integrate(void* e, uint64_t n, const float* x, const float* y, const float* z,
                     const float* intensity, const uint32_t* rgb, const double* T_base_sensor,
                     const double* T_world_base, fdmref_stats* out);

/* ElevationMapping::update(cloud, robot_xy) on a cloud already in the map frame;
 * z_var may be NULL (cloud without covariance channel -> 0). */
int fdmref_update(void* e, uint64_t n, const float* x, const float* y, const float* z,
                  const float* z_var, const float* intensity, const uint32_t* rgb, double robot_x,
                  double robot_y, fdmref_stats* out);

/* Time `iters` integrate() calls on a prebuilt AoS cloud (AoS build excluded).
 * Poses: T_world_base[k] for k in [0, n_poses) cycled; returns seconds total.
 * stage_seconds (nullable) receives the 5 Jetson-figure stages accumulated. */
double fdmref_time_integrate(void* e, uint64_t n, const float* x, const float* y, const float* z,
                             const float* intensity, const uint32_t* rgb,
                             const double* T_base_sensor, const double* T_world_base_seq,
                             int n_poses, int iters, double* stage_seconds);

int fdmref_move(void* e, double x, double y, int32_t* shift2);
void fdmref_get_geometry(void* e, fdmref_geometry* g);
void fdmref_set_position(void* e, double x, double y);
void fdmref_set_start_index(void* e, int r, int c);
int fdmref_get_index(void* e, double x, double y, int32_t* rc2);    /* 1 inside */
int fdmref_get_position(void* e, int r, int c, double* xy2);        /* 1 valid */

int fdmref_num_layers(void* e);
const char* fdmref_layer_name(void* e, int i);
int fdmref_layer_exists(void* e, const char* name);
int fdmref_layer_get(void* e, const char* name, float* out);        /* rows*cols, col-major */
int fdmref_layer_set(void* e, const char* name, const float* in);   /* adds if missing */
int fdmref_layer_add(void* e, const char* name, float value);
int fdmref_clear(void* e, const char* name /* NULL = all */);
int fdmref_last_cell_ids(void* e, int32_t* out, uint64_t n);

/* scan callbacks: keep the preprocessed cloud / rasterized observations of the last integrate() */
void fdmref_keep_scan(void* e, int on);
uint64_t fdmref_last_preprocessed(void* e, uint64_t cap, float* x, float* y, float* z, float* var);
/* the covariance channel of that cloud, 9 floats per point, column-major */
uint64_t fdmref_last_preprocessed_cov(void* e, uint64_t cap, float* cov9);
uint64_t fdmref_last_rasterized(void* e, uint64_t cap, float* x, float* y, float* z);

/* raycasting stage (fdm_ref_raycast.hpp).  stats5 = {n_rays, n_observed, n_ray_cells, n_conflicts,
 * n_cleared}.  voxel tie order: 1 = by original index (default; what the engine reproduces),
 * 0 = std::sort on the key exactly as the reference. */
void fdmref_set_voxel_stable(void* e, int on);
/* GridMap::move(): 1 = the vacated strips clear the basic layers {elevation, elevation_min, elevation_max} only
 * (fdm_grid.hpp clearStrip; default 0 = every layer) */
void fdmref_set_move_clear_basic(void* e, int on);
void fdmref_last_ray_stats(void* e, uint32_t* stats5);
/* applyRaycasting(map, scan, sensor_origin, cfg) on a map-frame cloud (raycasting.cpp:204-249) */
int fdmref_apply_raycasting(void* e, uint64_t n, const float* x, const float* y, const float* z,
                            const float* origin3, uint32_t* stats5);
/* filters::voxelGrid(cloud, size, VoxelMode::ANY): original indices of the selected points in
 * output order; returns their count, -1 if voxel_size is outside [0.001, 100] (the reference throws) */
int64_t fdmref_voxel_any(uint64_t n, const float* x, const float* y, const float* z, float voxel_size,
                         int stable, uint32_t* out_idx);
/* voxel::pack (nanopcl/core/voxel.hpp:28-43) */
uint64_t fdmref_voxel_pack(float x, float y, float z, float inv_voxel_size);
void fdmref_sensor_origin(const double* T_base_sensor, const double* T_world_base, float* out3);

/* map -> PointCloud2 egress (fdm_ref_egress.hpp; bridge/ros/impl.hpp:28-166).  sub_rows < 0 = full map
 * (sub_start = start index, sub_size = size).  fields: '\n'-separated names into fields_buf.
 * Returns the number of points; data (nullable) receives n_points*point_step bytes if cap allows. */
int64_t fdmref_pack_cloud(void* e, const char* elevation_layer, int sub_r0, int sub_c0, int sub_rows,
                          int sub_cols, uint8_t* data, uint64_t cap_bytes, uint32_t* point_step,
                          char* fields_buf, uint64_t fields_cap);

/* PointCloud2 ingest (fdm_ref_ingest.hpp; nanopcl/bridge/ros/impl.hpp:174-246).  Offsets are byte
 * offsets inside one point record, -1 = field absent; intensity_type = PointField datatype code. */
typedef struct fdmref_cloud2_layout {
  uint32_t point_step;
  int32_t off_x, off_y, off_z;
  int32_t off_intensity, intensity_type;
  int32_t off_rgb;
} fdmref_cloud2_layout;
/* from_impl: writes the kept points' channels (nullable outputs, capacity n_points); returns their count */
uint64_t fdmref_from_cloud2(const void* data, uint64_t n_points, const fdmref_cloud2_layout* layout,
                            float* x, float* y, float* z, float* intensity, uint32_t* rgb);
/* from_impl + FastDEM::integrate */
int fdmref_integrate_cloud2(void* e, const void* data, uint64_t n_points,
                            const fdmref_cloud2_layout* layout, const double* T_base_sensor,
                            const double* T_world_base, fdmref_stats* out);

/* stencil post-processing (fdm_ref_post.hpp): inpainting.cpp:21-67, spatial_smoothing.hpp:38-67,
 * uncertainty_fusion.cpp:103-186, feature_extraction.cpp:28-118 */
void fdmref_apply_inpainting(void* e, int max_iterations, int min_valid_neighbors, int inplace);
void fdmref_apply_spatial_smoothing(void* e, const char* layer, int kernel_size, int min_valid_neighbors);
void fdmref_apply_uncertainty_fusion(void* e, int enabled, float search_radius, float spatial_sigma,
                                     float quantile_lower, float quantile_upper, int min_valid_neighbors);
/* process-wide: 0 = platform float libm (default, the reference as built here), 1 = correctly rounded trig */
void fdmref_set_trig_mode(int mode);
void fdmref_apply_feature_extraction(void* e, float analysis_radius, int min_valid_neighbors,
                                     float step_lower_percentile, float step_upper_percentile);
/* Eigen SelfAdjointEigenSolver<Matrix3f>::computeDirect restated: cov9 column-major -> val3 ascending, vec9 */
void fdmref_eig3(const float* cov9, float* val3, float* vec9);

/* unit-level entry points for the reference's known-answer tests */
void fdmref_sensor_covariance(const fdmref_config* cfg, const float* p3, float* cov9_colmajor);
/* state8 = {x, P, count, sample_mean, sample_var, m2, upper, lower} */
void fdmref_kalman_update(float min_var, float max_var, float q, float* state8, float z, float var,
                          int compute_bounds);
/* state15 = {elevation, variance, count, upper, lower, q0..q4, n0..n4} */
void fdmref_p2_update(const float* dn5, int marker, float max_count, float* state15, float x,
                      int compute_bounds);
float fdmref_sigma_z2(const fdmref_config* cfg, const float* p3, const double* T_base_sensor,
                      const double* T_world_base);

#ifdef __cplusplus
}
#endif
#endif
