// fdm_ref_capi.cpp — extern "C" surface of the CPU oracle.
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  (see fdm_ref.hpp header)
#include "fdm_ref.h"

#include <cstdio>
#include <cstring>

#include "fdm_ref.hpp"
#include "fdm_ref_ingest.hpp"
#include "fdm_ref_post.hpp"

using namespace fdmref;

static_assert(sizeof(fdmref_config) == sizeof(Config), "config layout must match");

namespace {
Config toConfig(const fdmref_config* c) {
  Config out;
  std::memcpy(static_cast<void*>(&out), c, sizeof(Config));
  return out;
}
Engine* E(void* e) { return static_cast<Engine*>(e); }

Cloud buildCloud(uint64_t n, const float* x, const float* y, const float* z, const float* intensity,
                 const uint32_t* rgb) {
  Cloud c;
  c.has_intensity = intensity != nullptr;
  c.has_color = rgb != nullptr;
  c.resize(n);
  for (uint64_t i = 0; i < n; ++i) {
    c.pts[i] = {x[i], y[i], z[i], 1.0f};
    if (intensity) c.intensity[i] = intensity[i];
    if (rgb) c.color[i] = {uint8_t(rgb[i] >> 16), uint8_t(rgb[i] >> 8), uint8_t(rgb[i])};
  }
  return c;
}
void copyStats(const ScanStats& s, fdmref_stats* out) {
  if (!out) return;
  out->n_input = s.n_input;
  out->n_after_filter = s.n_after_filter;
  out->n_in_map = s.n_in_map;
  out->n_cells_touched = s.n_cells_touched;
  out->shift_rows = s.shift_rows;
  out->shift_cols = s.shift_cols;
}
}  // namespace

extern "C" {

void fdmref_default_config(fdmref_config* cfg) {
  Config d;
  std::memcpy(cfg, &d, sizeof(Config));
}

void* fdmref_create(float width, float height, float resolution, const fdmref_config* cfg) {
  return new Engine(width, height, resolution, toConfig(cfg));
}
void fdmref_destroy(void* e) { delete E(e); }
void fdmref_set_config(void* e, const fdmref_config* cfg) { E(e)->setConfig(toConfig(cfg)); }
void fdmref_reset(void* e) { E(e)->reset(); }
void fdmref_track_ids(void* e, int on) { E(e)->track_ids = on != 0; }

int fdmref_integrate(void* e, uint64_t n, const float* x, const float* y, const float* z,
                     const float* intensity, const uint32_t* rgb, const double* T_bs,
                     const double* T_wb, fdmref_stats* out) {
  Cloud c = buildCloud(n, x, y, z, intensity, rgb);
  ScanStats s;
  Status st;
  try {
    st = E(e)->integrate(c, T_bs, T_wb, &s);
  } catch (const std::invalid_argument&) {  // voxelGrid range check (voxel_grid_impl.hpp:31-33)
    copyStats(s, out);
    return -1;
  }
  copyStats(s, out);
  return st;
}

int fdmref_update(void* e, uint64_t n, const float* x, const float* y, const float* z,
                  const float* z_var, const float* intensity, const uint32_t* rgb, double robot_x,
                  double robot_y, fdmref_stats* out) {
  Cloud c = buildCloud(n, x, y, z, intensity, rgb);
  if (z_var) {
    c.has_cov = true;
    c.cov.assign(n, Mat3f{});
    for (uint64_t i = 0; i < n; ++i) M3(c.cov[i], 2, 2) = z_var[i];
  }
  Engine* en = E(e);
  if (en->track_ids) {
    en->last_cell_ids.assign(n, -1);
    c.track_orig = true;
    c.orig.resize(n);
    for (uint64_t i = 0; i < n; ++i) c.orig[i] = uint32_t(i);
  }
  ScanStats s;
  s.n_input = s.n_after_filter = uint32_t(n);
  en->update(c, robot_x, robot_y, &s);
  copyStats(s, out);
  return OK;
}

double fdmref_time_integrate(void* e, uint64_t n, const float* x, const float* y, const float* z,
                             const float* intensity, const uint32_t* rgb, const double* T_bs,
                             const double* T_wb_seq, int n_poses, int iters, double* stage_seconds) {
  Engine* en = E(e);
  Cloud c = buildCloud(n, x, y, z, intensity, rgb);
  const bool was_tracking = en->track_ids;
  en->track_ids = false;
  en->time_stages = stage_seconds != nullptr;
  en->times = StageTimes{};
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0; it < iters; ++it) {
    ScanStats s;
    en->integrate(c, T_bs, T_wb_seq + 16 * (it % n_poses), &s);
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (stage_seconds) {
    stage_seconds[0] = en->times.sensor_cov;
    stage_seconds[1] = en->times.transform_filter;
    stage_seconds[2] = en->times.cov_transform;
    stage_seconds[3] = en->times.rasterize;
    stage_seconds[4] = en->times.map_update;
  }
  en->time_stages = false;
  en->track_ids = was_tracking;
  return dt;
}

int fdmref_move(void* e, double x, double y, int32_t* shift2) {
  int sh[2] = {0, 0};
  const bool moved = E(e)->map().move(x, y, sh);
  if (shift2) {
    shift2[0] = sh[0];
    shift2[1] = sh[1];
  }
  return moved ? 1 : 0;
}

void fdmref_get_geometry(void* e, fdmref_geometry* g) {
  const Grid& m = E(e)->map();
  g->length_x = m.length()[0];
  g->length_y = m.length()[1];
  g->resolution = m.resolution();
  g->position_x = m.position()[0];
  g->position_y = m.position()[1];
  g->rows = m.rows();
  g->cols = m.cols();
  g->start_row = m.startIndex()[0];
  g->start_col = m.startIndex()[1];
}
void fdmref_set_position(void* e, double x, double y) { E(e)->map().setPosition(x, y); }
void fdmref_set_start_index(void* e, int r, int c) { E(e)->map().setStartIndex(r, c); }

int fdmref_get_index(void* e, double x, double y, int32_t* rc2) {
  Index2 i;
  const bool ok = E(e)->map().getIndex(x, y, i);
  rc2[0] = i.r;
  rc2[1] = i.c;
  return ok ? 1 : 0;
}
int fdmref_get_position(void* e, int r, int c, double* xy2) {
  return E(e)->map().getPosition(Index2{r, c}, xy2[0], xy2[1]) ? 1 : 0;
}

int fdmref_num_layers(void* e) { return int(E(e)->map().layers().size()); }
const char* fdmref_layer_name(void* e, int i) { return E(e)->map().layers()[size_t(i)].c_str(); }
int fdmref_layer_exists(void* e, const char* name) { return E(e)->map().exists(name) ? 1 : 0; }
int fdmref_layer_get(void* e, const char* name, float* out) {
  Grid& m = E(e)->map();
  if (!m.exists(name)) return -1;
  const auto& v = m.get(name);
  std::memcpy(out, v.data(), v.size() * sizeof(float));
  return 0;
}
int fdmref_layer_set(void* e, const char* name, const float* in) {
  Grid& m = E(e)->map();
  if (!m.exists(name)) m.add(name);
  auto& v = m.get(name);
  std::memcpy(v.data(), in, v.size() * sizeof(float));
  return 0;
}
int fdmref_layer_add(void* e, const char* name, float value) {
  E(e)->map().add(name, value);
  return 0;
}
int fdmref_clear(void* e, const char* name) {
  Grid& m = E(e)->map();
  if (!name) {
    m.clearAll();
    return 0;
  }
  if (!m.exists(name)) return -1;
  m.clear(name);
  return 0;
}
int fdmref_last_cell_ids(void* e, int32_t* out, uint64_t n) {
  const auto& v = E(e)->last_cell_ids;
  if (v.size() != n) return -1;
  std::memcpy(out, v.data(), n * sizeof(int32_t));
  return 0;
}

void fdmref_keep_scan(void* e, int on) { E(e)->keep_scan = on != 0; }
uint64_t fdmref_last_preprocessed(void* e, uint64_t cap, float* x, float* y, float* z, float* var) {
  const Cloud& c = E(e)->last_preprocessed;
  for (uint64_t i = 0; i < c.size() && i < cap; ++i) {
    x[i] = c.pts[i][0];
    y[i] = c.pts[i][1];
    z[i] = c.pts[i][2];
    if (var) var[i] = M3(c.cov[i], 2, 2);
  }
  return c.size();
}
uint64_t fdmref_last_preprocessed_cov(void* e, uint64_t cap, float* cov9) {
  const Cloud& c = E(e)->last_preprocessed;
  for (uint64_t i = 0; i < c.size() && i < cap; ++i)
    for (int k = 0; k < 9; ++k) cov9[i * 9 + k] = c.cov[i][k];
  return c.size();
}
uint64_t fdmref_last_rasterized(void* e, uint64_t cap, float* x, float* y, float* z) {
  const auto& r = E(e)->last_rasterized;
  for (uint64_t i = 0; i < r.size() && i < cap; ++i) {
    x[i] = r[i][0];
    y[i] = r[i][1];
    z[i] = r[i][2];
  }
  return r.size();
}

int64_t fdmref_pack_cloud(void* e, const char* elevation_layer, int sub_r0, int sub_c0, int sub_rows,
                          int sub_cols, uint8_t* data, uint64_t cap_bytes, uint32_t* point_step,
                          char* fields_buf, uint64_t fields_cap) {
  Grid& m = E(e)->map();
  if (!m.exists(elevation_layer)) return -1;
  if (sub_rows < 0) {
    sub_r0 = m.startIndex()[0]; sub_c0 = m.startIndex()[1];
    sub_rows = m.rows(); sub_cols = m.cols();
  }
  const PackedCloud pc = packCloud(m, elevation_layer, sub_r0, sub_c0, sub_rows, sub_cols);
  if (point_step) *point_step = pc.point_step;
  if (fields_buf && fields_cap) {
    std::string joined;
    for (size_t k = 0; k < pc.fields.size(); ++k) joined += (k ? "\n" : "") + pc.fields[k];
    std::snprintf(fields_buf, fields_cap, "%s", joined.c_str());
  }
  if (data && !pc.data.empty() && cap_bytes >= pc.data.size()) std::memcpy(data, pc.data.data(), pc.data.size());
  return int64_t(pc.n_points);
}

static Cloud2Layout toLayout(const fdmref_cloud2_layout* l) {
  Cloud2Layout o;
  o.point_step = l->point_step;
  o.off_x = l->off_x; o.off_y = l->off_y; o.off_z = l->off_z;
  o.off_intensity = l->off_intensity; o.intensity_type = l->intensity_type;
  o.off_rgb = l->off_rgb;
  return o;
}
uint64_t fdmref_from_cloud2(const void* data, uint64_t n_points, const fdmref_cloud2_layout* layout,
                            float* x, float* y, float* z, float* intensity, uint32_t* rgb) {
  const Cloud c = fromCloud2(static_cast<const uint8_t*>(data), n_points, toLayout(layout));
  for (size_t i = 0; i < c.size(); ++i) {
    if (x) x[i] = c.pts[i][0];
    if (y) y[i] = c.pts[i][1];
    if (z) z[i] = c.pts[i][2];
    if (intensity && c.has_intensity) intensity[i] = c.intensity[i];
    if (rgb && c.has_color) rgb[i] = packColor(c.color[i][0], c.color[i][1], c.color[i][2]);
  }
  return c.size();
}
int fdmref_integrate_cloud2(void* e, const void* data, uint64_t n_points,
                            const fdmref_cloud2_layout* layout, const double* T_bs, const double* T_wb,
                            fdmref_stats* out) {
  const Cloud c = fromCloud2(static_cast<const uint8_t*>(data), n_points, toLayout(layout));
  ScanStats s;
  const Status st = E(e)->integrate(c, T_bs, T_wb, &s);
  copyStats(s, out);
  return st;
}

void fdmref_apply_inpainting(void* e, int it, int mv, int inplace) {
  applyInpainting(E(e)->map(), it, mv, inplace != 0);
}
void fdmref_apply_spatial_smoothing(void* e, const char* layer, int k, int mv) {
  applySpatialSmoothing(E(e)->map(), layer, k, mv);
}
void fdmref_apply_uncertainty_fusion(void* e, int enabled, float r, float s, float ql, float qu, int mv) {
  FusionConfig c;
  c.enabled = enabled != 0; c.search_radius = r; c.spatial_sigma = s; c.quantile_lower = ql;
  c.quantile_upper = qu; c.min_valid_neighbors = mv;
  applyUncertaintyFusion(E(e)->map(), c);
}
// 0 = the platform's float libm (the reference as built here), 1 = correctly rounded trig (fdm_ref_post.hpp)
void fdmref_set_trig_mode(int mode) { fdmref::trig_mode() = mode ? 1 : 0; }
void fdmref_apply_feature_extraction(void* e, float r, int mv, float lo, float hi) {
  applyFeatureExtraction(E(e)->map(), r, mv, lo, hi);
}
void fdmref_eig3(const float* cov9, float* val3, float* vec9) {
  const Eig3 r = computeDirect3(cov9);
  std::memcpy(val3, r.val, sizeof(r.val));
  std::memcpy(vec9, r.vec, sizeof(r.vec));
}

void fdmref_set_voxel_stable(void* e, int on) { E(e)->voxel_stable = on != 0; }
void fdmref_set_move_clear_basic(void* e, int on) { E(e)->map().setMoveClearBasic(on != 0); }
static void copyRay(const RayStats& r, uint32_t* s5) {
  if (!s5) return;
  s5[0] = r.n_rays; s5[1] = r.n_observed; s5[2] = r.n_ray_cells; s5[3] = r.n_conflicts; s5[4] = r.n_cleared;
}
void fdmref_last_ray_stats(void* e, uint32_t* stats5) { copyRay(E(e)->last_ray, stats5); }
int fdmref_apply_raycasting(void* e, uint64_t n, const float* x, const float* y, const float* z,
                            const float* origin3, uint32_t* stats5) {
  std::vector<std::array<float, 4>> scan(n);
  for (uint64_t i = 0; i < n; ++i) scan[i] = {x[i], y[i], z[i], 1.0f};
  Engine* en = E(e);
  en->last_ray = applyRaycasting(en->map(), scan, origin3, en->config().raycasting());
  copyRay(en->last_ray, stats5);
  return 0;
}
int64_t fdmref_voxel_any(uint64_t n, const float* x, const float* y, const float* z, float voxel_size,
                         int stable, uint32_t* out_idx) {
  std::vector<std::array<float, 4>> pts(n);
  for (uint64_t i = 0; i < n; ++i) pts[i] = {x[i], y[i], z[i], 1.0f};
  try {
    const auto sel = voxelGridAny(pts, voxel_size, stable != 0);
    if (!sel.empty()) std::memcpy(out_idx, sel.data(), sel.size() * sizeof(uint32_t));  // (memcpy(_, nullptr, 0) is UB)
    return int64_t(sel.size());
  } catch (const std::invalid_argument&) {
    return -1;
  }
}
uint64_t fdmref_voxel_pack(float x, float y, float z, float inv) { return voxel::pack(x, y, z, inv); }
void fdmref_sensor_origin(const double* T_bs, const double* T_wb, float* out3) {
  sensorOrigin(T_wb, T_bs, out3);
}

void fdmref_sensor_covariance(const fdmref_config* cfg, const float* p3, float* cov9) {
  const Mat3f m = sensorCovariance(toConfig(cfg), p3);
  std::memcpy(cov9, m.data(), sizeof(float) * 9);
}

void fdmref_kalman_update(float min_var, float max_var, float q, float* s, float z, float var,
                          int compute_bounds) {
  KalmanCell c{s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]};
  kalmanUpdate(KalmanParams{min_var, max_var, q}, c, z, var);
  if (compute_bounds) kalmanBounds(c);
}

void fdmref_p2_update(const float* dn5, int marker, float max_count, float* s, float x,
                      int compute_bounds) {
  const P2Params p = P2Params::make(dn5, marker, max_count);
  P2Cell c{s[0], s[1], s[2], s[3], s[4], {}, {}};
  for (int k = 0; k < 5; ++k) {
    c.q[k] = s + 5 + k;
    c.n[k] = s + 10 + k;
  }
  p2Update(p, c, x);
  if (compute_bounds) p2Bounds(p, c);
}

float fdmref_sigma_z2(const fdmref_config* cfg, const float* p3, const double* T_bs,
                      const double* T_wb) {
  const Mat3f S = sensorCovariance(toConfig(cfg), p3);
  const Mat3f R = rotationOfProduct(T_wb, T_bs);
  return M3(rotateCovariance(R, S), 2, 2);
}

}  // extern "C"
