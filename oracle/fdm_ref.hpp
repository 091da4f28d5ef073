// fdm_ref.hpp — single-threaded CPU restatement of FastDEM::integrate().
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  (see fdm_grid.hpp header)
// It is the checker for the HIP engine and the timed "port" CPU baseline of
// bench.py.  It mirrors the reference's data structures and OPERATION ORDER
// (AoS xyz1 points, per-point 3x3 covariance, two float 4x4 transforms with the
// crops in between, unordered_map binning, one estimator update per touched
// cell) so that timing it is a fair stand-in for the reference, which cannot be
// built here (Eigen3 / yaml-cpp / spdlog / nanoGrid absent — SURVEY.md §8c).
//
// PARITY STATUS: pinned against every known-answer value the reference's own
// tests hold for this path (tests/test_oracle_reference_spec.py re-expresses
// fastdem/tests/test_{kalman_estimation,quantile_estimation,sensor_models,
// dual_layer,fastdem_integration,elevation_map}.cpp); index arithmetic is
// "parity unpinned" (nanoGrid absent — see fdm_grid.hpp).
//
// Float evaluation orders below restate Eigen 3.3/3.4 x86-64 SSE2 (no FMA:
// the reference builds plain Release, fastdem/CMakeLists.txt:4-11) and are
// NORMATIVE for this repo; build with -ffp-contract=off and no -ffast-math.
//
// Follows, in order:
//   fastdem/src/fastdem.cpp:122-190                      integrate / preprocessScan
//   fastdem/include/fastdem/sensors/{sensor_model,lidar_model,rgbd_model}.hpp
//   fastdem/lib/nanoPCL/include/nanopcl/core/transform.hpp:19-37,78-82
//   fastdem/lib/nanoPCL/include/nanopcl/filters/core.hpp:21-68
//   fastdem/lib/nanoPCL/include/nanopcl/filters/impl/crop_impl.hpp:79-96,167-178
//   fastdem/src/elevation_mapping.cpp                    (all)
//   fastdem/include/fastdem/mapping/{kalman,quantile}_estimation.hpp
//   fastdem/include/fastdem/elevation_map.hpp
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "fdm_grid.hpp"
#include "fdm_ref_raycast.hpp"
#include "fdm_ref_egress.hpp"

namespace fdmref {

// ---------------------------------------------------------------- config ----
// Field-for-field mirror of fastdem::Config for this path
// (config/fastdem.hpp:23-38, config/mapping.hpp:10-48, config/sensor_model.hpp:10-37).
enum SensorType : int32_t { SENSOR_CONSTANT = 0, SENSOR_LIDAR = 1, SENSOR_RGBD = 2 };
enum MappingMode : int32_t { MODE_LOCAL = 0, MODE_GLOBAL = 1 };
enum EstimationType : int32_t { EST_KALMAN = 0, EST_P2 = 1 };

struct Config {
  float z_min = -std::numeric_limits<float>::max();
  float z_max = std::numeric_limits<float>::max();
  float range_min = 0.0f;
  float range_max = std::numeric_limits<float>::max();
  int32_t sensor_type = SENSOR_LIDAR;
  float lidar_range_noise = 0.02f, lidar_angular_noise = 0.001f;
  float rgbd_normal_a = 0.001f, rgbd_normal_b = 0.002f, rgbd_normal_c = 0.4f,
        rgbd_lateral_factor = 0.001f;
  float constant_uncertainty = 0.03f;
  int32_t mode = MODE_LOCAL;
  int32_t estimation_type = EST_KALMAN;
  float kalman_min_variance = 0.0001f, kalman_max_variance = 0.01f,
        kalman_process_noise = 0.0f;
  float p2_dn[5] = {0.01f, 0.16f, 0.50f, 0.84f, 0.99f};
  int32_t p2_elevation_marker = 3;
  float p2_max_sample_count = 0.0f;
  // config::Raycasting (config/postprocess.hpp:16-23)
  int32_t raycast_enabled = 0;
  float rc_height_conflict_threshold = 0.05f, rc_log_odds_observed = 0.4f, rc_log_odds_ghost = 0.2f,
        rc_log_odds_max = 2.0f, rc_clear_threshold = -1.0f;
  RaycastConfig raycasting() const {
    RaycastConfig r;
    r.enabled = raycast_enabled != 0;
    r.height_conflict_threshold = rc_height_conflict_threshold;
    r.log_odds_observed = rc_log_odds_observed;
    r.log_odds_ghost = rc_log_odds_ghost;
    r.log_odds_max = rc_log_odds_max;
    r.clear_threshold = rc_clear_threshold;
    return r;
  }
};

// layer names (elevation_map.hpp:28-46, kalman_estimation.hpp:27-31,
// quantile_estimation.hpp:25-36)
namespace layer {
constexpr auto elevation = "elevation";
constexpr auto elevation_min = "elevation_min";
constexpr auto elevation_max = "elevation_max";
constexpr auto variance = "variance";
constexpr auto n_points = "n_points";
constexpr auto upper_bound = "upper_bound";
constexpr auto lower_bound = "lower_bound";
constexpr auto obstacle = "obstacle";
constexpr auto intensity = "intensity";
constexpr auto color = "color";
constexpr auto kalman_p = "_kalman_p";
constexpr auto sample_mean = "_sample_mean";
constexpr auto sample_m2 = "_sample_m2";
inline const char* p2_q(int k) {
  static const char* n[5] = {"_p2_q0", "_p2_q1", "_p2_q2", "_p2_q3", "_p2_q4"};
  return n[k];
}
inline const char* p2_n(int k) {
  static const char* n[5] = {"_p2_n0", "_p2_n1", "_p2_n2", "_p2_n3", "_p2_n4"};
  return n[k];
}
}  // namespace layer

// ------------------------------------------------------------ small math ----
using Mat3f = std::array<float, 9>;   // column-major: m[c*3+r]
using Mat4f = std::array<float, 16>;  // column-major: m[c*4+r]
inline float& M3(Mat3f& m, int r, int c) { return m[c * 3 + r]; }
inline float M3(const Mat3f& m, int r, int c) { return m[c * 3 + r]; }

// Eigen 3-term redux (redux_novec_unroller): a0 + (a1 + a2).
inline float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }
inline double sum3(double a0, double a1, double a2) { return a0 + (a1 + a2); }

// Isometry3d::matrix().cast<float>() — T given column-major double[16].
inline Mat4f castTransform(const double* T) {
  Mat4f m;
  for (int i = 0; i < 16; ++i) m[i] = static_cast<float>(T[i]);
  return m;
}

// R = (T_world_base * T_base_sensor).rotation().cast<float>() (fastdem.cpp:182-183).
// Isometry product: linear = L1 * L2 (3x3 double, coeff-based product).
inline Mat3f rotationOfProduct(const double* Twb, const double* Tbs) {
  Mat3f R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double v = sum3(Twb[0 * 4 + i] * Tbs[j * 4 + 0], Twb[1 * 4 + i] * Tbs[j * 4 + 1],
                            Twb[2 * 4 + i] * Tbs[j * 4 + 2]);
      M3(R, i, j) = static_cast<float>(v);
    }
  return R;
}

// Eigen Matrix4f * Vector4f, SSE packet path (etor_product_packet_impl):
// r = c0*x; r = c1*y + r; r = c2*z + r; r = c3*w + r  (separate mul / add).
inline void transformPoint(const Mat4f& T, float* p) {
  const float x = p[0], y = p[1], z = p[2], w = p[3];
  for (int r = 0; r < 4; ++r) {
    float acc = T[0 * 4 + r] * x;
    acc = T[1 * 4 + r] * y + acc;
    acc = T[2 * 4 + r] * z + acc;
    acc = T[3 * 4 + r] * w + acc;
    p[r] = acc;
  }
}

inline float squaredNorm3(const float* p) { return sum3(p[0] * p[0], p[1] * p[1], p[2] * p[2]); }

// ---------------------------------------------------------- sensor models ----
inline Mat3f scaledIdentity(float v) {
  Mat3f m{};
  M3(m, 0, 0) = v;  // Identity() * v : 1*v on the diagonal, 0*v off it
  M3(m, 1, 1) = v;
  M3(m, 2, 2) = v;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      if (r != c) M3(m, r, c) = 0.0f * v;
  return m;
}

// LiDARSensorModel::computeCovariance (lidar_model.hpp:64-89)
inline Mat3f lidarCovariance(const float* p, float range_noise, float angular_noise) {
  range_noise = std::abs(range_noise);      // ctor, lidar_model.hpp:58-62
  angular_noise = std::abs(angular_noise);
  const float dist_sq = squaredNorm3(p);
  if (dist_sq < 1e-6f) return scaledIdentity(0.01f);
  const float distance = std::sqrt(dist_sq);
  const float dir[3] = {p[0] / distance, p[1] / distance, p[2] / distance};
  const float var_radial = std::max(range_noise * range_noise, 1e-6f);
  const float var_lateral =
      std::max((distance * angular_noise) * (distance * angular_noise), 1e-6f);
  Mat3f cov = scaledIdentity(var_lateral);
  // cov += s * (dir * dir^T): Eigen rewrites "scalar * (A*B)" as "(scalar*A) * B"
  // (ProductEvaluators.h) and evaluates the outer product column by column into a
  // temporary: tmp(i,j) = dir[j] * (s*dir[i]); then cov += tmp.
  const float s = var_radial - var_lateral;
  const float t[3] = {s * dir[0], s * dir[1], s * dir[2]};
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) M3(cov, r, c) = M3(cov, r, c) + dir[c] * t[r];
  return cov;
}

// RGBDSensorModel::computeCovariance (rgbd_model.hpp:82-101)
inline Mat3f rgbdCovariance(const float* p, float a, float b, float c0, float k) {
  const float depth = p[2];
  if (depth <= 0.0f) return scaledIdentity(0.01f);
  const float diff = depth - c0;
  const float sigma_norm = a + b * diff * diff;
  const float var_norm = sigma_norm * sigma_norm;
  const float sigma_lat = k * depth;
  const float var_lat = sigma_lat * sigma_lat;
  Mat3f m{};
  M3(m, 0, 0) = var_lat;
  M3(m, 1, 1) = var_lat;
  M3(m, 2, 2) = var_norm;
  return m;
}

// ConstantUncertaintyModel (sensor_model.hpp:87-93)
inline Mat3f constantCovariance(float uncertainty) {
  return scaledIdentity(uncertainty * uncertainty);
}

inline Mat3f sensorCovariance(const Config& c, const float* p) {
  switch (c.sensor_type) {
    case SENSOR_RGBD:
      return rgbdCovariance(p, c.rgbd_normal_a, c.rgbd_normal_b, c.rgbd_normal_c,
                            c.rgbd_lateral_factor);
    case SENSOR_CONSTANT:
      return constantCovariance(c.constant_uncertainty);
    case SENSOR_LIDAR:
    default:  // unknown -> LiDAR (sensor_model.cpp:34-38)
      return lidarCovariance(p, c.lidar_range_noise, c.lidar_angular_noise);
  }
}

// cov = R * cov * R^T (fastdem.cpp:184-187): M = R*cov to a temporary, then M*R^T,
// every coefficient a 3-term dot a0b0 + (a1b1 + a2b2).
inline Mat3f rotateCovariance(const Mat3f& R, const Mat3f& S) {
  Mat3f M, out;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      M3(M, i, j) = sum3(M3(R, i, 0) * M3(S, 0, j), M3(R, i, 1) * M3(S, 1, j),
                         M3(R, i, 2) * M3(S, 2, j));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      M3(out, i, j) = sum3(M3(M, i, 0) * M3(R, j, 0), M3(M, i, 1) * M3(R, j, 1),
                           M3(M, i, 2) * M3(R, j, 2));
  return out;
}

// ------------------------------------------------------------- estimators ----
// Kalman::update + computeBounds on one cell (kalman_estimation.hpp:98-153).
struct KalmanParams {
  float min_variance = 0.0001f, max_variance = 0.01f, process_noise = 0.0f;
};
struct KalmanCell {  // references into 8 layers
  float &x, &P, &count, &sample_mean, &sample_var, &m2, &upper, &lower;
};
inline float clampf(float v, float lo, float hi) {  // std::clamp semantics
  return (v < lo) ? lo : (hi < v) ? hi : v;
}
inline void kalmanUpdate(const KalmanParams& k, KalmanCell c, float z, float meas_var) {
  const float R = (meas_var > 0.0f) ? meas_var : k.max_variance;
  if (std::isnan(c.x)) {
    c.x = z;
    c.P = R;
    c.count = 1.0f;
  } else {
    c.P += k.process_noise;
    const float K = c.P / (c.P + R);
    c.x = c.x + K * (z - c.x);
    c.P = (1.0f - K) * c.P;
    c.P = clampf(c.P, k.min_variance, k.max_variance);
    c.count += 1.0f;
  }
  if (std::isnan(c.sample_mean)) {
    c.sample_mean = z;
    c.sample_var = 0.0f;
    c.m2 = 0.0f;
  } else {
    const float delta = z - c.sample_mean;
    const float new_mean = c.sample_mean + (delta / c.count);
    const float delta2 = z - new_mean;
    c.m2 += delta * delta2;
    c.sample_var = (c.count > 1.0f) ? c.m2 / (c.count - 1.0f) : 0.0f;
    c.sample_mean = new_mean;
  }
}
inline void kalmanBounds(KalmanCell c) {
  const float v = c.sample_var;
  const float sigma = std::sqrt((0.0f < v) ? v : 0.0f);  // std::max(0.0f, v)
  c.upper = c.x + 2.0f * sigma;
  c.lower = c.x - 2.0f * sigma;
}

// P2Quantile (quantile_estimation.hpp:72-258)
struct P2Params {
  float dn[5] = {0.01f, 0.16f, 0.50f, 0.84f, 0.99f};
  int elevation_marker = 3;
  float max_sample_count = 0.0f;
  static P2Params make(const float* dn_in, int marker, float max_count) {
    P2Params p;
    p.elevation_marker = std::clamp(marker, 0, 4);
    p.max_sample_count = std::max(max_count, 0.0f);
    for (int i = 0; i < 5; ++i) p.dn[i] = std::clamp(dn_in[i], 0.0f, 1.0f);
    for (int i = 1; i < 5; ++i) p.dn[i] = std::max(p.dn[i], p.dn[i - 1]);
    return p;
  }
};
inline float p2Parabolic(const float* q, const float* n, int i, int sign) {
  const float d_right = n[i + 1] - n[i];
  const float d_left = n[i] - n[i - 1];
  const float d_span = n[i + 1] - n[i - 1];
  if (d_right == 0.0f || d_left == 0.0f || d_span == 0.0f) return q[i];
  const float s = static_cast<float>(sign);
  const float t1 = (d_left + s) * (q[i + 1] - q[i]) / d_right;
  const float t2 = (d_right - s) * (q[i] - q[i - 1]) / d_left;
  return q[i] + s * (t1 + t2) / d_span;
}
inline float p2Linear(const float* q, const float* n, int i, int sign) {
  const int j = i + sign;
  const float dn = n[j] - n[i];
  if (dn == 0.0f) return q[i];
  return q[i] + static_cast<float>(sign) * (q[j] - q[i]) / dn;
}
inline void p2Core(const P2Params& p, float* q, float* n, float& count, float x) {
  if (std::isnan(count) || count < 0.0f) count = 0.0f;
  if (count < 5.0f) {
    q[static_cast<int>(count)] = x;
    count += 1.0f;
    if (count >= 5.0f) {
      std::sort(q, q + 5);
      for (int i = 0; i < 5; ++i) n[i] = static_cast<float>(i);
    }
    return;
  }
  int k;
  if (x < q[0]) {
    q[0] = x;
    k = 0;
  } else if (x < q[1]) {
    k = 0;
  } else if (x < q[2]) {
    k = 1;
  } else if (x < q[3]) {
    k = 2;
  } else if (x <= q[4]) {
    k = 3;
  } else {
    q[4] = x;
    k = 3;
  }
  for (int i = k + 1; i < 5; ++i) n[i] += 1.0f;
  float n_prime[5];
  for (int i = 0; i < 5; ++i) n_prime[i] = p.dn[i] * count;
  count += 1.0f;
  if (p.max_sample_count > 0.0f && count > p.max_sample_count) {
    const float scale = p.max_sample_count / count;
    for (int i = 0; i < 5; ++i) n[i] *= scale;
    count = p.max_sample_count;
  }
  for (int i = 1; i < 4; ++i) {
    const float d = n_prime[i] - n[i];
    if ((d >= 1.0f && n[i + 1] - n[i] > 1.0f) || (d <= -1.0f && n[i - 1] - n[i] < -1.0f)) {
      const int sign = (d >= 0.0f) ? 1 : -1;
      const float q_new = p2Parabolic(q, n, i, sign);
      q[i] = (q[i - 1] < q_new && q_new < q[i + 1]) ? q_new : p2Linear(q, n, i, sign);
      n[i] += static_cast<float>(sign);
    }
  }
}
struct P2Cell {  // references into 15 layers
  float &elevation, &variance, &count, &upper, &lower;
  float* q[5];
  float* n[5];
};
inline void p2Update(const P2Params& p, P2Cell c, float x) {
  float q[5], n[5];
  for (int k = 0; k < 5; ++k) {
    q[k] = *c.q[k];
    n[k] = *c.n[k];
  }
  p2Core(p, q, n, c.count, x);
  for (int k = 0; k < 5; ++k) {
    *c.q[k] = q[k];
    *c.n[k] = n[k];
  }
  c.elevation = (c.count >= 5.0f) ? q[p.elevation_marker] : x;
}
inline void p2Bounds(const P2Params& p, P2Cell c) {
  c.elevation = *c.q[p.elevation_marker];
  const float sigma = (*c.q[3] - *c.q[1]) / 2.0f;
  c.variance = sigma * sigma;
  c.lower = *c.q[0];
  c.upper = *c.q[4];
}

// ------------------------------------------------------------ point cloud ----
// nanopcl::PointCloud subset (point_cloud.hpp:126-147): AoS xyz1 + channels.
struct Cloud {
  std::vector<std::array<float, 4>> pts;
  std::vector<float> intensity;
  std::vector<std::array<uint8_t, 3>> color;
  std::vector<Mat3f> cov;
  std::vector<uint32_t> orig;  // bookkeeping for parity tests only (not in the reference)
  bool has_intensity = false, has_color = false, has_cov = false, track_orig = false;
  size_t size() const { return pts.size(); }
  bool empty() const { return pts.empty(); }
  void resize(size_t n) {
    pts.resize(n);
    if (has_intensity) intensity.resize(n);
    if (has_color) color.resize(n);
    if (has_cov) cov.resize(n);
    if (track_orig) orig.resize(n);
  }
};

// filters::detail::filterInPlace (filters/core.hpp:21-68): stable compaction of all channels.
template <typename Pred>
inline void filterInPlace(Cloud& c, Pred pred) {
  if (c.empty()) return;
  const size_t n = c.size();
  size_t write = 0;
  for (size_t read = 0; read < n; ++read) {
    if (pred(read)) {
      if (write != read) {
        c.pts[write] = c.pts[read];
        if (c.has_intensity) c.intensity[write] = c.intensity[read];
        if (c.has_color) c.color[write] = c.color[read];
        if (c.has_cov) c.cov[write] = c.cov[read];
        if (c.track_orig) c.orig[write] = c.orig[read];
      }
      ++write;
    }
  }
  c.resize(write);
}

// --------------------------------------------------------------- mapping ----
// ElevationMapping::CellObservation (elevation_mapping.hpp:26-34)
struct CellObservation {
  float min_z = std::numeric_limits<float>::max();
  float min_z_var = 0.0f;
  float max_z = std::numeric_limits<float>::lowest();
  float max_intensity = std::numeric_limits<float>::lowest();
  uint32_t color_packed = 0;  // bit pattern of the packed float
  bool has_intensity = false;
  bool has_color = false;
};
using CellObservations = std::unordered_map<Index2, CellObservation, IndexHash>;

enum Status : int32_t {
  OK = 0,
  SKIP_EMPTY_CLOUD = 1,   // fastdem.cpp:125-128
  SKIP_ALL_FILTERED = 2,  // fastdem.cpp:138
};

struct ScanStats {
  uint32_t n_input = 0, n_after_filter = 0, n_in_map = 0, n_cells_touched = 0;
  int32_t shift_rows = 0, shift_cols = 0;
};

struct StageTimes {  // the five stages of assets/fastdem_jetson_benchmark.svg
  double sensor_cov = 0, transform_filter = 0, cov_transform = 0, rasterize = 0, map_update = 0;
};

class Engine {
 public:
  Engine(float width, float height, float resolution, const Config& cfg) : cfg_(cfg) {
    // ElevationMap ctor (elevation_map.hpp:101-116): 3 default layers, setGeometry, clearAll
    map_.add(layer::elevation);
    map_.add(layer::elevation_min);
    map_.add(layer::elevation_max);
    map_.setGeometry(double(width), double(height), double(resolution));
    rebuildMapping();
  }

  Grid& map() { return map_; }
  const Config& config() const { return cfg_; }
  void setConfig(const Config& c) {  // fluent setters rebuild ElevationMapping (fastdem.cpp:28-38);
    cfg_ = c;                         // ensureLayers only adds what is missing, so this is idempotent
    rebuildMapping();
  }
  void reset() { map_.clearAll(); }  // fastdem.cpp:26

  bool track_ids = false;            // parity bookkeeping (off while timing)
  std::vector<int32_t> last_cell_ids;  // per INPUT point: linear id, -1 cropped, -2 outside map
  bool time_stages = false;
  StageTimes times;
  bool keep_scan = false;             // parity bookkeeping for the scan callbacks
  Cloud last_preprocessed;            // what on_preprocessed_ receives (fastdem.cpp:139-141)
  std::vector<std::array<float, 3>> last_rasterized;  // toPointCloud(obs) (fastdem.cpp:200-214)

  // FastDEM::integrate(cloud, T_base_sensor, T_world_base)  (fastdem.cpp:122-162)
  Status integrate(const Cloud& cloud, const double* T_bs, const double* T_wb, ScanStats* st) {
    ScanStats s;
    s.n_input = uint32_t(cloud.size());
    if (track_ids) last_cell_ids.assign(cloud.size(), -1);
    if (cloud.empty()) {
      if (st) *st = s;
      return SKIP_EMPTY_CLOUD;
    }
    Cloud points = preprocessScan(cloud, T_bs, T_wb);
    s.n_after_filter = uint32_t(points.size());
    if (points.empty()) {
      if (st) *st = s;
      return SKIP_ALL_FILTERED;
    }
    if (keep_scan) last_preprocessed = points;
    // robot_position = T_world_base.translation().head<2>()
    const auto obs = update(points, T_wb[12], T_wb[13], &s);
    if (keep_scan) {
      last_rasterized.clear();
      for (const auto& [index, cell] : obs) {  // FastDEM::toPointCloud
        double px = 0.0, py = 0.0;
        map_.getPosition(index, px, py);
        last_rasterized.push_back({float(px), float(py), cell.min_z});
      }
    }
    // 3. raycasting (fastdem.cpp:152-159)
    if (cfg_.raycast_enabled) {
      float origin[3];
      sensorOrigin(T_wb, T_bs, origin);
      const auto sel = voxelGridAny(points.pts, static_cast<float>(map_.resolution()), voxel_stable);
      std::vector<std::array<float, 4>> ray_scan(sel.size());
      for (size_t k = 0; k < sel.size(); ++k) ray_scan[k] = points.pts[sel[k]];
      last_ray = applyRaycasting(map_, ray_scan, origin, cfg_.raycasting());
    }
    if (st) *st = s;
    return OK;
  }

  bool voxel_stable = true;  // tie order inside a voxel for VoxelMode::ANY (see fdm_ref_raycast.hpp)
  RayStats last_ray;

  // ElevationMapping::update (elevation_mapping.cpp:110-125) — also the public entry
  // tests/test_dual_layer.cpp:71 drives directly (cloud in map frame, no covariance).
  CellObservations update(const Cloud& cloud, double robot_x, double robot_y, ScanStats* st) {
    auto t0 = now();
    if (cfg_.mode == MODE_LOCAL) {
      int sh[2] = {0, 0};
      map_.move(robot_x, robot_y, sh);
      if (st) {
        st->shift_rows = sh[0];
        st->shift_cols = sh[1];
      }
    }
    auto obs = rasterize(cloud, st);
    auto t1 = now();
    if (time_stages) times.rasterize += secs(t0, t1);
    if (obs.empty()) return obs;
    estimate(obs);
    updateMinMax(obs);
    updateObstacle(obs);
    if (cloud.has_intensity) updateIntensity(obs);
    if (cloud.has_color) updateColor(obs);
    if (time_stages) times.map_update += secs(t1, now());
    if (st) st->n_cells_touched = uint32_t(obs.size());
    return obs;
  }

  // preprocessScan (fastdem.cpp:164-190)
  Cloud preprocessScan(const Cloud& cloud, const double* T_bs, const double* T_wb) {
    auto t0 = now();
    // SensorModel::computeCovariances(PointCloud scan) takes the cloud BY VALUE (sensor_model.hpp:76-85)
    Cloud points = cloud;
    points.track_orig = track_ids;
    if (track_ids) {
      points.orig.resize(points.size());
      for (size_t i = 0; i < points.size(); ++i) points.orig[i] = uint32_t(i);
    }
    points.has_cov = true;
    points.cov.resize(points.size());
    for (size_t i = 0; i < points.size(); ++i)
      points.cov[i] = sensorCovariance(cfg_, points.pts[i].data());
    auto t1 = now();

    const Mat4f Tbs = castTransform(T_bs);
    for (auto& p : points.pts) transformPoint(Tbs, p.data());
    // cropRange (crop_impl.hpp:79-96): min_sq/max_sq in fp32; FLT_MAX^2 = +inf
    const float min_sq = cfg_.range_min * cfg_.range_min;
    const float max_sq = cfg_.range_max * cfg_.range_max;
    filterInPlace(points, [&](size_t i) {
      const float d2 = squaredNorm3(points.pts[i].data());
      return d2 >= min_sq && d2 <= max_sq;
    });
    // cropZ (crop_impl.hpp:167-178)
    filterInPlace(points, [&](size_t i) {
      const float v = points.pts[i][2];
      return v >= cfg_.z_min && v <= cfg_.z_max;
    });
    const Mat4f Twb = castTransform(T_wb);
    for (auto& p : points.pts) transformPoint(Twb, p.data());
    auto t2 = now();

    const Mat3f R = rotationOfProduct(T_wb, T_bs);
    for (auto& c : points.cov) c = rotateCovariance(R, c);
    if (time_stages) {
      times.sensor_cov += secs(t0, t1);
      times.transform_filter += secs(t1, t2);
      times.cov_transform += secs(t2, now());
    }
    return points;
  }

  // rasterize (elevation_mapping.cpp:41-92)
  CellObservations rasterize(const Cloud& cloud, ScanStats* st) {
    if (cloud.empty()) return {};
    CellObservations cells;
    cells.reserve(cloud.size());
    uint32_t n_in = 0;
    for (size_t i = 0; i < cloud.size(); ++i) {
      const auto& pt = cloud.pts[i];
      Index2 index;
      const bool inside = map_.getIndex(double(pt[0]), double(pt[1]), index);
      if (track_ids && cloud.track_orig)
        last_cell_ids[cloud.orig[i]] = inside ? index.c * map_.rows() + index.r : -2;
      if (!inside) continue;
      ++n_in;
      float pt_z_var = 0.0f;
      if (cloud.has_cov) pt_z_var = M3(cloud.cov[i], 2, 2);
      auto& cell = cells[index];
      const float z = pt[2];
      if (z < cell.min_z) {
        cell.min_z = z;
        cell.min_z_var = pt_z_var;
      }
      if (z > cell.max_z) cell.max_z = z;
      if (cloud.has_intensity) {
        const float val = cloud.intensity[i];
        if (!cell.has_intensity || val > cell.max_intensity) {
          cell.max_intensity = val;
          cell.has_intensity = true;
        }
      }
      if (cloud.has_color) {
        const auto& c = cloud.color[i];
        cell.color_packed = packColor(c[0], c[1], c[2]);
        cell.has_color = true;
      }
    }
    if (st) st->n_in_map = n_in;
    return cells;
  }

  // estimate (elevation_mapping.cpp:94-108)
  void estimate(const CellObservations& obs) {
    if (obs.empty()) return;
    if (cfg_.estimation_type == EST_P2) {
      const P2Params p = P2Params::make(cfg_.p2_dn, cfg_.p2_elevation_marker, cfg_.p2_max_sample_count);
      float* el = map_.get(layer::elevation).data();
      float* va = map_.get(layer::variance).data();
      float* cn = map_.get(layer::n_points).data();
      float* up = map_.get(layer::upper_bound).data();
      float* lo = map_.get(layer::lower_bound).data();
      float *q[5], *n[5];
      for (int k = 0; k < 5; ++k) {
        q[k] = map_.get(layer::p2_q(k)).data();
        n[k] = map_.get(layer::p2_n(k)).data();
      }
      const int R = map_.rows();
      for (const auto& [index, cell] : obs) {
        const size_t o = size_t(index.c) * R + index.r;
        P2Cell c{el[o], va[o], cn[o], up[o], lo[o], {}, {}};
        for (int k = 0; k < 5; ++k) {
          c.q[k] = q[k] + o;
          c.n[k] = n[k] + o;
        }
        p2Update(p, c, cell.min_z);
        p2Bounds(p, c);
      }
    } else {
      const KalmanParams k{cfg_.kalman_min_variance, cfg_.kalman_max_variance,
                           cfg_.kalman_process_noise};
      float* el = map_.get(layer::elevation).data();
      float* kp = map_.get(layer::kalman_p).data();
      float* cn = map_.get(layer::n_points).data();
      float* sm = map_.get(layer::sample_mean).data();
      float* va = map_.get(layer::variance).data();
      float* m2 = map_.get(layer::sample_m2).data();
      float* up = map_.get(layer::upper_bound).data();
      float* lo = map_.get(layer::lower_bound).data();
      const int R = map_.rows();
      for (const auto& [index, cell] : obs) {
        const size_t o = size_t(index.c) * R + index.r;
        KalmanCell c{el[o], kp[o], cn[o], sm[o], va[o], m2[o], up[o], lo[o]};
        kalmanUpdate(k, c, cell.min_z, cell.min_z_var);
        kalmanBounds(c);
      }
    }
  }

 private:
  using Clock = std::chrono::steady_clock;
  static Clock::time_point now() { return Clock::now(); }
  static double secs(Clock::time_point a, Clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  }

  // ElevationMapping ctor (elevation_mapping.cpp:11-39) + ensureLayers
  // (kalman_estimation.hpp:64-82, quantile_estimation.hpp:97-115): add only if missing.
  void rebuildMapping() {
    auto ensure = [&](const char* n, float v) {
      if (!map_.exists(n)) map_.add(n, v);
    };
    if (cfg_.estimation_type == EST_P2) {
      ensure(layer::variance, NAN);
      ensure(layer::n_points, 0.0f);
      for (int k = 0; k < 5; ++k) ensure(layer::p2_q(k), NAN);
      for (int k = 0; k < 5; ++k) ensure(layer::p2_n(k), float(k));
      ensure(layer::upper_bound, NAN);
      ensure(layer::lower_bound, NAN);
    } else {
      ensure(layer::variance, 0.0f);
      ensure(layer::n_points, 0.0f);
      ensure(layer::kalman_p, 0.0f);
      ensure(layer::sample_mean, NAN);
      ensure(layer::sample_m2, 0.0f);
      ensure(layer::upper_bound, NAN);
      ensure(layer::lower_bound, NAN);
    }
    ensure(layer::obstacle, NAN);
  }

  // updateMinMax / updateObstacle / updateIntensity / updateColor (elevation_mapping.cpp:127-175)
  void updateMinMax(const CellObservations& obs) {
    float* mn = map_.get(layer::elevation_min).data();
    float* mx = map_.get(layer::elevation_max).data();
    const int R = map_.rows();
    for (const auto& [index, cell] : obs) {
      const size_t o = size_t(index.c) * R + index.r;
      if (std::isnan(mn[o]) || cell.min_z < mn[o]) mn[o] = cell.min_z;
      if (std::isnan(mx[o]) || cell.max_z > mx[o]) mx[o] = cell.max_z;
    }
  }
  void updateObstacle(const CellObservations& obs) {
    map_.clear(layer::obstacle);
    float* ob = map_.get(layer::obstacle).data();
    const int R = map_.rows();
    for (const auto& [index, cell] : obs)
      ob[size_t(index.c) * R + index.r] = (cell.max_z > cell.min_z) ? cell.max_z : NAN;
  }
  void updateIntensity(const CellObservations& obs) {
    if (!map_.exists(layer::intensity)) map_.add(layer::intensity, NAN);
    float* in = map_.get(layer::intensity).data();
    const int R = map_.rows();
    for (const auto& [index, cell] : obs) {
      if (!cell.has_intensity) continue;
      float& stored = in[size_t(index.c) * R + index.r];
      if (std::isnan(stored) || cell.max_intensity > stored) stored = cell.max_intensity;
    }
  }
  void updateColor(const CellObservations& obs) {
    if (!map_.exists(layer::color)) map_.add(layer::color, NAN);
    float* co = map_.get(layer::color).data();
    const int R = map_.rows();
    for (const auto& [index, cell] : obs) {
      if (!cell.has_color) continue;
      std::memcpy(&co[size_t(index.c) * R + index.r], &cell.color_packed, 4);
    }
  }

  Grid map_;
  Config cfg_;
};

}  // namespace fdmref
