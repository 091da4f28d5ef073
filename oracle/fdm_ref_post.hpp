// fdm_ref_post.hpp — CPU restatement of the stencil post-processing stages.
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  (same rules as fdm_ref.hpp)
//
// Follows
//   fastdem/src/inpainting.cpp:21-67                                 applyInpainting
//   fastdem/include/fastdem/postprocess/spatial_smoothing.hpp:38-67  applySpatialSmoothing
//   fastdem/src/uncertainty_fusion.cpp:28-186                        SimpleWeightedECDF, applyUncertaintyFusion
//   fastdem/src/feature_extraction.cpp:28-118                        applyFeatureExtraction
//   fastdem/lib/nanoPCL/include/nanopcl/geometry/impl/pca.hpp:66-88  computePCA
//
// PARITY STATUS: "parity unpinned" for the neighbourhood iteration.  All four functions walk the
// map through nanoGrid's cells() / region() / neighbors(), and nanoGrid is not on disk
// (fastdem/CMakeLists.txt:24-28).  What the call sites and tests pin, and what is ASSUMED here:
//   * neighbours live in LOGICAL (unwrapped) coordinates, are clipped at the map border, carry
//     row / col such that `n.row - cell.row` is the world offset in cells
//     (feature_extraction.cpp:73-76) and include the centre cell (inpainting.cpp:50 skips it by hand);
//   * region(Size(k,k)) is the k x k box; region(radius) is the disc of offsets with
//     (dr^2+dc^2)*res^2 <= radius^2, so that 0.6 m at 0.5 m resolution holds the 4-neighbours but not
//     the diagonals ("slightly more than 1 cell", tests/test_postprocess.cpp:212) — ASSUMED inclusive;
//   * dist_sq is in m^2 (it is multiplied by 1/(2 sigma^2) with sigma in metres,
//     uncertainty_fusion.cpp:122-123,158) — ASSUMED computed as float((dr^2+dc^2)) * float(res)^2;
//   * entries are visited dr-major, dc-minor, ascending — ASSUMED; it only fixes the order of the
//     float sums (a last-ulp effect), every result is otherwise order-free.
// The Eigen 3x3 direct eigen-solver (SelfAdjointEigenSolver::computeDirect) is restated from Eigen
// 3.4's published algorithm (closed-form roots + cross-product kernels); the reference's known-answer
// tests for these stages (tests/test_postprocess.cpp:37-72,192-420) are re-expressed in
// tests/test_oracle_post_spec.py.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

#include "fdm_grid.hpp"

namespace fdmref {

namespace layer {
constexpr auto elevation_inpainted = "elevation_inpainted";  // postprocess/inpainting.hpp:23
constexpr auto step = "step";                                // postprocess/feature_extraction.hpp:12-18
constexpr auto slope = "slope";
constexpr auto roughness = "roughness";
constexpr auto curvature = "curvature";
constexpr auto normal_x = "_normal_x";
constexpr auto normal_y = "_normal_y";
constexpr auto normal_z = "_normal_z";
}  // namespace layer

// ------------------------------------------------------------ neighbourhoods ----
struct RegionEntry { int dr, dc; float dist_sq; };
struct Region { std::vector<RegionEntry> entries; };

inline Region regionBox(const Grid& map, int kr, int kc) {
  Region reg;
  const float res = static_cast<float>(map.resolution());
  for (int dr = -(kr / 2); dr <= kr / 2; ++dr)
    for (int dc = -(kc / 2); dc <= kc / 2; ++dc)
      reg.entries.push_back({dr, dc, static_cast<float>(dr * dr + dc * dc) * (res * res)});
  return reg;
}
inline Region regionDisc(const Grid& map, float radius) {
  Region reg;
  const float res = static_cast<float>(map.resolution());
  const int k = static_cast<int>(std::floor(radius / res + 1e-4f));
  const float r2 = radius * radius;
  for (int dr = -k; dr <= k; ++dr)
    for (int dc = -k; dc <= k; ++dc) {
      const float d2 = static_cast<float>(dr * dr + dc * dc) * (res * res);
      if (d2 <= r2 * (1.0f + 1e-5f)) reg.entries.push_back({dr, dc, d2});
    }
  return reg;
}

// logical (row, col) -> linear storage index (column-major), through the circular buffer
struct CellWalker {
  const Grid& map;
  int rows, cols, sr, sc;
  explicit CellWalker(const Grid& m) : map(m), rows(m.rows()), cols(m.cols()), sr(m.startIndex()[0]), sc(m.startIndex()[1]) {}
  size_t index(int lr, int lc) const {
    int r = lr + sr, c = lc + sc;
    if (r >= rows) r -= rows;
    if (c >= cols) c -= cols;
    return size_t(c) * rows + r;
  }
  bool inside(int lr, int lc) const { return lr >= 0 && lc >= 0 && lr < rows && lc < cols; }
};

// ------------------------------------------------------------ inpainting ----
inline void applyInpainting(Grid& map, int max_iterations, int min_valid_neighbors, bool inplace) {
  const char* output = inplace ? "elevation" : layer::elevation_inpainted;
  if (!map.exists(output)) map.add(output, NAN);
  auto& inpainted = map.get(output);
  if (!inplace) inpainted = map.get("elevation");
  const Region reg8 = regionBox(map, 3, 3);
  const CellWalker w(map);
  std::vector<float> buffer(inpainted.size());
  for (int iter = 0; iter < max_iterations; ++iter) {
    bool changed = false;
    buffer = inpainted;
    for (int lc = 0; lc < w.cols; ++lc)
      for (int lr = 0; lr < w.rows; ++lr) {
        const size_t ci = w.index(lr, lc);
        if (!std::isnan(inpainted[ci])) continue;
        float sum = 0.0f;
        int count = 0;
        for (const auto& e : reg8.entries) {
          if (e.dr == 0 && e.dc == 0) continue;
          if (!w.inside(lr + e.dr, lc + e.dc)) continue;
          const float val = inpainted[w.index(lr + e.dr, lc + e.dc)];
          if (std::isfinite(val)) {
            sum += val;
            ++count;
          }
        }
        if (count >= min_valid_neighbors) {
          buffer[ci] = sum / static_cast<float>(count);
          changed = true;
        }
      }
    inpainted = buffer;
    if (!changed) break;
  }
}

// ------------------------------------------------------------ spatial smoothing ----
inline void applySpatialSmoothing(Grid& map, const std::string& layer_name, int kernel_size,
                                  int min_valid_neighbors) {
  if (!map.exists(layer_name)) return;
  const std::vector<float> input = map.get(layer_name);
  auto& output = map.get(layer_name);
  const Region reg = regionBox(map, kernel_size, kernel_size);
  const CellWalker w(map);
  std::vector<float> window;
  for (int lc = 0; lc < w.cols; ++lc)
    for (int lr = 0; lr < w.rows; ++lr) {
      const size_t ci = w.index(lr, lc);
      if (!std::isfinite(input[ci])) continue;
      window.clear();
      for (const auto& e : reg.entries) {
        if (!w.inside(lr + e.dr, lc + e.dc)) continue;
        const float val = input[w.index(lr + e.dr, lc + e.dc)];
        if (std::isfinite(val)) window.push_back(val);
      }
      if (static_cast<int>(window.size()) < min_valid_neighbors) continue;
      const size_t mid = window.size() / 2;
      std::nth_element(window.begin(), window.begin() + mid, window.end());
      output[ci] = window[mid];
    }
}

// ------------------------------------------------------------ uncertainty fusion ----
struct FusionConfig {  // config/postprocess.hpp:32-39
  bool enabled = false;
  float search_radius = 0.15f, spatial_sigma = 0.05f, quantile_lower = 0.01f, quantile_upper = 0.99f;
  int min_valid_neighbors = 3;
};

class WeightedECDF {  // uncertainty_fusion.cpp:36-99
 public:
  void add(float value, float weight) {
    if (weight > 1e-6f && std::isfinite(value)) samples_.push_back({value, weight});
  }
  void clear() { samples_.clear(); }
  float quantile(float p) {
    if (samples_.empty()) return NAN;
    if (samples_.size() == 1) return samples_[0].value;
    std::sort(samples_.begin(), samples_.end(), [](const S& a, const S& b) { return a.value < b.value; });
    float total = 0.0f;
    for (const auto& s : samples_) total += s.weight;
    if (total <= 0.0f) return NAN;
    const float target = p * total;
    float cumulative = 0.0f;
    for (const auto& s : samples_) {
      cumulative += s.weight;
      if (cumulative >= target) return s.value;
    }
    return samples_.back().value;
  }

 private:
  struct S { float value, weight; };
  std::vector<S> samples_;
};

inline void applyUncertaintyFusion(Grid& map, const FusionConfig& cfg) {
  if (!cfg.enabled) return;
  if (!map.exists("upper_bound") || !map.exists("lower_bound")) return;
  auto& upper = map.get("upper_bound");
  auto& lower = map.get("lower_bound");
  const Region reg = regionDisc(map, cfg.search_radius);
  const float inv_2s2 = 1.0f / (2.0f * cfg.spatial_sigma * cfg.spatial_sigma);
  std::vector<float> ub = upper, lb = lower;
  WeightedECDF lo, up;
  const CellWalker w(map);
  for (int lc = 0; lc < w.cols; ++lc)
    for (int lr = 0; lr < w.rows; ++lr) {
      const size_t ci = w.index(lr, lc);
      if (!std::isfinite(upper[ci]) || !std::isfinite(lower[ci])) continue;
      lo.clear();
      up.clear();
      int valid = 0;
      for (const auto& e : reg.entries) {
        if (!w.inside(lr + e.dr, lc + e.dc)) continue;
        const size_t ni = w.index(lr + e.dr, lc + e.dc);
        const float nu = upper[ni], nl = lower[ni];
        if (!std::isfinite(nu) || !std::isfinite(nl)) continue;
        const float w_spatial = std::exp(-e.dist_sq * inv_2s2);
        constexpr float epsilon = 1e-4f;
        const float range = nu - nl;
        const float w_range = 1.0f / (range + epsilon);
        const float weight = w_spatial * w_range;
        lo.add(nl, weight);
        up.add(nu, weight);
        ++valid;
      }
      if (valid >= cfg.min_valid_neighbors) {
        const float l = lo.quantile(cfg.quantile_lower);
        const float u = up.quantile(cfg.quantile_upper);
        if (std::isfinite(l) && std::isfinite(u)) {
          ub[ci] = u;
          lb[ci] = l;
        }
      }
    }
  upper = ub;
  lower = lb;
}

// ------------------------------------------------------------ 3x3 symmetric eigen-solver ----
// Eigen::SelfAdjointEigenSolver<Matrix3f>::computeDirect (Eigen 3.4,
// direct_selfadjoint_eigenvalues<SolverType,3,false>): shift by trace/3, scale to [-1,1],
// trigonometric roots, eigenvectors from cross-product kernels.  m is column-major, lower triangle read.
struct Eig3 { float val[3]; float vec[9]; };  // vec column k = eigenvector of val[k], ascending

// The reference calls the platform's float libm (atan2f / cosf / sinf / acosf).  Which float comes back for an
// argument is a property of that libm: glibc 2.35 (this image) ships the fdlibm atan2f (errors up to ~1 ulp), glibc
// >= 2.41 the correctly rounded CORE-MATH versions.  trig_mode() selects what the oracle restates:
//   0  the platform's float functions — the reference as built on this machine (default);
//   1  the function evaluated in double and rounded once to float, i.e. the correctly rounded result (up to
//      double rounding) — what the device computes, and what a CORE-MATH libm returns.
// Parity tests run both: mode 1 pins every OTHER float operation of the stage bit for bit, mode 0 bounds what
// the libm's last-ulp differences become after the cancellation in the roots.
inline int& trig_mode() { static int mode = 0; return mode; }
namespace trig {
inline float atan2_(float y, float x) {
  return trig_mode() ? static_cast<float>(std::atan2(static_cast<double>(y), static_cast<double>(x))) : std::atan2(y, x);
}
inline float cos_(float v) { return trig_mode() ? static_cast<float>(std::cos(static_cast<double>(v))) : std::cos(v); }
inline float sin_(float v) { return trig_mode() ? static_cast<float>(std::sin(static_cast<double>(v))) : std::sin(v); }
inline float acos_(float v) { return trig_mode() ? static_cast<float>(std::acos(static_cast<double>(v))) : std::acos(v); }
}  // namespace trig

namespace eig3 {
inline float& M(float* m, int r, int c) { return m[c * 3 + r]; }
inline void cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
inline float sqnorm(const float* a) { return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]); }
inline void computeRoots(const float* m, float* roots) {
  auto A = [&](int r, int c) { return m[c * 3 + r]; };
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = std::sqrt(3.0f);
  const float c0 = A(0, 0) * A(1, 1) * A(2, 2) + 2.0f * A(1, 0) * A(2, 0) * A(2, 1) - A(0, 0) * A(2, 1) * A(2, 1) -
                   A(1, 1) * A(2, 0) * A(2, 0) - A(2, 2) * A(1, 0) * A(1, 0);
  const float c1 = A(0, 0) * A(1, 1) - A(1, 0) * A(1, 0) + A(0, 0) * A(2, 2) - A(2, 0) * A(2, 0) +
                   A(1, 1) * A(2, 2) - A(2, 1) * A(2, 1);
  const float c2 = A(0, 0) + A(1, 1) + A(2, 2);
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = std::max(a_over_3, 0.0f);
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = std::max(q, 0.0f);
  const float rho = std::sqrt(a_over_3);
  const float theta = trig::atan2_(std::sqrt(q), half_b) * s_inv3;
  const float cos_theta = trig::cos_(theta), sin_theta = trig::sin_(theta);
  roots[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 + 2.0f * rho * cos_theta;
}
// res = unit vector in the kernel of mat; representative = the column with the largest |diagonal|
inline void extractKernel(const float* mat, float* res, float* representative) {
  int i0 = 0;
  float best = std::fabs(mat[0]);
  for (int k = 1; k < 3; ++k)
    if (std::fabs(mat[k * 3 + k]) > best) { best = std::fabs(mat[k * 3 + k]); i0 = k; }
  for (int r = 0; r < 3; ++r) representative[r] = mat[i0 * 3 + r];
  float c0[3], c1[3];
  cross(representative, mat + ((i0 + 1) % 3) * 3, c0);
  cross(representative, mat + ((i0 + 2) % 3) * 3, c1);
  const float n0 = sqnorm(c0), n1 = sqnorm(c1);
  if (n0 > n1) {
    const float s = std::sqrt(n0);
    for (int r = 0; r < 3; ++r) res[r] = c0[r] / s;
  } else {
    const float s = std::sqrt(n1);
    for (int r = 0; r < 3; ++r) res[r] = c1[r] / s;
  }
}
}  // namespace eig3

inline Eig3 computeDirect3(const float* cov) {
  using namespace eig3;
  Eig3 out{};
  const float shift = (cov[0] + cov[4] + cov[8]) / 3.0f;
  float sm[9];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) sm[c * 3 + r] = r >= c ? cov[c * 3 + r] : cov[r * 3 + c];  // selfadjointView<Lower>
  sm[0] -= shift; sm[4] -= shift; sm[8] -= shift;
  float scale = 0.0f;
  for (float v : sm) scale = std::max(scale, std::fabs(v));
  if (scale > 0.0f)
    for (float& v : sm) v /= scale;
  float* ev = out.val;
  computeRoots(sm, ev);
  float* V = out.vec;
  if ((ev[2] - ev[0]) <= std::numeric_limits<float>::epsilon()) {
    for (int k = 0; k < 9; ++k) V[k] = (k % 4 == 0) ? 1.0f : 0.0f;
  } else {
    float tmp[9];
    float d0 = ev[2] - ev[1];
    const float d1 = ev[1] - ev[0];
    int k = 0, l = 2;
    if (d0 > d1) { std::swap(k, l); d0 = d1; }
    std::copy(sm, sm + 9, tmp);
    tmp[0] -= ev[k]; tmp[4] -= ev[k]; tmp[8] -= ev[k];
    extractKernel(tmp, V + k * 3, V + l * 3);
    if (d0 <= 2.0f * std::numeric_limits<float>::epsilon() * d1) {
      const float dot = V[k * 3] * V[l * 3] + (V[k * 3 + 1] * V[l * 3 + 1] + V[k * 3 + 2] * V[l * 3 + 2]);
      for (int r = 0; r < 3; ++r) V[l * 3 + r] -= dot * V[l * 3 + r];
      const float n = std::sqrt(sqnorm(V + l * 3));
      for (int r = 0; r < 3; ++r) V[l * 3 + r] /= n;
    } else {
      std::copy(sm, sm + 9, tmp);
      tmp[0] -= ev[l]; tmp[4] -= ev[l]; tmp[8] -= ev[l];
      float dummy[3];
      extractKernel(tmp, V + l * 3, dummy);
    }
    float c[3];
    cross(V + 6, V + 0, c);
    const float n = std::sqrt(sqnorm(c));
    for (int r = 0; r < 3; ++r) V[3 + r] = c[r] / n;
  }
  for (int i = 0; i < 3; ++i) ev[i] = ev[i] * scale + shift;
  return out;
}

// ------------------------------------------------------------ feature extraction ----
inline void applyFeatureExtraction(Grid& map, float analysis_radius, int min_valid_neighbors,
                                   float step_lower_percentile, float step_upper_percentile) {
  if (!map.exists("elevation")) return;
  for (const char* n : {layer::step, layer::slope, layer::roughness, layer::curvature, layer::normal_x,
                        layer::normal_y, layer::normal_z})
    if (!map.exists(n)) map.add(n, NAN);
  if (map.rows() == 0 || map.cols() == 0) return;
  const auto& elev = map.get("elevation");
  auto& step_mat = map.get(layer::step);
  auto& slope_mat = map.get(layer::slope);
  auto& rough_mat = map.get(layer::roughness);
  auto& curv_mat = map.get(layer::curvature);
  auto& nx_mat = map.get(layer::normal_x);
  auto& ny_mat = map.get(layer::normal_y);
  auto& nz_mat = map.get(layer::normal_z);
  const Region reg = regionDisc(map, analysis_radius);
  const float resf = static_cast<float>(map.resolution());
  const CellWalker w(map);
  std::vector<float> z_vals;
  for (int lc = 0; lc < w.cols; ++lc)
    for (int lr = 0; lr < w.rows; ++lr) {
      const size_t ci = w.index(lr, lc);
      const float center_z = elev[ci];
      if (!std::isfinite(center_z)) continue;
      float sum[3] = {0.f, 0.f, 0.f};
      float sq[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      z_vals.clear();
      int count = 0;
      for (const auto& e : reg.entries) {
        if (!w.inside(lr + e.dr, lc + e.dc)) continue;
        const float nz = elev[w.index(lr + e.dr, lc + e.dc)];
        if (!std::isfinite(nz)) continue;
        const float d[3] = {static_cast<float>(-e.dr) * resf, static_cast<float>(-e.dc) * resf, nz - center_z};
        for (int k = 0; k < 3; ++k) sum[k] += d[k];
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 3; ++r) sq[c * 3 + r] += d[r] * d[c];
        z_vals.push_back(nz);
        ++count;
      }
      if (count < min_valid_neighbors) continue;
      const float inv_n = 1.0f / static_cast<float>(count);
      const float mean[3] = {sum[0] * inv_n, sum[1] * inv_n, sum[2] * inv_n};
      float cov[9];
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) cov[c * 3 + r] = sq[c * 3 + r] * inv_n - mean[r] * mean[c];
      // computePCA (pca.hpp:66-88)
      const float trace = cov[0] + cov[4] + cov[8];
      if (trace < std::numeric_limits<float>::epsilon()) continue;
      const Eig3 pca = computeDirect3(cov);
      constexpr float kMinEigenvalue = 1e-8f;
      if (pca.val[1] < kMinEigenvalue) continue;
      float normal[3] = {pca.vec[0], pca.vec[1], pca.vec[2]};
      if (normal[2] < 0.0f)
        for (float& v : normal) v = -v;
      std::sort(z_vals.begin(), z_vals.end());
      const int lo = static_cast<int>(step_lower_percentile * static_cast<float>(count - 1));
      const int hi = static_cast<int>(step_upper_percentile * static_cast<float>(count - 1));
      step_mat[ci] = z_vals[hi] - z_vals[lo];
      slope_mat[ci] = trig::acos_(std::abs(normal[2])) * 180.0f / static_cast<float>(M_PI);
      rough_mat[ci] = std::sqrt(pca.val[0]);
      curv_mat[ci] = (trace > 0.0f) ? std::abs(pca.val[0] / trace) : 0.0f;
      nx_mat[ci] = normal[0];
      ny_mat[ci] = normal[1];
      nz_mat[ci] = normal[2];
    }
}

}  // namespace fdmref
