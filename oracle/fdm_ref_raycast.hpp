// fdm_ref_raycast.hpp — CPU restatement of the raycasting stage of FastDEM::integrateImpl
// (SURVEY.md §8 row f1).
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  (same rules as fdm_ref.hpp)
//
// Follows, in operation order:
//   fastdem/src/fastdem.cpp:152-159                      sensor origin, voxelGrid(ANY), applyRaycasting
//   fastdem/lib/nanoPCL/include/nanopcl/core/voxel.hpp:28-43,98-102      voxel key packing
//   fastdem/lib/nanoPCL/include/nanopcl/filters/impl/voxel_grid_impl.hpp:30-60,171-189  ANY mode
//   fastdem/src/raycasting.cpp:46-249                    traceRay / processScan / resolveGhostCells
//   fastdem/include/fastdem/elevation_map.hpp:131-135    clearAt = NaN in EVERY layer
//
// PARITY STATUS: the raycasting arithmetic is pinned by the reference's known-answer tests
// (fastdem/tests/test_postprocess.cpp:73-190, re-expressed in tests/test_oracle_raycast_spec.py).
// VoxelMode::ANY is NOT a function of the input alone in the reference: the representative is
// `idx[start + (count*7 + start*13) % count]` AFTER an unstable std::sort on the voxel key only,
// so which of a voxel's points sits at that position depends on the standard library's introsort.
// Two selections are restated here:
//   stable=false  std::sort on {key} exactly as the reference (this libstdc++ == what a g++ build
//                 of the reference does on this machine);
//   stable=true   ties broken by original point index (what a stable sort gives).  This is the one
//                 the HIP engine reproduces bit-exactly; it coincides with the reference whenever a
//                 voxel holds one point or introsort happens to keep the tie order.
// Either is a valid outcome of "ANY"; tests check GPU == stable oracle bit-exactly and that the
// std::sort variant selects a point of the same voxel for every voxel.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "fdm_grid.hpp"

namespace fdmref {

// config/postprocess.hpp:16-23
struct RaycastConfig {
  bool enabled = false;
  float height_conflict_threshold = 0.05f;
  float log_odds_observed = 0.4f;
  float log_odds_ghost = 0.2f;
  float log_odds_max = 2.0f;
  float clear_threshold = -1.0f;
};

namespace layer {
constexpr auto ghost_removal = "ghost_removal";            // postprocess/raycasting.hpp:27-31
constexpr auto raycasting = "raycasting";
constexpr auto visibility_logodds = "_visibility_logodds";
}  // namespace layer

// ---------------------------------------------------------------- voxel key ----
namespace voxel {
constexpr float MIN_SIZE = 0.001f, MAX_SIZE = 100.0f;
constexpr int32_t COORD_OFFSET = 1 << 20;
constexpr int32_t COORD_MIN = -COORD_OFFSET, COORD_MAX = COORD_OFFSET - 1;

// static_cast<int32_t>(float) outside the int range is UB in C++; the reference's x86 build
// (cvttss2si) yields INT_MIN for both signs and for NaN.  Stated explicitly so every compiler
// and the GPU agree.
inline int32_t cvt_x86(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT32_MIN;
  return static_cast<int32_t>(v);
}
inline uint64_t pack(float x, float y, float z, float inv) {  // voxel.hpp:28-43: [z:21][y:21][x:21]
  int32_t ix = cvt_x86(std::floor(x * inv));
  int32_t iy = cvt_x86(std::floor(y * inv));
  int32_t iz = cvt_x86(std::floor(z * inv));
  ix = std::clamp(ix, COORD_MIN, COORD_MAX);
  iy = std::clamp(iy, COORD_MIN, COORD_MAX);
  iz = std::clamp(iz, COORD_MIN, COORD_MAX);
  const uint64_t ux = uint64_t(ix + COORD_OFFSET), uy = uint64_t(iy + COORD_OFFSET),
                 uz = uint64_t(iz + COORD_OFFSET);
  return (uz << 42) | (uy << 21) | ux;
}
struct IndexedPoint {  // voxel.hpp:98-102: ordering looks at the key only
  uint64_t key;
  uint32_t index;
  bool operator<(const IndexedPoint& o) const { return key < o.key; }
};
}  // namespace voxel

// filters::voxelGrid(cloud, voxel_size, VoxelMode::ANY) (voxel_grid_impl.hpp:30-60,171-189).
// Returns the ORIGINAL indices of the selected points, in output (voxel-key) order.
inline std::vector<uint32_t> voxelGridAny(const std::vector<std::array<float, 4>>& pts, float voxel_size,
                                          bool stable) {
  if (voxel_size < voxel::MIN_SIZE || voxel_size > voxel::MAX_SIZE)
    throw std::invalid_argument("voxel_size must be in [0.001, 100]");
  std::vector<uint32_t> out;
  if (pts.empty()) return out;
  const float inv = 1.0f / voxel_size;
  std::vector<voxel::IndexedPoint> idx;
  idx.reserve(pts.size());
  for (size_t i = 0; i < pts.size(); ++i) {
    const auto& p = pts[i];
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    idx.push_back({voxel::pack(p[0], p[1], p[2], inv), uint32_t(i)});
  }
  if (idx.empty()) return out;
  if (stable)
    std::stable_sort(idx.begin(), idx.end());
  else
    std::sort(idx.begin(), idx.end());
  size_t start = 0;
  while (start < idx.size()) {
    const uint64_t key = idx[start].key;
    size_t end = start + 1;
    while (end < idx.size() && idx[end].key == key) ++end;
    const size_t count = end - start;
    out.push_back(idx[start + (count * 7 + start * 13) % count].index);
    start = end;
  }
  return out;
}

// ---------------------------------------------------------------- raycasting ----
struct RayStats {
  uint32_t n_rays = 0, n_observed = 0, n_ray_cells = 0, n_conflicts = 0, n_cleared = 0;
};

namespace detail {
constexpr float kMinRayLength = 1e-4f;
constexpr float kInfinity = 1e30f;

inline float& cell(std::vector<float>& m, int rows, int r, int c) { return m[size_t(c) * rows + r]; }

// traceRay (raycasting.cpp:46-140): 2-D DDA in fp32 grid coordinates, min exit height per cell.
inline void traceRay(const Grid& map, float resolution, const float* start, const float* end,
                     std::vector<float>& ray_min, std::vector<Index2>& ray_cells) {
  const float dx = end[0] - start[0];
  const float dy = end[1] - start[1];
  const float ray_len_2d = std::sqrt(dx * dx + dy * dy);
  if (ray_len_2d < kMinRayLength) return;
  const float dz = end[2] - start[2];

  const int nrows = map.rows(), ncols = map.cols();
  const int* buf_start = map.startIndex();
  const float origin_x = static_cast<float>(map.position()[0]) + nrows * resolution * 0.5f;
  const float origin_y = static_cast<float>(map.position()[1]) + ncols * resolution * 0.5f;

  const float gr0 = (origin_x - start[0]) / resolution;
  const float gc0 = (origin_y - start[1]) / resolution;
  const float gr1 = (origin_x - end[0]) / resolution;
  const float gc1 = (origin_y - end[1]) / resolution;
  const float dr = gr1 - gr0;
  const float dc = gc1 - gc0;

  int r = static_cast<int>(std::floor(gr0));
  int c = static_cast<int>(std::floor(gc0));

  int step_r, step_c;
  float t_max_r, t_max_c, t_delta_r, t_delta_c;
  if (std::abs(dr) > 1e-8f) {
    step_r = (dr > 0) ? 1 : -1;
    const float boundary = (step_r > 0) ? (r + 1.0f) : static_cast<float>(r);
    t_max_r = (boundary - gr0) / dr;
    t_delta_r = static_cast<float>(step_r) / dr;
  } else {
    step_r = 0;
    t_max_r = kInfinity;
    t_delta_r = kInfinity;
  }
  if (std::abs(dc) > 1e-8f) {
    step_c = (dc > 0) ? 1 : -1;
    const float boundary = (step_c > 0) ? (c + 1.0f) : static_cast<float>(c);
    t_max_c = (boundary - gc0) / dc;
    t_delta_c = static_cast<float>(step_c) / dc;
  } else {
    step_c = 0;
    t_max_c = kInfinity;
    t_delta_c = kInfinity;
  }

  const int max_steps = nrows + ncols;
  for (int s = 0; s < max_steps; ++s) {
    if (r >= 0 && r < nrows && c >= 0 && c < ncols) {
      const int mr = (r + buf_start[0]) % nrows;
      const int mc = (c + buf_start[1]) % ncols;
      const float t_exit = std::min(t_max_r, t_max_c);
      const float height = start[2] + std::min(t_exit, 1.0f) * dz;
      float& cur_min = cell(ray_min, nrows, mr, mc);
      if (std::isnan(cur_min)) {
        cur_min = height;
        ray_cells.push_back({mr, mc});
      } else if (height < cur_min) {
        cur_min = height;
      }
    }
    if (t_max_r < t_max_c) {
      if (t_max_r >= 1.0f) break;
      r += step_r;
      t_max_r += t_delta_r;
    } else {
      if (t_max_c >= 1.0f) break;
      c += step_c;
      t_max_c += t_delta_c;
    }
  }
}
}  // namespace detail

// applyRaycasting (raycasting.cpp:204-249).  `scan` = xyz1 points in the map frame.
// Points with a non-finite coordinate never reach this function through integrate() (voxelGrid
// drops them); a direct caller passing one hits float->int UB in traceRay, so they are skipped here
// (and by the engine) — stated, not inherited.
inline RayStats applyRaycasting(Grid& map, const std::vector<std::array<float, 4>>& scan,
                                const float* sensor_origin, const RaycastConfig& cfg) {
  RayStats st;
  if (!cfg.enabled || scan.empty()) return st;
  if (!map.exists("elevation")) return st;
  if (!map.isInside(double(sensor_origin[0]), double(sensor_origin[1]))) return st;
  if (!map.exists(layer::ghost_removal)) map.add(layer::ghost_removal);
  if (!map.exists(layer::raycasting)) map.add(layer::raycasting);
  if (!map.exists(layer::visibility_logodds)) map.add(layer::visibility_logodds);
  map.clear(layer::raycasting);

  // processScan (raycasting.cpp:142-173)
  const int rows = map.rows();
  const float resolution = static_cast<float>(map.resolution());
  std::vector<Index2> ray_cells;
  {
    auto& logodds_mat = map.get(layer::visibility_logodds);
    auto& min_height_mat = map.get(layer::raycasting);
    for (const auto& pt : scan) {
      if (!std::isfinite(pt[0]) || !std::isfinite(pt[1]) || !std::isfinite(pt[2])) continue;
      Index2 idx;
      if (map.getIndex(double(pt[0]), double(pt[1]), idx)) {
        float& logodds = detail::cell(logodds_mat, rows, idx.r, idx.c);
        if (std::isnan(logodds)) logodds = 0.0f;
        logodds = std::min(logodds + cfg.log_odds_observed, cfg.log_odds_max);
        ++st.n_observed;
      }
      if (pt[2] >= sensor_origin[2]) continue;  // upward ray
      ++st.n_rays;
      detail::traceRay(map, resolution, sensor_origin, pt.data(), min_height_mat, ray_cells);
    }
  }
  st.n_ray_cells = uint32_t(ray_cells.size());

  // resolveGhostCells (raycasting.cpp:175-202)
  for (const auto& idx : ray_cells) {
    const float elev = map.at("elevation", idx);
    if (std::isnan(elev)) continue;
    if (elev > map.at(layer::raycasting, idx) + cfg.height_conflict_threshold) {
      ++st.n_conflicts;
      float& logodds = map.at(layer::visibility_logodds, idx);
      if (std::isnan(logodds)) logodds = 0.0f;
      logodds -= cfg.log_odds_ghost;
      if (logodds < cfg.clear_threshold) {
        for (const auto& name : map.layers()) map.at(name, idx) = NAN;  // ElevationMap::clearAt
        map.at(layer::ghost_removal, idx) = 1.0f;
        ++st.n_cleared;
      }
    }
  }
  return st;
}

// sensor_origin = (T_world_base * T_base_sensor).translation().cast<float>() (fastdem.cpp:153-154).
// Isometry3d product: translation = L_wb * t_bs + t_wb, 3x3*3x1 double coeff-based product
// (3-term redux a0 + (a1 + a2)), then the sum, then the cast.
inline void sensorOrigin(const double* Twb, const double* Tbs, float* out3) {
  for (int i = 0; i < 3; ++i) {
    const double a0 = Twb[0 * 4 + i] * Tbs[12], a1 = Twb[1 * 4 + i] * Tbs[13], a2 = Twb[2 * 4 + i] * Tbs[14];
    out3[i] = static_cast<float>((a0 + (a1 + a2)) + Twb[12 + i]);
  }
}

}  // namespace fdmref
