// fdm_ref_egress.hpp — CPU restatement of the map -> PointCloud2 egress.
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  (same rules as fdm_ref.hpp)
//
// Follows fastdem/include/fastdem/bridge/ros/impl.hpp:28-166 (toPointCloud2Impl) and
// elevation_map.hpp:42-45 (layer::isInternal).  PARITY STATUS: no reference test pins the byte
// stream (the ROS bridge has no unit test); the restatement is a line-by-line statement of the
// function and is checked here against hand-derived values (tests/test_oracle_egress_spec.py).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "fdm_grid.hpp"

namespace fdmref {

struct PackedCloud {
  std::vector<std::string> fields;  // "x","y","z", float layers in getLayers() order, then "rgb"
  uint32_t point_step = 0;
  uint64_t n_points = 0;
  std::vector<uint8_t> data;
};

inline bool isInternalLayer(const std::string& n) { return !n.empty() && n[0] == '_'; }

// toPointCloud2Impl(map, stamp, elevation_layer, sub_start, sub_size)
inline PackedCloud packCloud(const Grid& map, const std::string& elevation_layer, int sub_r0, int sub_c0,
                             int sub_rows, int sub_cols) {
  PackedCloud out;
  const auto& elev = map.get(elevation_layer);
  const int rows = map.rows(), cols = map.cols();
  const int* start = map.startIndex();
  const double res = map.resolution();
  const double origin_x = map.position()[0] + map.length()[0] / 2.0 - res / 2.0;
  const double origin_y = map.position()[1] + map.length()[1] / 2.0 - res / 2.0;

  std::vector<float> row_x(sub_rows), col_y(sub_cols);
  std::vector<int> buf_row(sub_rows), buf_col(sub_cols);
  for (int i = 0; i < sub_rows; ++i) {
    const int r = (sub_r0 + i) % rows;
    buf_row[i] = r;
    const int unwrapped = (r - start[0] + rows) % rows;
    row_x[i] = static_cast<float>(origin_x - unwrapped * res);
  }
  for (int j = 0; j < sub_cols; ++j) {
    const int c = (sub_c0 + j) % cols;
    buf_col[j] = c;
    const int unwrapped = (c - start[1] + cols) % cols;
    col_y[j] = static_cast<float>(origin_y - unwrapped * res);
  }

  std::vector<std::string> float_layers;
  bool has_color = false;
  for (const auto& l : map.layers()) {
    if (isInternalLayer(l)) continue;
    if (l == elevation_layer) continue;
    if (l == "color") { has_color = true; continue; }
    float_layers.push_back(l);
  }
  out.fields = {"x", "y", "z"};
  for (const auto& l : float_layers) out.fields.push_back(l);
  if (has_color) out.fields.push_back("rgb");
  out.point_step = uint32_t(out.fields.size()) * 4u;

  std::vector<const float*> ptrs;
  for (const auto& l : float_layers) ptrs.push_back(map.get(l).data());
  const float* color = has_color ? map.get("color").data() : nullptr;

  size_t valid = 0;
  for (int j = 0; j < sub_cols; ++j) {
    const size_t base = size_t(buf_col[j]) * rows;
    for (int i = 0; i < sub_rows; ++i)
      if (std::isfinite(elev[base + buf_row[i]])) ++valid;
  }
  out.n_points = valid;
  out.data.resize(valid * out.point_step);
  uint8_t* o = out.data.data();
  auto put = [&](const float& v) { std::memcpy(o, &v, 4); o += 4; };
  for (int j = 0; j < sub_cols; ++j) {
    const float y = col_y[j];
    const size_t base = size_t(buf_col[j]) * rows;
    for (int i = 0; i < sub_rows; ++i) {
      const size_t idx = base + buf_row[i];
      const float z = elev[idx];
      if (!std::isfinite(z)) continue;
      put(row_x[i]);
      put(y);
      put(z);
      for (const float* p : ptrs) put(p[idx]);
      if (color) put(color[idx]);
    }
  }
  return out;
}

}  // namespace fdmref
