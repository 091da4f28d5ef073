// fdm_ref_ingest.hpp — CPU restatement of the PointCloud2 -> PointCloud ingest.
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  (same rules as fdm_ref.hpp)
//
// Follows fastdem/lib/nanoPCL/include/nanopcl/bridge/ros/impl.hpp:41-100 (field offsets),
// :104-118 (readIntensity), :163-171 (readRgb), :174-246 (from_impl: points with a non-finite
// coordinate are skipped, order kept).  Only the channels the integrate() path consumes are kept
// (xyz, intensity, colour); ring / time / label / normals never reach the map.
// PARITY STATUS: no reference test pins this function (the ROS bridges have no unit tests);
// hand-derived values in tests/test_oracle_ingest_spec.py.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "fdm_ref.hpp"

namespace fdmref {

// sensor_msgs/PointField datatype codes (impl.hpp:22-31)
enum : int32_t { PF_INT8 = 1, PF_UINT8 = 2, PF_INT16 = 3, PF_UINT16 = 4, PF_INT32 = 5, PF_UINT32 = 6,
                 PF_FLOAT32 = 7, PF_FLOAT64 = 8 };

struct Cloud2Layout {
  uint32_t point_step = 0;
  int32_t off_x = -1, off_y = -1, off_z = -1;
  int32_t off_intensity = -1, intensity_type = 0;
  int32_t off_rgb = -1;
};

inline float readIntensity(const uint8_t* p, int32_t type) {  // impl.hpp:104-118
  switch (type) {
    case PF_UINT8: return static_cast<float>(*p);
    case PF_UINT16: { uint16_t v; std::memcpy(&v, p, 2); return static_cast<float>(v); }
    case PF_FLOAT32: { float v; std::memcpy(&v, p, 4); return v; }
    case PF_FLOAT64: { double v; std::memcpy(&v, p, 8); return static_cast<float>(v); }
    default: return 0.0f;
  }
}

// from_impl (impl.hpp:174-246)
inline Cloud fromCloud2(const uint8_t* data, uint64_t num_points, const Cloud2Layout& L) {
  Cloud c;
  if (num_points == 0) return c;
  if (!(L.off_x >= 0 && L.off_y >= 0 && L.off_z >= 0)) return c;
  c.has_intensity = L.off_intensity >= 0;
  c.has_color = L.off_rgb >= 0;
  for (uint64_t i = 0; i < num_points; ++i) {
    const uint8_t* pt = data + i * L.point_step;
    float x, y, z;
    std::memcpy(&x, pt + L.off_x, 4);
    std::memcpy(&y, pt + L.off_y, 4);
    std::memcpy(&z, pt + L.off_z, 4);
    if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z)) continue;
    c.pts.push_back({x, y, z, 1.0f});
    if (c.has_intensity) c.intensity.push_back(readIntensity(pt + L.off_intensity, L.intensity_type));
    if (c.has_color) {
      uint32_t rgb;
      std::memcpy(&rgb, pt + L.off_rgb, 4);
      c.color.push_back({uint8_t((rgb >> 16) & 0xFF), uint8_t((rgb >> 8) & 0xFF), uint8_t(rgb & 0xFF)});
    }
  }
  return c;
}

}  // namespace fdmref
