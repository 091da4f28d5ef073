// fastdem/config/fastdem.hpp (fastdem/include/fastdem/config/fastdem.hpp:23-38) + the
// validation semantics of fastdem/src/config_fastdem.cpp:128-260 (throw on min_var >= max_var and
// on unsorted P2 markers, otherwise warn + clamp).  The YAML loader (parseConfig / loadConfig)
// is the "next" row f4: yaml-cpp is not available here, so configs are filled programmatically.
#pragma once
#include <algorithm>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>

#include "fastdem/config/mapping.hpp"
#include "fastdem/config/postprocess.hpp"
#include "fastdem/config/sensor_model.hpp"

namespace fastdem {
namespace config {
struct PointFilter {
  float z_min = -std::numeric_limits<float>::max();
  float z_max = std::numeric_limits<float>::max();
  float range_min = 0.0f;
  float range_max = std::numeric_limits<float>::max();
};
}  // namespace config

struct Config {
  config::PointFilter point_filter;
  config::SensorModel sensor_model;
  config::Mapping mapping;
  config::Raycasting raycasting;
};

namespace detail {
inline void warn(const std::string& m) { std::fprintf(stderr, "[warn] [Config] %s\n", m.c_str()); }

// detail::validate of config_fastdem.cpp:128-260
inline void validate(Config& m) {
  if (m.mapping.kalman.min_variance >= m.mapping.kalman.max_variance)
    throw std::invalid_argument("mapping.kalman: min_variance (" +
                                std::to_string(m.mapping.kalman.min_variance) + ") >= max_variance (" +
                                std::to_string(m.mapping.kalman.max_variance) + ")");
  if (m.raycasting.enabled) {
    if (m.raycasting.height_conflict_threshold <= 0.0f) { warn("raycasting.height_conflict_threshold must be > 0, clamping to 0.05"); m.raycasting.height_conflict_threshold = 0.05f; }
    if (m.raycasting.log_odds_observed <= 0.0f) { warn("raycasting.log_odds_observed must be > 0, clamping to 0.4"); m.raycasting.log_odds_observed = 0.4f; }
    if (m.raycasting.log_odds_ghost <= 0.0f) { warn("raycasting.log_odds_ghost must be > 0, clamping to 0.2"); m.raycasting.log_odds_ghost = 0.2f; }
    if (m.raycasting.log_odds_max <= 0.0f) { warn("raycasting.log_odds_max must be > 0, clamping to 2.0"); m.raycasting.log_odds_max = 2.0f; }
    if (m.raycasting.clear_threshold >= 0.0f) { warn("raycasting.clear_threshold must be < 0, clamping to -1.0"); m.raycasting.clear_threshold = -1.0f; }
  }
  if (m.mapping.kalman.min_variance <= 0.0f) { warn("estimation.kalman.min_variance must be > 0, clamping to 0.0001"); m.mapping.kalman.min_variance = 0.0001f; }
  if (m.mapping.kalman.process_noise < 0.0f) { warn("estimation.kalman.process_noise must be >= 0, clamping to 0"); m.mapping.kalman.process_noise = 0.0f; }
  if (m.mapping.p2.elevation_marker < 0 || m.mapping.p2.elevation_marker > 4) {
    warn("mapping.p2.elevation_marker out of range [0, 4], clamping");
    m.mapping.p2.elevation_marker = std::clamp(m.mapping.p2.elevation_marker, 0, 4);
  }
  auto& p2 = m.mapping.p2;
  float* dns[] = {&p2.dn0, &p2.dn1, &p2.dn2, &p2.dn3, &p2.dn4};
  for (int i = 0; i < 5; ++i)
    if (*dns[i] < 0.0f || *dns[i] > 1.0f) {
      warn("mapping.p2.dn" + std::to_string(i) + " out of [0, 1], clamping");
      *dns[i] = std::clamp(*dns[i], 0.0f, 1.0f);
    }
  if (p2.dn0 > p2.dn1 || p2.dn1 > p2.dn2 || p2.dn2 > p2.dn3 || p2.dn3 > p2.dn4)
    throw std::invalid_argument("mapping.p2: markers must be sorted (dn0 <= dn1 <= dn2 <= dn3 <= dn4)");
  auto& s = m.sensor_model;
  if (s.lidar.range_noise <= 0.0f) { warn("sensor.lidar.range_noise must be > 0, clamping to 0.02"); s.lidar.range_noise = 0.02f; }
  if (s.lidar.angular_noise < 0.0f) { warn("sensor.lidar.angular_noise must be >= 0, clamping to 0"); s.lidar.angular_noise = 0.0f; }
  if (s.constant.uncertainty <= 0.0f) { warn("sensor.constant.uncertainty must be > 0, clamping to 0.1"); s.constant.uncertainty = 0.1f; }
  if (s.rgbd.normal_a < 0.0f) { warn("sensor.rgbd.normal_a must be >= 0, clamping to 0"); s.rgbd.normal_a = 0.0f; }
  if (s.rgbd.normal_b < 0.0f) { warn("sensor.rgbd.normal_b must be >= 0, clamping to 0"); s.rgbd.normal_b = 0.0f; }
  if (s.rgbd.normal_c < 0.0f) { warn("sensor.rgbd.normal_c must be >= 0, clamping to 0"); s.rgbd.normal_c = 0.0f; }
  if (s.rgbd.lateral_factor < 0.0f) { warn("sensor.rgbd.lateral_factor must be >= 0, clamping to 0"); s.rgbd.lateral_factor = 0.0f; }
}
}  // namespace detail

/// Validate + clamp a programmatically built config exactly like parseConfig() does after parsing.
inline Config validated(Config cfg) {
  detail::validate(cfg);
  return cfg;
}

}  // namespace fastdem
