// fastdem/config/fastdem.hpp (fastdem/include/fastdem/config/fastdem.hpp:23-38) + the
// validation semantics of fastdem/src/config_fastdem.cpp:128-260 (throw on min_var >= max_var and
// on unsorted P2 markers, otherwise warn + clamp), and the YAML loader parseConfig / loadConfig
// (config_fastdem.cpp:57-126,262-277) over the YAML subset in yaml_lite.hpp.
#pragma once
#include <algorithm>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <string>

#include "fastdem/config/mapping.hpp"
#include "fastdem/config/postprocess.hpp"
#include "fastdem/config/sensor_model.hpp"
#include "fastdem/config/yaml_lite.hpp"

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

namespace detail {
template <typename T>
inline void load(const yaml::Node& node, const std::string& key, T& value) {  // config_fastdem.cpp:26-31
  if (node[key]) value = node[key].as<T>();
}
// detail::parse of config_fastdem.cpp:57-126: every key optional, unknown enum strings warn + default
inline Config parse(const yaml::Node& root) {
  Config cfg;
  if (const auto& n = root["mapping"]) {
    auto& m = cfg.mapping;
    std::string mode, type;
    load(n, "mode", mode);
    if (!mode.empty()) {
      if (mode == "local") m.mode = MappingMode::LOCAL;
      else if (mode == "global") m.mode = MappingMode::GLOBAL;
      else { warn("Unknown mapping mode '" + mode + "', defaulting to local"); m.mode = MappingMode::LOCAL; }
    }
    load(n, "type", type);
    if (!type.empty()) {
      if (type == "kalman_filter") m.estimation_type = EstimationType::Kalman;
      else if (type == "p2_quantile") m.estimation_type = EstimationType::P2Quantile;
      else { warn("Unknown estimation type '" + type + "', defaulting to kalman_filter"); m.estimation_type = EstimationType::Kalman; }
    }
    if (const auto& k = n["kalman"]) {
      load(k, "min_variance", m.kalman.min_variance);
      load(k, "max_variance", m.kalman.max_variance);
      load(k, "process_noise", m.kalman.process_noise);
    }
    if (const auto& p = n["p2"]) {
      load(p, "dn0", m.p2.dn0);
      load(p, "dn1", m.p2.dn1);
      load(p, "dn2", m.p2.dn2);
      load(p, "dn3", m.p2.dn3);
      load(p, "dn4", m.p2.dn4);
      load(p, "elevation_marker", m.p2.elevation_marker);
      load(p, "max_sample_count", m.p2.max_sample_count);
    }
  }
  if (const auto& n = root["point_filter"]) {
    load(n, "z_min", cfg.point_filter.z_min);
    load(n, "z_max", cfg.point_filter.z_max);
    load(n, "range_min", cfg.point_filter.range_min);
    load(n, "range_max", cfg.point_filter.range_max);
  }
  if (const auto& n = root["raycasting"]) {
    load(n, "enabled", cfg.raycasting.enabled);
    load(n, "height_conflict_threshold", cfg.raycasting.height_conflict_threshold);
    load(n, "log_odds_observed", cfg.raycasting.log_odds_observed);
    load(n, "log_odds_ghost", cfg.raycasting.log_odds_ghost);
    load(n, "log_odds_max", cfg.raycasting.log_odds_max);
    load(n, "clear_threshold", cfg.raycasting.clear_threshold);
  }
  if (const auto& n = root["sensor_model"]) {
    std::string type;
    load(n, "type", type);
    if (!type.empty()) {
      if (type == "lidar" || type == "laser") cfg.sensor_model.type = SensorType::LiDAR;
      else if (type == "rgbd") cfg.sensor_model.type = SensorType::RGBD;
      else if (type == "constant" || type == "none") cfg.sensor_model.type = SensorType::Constant;
      else { warn("Unknown sensor_model.type '" + type + "', defaulting to LiDAR"); cfg.sensor_model.type = SensorType::LiDAR; }
    }
    if (const auto& l = n["lidar"]) {
      load(l, "range_noise", cfg.sensor_model.lidar.range_noise);
      load(l, "angular_noise", cfg.sensor_model.lidar.angular_noise);
    }
    if (const auto& r = n["rgbd"]) {
      load(r, "normal_a", cfg.sensor_model.rgbd.normal_a);
      load(r, "normal_b", cfg.sensor_model.rgbd.normal_b);
      load(r, "normal_c", cfg.sensor_model.rgbd.normal_c);
      load(r, "lateral_factor", cfg.sensor_model.rgbd.lateral_factor);
    }
    if (const auto& c = n["constant"]) load(c, "uncertainty", cfg.sensor_model.constant.uncertainty);
  }
  return cfg;
}
}  // namespace detail

/// parseConfig (config_fastdem.cpp:262-266): parse + validate.
inline Config parseConfig(const yaml::Node& root) {
  Config cfg = detail::parse(root);
  detail::validate(cfg);
  return cfg;
}
/// The same from YAML text.
inline Config parseConfigText(const std::string& text) { return parseConfig(yaml::parse(text)); }
/// loadConfig (config_fastdem.cpp:268-275): std::runtime_error when the file cannot be read or parsed.
inline Config loadConfig(const std::string& path) {
  try {
    return parseConfig(yaml::loadFile(path));
  } catch (const yaml::Error& e) {
    throw std::runtime_error("Failed to load config: " + path + " - " + e.what());
  }
}

/// Validate + clamp a programmatically built config exactly like parseConfig() does after parsing.
inline Config validated(Config cfg) {
  detail::validate(cfg);
  return cfg;
}

}  // namespace fastdem
