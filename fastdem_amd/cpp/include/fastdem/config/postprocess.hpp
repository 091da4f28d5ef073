// fastdem/config/postprocess.hpp — only config::Raycasting is part of fastdem::Config
// (fastdem/include/fastdem/config/postprocess.hpp:16-23); the stage runs on the device
// (fdm_engine.h raycasting section).
#pragma once
namespace fastdem::config {
struct Raycasting {
  bool enabled = false;
  float height_conflict_threshold = 0.05f;
  float log_odds_observed = 0.4f;
  float log_odds_ghost = 0.2f;
  float log_odds_max = 2.0f;
  float clear_threshold = -1.0f;
};
}  // namespace fastdem::config

// config/postprocess.hpp:25-52 — parameters of the stencil stages
namespace fastdem::config {
struct Inpainting {
  bool enabled = false;
  int max_iterations = 3;
  int min_valid_neighbors = 2;
};
struct UncertaintyFusion {
  bool enabled = false;
  float search_radius = 0.15f;
  float spatial_sigma = 0.05f;
  float quantile_lower = 0.01f;
  float quantile_upper = 0.99f;
  int min_valid_neighbors = 3;
};
struct FeatureExtraction {
  bool enabled = false;
  float analysis_radius = 0.3f;
  int min_valid_neighbors = 4;
  float step_lower_percentile = 0.05f;
  float step_upper_percentile = 0.95f;
};
struct PostProcess {
  Inpainting inpainting;
  UncertaintyFusion uncertainty_fusion;
  FeatureExtraction feature_extraction;
};
}  // namespace fastdem::config

// parsePostProcess / loadPostProcess (fastdem/src/config_postprocess.cpp:20-140): every key optional,
// out-of-range values warn and are clamped (nothing throws but an unreadable / malformed file).
#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "fastdem/config/yaml_lite.hpp"

namespace fastdem::config {
namespace detail {
template <typename T>
inline void ppLoad(const yaml::Node& node, const std::string& key, T& value) {
  if (node[key]) value = node[key].as<T>();
}
inline void ppWarn(const std::string& m) { std::fprintf(stderr, "[warn] [PostProcess] %s\n", m.c_str()); }
inline void validate(PostProcess& cfg) {
  auto positive = [](const char* name, float& v, float fallback) {
    if (v <= 0.0f) { ppWarn(std::string(name) + " must be > 0, clamping to " + std::to_string(fallback)); v = fallback; }
  };
  auto at_least = [](const char* name, int& v, int lo) {
    if (v < lo) { ppWarn(std::string(name) + " must be >= " + std::to_string(lo) + ", clamping"); v = lo; }
  };
  at_least("inpainting.max_iterations", cfg.inpainting.max_iterations, 1);
  at_least("inpainting.min_valid_neighbors", cfg.inpainting.min_valid_neighbors, 1);
  positive("uncertainty_fusion.search_radius", cfg.uncertainty_fusion.search_radius, 0.15f);
  positive("uncertainty_fusion.spatial_sigma", cfg.uncertainty_fusion.spatial_sigma, 0.05f);
  at_least("uncertainty_fusion.min_valid_neighbors", cfg.uncertainty_fusion.min_valid_neighbors, 1);
  auto& ql = cfg.uncertainty_fusion.quantile_lower;
  auto& qu = cfg.uncertainty_fusion.quantile_upper;
  ql = std::clamp(ql, 0.0f, 1.0f);
  qu = std::clamp(qu, 0.0f, 1.0f);
  if (ql >= qu) { ppWarn("uncertainty_fusion.quantile_lower >= quantile_upper, resetting to defaults"); ql = 0.01f; qu = 0.99f; }
  positive("feature_extraction.analysis_radius", cfg.feature_extraction.analysis_radius, 0.3f);
  at_least("feature_extraction.min_valid_neighbors", cfg.feature_extraction.min_valid_neighbors, 3);
  auto& sl = cfg.feature_extraction.step_lower_percentile;
  auto& su = cfg.feature_extraction.step_upper_percentile;
  sl = std::clamp(sl, 0.0f, 1.0f);
  su = std::clamp(su, 0.0f, 1.0f);
  if (sl >= su) { ppWarn("feature_extraction.step_lower_percentile >= step_upper_percentile, resetting to defaults"); sl = 0.05f; su = 0.95f; }
}
}  // namespace detail

inline PostProcess parsePostProcess(const yaml::Node& root) {
  PostProcess cfg;
  if (const auto& n = root["inpainting"]) {
    detail::ppLoad(n, "enabled", cfg.inpainting.enabled);
    detail::ppLoad(n, "max_iterations", cfg.inpainting.max_iterations);
    detail::ppLoad(n, "min_valid_neighbors", cfg.inpainting.min_valid_neighbors);
  }
  if (const auto& n = root["uncertainty_fusion"]) {
    detail::ppLoad(n, "enabled", cfg.uncertainty_fusion.enabled);
    detail::ppLoad(n, "search_radius", cfg.uncertainty_fusion.search_radius);
    detail::ppLoad(n, "spatial_sigma", cfg.uncertainty_fusion.spatial_sigma);
    detail::ppLoad(n, "quantile_lower", cfg.uncertainty_fusion.quantile_lower);
    detail::ppLoad(n, "quantile_upper", cfg.uncertainty_fusion.quantile_upper);
    detail::ppLoad(n, "min_valid_neighbors", cfg.uncertainty_fusion.min_valid_neighbors);
  }
  if (const auto& n = root["feature_extraction"]) {
    detail::ppLoad(n, "enabled", cfg.feature_extraction.enabled);
    detail::ppLoad(n, "analysis_radius", cfg.feature_extraction.analysis_radius);
    detail::ppLoad(n, "min_valid_neighbors", cfg.feature_extraction.min_valid_neighbors);
    detail::ppLoad(n, "step_lower_percentile", cfg.feature_extraction.step_lower_percentile);
    detail::ppLoad(n, "step_upper_percentile", cfg.feature_extraction.step_upper_percentile);
  }
  detail::validate(cfg);
  return cfg;
}
inline PostProcess loadPostProcess(const std::string& path) {
  try {
    return parsePostProcess(yaml::loadFile(path));
  } catch (const yaml::Error& e) {
    throw std::runtime_error("Failed to load postprocess config: " + path + " - " + e.what());
  }
}
}  // namespace fastdem::config
