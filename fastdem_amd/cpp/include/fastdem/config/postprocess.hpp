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
