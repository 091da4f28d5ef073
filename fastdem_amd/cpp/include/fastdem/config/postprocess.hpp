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
