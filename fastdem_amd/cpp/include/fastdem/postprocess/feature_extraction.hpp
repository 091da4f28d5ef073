// fastdem/postprocess/feature_extraction.hpp — applyFeatureExtraction over the device engine
// (fastdem/include/fastdem/postprocess/feature_extraction.hpp:11-36, src/feature_extraction.cpp:28-118).
#pragma once
#include "fastdem/elevation_map.hpp"

namespace fastdem {
namespace layer {
constexpr auto step = "step";
constexpr auto slope = "slope";
constexpr auto roughness = "roughness";
constexpr auto curvature = "curvature";
constexpr auto normal_x = "_normal_x";
constexpr auto normal_y = "_normal_y";
constexpr auto normal_z = "_normal_z";
}  // namespace layer
inline void applyFeatureExtraction(ElevationMap& map, float analysis_radius = 0.3f, int min_valid_neighbors = 4,
                                   float step_lower_percentile = 0.05f, float step_upper_percentile = 0.95f) {
  if (!map.hasEngine() || !map.exists(layer::elevation)) return;  // also the default-constructed map
  map.flushToDevice();
  const int rc = fdm_engine_apply_feature_extraction(map.engine(), analysis_radius, min_valid_neighbors,
                                                     step_lower_percentile, step_upper_percentile);
  if (rc < 0) throw nanogrid::EngineError(std::string("fdm_engine_apply_feature_extraction: ") + fdm_last_error());
  map.invalidateHost();
}
}  // namespace fastdem
