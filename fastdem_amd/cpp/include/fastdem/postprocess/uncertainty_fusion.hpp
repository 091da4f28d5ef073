// fastdem/postprocess/uncertainty_fusion.hpp — applyUncertaintyFusion over the device engine
// (fastdem/src/uncertainty_fusion.cpp:103-186): weighted-ECDF fusion of the estimator bounds.
#pragma once
#include <cstdio>

#include "fastdem/config/postprocess.hpp"
#include "fastdem/elevation_map.hpp"

namespace fastdem {
inline void applyUncertaintyFusion(ElevationMap& map, const config::UncertaintyFusion& config) {
  if (!config.enabled) return;
  if (!map.hasEngine() || !map.exists(layer::upper_bound) || !map.exists(layer::lower_bound)) {
    std::fprintf(stderr, "[warn] [UncertaintyFusion] Missing required layers (upper_bound, lower_bound).\n");
    return;
  }
  fdm_fusion_config c;
  c.enabled = 1;
  c.search_radius = config.search_radius;
  c.spatial_sigma = config.spatial_sigma;
  c.quantile_lower = config.quantile_lower;
  c.quantile_upper = config.quantile_upper;
  c.min_valid_neighbors = config.min_valid_neighbors;
  map.flushToDevice();
  const int rc = fdm_engine_apply_uncertainty_fusion(map.engine(), &c);
  if (rc < 0) throw nanogrid::EngineError(std::string("fdm_engine_apply_uncertainty_fusion: ") + fdm_last_error());
  map.invalidateHost();
}
}  // namespace fastdem
