// fastdem/postprocess/spatial_smoothing.hpp — applySpatialSmoothing over the device engine
// (fastdem/include/fastdem/postprocess/spatial_smoothing.hpp:38-67): in-place median filter.
#pragma once
#include <string>

#include "fastdem/elevation_map.hpp"

namespace fastdem {
inline void applySpatialSmoothing(ElevationMap& map, const std::string& layer_name, int kernel_size = 3,
                                  int min_valid_neighbors = 5) {
  if (!map.hasEngine() || !map.exists(layer_name)) return;
  map.flushToDevice();
  const int rc = fdm_engine_apply_spatial_smoothing(map.engine(), layer_name.c_str(), kernel_size, min_valid_neighbors);
  if (rc < 0) throw nanogrid::EngineError(std::string("fdm_engine_apply_spatial_smoothing: ") + fdm_last_error());
  map.invalidateHost();
}
}  // namespace fastdem
