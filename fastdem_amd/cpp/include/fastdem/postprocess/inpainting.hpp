// fastdem/postprocess/inpainting.hpp — applyInpainting over the device engine
// (fastdem/include/fastdem/postprocess/inpainting.hpp:21-43, src/inpainting.cpp:21-67).
#pragma once
#include "fastdem/elevation_map.hpp"

namespace fastdem {
namespace layer {
constexpr auto elevation_inpainted = "elevation_inpainted";
}
inline void applyInpainting(ElevationMap& map, int max_iterations = 3, int min_valid_neighbors = 2,
                            bool inplace = false) {
  map.flushToDevice();
  const int rc = fdm_engine_apply_inpainting(map.engine(), max_iterations, min_valid_neighbors, inplace ? 1 : 0);
  if (rc < 0) throw nanogrid::EngineError(std::string("fdm_engine_apply_inpainting: ") + fdm_last_error());
  map.invalidateHost();
}
}  // namespace fastdem
