// fastdem/postprocess/raycasting.hpp — applyRaycasting over the device engine
// (fastdem/include/fastdem/postprocess/raycasting.hpp:27-49, src/raycasting.cpp:204-249).
// The whole stage (observed evidence, DDA min ray height, ghost resolution, clearAt) runs in HBM;
// host-side writes to the map are flushed first and the host mirror is dropped afterwards.
#pragma once
#include "fastdem/config/postprocess.hpp"
#include "fastdem/elevation_map.hpp"
#include "fastdem/point_types.hpp"

namespace fastdem {

namespace layer {
constexpr auto ghost_removal = "ghost_removal";
constexpr auto raycasting = "raycasting";
constexpr auto visibility_logodds = "_visibility_logodds";
}  // namespace layer

inline void applyRaycasting(ElevationMap& map, const PointCloud& scan, const Eigen::Vector3f& sensor_origin,
                            const config::Raycasting& config) {
  if (!config.enabled || scan.empty()) return;
  fdm_raycast_config rc;
  rc.enabled = 1;
  rc.height_conflict_threshold = config.height_conflict_threshold;
  rc.log_odds_observed = config.log_odds_observed;
  rc.log_odds_ghost = config.log_odds_ghost;
  rc.log_odds_max = config.log_odds_max;
  rc.clear_threshold = config.clear_threshold;
  const float origin[3] = {sensor_origin(0), sensor_origin(1), sensor_origin(2)};
  map.flushToDevice();
  const int st = fdm_engine_apply_raycasting(map.engine(), scan.size(), scan.xData(), scan.yData(), scan.zData(),
                                             origin, &rc);
  if (st < 0) throw nanogrid::EngineError(std::string("fdm_engine_apply_raycasting: ") + fdm_last_error());
  map.invalidateHost();
}

}  // namespace fastdem
