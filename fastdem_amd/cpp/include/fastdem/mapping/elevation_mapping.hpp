// fastdem/mapping/elevation_mapping.hpp — ElevationMapping over the device engine
// (fastdem/include/fastdem/mapping/elevation_mapping.hpp:21-59, src/elevation_mapping.cpp).
// update() = LOCAL-mode move + rasterize + estimate + min/max/obstacle/intensity/colour, all on
// the GPU.  The per-cell observations stay on the device; what comes back is their count.
#pragma once
#include <memory>

#include "fastdem/config/fastdem.hpp"
#include "fastdem/elevation_map.hpp"
#include "fastdem/point_types.hpp"

namespace fastdem {

namespace detail {
inline fdm_config toEngineConfig(const Config& c) {
  fdm_config f;
  fdm_default_config(&f);
  f.z_min = c.point_filter.z_min;
  f.z_max = c.point_filter.z_max;
  f.range_min = c.point_filter.range_min;
  f.range_max = c.point_filter.range_max;
  f.sensor_type = c.sensor_model.type == SensorType::Constant ? 0 : (c.sensor_model.type == SensorType::LiDAR ? 1 : 2);
  f.lidar_range_noise = c.sensor_model.lidar.range_noise;
  f.lidar_angular_noise = c.sensor_model.lidar.angular_noise;
  f.rgbd_normal_a = c.sensor_model.rgbd.normal_a;
  f.rgbd_normal_b = c.sensor_model.rgbd.normal_b;
  f.rgbd_normal_c = c.sensor_model.rgbd.normal_c;
  f.rgbd_lateral_factor = c.sensor_model.rgbd.lateral_factor;
  f.constant_uncertainty = c.sensor_model.constant.uncertainty;
  f.mode = c.mapping.mode == MappingMode::LOCAL ? 0 : 1;
  f.estimation_type = c.mapping.estimation_type == EstimationType::Kalman ? 0 : 1;
  f.kalman_min_variance = c.mapping.kalman.min_variance;
  f.kalman_max_variance = c.mapping.kalman.max_variance;
  f.kalman_process_noise = c.mapping.kalman.process_noise;
  f.p2_dn[0] = c.mapping.p2.dn0;
  f.p2_dn[1] = c.mapping.p2.dn1;
  f.p2_dn[2] = c.mapping.p2.dn2;
  f.p2_dn[3] = c.mapping.p2.dn3;
  f.p2_dn[4] = c.mapping.p2.dn4;
  f.p2_elevation_marker = c.mapping.p2.elevation_marker;
  f.p2_max_sample_count = c.mapping.p2.max_sample_count;
  f.raycast_enabled = c.raycasting.enabled ? 1 : 0;
  f.rc_height_conflict_threshold = c.raycasting.height_conflict_threshold;
  f.rc_log_odds_observed = c.raycasting.log_odds_observed;
  f.rc_log_odds_ghost = c.raycasting.log_odds_ghost;
  f.rc_log_odds_max = c.raycasting.log_odds_max;
  f.rc_clear_threshold = c.raycasting.clear_threshold;
  return f;
}
inline void ck(int rc, const char* what) {
  if (rc < 0) throw nanogrid::EngineError(std::string(what) + ": " + fdm_last_error());
}
}  // namespace detail

class ElevationMapping {
 public:
  /// What update() reports back: the number of cells the scan observed (the reference returns
  /// the CellMap itself; its entries stay in HBM here).
  struct CellObservations {
    size_t n_cells = 0;
    size_t n_points_in_map = 0;
    size_t size() const { return n_cells; }
    bool empty() const { return n_cells == 0; }
  };

  ElevationMapping(ElevationMap& map, const config::Mapping& cfg) : map_(map) {
    cfg_.mapping = cfg;
    apply();
  }
  /// Used by FastDEM: whole Config (filters + sensor model + mapping).
  ElevationMapping(ElevationMap& map, const Config& cfg) : map_(map), cfg_(cfg) { apply(); }

  void setConfig(const Config& cfg) {
    cfg_ = cfg;
    apply();
  }

  /// ElevationMapping::update(cloud, robot_position) on a cloud already in the map frame
  /// (elevation_mapping.cpp:110-125); clouds carry no covariance channel here, so pt_z_var = 0.
  CellObservations update(const PointCloud& cloud, const Eigen::Vector2d& robot_position) {
    map_.flushToDevice();
    fdm_scan_stats st{};
    detail::ck(fdm_engine_update(map_.engine(), cloud.size(), cloud.xData(), cloud.yData(), cloud.zData(),
                                 nullptr, cloud.intensityData(), cloud.rgbData(), robot_position(0),
                                 robot_position(1), &st),
               "fdm_engine_update");
    map_.invalidateHost();
    CellObservations o;
    o.n_cells = st.n_cells_touched;
    o.n_points_in_map = st.n_in_map;
    return o;
  }

 private:
  void apply() {
    if (!map_.hasEngine()) throw nanogrid::EngineError("ElevationMapping: the map has no geometry yet");
    const fdm_config f = detail::toEngineConfig(cfg_);
    detail::ck(fdm_engine_set_config(map_.engine(), &f), "fdm_engine_set_config");
  }
  ElevationMap& map_;
  Config cfg_;
};

inline std::unique_ptr<ElevationMapping> createElevationMapping(ElevationMap& map, const config::Mapping& cfg) {
  return std::make_unique<ElevationMapping>(map, cfg);
}

}  // namespace fastdem
