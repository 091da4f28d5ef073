// fastdem/elevation_map.hpp — fastdem::ElevationMap (fastdem/include/fastdem/elevation_map.hpp:28-181):
// layer-name constants, CellMap, and the 2.5-D map type, over the device-resident GridMap.
#pragma once
#include <cmath>
#include <initializer_list>
#include <string>
#include <unordered_map>

#include "nanogrid/nanogrid.hpp"

namespace fastdem {

namespace layer {
constexpr auto elevation = "elevation";
constexpr auto elevation_min = "elevation_min";
constexpr auto elevation_max = "elevation_max";
constexpr auto variance = "variance";
constexpr auto n_points = "n_points";
constexpr auto upper_bound = "upper_bound";
constexpr auto lower_bound = "lower_bound";
constexpr auto obstacle = "obstacle";
constexpr auto intensity = "intensity";
constexpr auto color = "color";
// estimator-internal layers (kalman_estimation.hpp:27-31, quantile_estimation.hpp:25-36)
constexpr auto kalman_p = "_kalman_p";
constexpr auto sample_mean = "_sample_mean";
constexpr auto sample_m2 = "_sample_m2";
constexpr auto p2_q0 = "_p2_q0";
constexpr auto p2_q1 = "_p2_q1";
constexpr auto p2_q2 = "_p2_q2";
constexpr auto p2_q3 = "_p2_q3";
constexpr auto p2_q4 = "_p2_q4";
constexpr auto p2_n0 = "_p2_n0";
constexpr auto p2_n1 = "_p2_n1";
constexpr auto p2_n2 = "_p2_n2";
constexpr auto p2_n3 = "_p2_n3";
constexpr auto p2_n4 = "_p2_n4";
inline bool isInternal(const std::string& name) { return !name.empty() && name[0] == '_'; }
}  // namespace layer

template <typename T>
using CellMap = std::unordered_map<nanogrid::Index, T, nanogrid::IndexHash, nanogrid::IndexEqual>;

class ElevationMap : public nanogrid::GridMap {
 public:
  ElevationMap() : nanogrid::GridMap({layer::elevation, layer::elevation_min, layer::elevation_max}) {}
  ElevationMap(float width, float height, float resolution, const std::string& frame_id) : ElevationMap() {
    setGeometry(width, height, resolution);
    setFrameId(frame_id);
  }
  ElevationMap(ElevationMap&&) = default;
  ElevationMap& operator=(ElevationMap&&) = default;
  ElevationMap(const ElevationMap&) = default;             // a deep copy, device to device (nanogrid::GridMap::copyFrom)
  ElevationMap& operator=(const ElevationMap&) = default;

  /// float arguments are promoted to double exactly like the reference (elevation_map.hpp:112-116).
  void setGeometry(float width, float height, float resolution) {
    nanogrid::GridMap::setGeometry(nanogrid::Length(width, height), resolution);
    clearAll();
  }
  bool isInitialized() const { return rows_ > 0 && cols_ > 0; }
  bool isEmpty() const { return get(layer::elevation).allNaN(); }
  bool isEmptyAt(const nanogrid::Index& i) const { return std::isnan(at(layer::elevation, i)); }
  void clearAt(const nanogrid::Index& i) {
    for (const auto& l : getLayers()) at(l, i) = NAN;
  }
  float elevationAt(const nanogrid::Position& p) const {
    if (!isInside(p)) return NAN;
    return atPosition(layer::elevation, p);
  }
  float elevationAt(const nanogrid::Index& i) const { return at(layer::elevation, i); }
  bool hasElevationAt(const nanogrid::Position& p) const { return std::isfinite(elevationAt(p)); }
  bool hasElevationAt(const nanogrid::Index& i) const { return std::isfinite(elevationAt(i)); }
  nanogrid::Matrix isFinite(const std::string& l) const {
    const auto& d = get(l);
    nanogrid::Matrix m(d.rows(), d.cols(), 0.0f);
    for (size_t k = 0; k < d.size(); ++k) m.data()[k] = std::isnan(d.data()[k]) ? 0.0f : 1.0f;
    return m;
  }
  /// Lightweight copy with only the given layers (own device map).
  ElevationMap snapshot(std::initializer_list<std::string> layers) const {
    ElevationMap snap;
    snap.setGeometry(float(getLength()(0)), float(getLength()(1)), float(getResolution()));
    snap.setFrameId(getFrameId());
    snap.setPosition(getPosition());
    snap.setStartIndex(getStartIndex());
    snap.setTimestamp(getTimestamp());
    const_cast<ElevationMap*>(this)->flushToDevice();
    for (const auto& name : layers) {
      if (!exists(name)) continue;
      ck(fdm_engine_layer_copy(snap.eng_, eng_, name.c_str()), "layer_copy");  // device to device
    }
    return snap;
  }
};

}  // namespace fastdem
