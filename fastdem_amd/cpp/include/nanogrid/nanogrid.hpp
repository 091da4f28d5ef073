// nanogrid/nanogrid.hpp — the nanogrid::GridMap surface FastDEM callers use,
// backed by the DEVICE-RESIDENT map of libfdm_engine.so.  Layers live in HBM; the host keeps a
// lazily synchronised mirror so `get(layer)` can still hand out a mutable matrix:
//   device -> host : on the first get()/at() after a device-side change (one download per layer)
//   host -> device : non-const access marks the layer host-dirty; FastDEM::integrate() /
//                    ElevationMapping::update() upload dirty layers before launching kernels.
// A `Matrix&` keeps its ADDRESS for the life of the map (the reference hands out references that stay live across
// integrate(): estimators bind raw pointers, kalman_estimation.hpp:85-95, callers hold references): the host copy of a
// layer is refreshed IN PLACE at the first host access — get() / at() of ANY layer, by anyone — after the device
// changed the map, so a reference held across an integrate() reads fresh data from then on.  What it cannot do is
// refresh on a bare dereference; a reference dereferenced between the device-side change and the next get() / at()
// would read the old values — builds without NDEBUG (or with FDM_MIRROR_GUARD) throw std::logic_error there instead.
// GridMap is copyable (a deep copy, device to device): snapshot() by value, the ROS node's copy under a shared lock
// (elevation_map.hpp:95-99, ros1/src/fastdem_ros_node.cpp:192-199).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "fastdem/compat/mini_eigen.hpp"
#include "fdm_engine.h"

namespace nanogrid {

using Index = Eigen::Array2i;
using Size = Eigen::Array2i;
using Position = Eigen::Vector2d;
using Length = Eigen::Array2d;

struct IndexHash {
  size_t operator()(const Index& i) const { return std::hash<uint64_t>()((uint64_t(uint32_t(i(0))) << 32) | uint32_t(i(1))); }
};
struct IndexEqual {
  bool operator()(const Index& a, const Index& b) const { return a(0) == b(0) && a(1) == b(1); }
};

// Column-major float matrix (Eigen::MatrixXf storage order).
class Matrix {
 public:
  Matrix() = default;
  Matrix(int r, int c, float v = NAN) : r_(r), c_(c), d_(size_t(r) * size_t(c), v) {}
  // a copy is a value of its own (it watches no map); assigning INTO a layer's host copy keeps that copy's watch
  Matrix(const Matrix& o) : r_(o.r_), c_(o.c_), d_((o.guard(), o.d_)) {}
  Matrix(Matrix&& o) noexcept : r_(o.r_), c_(o.c_), d_(std::move(o.d_)) { o.r_ = o.c_ = 0; }
  Matrix& operator=(const Matrix& o) {
    if (this != &o) { o.guard(); guard(); r_ = o.r_; c_ = o.c_; d_ = o.d_; }
    return *this;
  }
  Matrix& operator=(Matrix&& o) noexcept {
    if (this != &o) { r_ = o.r_; c_ = o.c_; d_ = std::move(o.d_); o.r_ = o.c_ = 0; }
    return *this;
  }
  int rows() const { return r_; }
  int cols() const { return c_; }
  size_t size() const { return d_.size(); }
  float* data() { guard(); return d_.data(); }
  const float* data() const { guard(); return d_.data(); }
  float& operator()(int i, int j) { guard(); return d_[size_t(j) * r_ + i]; }
  float operator()(int i, int j) const { guard(); return d_[size_t(j) * r_ + i]; }
  float& operator()(const Index& i) { return (*this)(i(0), i(1)); }
  float operator()(const Index& i) const { return (*this)(i(0), i(1)); }
  void setConstant(float v) { guard(); std::fill(d_.begin(), d_.end(), v); }
  // (a layer's host copy: GridMap sets the flag when the device changes the map and clears it when it has refreshed the
  // copy in place; an access in between reads stale data — debug builds refuse)
  void watch(const bool* stale) { stale_ = stale; }
  float* raw() { return d_.data(); }
  bool allNaN() const { guard(); for (float v : d_) if (!std::isnan(v)) return false; return true; }
  size_t countFinite() const { guard(); size_t n = 0; for (float v : d_) n += std::isfinite(v) ? 1 : 0; return n; }

 private:
  void guard() const {
#if !defined(NDEBUG) || defined(FDM_MIRROR_GUARD)
    if (stale_ && *stale_)
      throw std::logic_error("nanogrid::Matrix: a reference obtained before integrate() / move() was dereferenced before the "
                             "next GridMap::get() / at(): it would read the map as it was (call get() again)");
#endif
  }
  int r_ = 0, c_ = 0;
  std::vector<float> d_;
  const bool* stale_ = nullptr;
};

// nanogrid::colorVectorToValue: 0x00RRGGBB bit-cast to float (bridge/ros/impl.hpp:20-21)
inline bool colorVectorToValue(const Eigen::Vector3i& rgb, float& value) {
  const uint32_t packed = (uint32_t(rgb[0]) << 16) | (uint32_t(rgb[1]) << 8) | uint32_t(rgb[2]);
  std::memcpy(&value, &packed, 4);
  return true;
}

class EngineError : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

class GridMap {
 public:
  GridMap() = default;
  explicit GridMap(const std::vector<std::string>& layers) : initial_layers_(layers) {}
  // a deep copy: an engine of its own, every layer copied device to device (host writes of `o` that have not reached
  // the device yet go there first)
  GridMap(const GridMap& o) { copyFrom(o); }
  GridMap& operator=(const GridMap& o) {
    if (this != &o) copyFrom(o);
    return *this;
  }
  GridMap(GridMap&& o) noexcept { *this = std::move(o); }
  GridMap& operator=(GridMap&& o) noexcept {
    if (this != &o) {
      release();
      eng_ = o.eng_; o.eng_ = nullptr;
      rows_ = o.rows_; cols_ = o.cols_; res_ = o.res_; length_ = o.length_;
      frame_id_ = std::move(o.frame_id_); timestamp_ = o.timestamp_;
      mirror_ = std::move(o.mirror_); initial_layers_ = std::move(o.initial_layers_);
      pending_pos_ = o.pending_pos_; move_clear_basic_ = o.move_clear_basic_; device_ = o.device_;
    }
    return *this;
  }
  ~GridMap() { release(); }

  // ---- geometry ----
  void setGeometry(const Length& length, double resolution, const Position& position = Position(0.0, 0.0)) {
    release();
    fdm_geometry g{};
    g.length_x = length(0); g.length_y = length(1); g.resolution = resolution;
    g.position_x = position(0); g.position_y = position(1);
    ck(fdm_engine_create_map(&g, nullptr, device_, &eng_), "fdm_engine_create_map");
    fdm_geometry out{};
    ck(fdm_engine_get_geometry(eng_, &out), "get_geometry");
    rows_ = out.rows; cols_ = out.cols; res_ = out.resolution;
    length_ = Length(out.length_x, out.length_y);
    mirror_.clear();
    for (const auto& n : initial_layers_)
      if (!exists(n)) add(n);
    if (move_clear_basic_) setMoveClearBasic(true);
  }
  // Which layers move() clears in the strips it vacates: every layer (default, what this repo assumes of nanoGrid) or
  // the basic layers {elevation, elevation_min, elevation_max} only — the other reading (scripts/conformance/probe.cpp
  // tells which one the real library implements)
  void setMoveClearBasic(bool on) {
    move_clear_basic_ = on;
    if (eng_) ck(fdm_engine_set_option(eng_, "move_clear_basic", on ? 1 : 0), "set_option(move_clear_basic)");
  }
  bool hasEngine() const { return eng_ != nullptr; }
  fdm_engine* engine() const { return eng_; }
  void setDevice(int d) { device_ = d; }

  Size getSize() const { return Size(rows_, cols_); }
  Length getLength() const { return length_; }
  double getResolution() const { return res_; }
  Position getPosition() const { const fdm_geometry g = geom(); return Position(g.position_x, g.position_y); }
  Index getStartIndex() const { const fdm_geometry g = geom(); return Index(g.start_row, g.start_col); }
  void setPosition(const Position& p) { need(); ck(fdm_engine_set_position(eng_, p(0), p(1)), "set_position"); }
  void setStartIndex(const Index& i) { need(); ck(fdm_engine_set_start_index(eng_, i(0), i(1)), "set_start_index"); }
  const std::string& getFrameId() const { return frame_id_; }
  void setFrameId(const std::string& f) { frame_id_ = f; }
  uint64_t getTimestamp() const { return timestamp_; }
  void setTimestamp(uint64_t t) { timestamp_ = t; }

  // grid_map_core arithmetic on the host copy of the geometry (same formulas as the kernels)
  bool isInside(const Position& p) const {
    const fdm_geometry g = geom();
    const double tx = -((p(0) - g.position_x) - 0.5 * g.length_x);
    const double ty = -((p(1) - g.position_y) - 0.5 * g.length_y);
    return tx >= 0.0 && ty >= 0.0 && tx < g.length_x && ty < g.length_y;
  }
  bool getIndex(const Position& p, Index& idx) const {
    if (!isInside(p)) return false;
    const fdm_geometry g = geom();
    int r = static_cast<int>(-(((p(0) - 0.5 * g.length_x) - g.position_x) / g.resolution));
    int c = static_cast<int>(-(((p(1) - 0.5 * g.length_y) - g.position_y) / g.resolution));
    if (g.start_row != 0 || g.start_col != 0) {
      r += g.start_row; c += g.start_col;
      if (r >= g.rows) r -= g.rows;
      if (c >= g.cols) c -= g.cols;
    }
    idx = Index(r, c);
    return r >= 0 && c >= 0 && r < g.rows && c < g.cols;
  }
  bool getPosition(const Index& idx, Position& p) const {
    const fdm_geometry g = geom();
    if (idx(0) < 0 || idx(1) < 0 || idx(0) >= g.rows || idx(1) >= g.cols) return false;
    int ur = idx(0) - g.start_row, uc = idx(1) - g.start_col;
    if (ur < 0) ur += g.rows;
    if (uc < 0) uc += g.cols;
    p = Position(g.position_x + (0.5 * g.length_x - 0.5 * g.resolution) + g.resolution * double(-ur),
                 g.position_y + (0.5 * g.length_y - 0.5 * g.resolution) + g.resolution * double(-uc));
    return true;
  }
  bool move(const Position& p) {
    need();
    flushToDevice();
    const Index before = getStartIndex();
    ck(fdm_engine_move(eng_, p(0), p(1)), "move");
    invalidateHost();
    const Index after = getStartIndex();
    return before(0) != after(0) || before(1) != after(1);
  }

  // ---- layers ----
  std::vector<std::string> getLayers() const {
    std::vector<std::string> out;
    if (!eng_) return out;
    const int n = fdm_engine_num_layers(eng_);
    for (int i = 0; i < n; ++i) out.emplace_back(fdm_engine_layer_name(eng_, i));
    return out;
  }
  bool exists(const std::string& n) const { return eng_ && fdm_engine_layer_exists(eng_, n.c_str()) == 1; }
  void add(const std::string& n, float value = NAN) {
    need();
    ck(fdm_engine_layer_add(eng_, n.c_str(), value), "layer_add");
    outdated(n);
  }
  void add(const std::string& n, const Matrix& m) {
    need();
    if (m.rows() != rows_ || m.cols() != cols_) throw std::invalid_argument("layer shape mismatch");
    ck(fdm_engine_layer_upload(eng_, n.c_str(), m.data(), rows_, cols_), "layer_upload");
    outdated(n);
  }
  Matrix& get(const std::string& n) {
    Mirror& m = fetch(n);
    m.dirty = true;  // the caller may write through the reference
    return m.host;
  }
  const Matrix& get(const std::string& n) const { return const_cast<GridMap*>(this)->fetch(n).host; }
  float& at(const std::string& n, const Index& i) { return get(n)(i); }
  float at(const std::string& n, const Index& i) const { return get(n)(i); }
  float& atPosition(const std::string& n, const Position& p) {
    Index i;
    if (!getIndex(p, i)) throw std::out_of_range("position outside the map");
    return at(n, i);
  }
  float atPosition(const std::string& n, const Position& p) const {
    Index i;
    if (!getIndex(p, i)) throw std::out_of_range("position outside the map");
    return at(n, i);
  }
  void clear(const std::string& n) {
    need();
    ck(fdm_engine_clear(eng_, n.c_str()), "clear");
    outdated(n);
  }
  void clearAll() {
    if (!eng_) return;
    ck(fdm_engine_clear(eng_, nullptr), "clearAll");
    invalidateHost();
  }

  // ---- coherence hooks used by FastDEM / ElevationMapping ----
  void flushToDevice() {
    if (!eng_) return;
    for (auto& kv : mirror_)
      if (kv.second->dirty && !kv.second->stale) {
        ck(fdm_engine_layer_upload(eng_, kv.first.c_str(), kv.second->host.raw(), rows_, cols_), "layer_upload");
        kv.second->dirty = false;
      }
  }
  // the device changed the map: every host copy is out of date until the next host access refreshes it in place
  void invalidateHost() {
    for (auto& kv : mirror_) { kv.second->stale = true; kv.second->dirty = false; }
    any_stale_ = !mirror_.empty();
  }

 protected:
  struct Mirror {
    Matrix host;
    bool dirty = false;
    bool stale = false;   // the device holds newer data (Matrix::guard watches this flag)
  };
  void outdated(const std::string& n) {
    auto it = mirror_.find(n);
    if (it != mirror_.end()) { it->second->stale = true; it->second->dirty = false; any_stale_ = true; }
  }
  // the first host access after a device-side change refreshes EVERY host copy in place (a layer that no longer exists
  // reads as NaN): whoever holds a Matrix& of any layer reads fresh data from here on
  void refresh() {
    if (!any_stale_) return;
    for (auto& kv : mirror_) {
      Mirror& m = *kv.second;
      if (!m.stale) continue;
      if (exists(kv.first)) ck(fdm_engine_layer_download(eng_, kv.first.c_str(), m.host.raw(), rows_, cols_), "layer_download");
      else std::fill(m.host.raw(), m.host.raw() + m.host.size(), NAN);
      m.stale = false;
    }
    any_stale_ = false;
  }
  Mirror& fetch(const std::string& n) {
    need();
    refresh();
    auto it = mirror_.find(n);
    if (it != mirror_.end()) {
      if (!exists(n)) throw std::out_of_range("GridMap::get(): no layer '" + n + "'");
      return *it->second;
    }
    if (!exists(n)) throw std::out_of_range("GridMap::get(): no layer '" + n + "'");
    std::unique_ptr<Mirror> m(new Mirror);
    m->host = Matrix(rows_, cols_);
    ck(fdm_engine_layer_download(eng_, n.c_str(), m->host.raw(), rows_, cols_), "layer_download");
    m->host.watch(&m->stale);
    return *mirror_.emplace(n, std::move(m)).first->second;
  }
  void copyFrom(const GridMap& o) {
    release();
    rows_ = cols_ = 0;
    frame_id_ = o.frame_id_; timestamp_ = o.timestamp_; initial_layers_ = o.initial_layers_;
    pending_pos_ = o.pending_pos_; device_ = o.device_; move_clear_basic_ = o.move_clear_basic_;
    if (!o.eng_) return;
    const_cast<GridMap&>(o).flushToDevice();
    const fdm_geometry g = o.geom();
    fdm_geometry mine{};
    mine.length_x = g.length_x; mine.length_y = g.length_y; mine.resolution = g.resolution;
    mine.position_x = g.position_x; mine.position_y = g.position_y;
    ck(fdm_engine_create_map(&mine, nullptr, device_, &eng_), "fdm_engine_create_map");
    rows_ = o.rows_; cols_ = o.cols_; res_ = o.res_; length_ = o.length_;
    ck(fdm_engine_set_start_index(eng_, g.start_row, g.start_col), "set_start_index");
    if (move_clear_basic_) ck(fdm_engine_set_option(eng_, "move_clear_basic", 1), "set_option(move_clear_basic)");
    for (const auto& n : o.getLayers()) ck(fdm_engine_layer_copy(eng_, o.eng_, n.c_str()), "layer_copy");
  }
  fdm_geometry geom() const {
    fdm_geometry g{};
    if (!eng_) return g;
    ck(fdm_engine_get_geometry(eng_, &g), "get_geometry");
    return g;
  }
  void need() const {
    if (!eng_) throw EngineError("GridMap: setGeometry() has not been called");
  }
  static void ck(int rc, const char* what) {
    if (rc < 0) throw EngineError(std::string(what) + ": " + fdm_last_error());
  }
  void release() {
    if (eng_) fdm_engine_destroy(eng_);
    eng_ = nullptr;
    mirror_.clear();
  }

  fdm_engine* eng_ = nullptr;
  int device_ = 0;
  int rows_ = 0, cols_ = 0;
  double res_ = 0.0;
  Length length_{0.0, 0.0};
  std::string frame_id_;
  uint64_t timestamp_ = 0;
  std::vector<std::string> initial_layers_;
  Position pending_pos_{0.0, 0.0};
  bool move_clear_basic_ = false;
  bool any_stale_ = false;
  // (heap nodes: a Matrix& — and the flag its guard watches — keep their address when the map is moved or rehashed)
  mutable std::unordered_map<std::string, std::unique_ptr<Mirror>> mirror_;
};

}  // namespace nanogrid
