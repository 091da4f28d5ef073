// fdm_grid.hpp — CPU restatement of the nanoGrid subset the integrate() path uses.
//
// *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// include, link or call anything under oracle/.  The product path
// (fastdem_amd/csrc) never does and fails loudly without its HIP library.
//
// PARITY STATUS: "parity unpinned" for the index arithmetic.
// nanoGrid (Ikhyeon-Cho/nanoGrid @ main, fetched by CMake FetchContent in the
// reference, fastdem/CMakeLists.txt:24-28) is NOT on disk and cannot be
// fetched.  No reference test pins an absolute (row, col) for a position
// (SURVEY.md §4 "Gap that matters").  This file restates the published
// ANYbotics grid_map_core algorithm nanoGrid descends from (header names
// GridMapMath.hpp / SubmapGeometry.hpp at fastdem/src/elevation_mapping.cpp:7,
// fastdem/include/fastdem/bridge/ros/impl.hpp:9) and is checked against the
// three independent in-tree restatements of the same geometry:
//   * fastdem/src/raycasting.cpp:63-80,112-113      (row/col from x/y, wrap)
//   * fastdem/include/fastdem/bridge/ros/impl.hpp:43-63 (cell centre, unwrap)
//   * fastdem/src/pcd_convert.cpp:335-348           (cell centre)
// and against the relative pins of fastdem/tests/test_elevation_map.cpp.
// By definition of this repo it is the index oracle.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

namespace fdmref {

struct Index2 {
  int r = 0, c = 0;
  bool operator==(const Index2& o) const { return r == o.r && c == o.c; }
};

struct IndexHash {  // stand-in for nanogrid::IndexHash (elevation_map.hpp:49-52)
  size_t operator()(const Index2& i) const {
    return std::hash<uint64_t>()((uint64_t(uint32_t(i.r)) << 32) | uint32_t(i.c));
  }
};

// grid_map_core wrapIndexToRange (GridMapMath): shortcuts, then modulo.
inline void wrapIndexToRange(int& index, int bufferSize) {
  if (index < bufferSize) {
    if (index >= 0) return;
    if (index >= -bufferSize) { index += bufferSize; return; }
    index = index % bufferSize;
    index += bufferSize;
  } else if (index < bufferSize * 2) {
    index -= bufferSize;
  } else {
    index = index % bufferSize;
  }
}

// Named float layers, column-major rows x cols (Eigen::MatrixXf storage;
// bridge/ros/impl.hpp:117-119, io_npz.cpp:144), circular buffer start index.
class Grid {
 public:
  // ---- geometry (nanogrid::GridMap::setGeometry; ElevationMap::setGeometry
  // promotes float -> double, elevation_map.hpp:112-116) ----
  void setGeometry(double len_x, double len_y, double resolution) {
    size_[0] = static_cast<int>(std::round(len_x / resolution));
    size_[1] = static_cast<int>(std::round(len_y / resolution));
    for (auto& kv : data_) kv.second.assign(size_t(size_[0]) * size_[1], NAN);
    res_ = resolution;
    length_[0] = double(size_[0]) * res_;
    length_[1] = double(size_[1]) * res_;
    // position unchanged by ElevationMap::setGeometry (default 0,0)
    start_[0] = start_[1] = 0;
  }

  int rows() const { return size_[0]; }
  int cols() const { return size_[1]; }
  double resolution() const { return res_; }
  const double* length() const { return length_; }
  const double* position() const { return pos_; }
  const int* startIndex() const { return start_; }
  void setPosition(double x, double y) { pos_[0] = x; pos_[1] = y; }
  void setStartIndex(int r, int c) { start_[0] = r; start_[1] = c; }

  // ---- layers ----
  bool exists(const std::string& n) const { return data_.count(n) != 0; }
  void add(const std::string& n, float value = NAN) {
    if (!exists(n)) names_.push_back(n);
    data_[n].assign(size_t(size_[0]) * size_[1], value);
  }
  std::vector<float>& get(const std::string& n) { return data_.at(n); }
  const std::vector<float>& get(const std::string& n) const { return data_.at(n); }
  const std::vector<std::string>& layers() const { return names_; }
  float& at(const std::string& n, const Index2& i) {
    return data_.at(n)[size_t(i.c) * size_[0] + i.r];
  }
  void clear(const std::string& n) {
    auto& v = data_.at(n);
    std::fill(v.begin(), v.end(), NAN);
  }
  void clearAll() {
    for (auto& kv : data_) std::fill(kv.second.begin(), kv.second.end(), NAN);
  }

  // ---- position <-> index (grid_map_core getIndexFromPosition /
  // checkIfPositionWithinMap / getPositionFromIndex) ----
  bool isInside(double x, double y) const {
    // positionTransformed = -(position - mapPosition - offset)
    const double ox = 0.5 * length_[0], oy = 0.5 * length_[1];
    const double tx = -((x - pos_[0]) - ox);
    const double ty = -((y - pos_[1]) - oy);
    return tx >= 0.0 && ty >= 0.0 && tx < length_[0] && ty < length_[1];
  }

  bool getIndex(double x, double y, Index2& out) const {
    if (!isInside(x, y)) return false;  // (int) cast below is only defined inside
    const double ox = 0.5 * length_[0], oy = 0.5 * length_[1];
    // indexVector = (position - offset - mapPosition) / resolution
    const double vx = ((x - ox) - pos_[0]) / res_;
    const double vy = ((y - oy) - pos_[1]) / res_;
    int r = static_cast<int>(-vx);  // transformMapFrameToBufferOrder + trunc
    int c = static_cast<int>(-vy);
    if (start_[0] != 0 || start_[1] != 0) {  // getBufferIndexFromIndex
      r += start_[0];
      c += start_[1];
      wrapIndexToRange(r, size_[0]);
      wrapIndexToRange(c, size_[1]);
    }
    out.r = r;
    out.c = c;
    return r >= 0 && c >= 0 && r < size_[0] && c < size_[1];  // checkIfIndexInRange
  }

  bool getPosition(const Index2& idx, double& x, double& y) const {
    if (idx.r < 0 || idx.c < 0 || idx.r >= size_[0] || idx.c >= size_[1]) return false;
    int ur = idx.r, uc = idx.c;
    if (start_[0] != 0 || start_[1] != 0) {  // getIndexFromBufferIndex
      ur -= start_[0];
      uc -= start_[1];
      wrapIndexToRange(ur, size_[0]);
      wrapIndexToRange(uc, size_[1]);
    }
    // offset = 0.5*length - 0.5*resolution ; position = mapPos + offset + res * (-unwrapped)
    x = pos_[0] + (0.5 * length_[0] - 0.5 * res_) + res_ * double(-ur);
    y = pos_[1] + (0.5 * length_[1] - 0.5 * res_) + res_ * double(-uc);
    return true;
  }

  // ---- move (grid_map_core GridMap::move) ----
  // Returns the index shift applied (buffer order) through shift_out[2].
  bool move(double x, double y, int* shift_out = nullptr) {
    const double ps[2] = {x - pos_[0], y - pos_[1]};
    int shift[2];
    for (int i = 0; i < 2; ++i) {  // getIndexShiftFromPositionShift
      const double t = ps[i] / res_;
      const int v = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
      shift[i] = -v;
    }
    for (int i = 0; i < 2; ++i) {
      if (shift[i] == 0) continue;
      if (std::abs(shift[i]) >= size_[i]) {
        clearAll();
      } else {
        const int sign = shift[i] > 0 ? 1 : -1;
        const int startIndex = start_[i] - (sign < 0 ? 1 : 0);
        const int endIndex = startIndex - sign + shift[i];
        const int nCells = std::abs(shift[i]);
        int index = sign > 0 ? startIndex : endIndex;
        wrapIndexToRange(index, size_[i]);
        if (index + nCells <= size_[i]) {
          clearStrip(i, index, nCells);
        } else {
          const int firstN = size_[i] - index;
          clearStrip(i, index, firstN);
          clearStrip(i, 0, nCells - firstN);
        }
      }
    }
    start_[0] += shift[0];
    start_[1] += shift[1];
    wrapIndexToRange(start_[0], size_[0]);
    wrapIndexToRange(start_[1], size_[1]);
    // getPositionShiftFromIndexShift: (-indexShift) * resolution
    pos_[0] += double(-shift[0]) * res_;
    pos_[1] += double(-shift[1]) * res_;
    if (shift_out) { shift_out[0] = shift[0]; shift_out[1] = shift[1]; }
    return shift[0] != 0 || shift[1] != 0;
  }

 private:
  // axis 0: rows [index, index+n) of the cleared layers; axis 1: cols.
  // WHICH layers: grid_map_core's clearRows / clearCols take `basicLayers_` when that list is not empty, every layer
  // otherwise.  ASSUMED (default, "all"): basicLayers is empty for ElevationMap.  The reference hints the other way —
  // it constructs nanogrid::GridMap({elevation, elevation_min, elevation_max}) (elevation_map.hpp:101-103) and its test
  // file speaks of "basicLayers = {elevation}" (tests/test_elevation_map.cpp:91) — and nanoGrid is not on disk to
  // settle it, so the other reading is a switch (setMoveClearBasic; the engine: option "move_clear_basic"): the strips
  // of a move shorter than the map then clear those three layers only (a move of >= the map's size is clearAll()
  // either way).  scripts/conformance/probe.cpp prints what the real library does.
  void clearStrip(int axis, int index, int n) {
    const int R = size_[0], C = size_[1];
    for (auto& kv : data_) {
      if (move_clear_basic_ && kv.first != "elevation" && kv.first != "elevation_min" && kv.first != "elevation_max") continue;
      float* d = kv.second.data();
      if (axis == 0) {
        for (int c = 0; c < C; ++c)
          for (int r = index; r < index + n; ++r) d[size_t(c) * R + r] = NAN;
      } else {
        for (int c = index; c < index + n; ++c)
          for (int r = 0; r < R; ++r) d[size_t(c) * R + r] = NAN;
      }
    }
  }

 public:
  void setMoveClearBasic(bool on) { move_clear_basic_ = on; }
  bool moveClearBasic() const { return move_clear_basic_; }

 private:
  bool move_clear_basic_ = false;
  std::vector<std::string> names_;
  std::unordered_map<std::string, std::vector<float>> data_;
  int size_[2] = {0, 0};
  int start_[2] = {0, 0};
  double res_ = 0.0;
  double length_[2] = {0.0, 0.0};
  double pos_[2] = {0.0, 0.0};
};

// nanogrid::colorVectorToValue (grid_map_core): 0x00RRGGBB bit-cast to float
// ("PCL packed float convention", bridge/ros/impl.hpp:20-21).
inline uint32_t packColor(uint8_t r, uint8_t g, uint8_t b) {
  return (uint32_t(r) << 16) | (uint32_t(g) << 8) | uint32_t(b);
}

}  // namespace fdmref
