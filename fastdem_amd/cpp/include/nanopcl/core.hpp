// nanopcl/core.hpp — the nanopcl::PointCloud surface FastDEM::integrate() consumes
// (fastdem/lib/nanoPCL/include/nanopcl/core/point_cloud.hpp:24-147, types.hpp), re-laid out as
// SoA channels: x / y / z / intensity / packed rgb are separate contiguous float arrays, which is
// exactly what the engine's C ABI takes — no AoS->SoA staging pass on the host.  The channels live in
// pinned memory from the engine's pool (fdm_host_alloc), so FastDEM::integrate() reads a cloud in
// place over PCIe instead of copying it first.
#pragma once
#include <cstddef>
#include <cstdint>
#include <new>
#include <string>
#include <vector>

#include "fastdem/compat/mini_eigen.hpp"
#include "fdm_engine.h"

namespace nanopcl {

// std::allocator over fdm_host_alloc / fdm_host_free (a free-list pop per cloud, not a driver call)
template <typename T>
struct HostAllocator {
  using value_type = T;
  HostAllocator() = default;
  template <typename U>
  HostAllocator(const HostAllocator<U>&) {}
  T* allocate(std::size_t n) {
    void* p = fdm_host_alloc(uint64_t(n) * sizeof(T));
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, std::size_t) { fdm_host_free(p); }
  template <typename U>
  bool operator==(const HostAllocator<U>&) const { return true; }
  template <typename U>
  bool operator!=(const HostAllocator<U>&) const { return false; }
};
template <typename T>
using HostVector = std::vector<T, HostAllocator<T>>;

using Point = Eigen::Vector3f;

struct Intensity {
  float val;
  explicit constexpr Intensity(float v = 0.0f) : val(v) {}
  constexpr operator float() const { return val; }
};

struct Color {
  uint8_t r, g, b;
  constexpr Color() : r(0), g(0), b(0) {}
  constexpr Color(uint8_t r_, uint8_t g_, uint8_t b_) : r(r_), g(g_), b(b_) {}
};

class PointCloud {
 public:
  // xyz view of one point (reads and writes go to the SoA channels)
  struct PointRef {
    float &x_, &y_, &z_;
    float x() const { return x_; }
    float y() const { return y_; }
    float z() const { return z_; }
    PointRef& operator=(const Eigen::Vector3f& p) { x_ = p[0]; y_ = p[1]; z_ = p[2]; return *this; }
    operator Eigen::Vector3f() const { return Eigen::Vector3f(x_, y_, z_); }
  };

  PointCloud() = default;
  explicit PointCloud(size_t n) { resize(n); }

  size_t size() const { return x_.size(); }
  bool empty() const { return x_.empty(); }
  void reserve(size_t n) { x_.reserve(n); y_.reserve(n); z_.reserve(n); }
  void resize(size_t n) {
    x_.resize(n); y_.resize(n); z_.resize(n);
    if (use_intensity_) intensity_.resize(n);
    if (use_color_) rgb_.resize(n);
    if (use_cov_) cov_.resize(n * 9);
  }
  void clear() { resize(0); }

  void add(float x, float y, float z) {
    x_.push_back(x); y_.push_back(y); z_.push_back(z);
    if (use_intensity_) intensity_.push_back(0.0f);
    if (use_color_) rgb_.push_back(0u);
    if (use_cov_) cov_.resize(cov_.size() + 9, 0.0f);
  }
  void add(float x, float y, float z, Intensity i) {
    if (!use_intensity_) useIntensity();
    add(x, y, z);
    intensity_.back() = i.val;
  }
  void add(float x, float y, float z, const Color& c) {
    if (!use_color_) useColor();
    add(x, y, z);
    rgb_.back() = pack(c);
  }

  PointRef point(size_t i) { return PointRef{x_[i], y_[i], z_[i]}; }
  Eigen::Vector3f point(size_t i) const { return Eigen::Vector3f(x_[i], y_[i], z_[i]); }

  bool hasIntensity() const { return use_intensity_; }
  void useIntensity() { use_intensity_ = true; intensity_.resize(size(), 0.0f); }
  float& intensity(size_t i) { return intensity_[i]; }
  float intensity(size_t i) const { return intensity_[i]; }

  // covariance channel (nanopcl/core/point_cloud.hpp:126-147): the cloud the preprocessed-scan callback
  // receives carries R * Sigma_sensor * R^T per point
  bool hasCovariance() const { return use_cov_; }
  void useCovariance() { use_cov_ = true; cov_.resize(size() * 9, 0.0f); }
  Eigen::Matrix3f covariance(size_t i) const {
    Eigen::Matrix3f m;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) m(r, c) = cov_[i * 9 + size_t(c) * 3 + size_t(r)];
    return m;
  }
  float* covarianceData() { return use_cov_ ? cov_.data() : nullptr; }  // [n][9], column-major 3x3 per point

  bool hasColor() const { return use_color_; }
  void useColor() { use_color_ = true; rgb_.resize(size(), 0u); }
  Color color(size_t i) const { return Color(uint8_t(rgb_[i] >> 16), uint8_t(rgb_[i] >> 8), uint8_t(rgb_[i])); }
  void setColor(size_t i, const Color& c) { rgb_[i] = pack(c); }

  const std::string& frameId() const { return frame_id_; }
  void setFrameId(const std::string& id) { frame_id_ = id; }
  uint64_t timestamp() const { return timestamp_ns_; }
  void setTimestamp(uint64_t ns) { timestamp_ns_ = ns; }

  // SoA channel access (what the C ABI binds)
  const float* xData() const { return x_.data(); }
  const float* yData() const { return y_.data(); }
  const float* zData() const { return z_.data(); }
  const float* intensityData() const { return use_intensity_ ? intensity_.data() : nullptr; }
  const uint32_t* rgbData() const { return use_color_ ? rgb_.data() : nullptr; }

 private:
  static uint32_t pack(const Color& c) { return (uint32_t(c.r) << 16) | (uint32_t(c.g) << 8) | uint32_t(c.b); }
  HostVector<float> x_, y_, z_, intensity_;
  HostVector<uint32_t> rgb_;  // 0x00RRGGBB
  std::vector<float> cov_;
  std::string frame_id_;
  uint64_t timestamp_ns_ = 0;
  bool use_intensity_ = false, use_color_ = false, use_cov_ = false;
};

}  // namespace nanopcl
