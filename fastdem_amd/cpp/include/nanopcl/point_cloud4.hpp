// nanopcl/point_cloud4.hpp — nanopcl::PointCloud4: a point cloud in the REFERENCE's own storage layout
// (fastdem/lib/nanoPCL/include/nanopcl/core/point_cloud.hpp:15-147, core/types.hpp:19-90,
// core/impl/point_cloud_impl.hpp): `points()` is one contiguous array of 16-byte {x, y, z, 1} records, the optional
// channels are separate arrays beside it.  FastDEM::integrate(const PointCloud4&, ...) hands `points().data()` to
// fdm_engine_integrate_points4 as it is — no host pass over the points.  The records (and the two channels the engine
// reads: intensity, packed colour) live in pinned memory from the engine's pool (fdm_host_alloc), so the bin kernel
// reads a cloud in place over PCIe with 16-byte loads.
//
// nanopcl::PointCloud (core.hpp) is the SoA sibling: 12 instead of 16 bytes per point over PCIe, and what the batch /
// queued entry points take.  A caller that already holds the reference's cloud uses this class; the surface — names,
// defaults, what clear() / reset() / resize() / extract() / erase() / operator+= keep and drop — is the reference's.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "nanopcl/core.hpp"

namespace nanopcl {

// core/types.hpp:19-22 — Eigen::Vector4f there; the same 16 bytes here
struct alignas(16) Point4 {
  float v[4];
  constexpr Point4() : v{0.0f, 0.0f, 0.0f, 0.0f} {}
  constexpr Point4(float x, float y, float z, float w) : v{x, y, z, w} {}
  float& x() { return v[0]; }
  float& y() { return v[1]; }
  float& z() { return v[2]; }
  float& w() { return v[3]; }
  float x() const { return v[0]; }
  float y() const { return v[1]; }
  float z() const { return v[2]; }
  float w() const { return v[3]; }
  float& operator[](int i) { return v[size_t(i)]; }
  float operator[](int i) const { return v[size_t(i)]; }
  float* data() { return v; }
  const float* data() const { return v; }
};
using Normal4 = Point4;
static_assert(sizeof(Point4) == 16 && alignof(Point4) == 16, "the engine reads 16-byte records");

struct Time {
  float val;
  explicit constexpr Time(float v = 0.0f) : val(v) {}
  constexpr operator float() const { return val; }
};
struct Ring {
  uint16_t val;
  explicit constexpr Ring(uint16_t v = 0) : val(v) {}
  constexpr operator uint16_t() const { return val; }
};
struct Label {
  uint32_t val;
  constexpr Label() : val(0) {}
  explicit constexpr Label(uint32_t v) : val(v) {}
  constexpr operator uint32_t() const { return val; }
};
inline double toSec(uint64_t ns) { return double(ns) * 1e-9; }
inline uint64_t fromSec(double s) { return static_cast<uint64_t>(s * 1e9); }

// core/types.hpp:70-90
class IndexRange {
  size_t n_;

 public:
  struct Iterator {
    size_t i;
    constexpr size_t operator*() const noexcept { return i; }
    constexpr Iterator& operator++() noexcept { ++i; return *this; }
    constexpr bool operator!=(Iterator o) const noexcept { return i != o.i; }
  };
  constexpr explicit IndexRange(size_t n) noexcept : n_(n) {}
  constexpr Iterator begin() const noexcept { return {0}; }
  constexpr Iterator end() const noexcept { return {n_}; }
  constexpr size_t size() const noexcept { return n_; }
};

class PointCloud4 {
 public:
  // point(i) of the reference is points_[i].head<3>(): a writable 3-D view of the record
  struct Point3Ref {
    Point4& p;
    float& x() { return p.v[0]; }
    float& y() { return p.v[1]; }
    float& z() { return p.v[2]; }
    float x() const { return p.v[0]; }
    float y() const { return p.v[1]; }
    float z() const { return p.v[2]; }
    Point3Ref& operator=(const Eigen::Vector3f& q) { p.v[0] = q[0]; p.v[1] = q[1]; p.v[2] = q[2]; return *this; }
    operator Eigen::Vector3f() const { return Eigen::Vector3f(p.v[0], p.v[1], p.v[2]); }
  };

  PointCloud4() = default;
  explicit PointCloud4(size_t n) { resize(n); }

  size_t size() const { return points_.size(); }
  bool empty() const { return points_.empty(); }
  size_t capacity() const { return points_.capacity(); }
  IndexRange indices() const { return IndexRange(size()); }

  void resize(size_t n) {  // point_cloud_impl.hpp:9-12
    points_.resize(n, Point4(0, 0, 0, 1));
    syncChannelSizes(n);
  }
  void reserve(size_t n) {
    points_.reserve(n);
    if (use_intensity_) intensity_.reserve(n);
    if (use_time_) time_.reserve(n);
    if (use_ring_) ring_.reserve(n);
    if (use_color_) rgb_.reserve(n);
    if (use_label_) label_.reserve(n);
    if (use_normal_) normal_.reserve(n);
  }
  void clear() {  // data only, the channel structure stays (:26-35)
    points_.clear();
    intensity_.clear(); time_.clear(); ring_.clear(); rgb_.clear(); label_.clear(); normal_.clear();
  }
  void reset() {  // everything, channels and metadata included (:37-67)
    points_ = HostVector<Point4>();
    intensity_ = HostVector<float>();
    rgb_ = HostVector<uint32_t>();
    time_ = std::vector<float>();
    ring_ = std::vector<uint16_t>();
    label_ = std::vector<Label>();
    normal_ = std::vector<Normal4>();
    use_intensity_ = use_time_ = use_ring_ = use_color_ = use_label_ = use_normal_ = false;
    frame_id_.clear();
    timestamp_ns_ = 0;
  }

  Point3Ref point(size_t i) { return Point3Ref{points_[i]}; }
  Eigen::Vector3f point(size_t i) const { return Eigen::Vector3f(points_[i].v[0], points_[i].v[1], points_[i].v[2]); }
  Point4& operator[](size_t i) { return points_[i]; }
  const Point4& operator[](size_t i) const { return points_[i]; }
  HostVector<Point4>& points() { return points_; }
  const HostVector<Point4>& points() const { return points_; }

  void add(float x, float y, float z) {
    points_.push_back(Point4(x, y, z, 1));
    pushDefaultChannelValues();
  }
  template <typename... Attrs>
  void add(float x, float y, float z, Attrs&&... attrs) {  // point_cloud.hpp:45-50
    points_.push_back(Point4(x, y, z, 1));
    pushDefaultChannelValues();
    (applyAttr(std::forward<Attrs>(attrs)), ...);
  }

  bool hasIntensity() const { return use_intensity_; }
  void useIntensity() { use_intensity_ = true; if (intensity_.empty() && !points_.empty()) intensity_.resize(points_.size(), 0.0f); }
  HostVector<float>& intensities() { return intensity_; }
  const HostVector<float>& intensities() const { return intensity_; }
  float& intensity(size_t i) { return intensity_[i]; }
  float intensity(size_t i) const { return intensity_[i]; }

  bool hasTime() const { return use_time_; }
  void useTime() { use_time_ = true; if (time_.empty() && !points_.empty()) time_.resize(points_.size(), 0.0f); }
  std::vector<float>& times() { return time_; }
  const std::vector<float>& times() const { return time_; }
  float& time(size_t i) { return time_[i]; }
  float time(size_t i) const { return time_[i]; }

  bool hasRing() const { return use_ring_; }
  void useRing() { use_ring_ = true; if (ring_.empty() && !points_.empty()) ring_.resize(points_.size(), 0); }
  std::vector<uint16_t>& rings() { return ring_; }
  const std::vector<uint16_t>& rings() const { return ring_; }
  uint16_t& ring(size_t i) { return ring_[i]; }
  uint16_t ring(size_t i) const { return ring_[i]; }

  // colour: kept packed (0x00RRGGBB, what the engine's colour layer stores) — color(i) / setColor(i, c) instead of the
  // reference's `Color& color(i)`
  bool hasColor() const { return use_color_; }
  void useColor() { use_color_ = true; if (rgb_.empty() && !points_.empty()) rgb_.resize(points_.size(), 0u); }
  Color color(size_t i) const { return Color(uint8_t(rgb_[i] >> 16), uint8_t(rgb_[i] >> 8), uint8_t(rgb_[i])); }
  void setColor(size_t i, const Color& c) { rgb_[i] = pack(c); }

  bool hasLabel() const { return use_label_; }
  void useLabel() { use_label_ = true; if (label_.empty() && !points_.empty()) label_.resize(points_.size(), Label()); }
  std::vector<Label>& labels() { return label_; }
  const std::vector<Label>& labels() const { return label_; }
  Label& label(size_t i) { return label_[i]; }
  const Label& label(size_t i) const { return label_[i]; }

  bool hasNormal() const { return use_normal_; }
  void useNormal() { use_normal_ = true; if (normal_.empty() && !points_.empty()) normal_.resize(points_.size(), Normal4(0, 0, 0, 0)); }
  Point3Ref normal(size_t i) { return Point3Ref{normal_[i]}; }
  Eigen::Vector3f normal(size_t i) const { return Eigen::Vector3f(normal_[i].v[0], normal_[i].v[1], normal_[i].v[2]); }
  std::vector<Normal4>& normals() { return normal_; }

  void copyChannelLayout(const PointCloud4& o) {
    if (o.hasIntensity()) useIntensity();
    if (o.hasTime()) useTime();
    if (o.hasRing()) useRing();
    if (o.hasColor()) useColor();
    if (o.hasLabel()) useLabel();
    if (o.hasNormal()) useNormal();
  }
  void copyChannelData(size_t dst, const PointCloud4& src, size_t at) {
    if (use_intensity_ && src.hasIntensity()) intensity_[dst] = src.intensity_[at];
    if (use_time_ && src.hasTime()) time_[dst] = src.time_[at];
    if (use_ring_ && src.hasRing()) ring_[dst] = src.ring_[at];
    if (use_color_ && src.hasColor()) rgb_[dst] = src.rgb_[at];
    if (use_label_ && src.hasLabel()) label_[dst] = src.label_[at];
    if (use_normal_ && src.hasNormal()) normal_[dst] = src.normal_[at];
  }

  const std::string& frameId() const { return frame_id_; }
  void setFrameId(const std::string& id) { frame_id_ = id; }
  uint64_t timestamp() const { return timestamp_ns_; }
  void setTimestamp(uint64_t ns) { timestamp_ns_ = ns; }

  PointCloud4 extract(const std::vector<size_t>& idx) const {  // point_cloud_impl.hpp:176-216
    PointCloud4 r;
    r.copyChannelLayout(*this);
    r.reserve(idx.size());
    for (size_t i : idx) {
      r.points_.push_back(points_[i]);
      if (use_intensity_) r.intensity_.push_back(intensity_[i]);
      if (use_time_) r.time_.push_back(time_[i]);
      if (use_ring_) r.ring_.push_back(ring_[i]);
      if (use_color_) r.rgb_.push_back(rgb_[i]);
      if (use_label_) r.label_.push_back(label_[i]);
      if (use_normal_) r.normal_.push_back(normal_[i]);
    }
    r.frame_id_ = frame_id_;
    r.timestamp_ns_ = timestamp_ns_;
    return r;
  }
  PointCloud4 extract(size_t start, size_t count) const {
    std::vector<size_t> idx(count);
    for (size_t k = 0; k < count; ++k) idx[k] = start + k;
    return extract(idx);
  }
  void erase(const std::vector<size_t>& idx) {  // :245-: duplicates and order of `idx` do not matter
    if (idx.empty()) return;
    std::vector<size_t> del = idx;
    std::sort(del.begin(), del.end());
    del.erase(std::unique(del.begin(), del.end()), del.end());
    size_t w = 0, d = 0;
    const size_t n = size();
    for (size_t i = 0; i < n; ++i) {
      if (d < del.size() && del[d] == i) { ++d; continue; }
      if (w != i) {
        points_[w] = points_[i];
        if (use_intensity_) intensity_[w] = intensity_[i];
        if (use_time_) time_[w] = time_[i];
        if (use_ring_) ring_[w] = ring_[i];
        if (use_color_) rgb_[w] = rgb_[i];
        if (use_label_) label_[w] = label_[i];
        if (use_normal_) normal_[w] = normal_[i];
      }
      ++w;
    }
    points_.resize(w, Point4(0, 0, 0, 1));
    syncChannelSizes(w);
  }
  PointCloud4& operator+=(const PointCloud4& o) {  // :160-174: an empty cloud adopts the other's channels
    if (o.empty()) return *this;
    if (empty()) copyChannelLayout(o);
    const size_t n = size() + o.size();
    reserve(n);
    points_.insert(points_.end(), o.points_.begin(), o.points_.end());
    if (use_intensity_ && o.hasIntensity()) intensity_.insert(intensity_.end(), o.intensity_.begin(), o.intensity_.end());
    if (use_time_ && o.hasTime()) time_.insert(time_.end(), o.time_.begin(), o.time_.end());
    if (use_ring_ && o.hasRing()) ring_.insert(ring_.end(), o.ring_.begin(), o.ring_.end());
    if (use_color_ && o.hasColor()) rgb_.insert(rgb_.end(), o.rgb_.begin(), o.rgb_.end());
    if (use_label_ && o.hasLabel()) label_.insert(label_.end(), o.label_.begin(), o.label_.end());
    if (use_normal_ && o.hasNormal()) normal_.insert(normal_.end(), o.normal_.begin(), o.normal_.end());
    syncChannelSizes(n);
    return *this;
  }

  // what the C ABI binds (fdm_engine_integrate_points4)
  const float* xyz1Data() const { return points_.empty() ? nullptr : points_.data()->v; }
  const float* intensityData() const { return use_intensity_ ? intensity_.data() : nullptr; }
  const uint32_t* rgbData() const { return use_color_ ? rgb_.data() : nullptr; }

 private:
  static uint32_t pack(const Color& c) { return (uint32_t(c.r) << 16) | (uint32_t(c.g) << 8) | uint32_t(c.b); }
  void pushDefaultChannelValues() {
    if (use_intensity_) intensity_.push_back(0.0f);
    if (use_time_) time_.push_back(0.0f);
    if (use_ring_) ring_.push_back(0);
    if (use_color_) rgb_.push_back(0u);
    if (use_label_) label_.push_back(Label());
    if (use_normal_) normal_.push_back(Normal4(0, 0, 0, 0));
  }
  void syncChannelSizes(size_t n) {
    if (use_intensity_) intensity_.resize(n, 0.0f);
    if (use_time_) time_.resize(n, 0.0f);
    if (use_ring_) ring_.resize(n, 0);
    if (use_color_) rgb_.resize(n, 0u);
    if (use_label_) label_.resize(n, Label());
    if (use_normal_) normal_.resize(n, Normal4(0, 0, 0, 0));
  }
  void applyAttr(Intensity a) { if (!use_intensity_) { useIntensity(); } intensity_.resize(points_.size(), 0.0f); intensity_.back() = a.val; }
  void applyAttr(Time a) { if (!use_time_) { useTime(); } time_.resize(points_.size(), 0.0f); time_.back() = a.val; }
  void applyAttr(Ring a) { if (!use_ring_) { useRing(); } ring_.resize(points_.size(), 0); ring_.back() = a.val; }
  void applyAttr(const Color& a) { if (!use_color_) { useColor(); } rgb_.resize(points_.size(), 0u); rgb_.back() = pack(a); }
  void applyAttr(Label a) { if (!use_label_) { useLabel(); } label_.resize(points_.size(), Label()); label_.back() = a; }
  void applyAttr(const Eigen::Vector3f& nrm) {
    if (!use_normal_) useNormal();
    normal_.resize(points_.size(), Normal4(0, 0, 0, 0));
    normal_.back() = Normal4(nrm[0], nrm[1], nrm[2], 0);
  }

  HostVector<Point4> points_;
  HostVector<float> intensity_;
  HostVector<uint32_t> rgb_;  // 0x00RRGGBB
  std::vector<float> time_;
  std::vector<uint16_t> ring_;
  std::vector<Label> label_;
  std::vector<Normal4> normal_;
  std::string frame_id_;
  uint64_t timestamp_ns_ = 0;
  bool use_intensity_ = false, use_time_ = false, use_ring_ = false, use_color_ = false, use_label_ = false,
       use_normal_ = false;
};

}  // namespace nanopcl
