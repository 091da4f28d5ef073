// fastdem/fastdem.hpp — fastdem::FastDEM, the drop-in facade (fastdem/include/fastdem/fastdem.hpp:54-158,
// fastdem/src/fastdem.cpp).  Same constructors, fluent setters, integrate() overloads, return
// values and log messages; the body of integrateImpl() is one call into the HIP engine.
//
// Not thread-safe (same contract as fastdem.hpp:48-53): the caller serialises integrate()
// against map reads.
#pragma once
#include <cstdio>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "fastdem/config/fastdem.hpp"
#include "fastdem/elevation_map.hpp"
#include "fastdem/mapping/elevation_mapping.hpp"
#include "fastdem/point_types.hpp"
#include "fastdem/postprocess/raycasting.hpp"
#include "fastdem/sensors/sensor_model.hpp"
#include "fastdem/transform_interface.hpp"
#include "nanopcl/point_cloud4.hpp"

namespace fastdem {

class FastDEM {
 public:
  using CloudCallback = std::function<void(const PointCloud&)>;

  explicit FastDEM(ElevationMap& map) : FastDEM(map, Config{}) {}
  FastDEM(ElevationMap& map, const Config& cfg) : map_(map), cfg_(cfg) {
    sensor_model_ = createSensorModel(cfg_.sensor_model);
    mapping_ = std::make_unique<ElevationMapping>(map_, cfg_);
  }
  ~FastDEM() {
    try { drain(); } catch (...) {}  // (queued clouds are borrowed: the GPU must be done with them)
  }
  FastDEM(const FastDEM&) = delete;
  FastDEM& operator=(const FastDEM&) = delete;

  FastDEM& setMappingMode(MappingMode mode) {
    cfg_.mapping.mode = mode;
    cfg_dirty_ = true;
    mapping_->setConfig(cfg_);
    return *this;
  }
  FastDEM& setEstimatorType(EstimationType type) {
    cfg_.mapping.estimation_type = type;
    cfg_dirty_ = true;
    mapping_->setConfig(cfg_);
    return *this;
  }
  FastDEM& setSensorModel(SensorType type) {
    cfg_.sensor_model.type = type;
    cfg_dirty_ = true;
    sensor_model_ = createSensorModel(cfg_.sensor_model);
    return *this;
  }
  FastDEM& setSensorModel(std::unique_ptr<SensorModel> model) noexcept {
    sensor_model_ = std::move(model);
    cfg_dirty_ = true;
    return *this;
  }
  FastDEM& setHeightFilter(float z_min, float z_max) noexcept {
    cfg_.point_filter.z_min = z_min;
    cfg_dirty_ = true;
    cfg_.point_filter.z_max = z_max;
    return *this;
  }
  FastDEM& setRangeFilter(float range_min, float range_max) noexcept {
    cfg_.point_filter.range_min = range_min;
    cfg_dirty_ = true;
    cfg_.point_filter.range_max = range_max;
    return *this;
  }
  FastDEM& enableRaycasting(bool enabled = true) noexcept {
    cfg_.raycasting.enabled = enabled;  // step 3 of integrateImpl (fastdem.cpp:152-159), on the device
    cfg_dirty_ = true;
    return *this;
  }
  FastDEM& setCalibrationProvider(std::shared_ptr<Calibration> c) noexcept {
    calibration_ = std::move(c);
    return *this;
  }
  FastDEM& setOdometryProvider(std::shared_ptr<Odometry> o) noexcept {
    odometry_ = std::move(o);
    return *this;
  }
  template <typename T>
  FastDEM& setTransformProvider(std::shared_ptr<T> system) {
    setCalibrationProvider(system);
    setOdometryProvider(system);
    return *this;
  }

  void reset() { drain(); map_.clearAll(); }
  const Config& config() const noexcept { return cfg_; }
  bool hasTransformProvider() const noexcept { return calibration_ != nullptr && odometry_ != nullptr; }

  /// Online mode (fastdem.cpp:83-120).
  bool integrate(std::shared_ptr<PointCloud> cloud) {
    if (!calibration_ || !odometry_) {
      std::fprintf(stderr, "[error] [FastDEM] Transform providers not set. Call setTransformProvider() or "
                           "setCalibrationProvider()/setOdometryProvider() first, or use integrate(cloud, "
                           "T_base_sensor, T_world_base) for explicit transforms.\n");
      return false;
    }
    if (!cloud || cloud->empty()) {
      std::fprintf(stderr, "[warn] [FastDEM] Received empty or null cloud. Skipping...\n");
      return false;
    }
    if (cloud->frameId().empty()) {
      std::fprintf(stderr, "[error] [FastDEM] Input cloud has no frameId. Skipping...\n");
      return false;
    }
    auto T_base_sensor = calibration_->getExtrinsic(cloud->frameId());
    if (!T_base_sensor) {
      std::fprintf(stderr, "[warn] [FastDEM] Calibration not available for '%s'. Skipping...\n", cloud->frameId().c_str());
      return false;
    }
    auto T_world_base = odometry_->getPoseAt(cloud->timestamp());
    if (!T_world_base) {
      std::fprintf(stderr, "[warn] [FastDEM] Odometry not available at %llu. Skipping...\n",
                   static_cast<unsigned long long>(cloud->timestamp()));
      return false;
    }
    return integrateImpl(*cloud, *T_base_sensor, *T_world_base);
  }

  /// Explicit transforms (fastdem.cpp:122-131).
  bool integrate(const PointCloud& cloud, const Eigen::Isometry3d& T_base_sensor,
                 const Eigen::Isometry3d& T_world_base) {
    if (cloud.empty()) {
      std::fprintf(stderr, "[warn] [FastDEM] Received empty cloud. Skipping...\n");
      return false;
    }
    return integrateImpl(cloud, T_base_sensor, T_world_base);
  }

  /// The same two calls for a cloud in the REFERENCE's storage layout (nanopcl::PointCloud4: `points()` = contiguous
  /// {x, y, z, 1} records, nanopcl/core/point_cloud.hpp:126-134): `points().data()` goes to the engine as it is
  /// (fdm_engine_integrate_points4; pinned records are read in place with 16-byte loads).  Synchronous.
  bool integrate(std::shared_ptr<nanopcl::PointCloud4> cloud) {
    if (!calibration_ || !odometry_) {
      std::fprintf(stderr, "[error] [FastDEM] Transform providers not set. Call setTransformProvider() or "
                           "setCalibrationProvider()/setOdometryProvider() first, or use integrate(cloud, "
                           "T_base_sensor, T_world_base) for explicit transforms.\n");
      return false;
    }
    if (!cloud || cloud->empty()) {
      std::fprintf(stderr, "[warn] [FastDEM] Received empty or null cloud. Skipping...\n");
      return false;
    }
    if (cloud->frameId().empty()) {
      std::fprintf(stderr, "[error] [FastDEM] Input cloud has no frameId. Skipping...\n");
      return false;
    }
    auto T_base_sensor = calibration_->getExtrinsic(cloud->frameId());
    if (!T_base_sensor) {
      std::fprintf(stderr, "[warn] [FastDEM] Calibration not available for '%s'. Skipping...\n", cloud->frameId().c_str());
      return false;
    }
    auto T_world_base = odometry_->getPoseAt(cloud->timestamp());
    if (!T_world_base) {
      std::fprintf(stderr, "[warn] [FastDEM] Odometry not available at %llu. Skipping...\n",
                   static_cast<unsigned long long>(cloud->timestamp()));
      return false;
    }
    return integrateImpl(*cloud, *T_base_sensor, *T_world_base);
  }
  bool integrate(const nanopcl::PointCloud4& cloud, const Eigen::Isometry3d& T_base_sensor,
                 const Eigen::Isometry3d& T_world_base) {
    if (cloud.empty()) {
      std::fprintf(stderr, "[warn] [FastDEM] Received empty cloud. Skipping...\n");
      return false;
    }
    return integrateImpl(cloud, T_base_sensor, T_world_base);
  }

  // ---- not in the reference: the two ways a host gets more than one synchronous call per scan out of the engine ----

  /// One scan of integrateBatch(): what integrate(cloud, T_base_sensor, T_world_base) takes.  The cloud is borrowed.
  struct Scan {
    const PointCloud* cloud;
    Eigen::Isometry3d T_base_sensor, T_world_base;
  };
  /// N consecutive integrate(cloud, T_base_sensor, T_world_base) calls as ONE call: the map afterwards is bit for bit
  /// what the N calls leave (per cell the reference fixes only the order of the scans, elevation_mapping.cpp:94-125),
  /// but the engine sees the scans up front and bins up to sixteen of them per launch (fdm_engine_integrate_host_batch;
  /// PointCloud channels live in pinned memory and are read in place).  Returns what the LAST call would have returned
  /// (false: its cloud was empty or entirely filtered); lastStats() are that scan's.  Empty clouds inside the batch are
  /// skipped with the reference's warning.  Scan callbacks and user SensorModel subclasses need host work per scan:
  /// with either, the scans are integrated one by one.
  bool integrateBatch(const std::vector<Scan>& scans) {
    if (scans.empty()) return false;
    SensorType builtin;
    const bool custom = sensor_model_ && !sensor_model_->builtin(builtin);
    if (on_preprocessed_ || on_rasterized_ || custom) {
      bool last = false;
      for (const Scan& s : scans) last = s.cloud && integrate(*s.cloud, s.T_base_sensor, s.T_world_base);
      return last;
    }
    drain();
    map_.flushToDevice();
    const fdm_config f = detail::toEngineConfig(effectiveConfig());
    detail::ck(fdm_engine_set_config(map_.engine(), &f), "fdm_engine_set_config");
    batch_.clear();
    batch_.reserve(scans.size());
    bool last_empty = false;
    for (const Scan& s : scans) {
      last_empty = !s.cloud || s.cloud->empty();
      if (last_empty) {
        std::fprintf(stderr, "[warn] [FastDEM] Received empty cloud. Skipping...\n");
        continue;
      }
      fdm_device_scan d{};
      d.n = s.cloud->size();
      d.x = s.cloud->xData(); d.y = s.cloud->yData(); d.z = s.cloud->zData();
      d.intensity = s.cloud->intensityData();
      d.rgb = s.cloud->rgbData();
      std::memcpy(d.T_base_sensor, s.T_base_sensor.matrix().data(), sizeof(d.T_base_sensor));
      std::memcpy(d.T_world_base, s.T_world_base.matrix().data(), sizeof(d.T_world_base));
      batch_.push_back(d);
    }
    if (batch_.empty()) return false;
    const int rc = fdm_engine_integrate_host_batch(map_.engine(), uint32_t(batch_.size()), batch_.data(), &last_);
    detail::ck(rc, "fdm_engine_integrate_host_batch");
    map_.invalidateHost();
    return rc == FDM_OK && !last_empty;
  }

  /// Queued mode.  integrate() then ENQUEUES the scan (the cloud's pinned channels are read in place by the launch) and
  /// returns without waiting for the GPU: `true` means "accepted", `false` keeps its host-decidable meanings (empty
  /// cloud, missing providers).  The one case the reference decides from the data — every point filtered: false, and
  /// the map does not move (fastdem.cpp:138) — is still honoured ON THE DEVICE (nothing moves, nothing is written);
  /// only its `false` arrives late: drain() waits for everything queued and returns what the LAST queued integrate()
  /// would have returned synchronously.  Contract for the caller: a queued cloud must stay alive and untouched until
  /// drain(), reset(), the next map access through ElevationMap, or the destructor — whichever comes first (all of
  /// them drain).  Scan callbacks and user SensorModel subclasses keep integrate() synchronous.
  FastDEM& setQueued(bool on) {
    if (!on) drain();
    queued_ = on;
    return *this;
  }
  bool queued() const noexcept { return queued_; }
  /// Wait for every queued scan; the status the last one would have returned (true when nothing was queued).
  bool drain() {
    if (!pending_) return true;
    pending_ = false;
    const int rc = fdm_engine_last_stats(map_.engine(), &last_);
    detail::ck(rc, "fdm_engine_last_stats");
    map_.invalidateHost();
    return rc == FDM_OK;
  }

  /// sensor_msgs/PointCloud2-shaped message straight to the device (what the ROS scan callback does
  /// with nanopcl::from + integrate, ros1/src/fastdem_ros_node.cpp:171-182): the raw bytes cross
  /// PCIe once; field decoding, the finite-point filter of from_impl
  /// (nanopcl/bridge/ros/impl.hpp:174-246) and the map update all run in HBM.
  /// Msg needs: width, height, point_step, data (contiguous bytes), fields[] with name / offset /
  /// datatype.  Built-in sensor models only (a user SensorModel subclass needs host points: decode
  /// with your bridge and call integrate(PointCloud, ...)).
  template <typename Msg>
  bool integrateCloud2(const Msg& msg, const Eigen::Isometry3d& T_base_sensor,
                       const Eigen::Isometry3d& T_world_base) {
    SensorType builtin;
    if (sensor_model_ && !sensor_model_->builtin(builtin))
      throw std::invalid_argument("integrateCloud2: custom SensorModel needs host points");
    fdm_cloud2_layout lay{};
    lay.point_step = msg.point_step;
    lay.off_x = lay.off_y = lay.off_z = lay.off_intensity = lay.off_rgb = -1;
    for (const auto& f : msg.fields) {  // FieldOffsets::parse (impl.hpp:65-99)
      const std::string name = f.name;
      if (name == "x") lay.off_x = int32_t(f.offset);
      else if (name == "y") lay.off_y = int32_t(f.offset);
      else if (name == "z") lay.off_z = int32_t(f.offset);
      else if (name == "intensity") { lay.off_intensity = int32_t(f.offset); lay.intensity_type = int32_t(f.datatype); }
      else if (name == "rgb" || name == "rgba") lay.off_rgb = int32_t(f.offset);
    }
    const uint64_t n = uint64_t(msg.width) * uint64_t(msg.height);
    map_.flushToDevice();
    const fdm_config f = detail::toEngineConfig(effectiveConfig());
    detail::ck(fdm_engine_set_config(map_.engine(), &f), "fdm_engine_set_config");
    const int rc = fdm_engine_integrate_cloud2(map_.engine(), msg.data.data(), 0, n, &lay,
                                               T_base_sensor.matrix().data(), T_world_base.matrix().data(), &last_);
    detail::ck(rc, "fdm_engine_integrate_cloud2");
    map_.invalidateHost();
    if (rc == FDM_SKIP_EMPTY_CLOUD) std::fprintf(stderr, "[warn] [FastDEM] Received empty cloud. Skipping...\n");
    if (rc != FDM_OK) return false;
    if (on_preprocessed_) on_preprocessed_(fetch(true, last_.n_input));
    if (on_rasterized_ && last_.n_cells_touched > 0) on_rasterized_(fetch(false, last_.n_cells_touched));
    return true;
  }

  /// Scan callbacks (fastdem.hpp:129-136).  The clouds are captured on the device and
  /// materialised on the host only while a callback is registered: the preprocessed scan (points
  /// that survived the filters, map frame, input order) and the rasterized scan (one point per
  /// observed cell at the cell centre, z = min_z; fastdem.cpp:200-214).
  void onScanPreprocessed(CloudCallback cb) { on_preprocessed_ = std::move(cb); syncCapture(); }
  void onScanRasterized(CloudCallback cb) { on_rasterized_ = std::move(cb); syncCapture(); }

  /// Statistics of the last integrate() (not in the reference).
  const fdm_scan_stats& lastStats() const { return last_; }

 private:
  // cfg_ with the built-in sensor model's parameters taken from the model OBJECT
  Config effectiveConfig() const {
    Config eff = cfg_;
    SensorType builtin;
    if (sensor_model_ && sensor_model_->builtin(builtin)) {
      eff.sensor_model.type = builtin;
      if (auto* l = dynamic_cast<const LiDARSensorModel*>(sensor_model_.get())) {
        eff.sensor_model.lidar.range_noise = l->rangeNoise();
        eff.sensor_model.lidar.angular_noise = l->angularNoise();
      } else if (auto* r = dynamic_cast<const RGBDSensorModel*>(sensor_model_.get())) {
        eff.sensor_model.rgbd.normal_a = r->a();
        eff.sensor_model.rgbd.normal_b = r->b();
        eff.sensor_model.rgbd.normal_c = r->c();
        eff.sensor_model.rgbd.lateral_factor = r->k();
      } else if (auto* c = dynamic_cast<const ConstantUncertaintyModel*>(sensor_model_.get())) {
        eff.sensor_model.constant.uncertainty = c->uncertainty();
      }
    }
    return eff;
  }
  // integrateImpl (fastdem.cpp:133-162): preprocessScan + ElevationMapping::update, on the device
  // (CLOUD: nanopcl::PointCloud — SoA channels — or nanopcl::PointCloud4 — the reference's xyz1 records)
  int engineIntegrate(const PointCloud& cloud, const float* sigma, const Eigen::Isometry3d& T_base_sensor,
                      const Eigen::Isometry3d& T_world_base) {
    return fdm_engine_integrate(map_.engine(), cloud.size(), cloud.xData(), cloud.yData(), cloud.zData(),
                                cloud.intensityData(), cloud.rgbData(), sigma, T_base_sensor.matrix().data(),
                                T_world_base.matrix().data(), &last_);
  }
  int engineIntegrate(const nanopcl::PointCloud4& cloud, const float* sigma, const Eigen::Isometry3d& T_base_sensor,
                      const Eigen::Isometry3d& T_world_base) {
    return fdm_engine_integrate_points4(map_.engine(), cloud.size(), cloud.xyz1Data(), cloud.intensityData(),
                                        cloud.rgbData(), sigma, T_base_sensor.matrix().data(),
                                        T_world_base.matrix().data(), &last_);
  }
  template <class CLOUD>
  bool integrateImpl(const CLOUD& cloud, const Eigen::Isometry3d& T_base_sensor,
                     const Eigen::Isometry3d& T_world_base) {
    SensorType bi;
    if (std::is_same<CLOUD, PointCloud>::value && queued_ && !on_preprocessed_ && !on_rasterized_ &&
        (!sensor_model_ || sensor_model_->builtin(bi))) {
      return integrateQueued(cloud, T_base_sensor, T_world_base);
    }
    return integrateSync(cloud, T_base_sensor, T_world_base);
  }
  bool integrateQueued(const nanopcl::PointCloud4&, const Eigen::Isometry3d&, const Eigen::Isometry3d&) { return false; }  // (never taken)
  bool integrateQueued(const PointCloud& cloud, const Eigen::Isometry3d& T_base_sensor,
                       const Eigen::Isometry3d& T_world_base) {
    {
      map_.flushToDevice();  // (host writes to the map since the last call; nothing to do when there were none)
      if (cfg_dirty_) {      // (the engine takes its parameters when a scan is ENQUEUED: scans already queued keep theirs)
        const fdm_config fq = detail::toEngineConfig(effectiveConfig());
        detail::ck(fdm_engine_set_config(map_.engine(), &fq), "fdm_engine_set_config");
        cfg_dirty_ = false;
      }
      detail::ck(fdm_engine_integrate_async(map_.engine(), cloud.size(), cloud.xData(), cloud.yData(), cloud.zData(),
                                            cloud.intensityData(), cloud.rgbData(), nullptr,
                                            T_base_sensor.matrix().data(), T_world_base.matrix().data()),
                 "fdm_engine_integrate_async");
      pending_ = true;
      map_.invalidateHost();
      return true;
    }
  }
  template <class CLOUD>
  bool integrateSync(const CLOUD& cloud, const Eigen::Isometry3d& T_base_sensor,
                     const Eigen::Isometry3d& T_world_base) {
    drain();
    map_.flushToDevice();
    Config eff = effectiveConfig();
    const float* sigma = nullptr;
    SensorType builtin;
    if (sensor_model_ && sensor_model_->builtin(builtin)) {
      // evaluated on the device with the parameters the OBJECT carries (effectiveConfig)
    } else if (sensor_model_) {
      // user SensorModel subclass: evaluate on the host, hand over sigma_z^2 = (R Sigma R^T)(2,2)
      const Eigen::Matrix3f R = (T_world_base * T_base_sensor).rotation().cast<float>();
      sigma_.resize(cloud.size());
      for (size_t i = 0; i < cloud.size(); ++i) {
        const Eigen::Matrix3f S = sensor_model_->computeCovariance(cloud.point(i));
        float m2[3];
        for (int j = 0; j < 3; ++j) m2[j] = R(2, 0) * S(0, j) + (R(2, 1) * S(1, j) + R(2, 2) * S(2, j));
        sigma_[i] = m2[0] * R(2, 0) + (m2[1] * R(2, 1) + m2[2] * R(2, 2));
      }
      sigma = sigma_.data();
    }
    const fdm_config f = detail::toEngineConfig(eff);
    detail::ck(fdm_engine_set_config(map_.engine(), &f), "fdm_engine_set_config");
    cfg_dirty_ = false;
    const int rc = engineIntegrate(cloud, sigma, T_base_sensor, T_world_base);
    detail::ck(rc, "fdm_engine_integrate");
    map_.invalidateHost();
    if (rc != FDM_OK) return false;  // FDM_SKIP_ALL_FILTERED == `if (points.empty()) return false`
    if (on_preprocessed_) {                                                         // fastdem.cpp:139-141
      R_last_ = (T_world_base * T_base_sensor).rotation().cast<float>();
      on_preprocessed_(fetch(true, cloud.size(), &cloud));
    }
    if (on_rasterized_ && last_.n_cells_touched > 0) on_rasterized_(fetch(false, last_.n_cells_touched));  // :148-150
    return true;
  }
  void syncCapture() {
    detail::ck(fdm_engine_capture(map_.engine(), on_preprocessed_ ? 2 : 0, on_rasterized_ ? 1 : 0),
               "fdm_engine_capture");
  }
  PointCloud fetch(bool preprocessed, size_t cap) { return fetch<PointCloud>(preprocessed, cap, nullptr); }
  template <class CLOUD>
  PointCloud fetch(bool preprocessed, size_t cap, const CLOUD* src) {
    std::vector<float> x(cap), y(cap), z(cap);
    uint64_t n = 0;
    if (preprocessed)
      detail::ck(fdm_engine_last_preprocessed(map_.engine(), cap, x.data(), y.data(), z.data(), nullptr, &n),
                 "fdm_engine_last_preprocessed");
    else
      detail::ck(fdm_engine_last_rasterized(map_.engine(), cap, x.data(), y.data(), z.data(), &n),
                 "fdm_engine_last_rasterized");
    PointCloud out;
    out.reserve(n);
    for (uint64_t i = 0; i < n && i < cap; ++i) out.add(x[i], y[i], z[i]);
    if (preprocessed) {  // the reference's preprocessed cloud carries the rotated covariances (fastdem.cpp:182-187)
      out.useCovariance();
      SensorType builtin;
      if (!sensor_model_ || sensor_model_->builtin(builtin)) {
        uint64_t nc = 0;
        detail::ck(fdm_engine_last_preprocessed_cov(map_.engine(), out.size(), out.covarianceData(), &nc),
                   "fdm_engine_last_preprocessed_cov");
      } else if (src) {
        // user SensorModel subclass: the device only ever saw sigma_z^2; R * Sigma * R^T of the surviving
        // points is evaluated here, in the reference's order (M = R * Sigma, then M * R^T)
        std::vector<int32_t> ids(src->size());
        detail::ck(fdm_engine_last_cell_ids(map_.engine(), ids.data(), ids.size()), "fdm_engine_last_cell_ids");
        const Eigen::Matrix3f Rt = R_last_.transpose();
        size_t w = 0;
        float* dst = out.covarianceData();
        for (size_t i = 0; i < src->size() && w < out.size(); ++i) {
          if (ids[i] == -1) continue;  // dropped by cropRange / cropZ
          const Eigen::Matrix3f C = (R_last_ * sensor_model_->computeCovariance(src->point(i))) * Rt;
          for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) dst[w * 9 + size_t(c) * 3 + size_t(r)] = C(r, c);
          ++w;
        }
      }
    }
    out.setFrameId(map_.getFrameId());
    return out;
  }

  ElevationMap& map_;
  Config cfg_;
  std::unique_ptr<SensorModel> sensor_model_;
  std::unique_ptr<ElevationMapping> mapping_;
  std::shared_ptr<Calibration> calibration_;
  std::shared_ptr<Odometry> odometry_;
  CloudCallback on_preprocessed_, on_rasterized_;
  std::vector<float> sigma_;
  Eigen::Matrix3f R_last_;
  fdm_scan_stats last_{};
  std::vector<fdm_device_scan> batch_;
  bool queued_ = false, pending_ = false, cfg_dirty_ = true;
};

}  // namespace fastdem
