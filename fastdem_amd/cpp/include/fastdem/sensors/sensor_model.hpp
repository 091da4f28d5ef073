// fastdem/sensors/sensor_model.hpp — host-side sensor models
// (fastdem/include/fastdem/sensors/{sensor_model,lidar_model,rgbd_model}.hpp, src/sensor_model.cpp).
// The three built-in models run ON THE DEVICE inside the engine; these classes exist (a) to keep
// the API (FastDEM::setSensorModel(unique_ptr<SensorModel>), fastdem.hpp:79-80) and (b) so a USER
// subclass can be evaluated here and handed to the engine as per-point sigma_z^2.
#pragma once
#include <cmath>
#include <cstdio>
#include <memory>

#include "fastdem/config/sensor_model.hpp"
#include "fastdem/point_types.hpp"

namespace fastdem {

class SensorModel {
 public:
  virtual ~SensorModel() = default;
  /// 3x3 measurement covariance of one point in the sensor frame.
  virtual Eigen::Matrix3f computeCovariance(const Eigen::Vector3f& point_sensor) const = 0;
  /// Built-in models report their SensorType so the engine evaluates them on the device.
  virtual bool builtin(SensorType& /*type*/) const { return false; }
};

class ConstantUncertaintyModel : public SensorModel {
 public:
  explicit ConstantUncertaintyModel(float uncertainty = 0.1f) : sigma_(uncertainty), variance_(uncertainty * uncertainty) {}
  Eigen::Matrix3f computeCovariance(const Eigen::Vector3f&) const override {
    return Eigen::Matrix3f::Identity() * variance_;
  }
  bool builtin(SensorType& t) const override { t = SensorType::Constant; return true; }
  float uncertainty() const { return sigma_; }

 private:
  float sigma_, variance_;
};

class LiDARSensorModel : public SensorModel {
 public:
  LiDARSensorModel(float range_noise = 0.02f, float angular_noise = 0.001f)
      : range_noise_(std::abs(range_noise)), angular_noise_(std::abs(angular_noise)) {}
  Eigen::Matrix3f computeCovariance(const Eigen::Vector3f& p) const override {
    const float dist_sq = p[0] * p[0] + (p[1] * p[1] + p[2] * p[2]);
    if (dist_sq < 1e-6f) return Eigen::Matrix3f::Identity() * 0.01f;
    const float distance = std::sqrt(dist_sq);
    const float dir[3] = {p[0] / distance, p[1] / distance, p[2] / distance};
    const float var_radial = std::max(range_noise_ * range_noise_, 1e-6f);
    const float var_lateral = std::max((distance * angular_noise_) * (distance * angular_noise_), 1e-6f);
    Eigen::Matrix3f cov = Eigen::Matrix3f::Identity() * var_lateral;
    const float s = var_radial - var_lateral;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) cov(r, c) = cov(r, c) + dir[c] * (s * dir[r]);
    return cov;
  }
  bool builtin(SensorType& t) const override { t = SensorType::LiDAR; return true; }
  float rangeNoise() const { return range_noise_; }
  float angularNoise() const { return angular_noise_; }

 private:
  float range_noise_, angular_noise_;
};

class RGBDSensorModel : public SensorModel {
 public:
  RGBDSensorModel(float normal_a = 0.001f, float normal_b = 0.002f, float normal_c = 0.4f,
                  float lateral_factor = 0.001f)
      : a_(normal_a), b_(normal_b), c_(normal_c), k_(lateral_factor) {}
  Eigen::Matrix3f computeCovariance(const Eigen::Vector3f& p) const override {
    const float depth = p[2];
    if (depth <= 0.0f) return Eigen::Matrix3f::Identity() * 0.01f;
    const float diff = depth - c_;
    const float sigma_norm = a_ + b_ * diff * diff;
    const float sigma_lat = k_ * depth;
    Eigen::Matrix3f m = Eigen::Matrix3f::Zero();
    m(0, 0) = m(1, 1) = sigma_lat * sigma_lat;
    m(2, 2) = sigma_norm * sigma_norm;
    return m;
  }
  bool builtin(SensorType& t) const override { t = SensorType::RGBD; return true; }
  float a() const { return a_; }
  float b() const { return b_; }
  float c() const { return c_; }
  float k() const { return k_; }

 private:
  float a_, b_, c_, k_;
};

// createSensorModel (fastdem/src/sensor_model.cpp:22-40)
inline std::unique_ptr<SensorModel> createSensorModel(const config::SensorModel& cfg) {
  switch (cfg.type) {
    case SensorType::LiDAR:
      return std::make_unique<LiDARSensorModel>(cfg.lidar.range_noise, cfg.lidar.angular_noise);
    case SensorType::RGBD:
      return std::make_unique<RGBDSensorModel>(cfg.rgbd.normal_a, cfg.rgbd.normal_b, cfg.rgbd.normal_c,
                                               cfg.rgbd.lateral_factor);
    case SensorType::Constant:
      return std::make_unique<ConstantUncertaintyModel>(cfg.constant.uncertainty);
    default:
      std::fprintf(stderr, "[warn] [SensorModel] Unknown type (%d), falling back to LiDAR\n", int(cfg.type));
      return std::make_unique<LiDARSensorModel>(cfg.lidar.range_noise, cfg.lidar.angular_noise);
  }
}

}  // namespace fastdem
