// fastdem/transform_interface.hpp — Calibration / Odometry providers
// (fastdem/include/fastdem/transform_interface.hpp:31-62).
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <string>

#include "fastdem/compat/mini_eigen.hpp"

namespace fastdem {

class Calibration {
 public:
  using Ptr = std::shared_ptr<Calibration>;
  virtual ~Calibration() = default;
  virtual std::optional<Eigen::Isometry3d> getExtrinsic(const std::string& sensor_frame) const = 0;
  virtual std::string getBaseFrame() const = 0;
};

class Odometry {
 public:
  using Ptr = std::shared_ptr<Odometry>;
  virtual ~Odometry() = default;
  virtual std::optional<Eigen::Isometry3d> getPoseAt(uint64_t timestamp_ns) const = 0;
  virtual std::string getWorldFrame() const = 0;
};

}  // namespace fastdem
