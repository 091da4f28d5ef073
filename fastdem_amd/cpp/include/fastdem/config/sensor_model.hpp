// fastdem/config/sensor_model.hpp (fastdem/include/fastdem/config/sensor_model.hpp:10-37)
#pragma once
namespace fastdem {
enum class SensorType { Constant, LiDAR, RGBD };
namespace config {
struct SensorModel {
  SensorType type = SensorType::LiDAR;
  struct LiDAR {
    float range_noise = 0.02f;
    float angular_noise = 0.001f;
  } lidar;
  struct RGBD {
    float normal_a = 0.001f;
    float normal_b = 0.002f;
    float normal_c = 0.4f;
    float lateral_factor = 0.001f;
  } rgbd;
  struct Constant {
    float uncertainty = 0.03f;
  } constant;
};
}  // namespace config
}  // namespace fastdem
