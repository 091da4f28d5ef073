// The reference's own API-level tests re-expressed against the MI355X engine through the C++
// mirror of fastdem::FastDEM / ElevationMap (each TEST cites the gtest it restates).  Needs a GPU.
//   fastdem/tests/test_elevation_map.cpp, test_fastdem_integration.cpp, test_online_mode.cpp,
//   test_dual_layer.cpp, test_config.cpp (validation), test_sensor_models.cpp (host classes),
//   test_postprocess.cpp:73-190 (raycasting)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <memory>
#include <optional>

#include "fastdem/fastdem.hpp"
#include "fastdem/io/npz.hpp"
#include "fastdem/postprocess/feature_extraction.hpp"
#include "fastdem/postprocess/inpainting.hpp"
#include "fastdem/postprocess/spatial_smoothing.hpp"
#include "fastdem/postprocess/uncertainty_fusion.hpp"
#include "mini_test.hpp"

using namespace fastdem;

namespace {
PointCloud makeGroundCloud(float height, int grid_half = 3, float spacing = 0.3f) {
  PointCloud cloud;  // 7x7 grid at 0.3 m (test_fastdem_integration.cpp:32-41)
  for (int i = -grid_half; i <= grid_half; ++i)
    for (int j = -grid_half; j <= grid_half; ++j) cloud.add(i * spacing, j * spacing, height);
  return cloud;
}
struct Fixture {
  ElevationMap map;
  Eigen::Isometry3d T_base_sensor = Eigen::Isometry3d::Identity();
  Eigen::Isometry3d T_world_base = Eigen::Isometry3d::Identity();
  Fixture() { map.setGeometry(10.0f, 10.0f, 0.5f); }
};
}  // namespace

// ---------------------------------------------------------------- test_elevation_map.cpp ----
TEST(ElevationMap, DefaultConstructorAndGeometry) {  // :17-38
  ElevationMap m;
  EXPECT_FALSE(m.isInitialized());
  m.setGeometry(10.0f, 10.0f, 0.5f);
  EXPECT_TRUE(m.isInitialized());
  EXPECT_EQ(m.getSize()(0), 20);
  EXPECT_EQ(m.getSize()(1), 20);
  EXPECT_TRUE(m.exists(layer::elevation));
  EXPECT_TRUE(m.exists(layer::elevation_min));
  EXPECT_TRUE(m.exists(layer::elevation_max));
  EXPECT_TRUE(m.isEmpty());
  EXPECT_TRUE(m.isInside(nanogrid::Position(0.0, 0.0)));
  EXPECT_FALSE(m.isInside(nanogrid::Position(100.0, 100.0)));
}
TEST(ElevationMap, AtAndElevationAtRoundTrip) {  // :40-71, :142-151
  ElevationMap m(10.0f, 10.0f, 0.5f, "map");
  EXPECT_EQ(m.getFrameId(), std::string("map"));
  nanogrid::Index idx;
  ASSERT_TRUE(m.getIndex(nanogrid::Position(1.0, 2.0), idx));
  m.at(layer::elevation, idx) = 5.0f;
  EXPECT_FLOAT_EQ(m.elevationAt(idx), 5.0f);
  EXPECT_FLOAT_EQ(m.elevationAt(nanogrid::Position(1.0, 2.0)), 5.0f);
  EXPECT_TRUE(m.hasElevationAt(idx));
  EXPECT_FALSE(m.isEmptyAt(idx));
  EXPECT_TRUE(std::isnan(m.elevationAt(nanogrid::Position(100.0, 100.0))));
  m.clearAt(idx);
  EXPECT_TRUE(m.isEmptyAt(idx));
  nanogrid::Position p;
  ASSERT_TRUE(m.getPosition(idx, p));
  nanogrid::Index back;
  ASSERT_TRUE(m.getIndex(p, back));
  EXPECT_EQ(back(0), idx(0));
  EXPECT_EQ(back(1), idx(1));
}
TEST(ElevationMap, HostWritesReachTheDeviceAndSnapshot) {
  Fixture f;
  nanogrid::Index idx;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0.0, 0.0), idx));
  f.map.at(layer::elevation, idx) = 7.0f;  // written on the host BEFORE the engine runs
  FastDEM mapper(f.map);
  mapper.setSensorModel(SensorType::Constant).setMappingMode(MappingMode::GLOBAL);
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  // The device saw the host-written 7.0: with the stored P = 0 the Kalman gain is 0, so the
  // estimate stays 7.0, P is clamped up to min_variance and the count goes 0 -> 1
  // (kalman_estimation.hpp:115-126).  Had the write been lost, the cell would read 1.0.
  const float fused = f.map.elevationAt(idx);
  EXPECT_FLOAT_EQ(fused, 7.0f);
  EXPECT_FLOAT_EQ(f.map.at(layer::kalman_p, idx), 0.0001f);
  EXPECT_FLOAT_EQ(f.map.at(layer::n_points, idx), 1.0f);
  ElevationMap snap = f.map.snapshot({layer::elevation, "does_not_exist"});
  EXPECT_FLOAT_EQ(snap.elevationAt(idx), fused);
  EXPECT_FALSE(snap.exists(layer::variance));
}

// -------------------------------------------------------- test_fastdem_integration.cpp ----
TEST(FastDEMIntegration, IntegrateUpdatesElevation) {  // :46-60
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-2.0f, 5.0f).setRangeFilter(0.0f, 20.0f).setSensorModel(SensorType::Constant)
      .setEstimatorType(EstimationType::Kalman);
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  nanogrid::Position center(0.0, 0.0);
  ASSERT_TRUE(f.map.hasElevationAt(center));
  EXPECT_NEAR(f.map.elevationAt(center), 1.0f, 0.1f);
}
TEST(FastDEMIntegration, EmptyCloudIsNoOp) {  // :62-70
  Fixture f;
  FastDEM mapper(f.map);
  PointCloud empty;
  mapper.integrate(empty, f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(FastDEMIntegration, HeightFilterRejectsOutOfRange) {  // :72-80
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(0.0f, 2.0f);
  mapper.integrate(makeGroundCloud(10.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(FastDEMIntegration, MultipleIntegrations) {  // :82-100
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-5.0f, 15.0f).setRangeFilter(0.0f, 20.0f).setSensorModel(SensorType::Constant)
      .setEstimatorType(EstimationType::Kalman);
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  mapper.integrate(makeGroundCloud(1.5f), f.T_base_sensor, f.T_world_base);
  nanogrid::Position center(0.0, 0.0);
  ASSERT_TRUE(f.map.hasElevationAt(center));
  EXPECT_GT(f.map.elevationAt(center), 0.9f);
  EXPECT_LT(f.map.elevationAt(center), 1.6f);
}
TEST(FastDEMIntegration, SensorModelsTimesEstimators) {  // :129-175
  for (SensorType s : {SensorType::Constant, SensorType::LiDAR, SensorType::RGBD})
    for (EstimationType e : {EstimationType::Kalman, EstimationType::P2Quantile}) {
      Fixture f;
      f.T_base_sensor.translation().z() = 0.5;
      FastDEM mapper(f.map);
      mapper.setHeightFilter(-5.0f, 15.0f).setSensorModel(s).setEstimatorType(e);
      for (int k = 0; k < 6; ++k) EXPECT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
      EXPECT_FALSE(f.map.isEmpty());
      EXPECT_TRUE(f.map.exists(layer::variance));
      EXPECT_TRUE(f.map.exists(e == EstimationType::Kalman ? layer::kalman_p : layer::p2_q0));
    }
}
TEST(FastDEMIntegration, GlobalModeFixedOrigin) {  // :179-196
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setMappingMode(MappingMode::GLOBAL).setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant);
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  f.T_world_base.translation().x() = 3.0;
  mapper.integrate(makeGroundCloud(2.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(f.map.hasElevationAt(nanogrid::Position(0.0, 0.0)));
}
TEST(FastDEMIntegration, LocalModeFollowsRobot) {  // :198-215
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setMappingMode(MappingMode::LOCAL).setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant);
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  f.T_world_base.translation().x() = 100.0;
  mapper.integrate(makeGroundCloud(2.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_FALSE(f.map.isInside(nanogrid::Position(0.0, 0.0)));
}
TEST(FastDEMIntegration, ConstructFromConfig) {  // :219-236
  Fixture f;
  Config cfg;
  cfg.mapping.estimation_type = EstimationType::Kalman;
  cfg.sensor_model.type = SensorType::Constant;
  cfg.point_filter.z_min = -2.0f;
  cfg.point_filter.z_max = 5.0f;
  cfg.raycasting.enabled = false;
  FastDEM mapper(f.map, cfg);
  EXPECT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  ASSERT_TRUE(f.map.hasElevationAt(nanogrid::Position(0.0, 0.0)));
  EXPECT_NEAR(f.map.elevationAt(nanogrid::Position(0.0, 0.0)), 1.0f, 0.1f);
}
TEST(FastDEMIntegration, ConfigPointFilterApplied) {  // :238-249
  Fixture f;
  Config cfg;
  cfg.point_filter.z_min = 0.0f;
  cfg.point_filter.z_max = 2.0f;
  FastDEM mapper(f.map, cfg);
  mapper.integrate(makeGroundCloud(5.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(FastDEMIntegration, SensorOffsetApplied) {  // :253-267
  Fixture f;
  f.T_base_sensor.translation().z() = 1.0;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant);
  mapper.integrate(makeGroundCloud(0.0f), f.T_base_sensor, f.T_world_base);
  ASSERT_TRUE(f.map.hasElevationAt(nanogrid::Position(0.0, 0.0)));
  EXPECT_NEAR(f.map.elevationAt(nanogrid::Position(0.0, 0.0)), 1.0f, 0.2f);
}
TEST(FastDEMIntegration, RotatedTransform) {  // :269-283
  Fixture f;
  f.T_world_base.rotate(Eigen::AngleAxisd(M_PI / 2, Eigen::Vector3d::UnitZ()));
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant);
  EXPECT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  EXPECT_FALSE(f.map.isEmpty());
}
TEST(FastDEMIntegration, RangeFilterRejectsClosePoints) {  // :287-296
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setRangeFilter(5.0f, 20.0f);
  mapper.integrate(makeGroundCloud(1.0f, 2, 0.3f), f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(FastDEMIntegration, CombinedHeightAndRangeFilter) {  // :298-316
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(0.0f, 3.0f).setRangeFilter(0.0f, 20.0f).setSensorModel(SensorType::Constant);
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_FALSE(f.map.isEmpty());
  f.map.clearAll();
  mapper.integrate(makeGroundCloud(5.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(FastDEMIntegration, ReturnValues) {  // :357-378
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant);
  EXPECT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  PointCloud empty;
  EXPECT_FALSE(mapper.integrate(empty, f.T_base_sensor, f.T_world_base));
  mapper.setHeightFilter(100.0f, 200.0f);
  EXPECT_FALSE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  mapper.reset();
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(FastDEMIntegration, ScanCallbacksFire) {  // :320-353
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant);
  bool pre = false, ras = false;
  size_t n_pre = 0, n_ras = 0;
  float z_ras = 0.f;
  bool has_cov = false;
  float c22 = -1.f, c00 = -1.f;
  mapper.onScanPreprocessed([&](const PointCloud& c) {
    pre = true; n_pre = c.size();
    has_cov = c.hasCovariance();                       // fastdem.cpp:182-187: the cloud carries R Sigma R^T
    if (has_cov && c.size()) { c22 = c.covariance(0)(2, 2); c00 = c.covariance(0)(0, 0); }
  });
  mapper.onScanRasterized([&](const PointCloud& c) { ras = true; n_ras = c.size(); z_ras = c.point(0)[2]; });
  mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base);
  EXPECT_TRUE(pre);
  EXPECT_EQ(n_pre, size_t(49));
  EXPECT_TRUE(has_cov);
  EXPECT_GT(c22, 0.0f);
  EXPECT_GT(c00, 0.0f);
  EXPECT_TRUE(ras);
  EXPECT_EQ(n_ras, size_t(mapper.lastStats().n_cells_touched));
  EXPECT_GT(n_ras, 0u);
  EXPECT_FLOAT_EQ(z_ras, 1.0f);
}
TEST(FastDEMIntegration, CustomSensorModelSubclass) {  // fastdem.hpp:79-80
  struct Wide : SensorModel {  // user model: 1 m^2 isotropic
    Eigen::Matrix3f computeCovariance(const Eigen::Vector3f&) const override { return Eigen::Matrix3f::Identity() * 0.005f; }
  };
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setMappingMode(MappingMode::GLOBAL).setSensorModel(std::make_unique<Wide>());
  EXPECT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  nanogrid::Index idx;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0.0, 0.0), idx));
  EXPECT_FLOAT_EQ(f.map.at(layer::kalman_p, idx), 0.005f);  // first update: P = R = sigma_z^2
}

// ---------------------------------------------------------------- test_online_mode.cpp ----
namespace {
class MockCalibration : public Calibration {
 public:
  explicit MockCalibration(Eigen::Isometry3d e = Eigen::Isometry3d::Identity()) : extrinsic_(e) {}
  std::optional<Eigen::Isometry3d> getExtrinsic(const std::string& frame) const override {
    if (frame == "unknown_sensor") return std::nullopt;
    return extrinsic_;
  }
  std::string getBaseFrame() const override { return "base_link"; }
 private:
  Eigen::Isometry3d extrinsic_;
};
class MockOdometry : public Odometry {
 public:
  std::optional<Eigen::Isometry3d> getPoseAt(uint64_t) const override {
    if (fail_) return std::nullopt;
    return pose_;
  }
  std::string getWorldFrame() const override { return "map"; }
  void setPose(const Eigen::Isometry3d& p) { pose_ = p; }
  void setFail(bool f) { fail_ = f; }
 private:
  Eigen::Isometry3d pose_ = Eigen::Isometry3d::Identity();
  bool fail_ = false;
};
std::shared_ptr<PointCloud> makeSharedCloud(float h = 1.0f, const std::string& frame = "lidar") {
  auto c = std::make_shared<PointCloud>(makeGroundCloud(h));
  c->setFrameId(frame);
  c->setTimestamp(1000000000ULL);
  return c;
}
}  // namespace

TEST(OnlineMode, IntegrateWithTransformProvider) {  // :98-114
  Fixture f;
  auto cal = std::make_shared<MockCalibration>();
  auto odo = std::make_shared<MockOdometry>();
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-2.0f, 5.0f).setRangeFilter(0.0f, 20.0f).setSensorModel(SensorType::Constant)
      .setCalibrationProvider(cal).setOdometryProvider(odo);
  ASSERT_TRUE(mapper.hasTransformProvider());
  EXPECT_TRUE(mapper.integrate(makeSharedCloud(1.0f)));
  ASSERT_TRUE(f.map.hasElevationAt(nanogrid::Position(0.0, 0.0)));
  EXPECT_NEAR(f.map.elevationAt(nanogrid::Position(0.0, 0.0)), 1.0f, 0.1f);
}
TEST(OnlineMode, ValidationFailuresReturnFalse) {  // :129-175
  Fixture f;
  auto cal = std::make_shared<MockCalibration>();
  auto odo = std::make_shared<MockOdometry>();
  FastDEM mapper(f.map);
  EXPECT_FALSE(mapper.hasTransformProvider());
  EXPECT_FALSE(mapper.integrate(makeSharedCloud()));                 // no providers
  mapper.setCalibrationProvider(cal);
  EXPECT_FALSE(mapper.hasTransformProvider());
  mapper.setOdometryProvider(odo);
  EXPECT_TRUE(mapper.hasTransformProvider());
  EXPECT_FALSE(mapper.integrate(std::shared_ptr<PointCloud>()));     // null cloud
  EXPECT_FALSE(mapper.integrate(std::make_shared<PointCloud>()));    // empty cloud
  EXPECT_FALSE(mapper.integrate(makeSharedCloud(1.0f, "")));         // no frame id
  EXPECT_FALSE(mapper.integrate(makeSharedCloud(1.0f, "unknown_sensor")));
  odo->setFail(true);
  EXPECT_FALSE(mapper.integrate(makeSharedCloud()));
  EXPECT_TRUE(f.map.isEmpty());
}
TEST(OnlineMode, PoseOffsetLandsDataAtRobot) {  // :221-241
  Fixture f;
  auto cal = std::make_shared<MockCalibration>();
  auto odo = std::make_shared<MockOdometry>();
  Eigen::Isometry3d pose = Eigen::Isometry3d::Identity();
  pose.translation().x() = 2.0;
  odo->setPose(pose);
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-5.0f, 15.0f).setSensorModel(SensorType::Constant).setCalibrationProvider(cal)
      .setOdometryProvider(odo);
  EXPECT_TRUE(mapper.integrate(makeSharedCloud(1.0f)));
  EXPECT_TRUE(f.map.hasElevationAt(nanogrid::Position(2.0, 0.0)));
}

// ------------------------------------------------------------------ test_dual_layer.cpp ----
namespace {
PointCloud cloudOf(std::initializer_list<std::array<float, 3>> pts) {
  PointCloud c;
  for (const auto& p : pts) c.add(p[0], p[1], p[2]);
  return c;
}
config::Mapping kalmanMapping() {
  config::Mapping cfg;
  cfg.mode = MappingMode::GLOBAL;
  cfg.estimation_type = EstimationType::Kalman;
  cfg.kalman.min_variance = 0.0001f;
  cfg.kalman.max_variance = 1.0f;
  cfg.kalman.process_noise = 0.0f;
  return cfg;
}
}  // namespace
TEST(DualLayer, GroundObstacleSeparationAndOverwrite) {  // :66-83, :121-143
  Fixture f;
  ElevationMapping mapping(f.map, kalmanMapping());
  Eigen::Vector2d robot(0.0, 0.0);
  mapping.update(cloudOf({{0, 0, 0.0f}, {0, 0, 3.0f}}), robot);
  nanogrid::Index idx;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0, 0), idx));
  EXPECT_NEAR(f.map.at(layer::elevation, idx), 0.0f, 0.1f);
  EXPECT_NEAR(f.map.at(layer::obstacle, idx), 3.0f, 0.1f);
  mapping.update(cloudOf({{0, 0, 0.1f}, {0, 0, 3.1f}}), robot);
  EXPECT_GT(f.map.at(layer::elevation, idx), -0.05f);
  EXPECT_LT(f.map.at(layer::elevation, idx), 0.15f);
  EXPECT_FLOAT_EQ(f.map.at(layer::obstacle, idx), 3.1f);
}
TEST(DualLayer, SinglePointOnlyGroundAndObstacleClears) {  // :106-119, :188-203
  Fixture f;
  ElevationMapping mapping(f.map, kalmanMapping());
  Eigen::Vector2d robot(0.0, 0.0);
  nanogrid::Index idx;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0, 0), idx));
  auto obs = mapping.update(cloudOf({{0, 0, 0.0f}, {0, 0, 2.0f}}), robot);
  EXPECT_EQ(obs.size(), size_t(1));
  EXPECT_FLOAT_EQ(f.map.at(layer::obstacle, idx), 2.0f);
  mapping.update(cloudOf({{0, 0, 0.0f}}), robot);
  EXPECT_TRUE(std::isnan(f.map.at(layer::obstacle, idx)));
}
TEST(DualLayer, ElevationMaxReflectsTrueMaxAndQuantile) {  // :145-186
  Fixture f;
  ElevationMapping mapping(f.map, kalmanMapping());
  Eigen::Vector2d robot(0.0, 0.0);
  nanogrid::Index idx;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0, 0), idx));
  mapping.update(cloudOf({{0, 0, 0.0f}, {0, 0, 3.0f}}), robot);
  EXPECT_FLOAT_EQ(f.map.at(layer::elevation_max, idx), 3.0f);
  mapping.update(cloudOf({{0, 0, 0.0f}, {0, 0, 5.0f}}), robot);
  EXPECT_FLOAT_EQ(f.map.at(layer::elevation_max, idx), 5.0f);
  Fixture g;
  config::Mapping q;
  q.mode = MappingMode::GLOBAL;
  q.estimation_type = EstimationType::P2Quantile;
  ElevationMapping qm(g.map, q);
  for (int i = 0; i < 10; ++i) {
    const float noise = (i % 2 == 0) ? 0.05f : -0.05f;
    qm.update(cloudOf({{0, 0, 0.0f + noise}, {0, 0, 5.0f + noise}}), robot);
  }
  ASSERT_TRUE(g.map.getIndex(nanogrid::Position(0, 0), idx));
  EXPECT_NEAR(g.map.at(layer::elevation, idx), 0.0f, 0.5f);
  EXPECT_NEAR(g.map.at(layer::obstacle, idx), 5.0f, 0.1f);
}

// ------------------------------------------------ test_postprocess.cpp (raycasting) ----
namespace {
struct PostFixture {  // test_postprocess.cpp:24-36
  ElevationMap map;
  PostFixture() { map.setGeometry(10.0f, 10.0f, 0.5f); }
  nanogrid::Index at(double x, double y) const {
    nanogrid::Index idx;
    map.getIndex(nanogrid::Position(x, y), idx);
    return idx;
  }
};
}  // namespace
TEST(Raycasting, CreatesLayers) {  // :75-92
  PostFixture f;
  f.map.at(layer::elevation, f.at(0.0, 0.0)) = 1.0f;
  PointCloud cloud;
  cloud.add(1.0f, 0.0f, 0.5f);
  config::Raycasting cfg;
  cfg.enabled = true;
  applyRaycasting(f.map, cloud, Eigen::Vector3f(0.0f, 0.0f, 5.0f), cfg);
  EXPECT_TRUE(f.map.exists(layer::ghost_removal));
  EXPECT_TRUE(f.map.exists(layer::raycasting));
  EXPECT_TRUE(f.map.exists(layer::visibility_logodds));
}
TEST(Raycasting, ClearsGhostCell) {  // :94-117
  PostFixture f;
  const nanogrid::Index ghost = f.at(2.0, 0.0);
  f.map.at(layer::elevation, ghost) = 10.0f;
  PointCloud cloud;
  cloud.add(4.0f, 0.0f, 0.0f);
  config::Raycasting cfg;
  cfg.enabled = true;
  cfg.height_conflict_threshold = 0.05f;
  cfg.log_odds_ghost = 0.5f;
  cfg.clear_threshold = -0.4f;
  applyRaycasting(f.map, cloud, Eigen::Vector3f(0.0f, 0.0f, 5.0f), cfg);
  EXPECT_TRUE(std::isnan(f.map.at(layer::elevation, ghost)));
  EXPECT_FLOAT_EQ(f.map.at(layer::ghost_removal, ghost), 1.0f);
}
TEST(Raycasting, ObservedCellProtected) {  // :119-146
  PostFixture f;
  const nanogrid::Index cell = f.at(2.0, 0.0);
  f.map.at(layer::elevation, cell) = 2.0f;
  PointCloud cloud;
  cloud.add(4.0f, 0.0f, 0.0f);
  cloud.add(2.0f, 0.0f, 0.3f);
  config::Raycasting cfg;
  cfg.enabled = true;
  cfg.log_odds_observed = 0.8f;
  cfg.log_odds_ghost = 0.5f;
  cfg.clear_threshold = -0.4f;
  applyRaycasting(f.map, cloud, Eigen::Vector3f(0.0f, 0.0f, 5.0f), cfg);
  EXPECT_FALSE(std::isnan(f.map.at(layer::elevation, cell)));
}
TEST(Raycasting, GhostRequiresAccumulation) {  // :148-175
  PostFixture f;
  const nanogrid::Index ghost = f.at(2.0, 0.0);
  PointCloud cloud;
  cloud.add(4.0f, 0.0f, 0.0f);
  config::Raycasting cfg;
  cfg.enabled = true;
  cfg.log_odds_ghost = 0.2f;
  cfg.clear_threshold = -0.9f;
  for (int i = 0; i < 4; ++i) {
    f.map.at(layer::elevation, ghost) = 10.0f;
    applyRaycasting(f.map, cloud, Eigen::Vector3f(0.0f, 0.0f, 5.0f), cfg);
  }
  EXPECT_FALSE(std::isnan(f.map.at(layer::elevation, ghost)));
  applyRaycasting(f.map, cloud, Eigen::Vector3f(0.0f, 0.0f, 5.0f), cfg);
  EXPECT_TRUE(std::isnan(f.map.at(layer::elevation, ghost)));
}
TEST(Raycasting, DisabledIsNoOp) {  // :177-190
  PostFixture f;
  PointCloud cloud;
  cloud.add(1.0f, 0.0f, 0.5f);
  config::Raycasting cfg;
  cfg.enabled = false;
  applyRaycasting(f.map, cloud, Eigen::Vector3f(0.0f, 0.0f, 5.0f), cfg);
  EXPECT_FALSE(f.map.exists(layer::ghost_removal));
  EXPECT_FALSE(f.map.exists(layer::raycasting));
  EXPECT_FALSE(f.map.exists(layer::visibility_logodds));
}
TEST(Raycasting, EnabledThroughIntegrate) {  // fastdem.cpp:152-159, default.yaml:40-41
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setSensorModel(SensorType::Constant).enableRaycasting(true);
  f.T_base_sensor.translation() = Eigen::Vector3d(0.0, 0.0, 5.0);
  PointCloud cloud;
  cloud.add(4.0f, 0.0f, -5.0f);  // ground point 4 m ahead, seen from 5 m up
  ASSERT_TRUE(mapper.integrate(cloud, f.T_base_sensor, f.T_world_base));
  EXPECT_TRUE(f.map.exists(layer::raycasting));
  nanogrid::Index mid;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(2.0, 0.0), mid));
  EXPECT_NEAR(f.map.at(layer::raycasting, mid), 2.5f, 1e-4f);  // x = 2.0 is the far edge of its cell: t = 0.5
  nanogrid::Index hit;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(4.0, 0.0), hit));
  EXPECT_FLOAT_EQ(f.map.at(layer::visibility_logodds, hit), 0.4f);
}

// -------------------------------- test_postprocess.cpp (inpainting / fusion / smoothing / features) ----
TEST(Postprocess, InpaintingFillsHoleAndPreservesValues) {  // :39-69
  PostFixture f;
  const nanogrid::Index c = f.at(0.0, 0.0);
  for (int dr = -1; dr <= 1; ++dr)
    for (int dc = -1; dc <= 1; ++dc)
      if (dr || dc) f.map.at(layer::elevation, nanogrid::Index(c(0) + dr, c(1) + dc)) = 1.0f;
  ASSERT_TRUE(std::isnan(f.map.at(layer::elevation, c)));
  applyInpainting(f.map, 3, 2);
  ASSERT_TRUE(f.map.exists(layer::elevation_inpainted));
  EXPECT_TRUE(std::isfinite(f.map.at(layer::elevation_inpainted, c)));
  EXPECT_NEAR(f.map.at(layer::elevation_inpainted, c), 1.0f, 0.01f);
  PostFixture g;
  g.map.get(layer::elevation).setConstant(2.0f);
  applyInpainting(g.map, 3, 2);
  EXPECT_FLOAT_EQ(g.map.at(layer::elevation_inpainted, g.at(0.0, 0.0)), 2.0f);
}
TEST(Postprocess, UncertaintyFusion) {  // :192-240
  PostFixture f;
  f.map.add(layer::upper_bound, NAN);
  f.map.add(layer::lower_bound, NAN);
  const nanogrid::Index c = f.at(0.0, 0.0);
  for (int dr = -1; dr <= 1; ++dr)
    for (int dc = -1; dc <= 1; ++dc) {
      const nanogrid::Index idx(c(0) + dr, c(1) + dc);
      const float h = 1.0f + 0.1f * dr;
      f.map.at(layer::elevation, idx) = h;
      f.map.at(layer::upper_bound, idx) = h + 0.2f;
      f.map.at(layer::lower_bound, idx) = h - 0.2f;
    }
  config::UncertaintyFusion cfg;
  cfg.enabled = true;
  cfg.search_radius = 0.6f;
  cfg.spatial_sigma = 0.3f;
  cfg.min_valid_neighbors = 1;
  applyUncertaintyFusion(f.map, cfg);
  const float upper = f.map.at(layer::upper_bound, c), lower = f.map.at(layer::lower_bound, c);
  EXPECT_TRUE(std::isfinite(upper));
  EXPECT_TRUE(std::isfinite(lower));
  EXPECT_GT(upper, lower);
  ElevationMap empty_map;  // missing bounds: returns early, no crash
  empty_map.setGeometry(10.0f, 10.0f, 0.5f);
  EXPECT_NO_THROW(applyUncertaintyFusion(empty_map, cfg));
  PostFixture g;
  cfg.enabled = false;
  applyUncertaintyFusion(g.map, cfg);
  EXPECT_FALSE(g.map.exists(layer::upper_bound));
}
TEST(Postprocess, SpatialSmoothing) {  // :243-264
  PostFixture f;
  const nanogrid::Index c = f.at(0.0, 0.0);
  for (int dr = -2; dr <= 2; ++dr)
    for (int dc = -2; dc <= 2; ++dc) f.map.at(layer::elevation, nanogrid::Index(c(0) + dr, c(1) + dc)) = 1.0f;
  f.map.at(layer::elevation, c) = 100.0f;
  applySpatialSmoothing(f.map, layer::elevation, 3, 5);
  EXPECT_NEAR(f.map.at(layer::elevation, c), 1.0f, 0.01f);
  EXPECT_NO_THROW(applySpatialSmoothing(f.map, "nonexistent_layer"));
}
TEST(Postprocess, FeatureExtraction) {  // :268-400
  PostFixture f;
  f.map.get(layer::elevation).setConstant(1.0f);
  applyFeatureExtraction(f.map, 0.6f, 4);
  for (const char* n : {layer::step, layer::slope, layer::roughness, layer::curvature, layer::normal_x,
                        layer::normal_y, layer::normal_z})
    EXPECT_TRUE(f.map.exists(n));
  const nanogrid::Index c = f.at(0.0, 0.0);
  EXPECT_NEAR(f.map.at(layer::slope, c), 0.0f, 1.0f);
  EXPECT_NEAR(f.map.at(layer::roughness, c), 0.0f, 0.001f);
  EXPECT_NEAR(f.map.at(layer::step, c), 0.0f, 0.001f);
  EXPECT_NEAR(f.map.at(layer::normal_z, c), 1.0f, 0.01f);
  PostFixture t;  // tilted plane: 0.5 rise / run along the rows
  auto& el = t.map.get(layer::elevation);
  for (int r = 0; r < el.rows(); ++r)
    for (int col = 0; col < el.cols(); ++col) el(r, col) = float(r) * 0.5f * 0.5f;
  applyFeatureExtraction(t.map, 0.6f, 4);
  EXPECT_GT(t.map.at(layer::slope, t.at(0.0, 0.0)), 10.0f);
  EXPECT_LT(t.map.at(layer::slope, t.at(0.0, 0.0)), 45.0f);
  EXPECT_GT(t.map.at(layer::normal_z, t.at(0.0, 0.0)), 0.0f);
  PostFixture s;  // step edge between the two halves
  auto& es = s.map.get(layer::elevation);
  for (int r = 0; r < es.rows(); ++r)
    for (int col = 0; col < es.cols(); ++col) es(r, col) = col < es.cols() / 2 ? 0.0f : 1.0f;
  applyFeatureExtraction(s.map, 0.6f, 4);
  EXPECT_GT(s.map.at(layer::step, s.at(0.0, 0.0)), 0.5f);
  ElevationMap empty_map;  // default-constructed: no crash
  EXPECT_NO_THROW(applyFeatureExtraction(empty_map));
  PostFixture n;  // all NaN: layers exist, nothing computed; a single cell has too few neighbours
  applyFeatureExtraction(n.map, 0.6f, 4);
  EXPECT_TRUE(n.map.exists(layer::slope));
  EXPECT_FALSE(std::isfinite(n.map.at(layer::slope, n.at(0.0, 0.0))));
  n.map.at(layer::elevation, n.at(0.0, 0.0)) = 1.0f;
  applyFeatureExtraction(n.map, 0.6f, 4);
  EXPECT_FALSE(std::isfinite(n.map.at(layer::slope, n.at(0.0, 0.0))));
}

// ------------------------------------------- PointCloud2-shaped message straight to the device ----
namespace {
struct FakeField { std::string name; uint32_t offset; uint8_t datatype; uint32_t count; };
struct FakeCloud2 {  // the members of sensor_msgs::PointCloud2 that nanopcl::from reads
  uint32_t height = 1, width = 0, point_step = 0;
  std::vector<FakeField> fields;
  std::vector<uint8_t> data;
};
}  // namespace
TEST(Cloud2, IntegrateMessageEqualsIntegrateCloud) {  // nanopcl/bridge/ros/impl.hpp:174-246
  Fixture a, b;
  FastDEM ma(a.map), mb(b.map);
  const PointCloud cloud = makeGroundCloud(0.75f);
  FakeCloud2 msg;
  msg.point_step = 20;
  msg.fields = {{"x", 0, 7, 1}, {"y", 4, 7, 1}, {"z", 8, 7, 1}, {"ring", 12, 4, 1}, {"intensity", 16, 7, 1}};
  const float nanv = NAN;
  for (size_t i = 0; i <= cloud.size(); ++i) {  // one extra NaN point: from_impl drops it
    const bool extra = i == cloud.size();
    float rec[5] = {extra ? nanv : cloud.xData()[i], extra ? 0.f : cloud.yData()[i], extra ? 0.f : cloud.zData()[i], 0.f,
                    0.5f};
    const uint8_t* p = reinterpret_cast<const uint8_t*>(rec);
    msg.data.insert(msg.data.end(), p, p + 20);
  }
  msg.width = uint32_t(cloud.size() + 1);
  ASSERT_TRUE(ma.integrateCloud2(msg, a.T_base_sensor, a.T_world_base));
  PointCloud with_i;
  for (size_t i = 0; i < cloud.size(); ++i) with_i.add(cloud.xData()[i], cloud.yData()[i], cloud.zData()[i], nanopcl::Intensity(0.5f));
  ASSERT_TRUE(mb.integrate(with_i, b.T_base_sensor, b.T_world_base));
  EXPECT_EQ(ma.lastStats().n_input, uint32_t(cloud.size()));
  ASSERT_TRUE(a.map.exists(layer::intensity));
  for (const auto& name : b.map.getLayers()) {
    const auto& x = a.map.get(name);
    const auto& y = b.map.get(name);
    size_t bad = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      const float u = x.data()[i], v = y.data()[i];
      if (std::isnan(u) ? !std::isnan(v) : !(u == v)) ++bad;
    }
    EXPECT_EQ(bad, size_t(0));
  }
  FakeCloud2 empty = msg;
  empty.width = 0;
  empty.data.clear();
  EXPECT_FALSE(ma.integrateCloud2(empty, a.T_base_sensor, a.T_world_base));
}

// ------------------------------------------------------- pinned input clouds ----
TEST(HostPool, BlocksAreReusedAndCloudsArePinned) {
  void* a = fdm_host_alloc(100000);
  ASSERT_TRUE(a != nullptr);
  EXPECT_EQ(fdm_host_is_pinned(a), 1);  // this suite runs on a GPU box
  fdm_host_free(a);
  void* b = fdm_host_alloc(70000);      // same 128 KiB class: the idle block comes back
  EXPECT_TRUE(a == b);
  EXPECT_EQ(fdm_host_is_pinned(b), 1);
  fdm_host_free(b);
  EXPECT_EQ(fdm_host_is_pinned(b), 0);  // idle blocks are not live
  fdm_host_free(b);                     // double free of a pooled block is ignored
  int on_stack = 0;
  fdm_host_free(&on_stack);             // foreign pointers are ignored
  EXPECT_EQ(fdm_host_is_pinned(&on_stack), 0);
  fdm_host_trim();
  const PointCloud cloud = makeGroundCloud(0.5f);
  EXPECT_EQ(fdm_host_is_pinned(cloud.xData()), 1);
  EXPECT_EQ(fdm_host_is_pinned(cloud.zData()), 1);
  PointCloud copy = cloud;              // copies get their own pinned blocks
  EXPECT_TRUE(copy.xData() != cloud.xData());
  EXPECT_EQ(fdm_host_is_pinned(copy.yData()), 1);
}

TEST(HostPool, InPlaceIntegrateEqualsCopiedIntegrate) {
  // the same scans through integrate(): pinned clouds read in place by the bin kernel vs the
  // copy path (option zero_copy = 0) — every layer identical
  Fixture a, b;
  FastDEM ma(a.map), mb(b.map);
  for (int k = 0; k < 3; ++k) {
    PointCloud cloud;
    for (float x = -2.0f; x <= 2.0f; x += 0.05f)
      for (float y = -2.0f; y <= 2.0f; y += 0.05f)
        cloud.add(x, y, 0.1f * float(k) + 0.3f * std::sin(3.0f * x) * std::cos(2.0f * y), nanopcl::Intensity(x * y));
    ASSERT_TRUE(ma.integrate(cloud, a.T_base_sensor, a.T_world_base));
    ASSERT_EQ(fdm_engine_set_option(b.map.engine(), "zero_copy", 0), FDM_OK);
    ASSERT_TRUE(mb.integrate(cloud, b.T_base_sensor, b.T_world_base));
  }
  for (const auto& name : b.map.getLayers()) {
    const auto& x = a.map.get(name);
    const auto& y = b.map.get(name);
    size_t bad = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      const float u = x.data()[i], v = y.data()[i];
      if (std::isnan(u) ? !std::isnan(v) : !(u == v)) ++bad;
    }
    EXPECT_EQ(bad, size_t(0));
  }
  EXPECT_TRUE(a.map.hasElevationAt(nanogrid::Position(0.0, 0.0)));
}

// ------------------------------------------------------- test_map_io.cpp (NPZ) ----
namespace {
struct NpzFixture {  // test_map_io.cpp:19-41
  ElevationMap map;
  std::string path;
  explicit NpzFixture(const char* file = "fdm_cpp_test_io.npz") {
    map = ElevationMap(10.0f, 8.0f, 0.5f, "map");
    auto& elev = map.get(layer::elevation);
    for (size_t i = 0; i < elev.size(); ++i) elev.data()[i] = static_cast<float>(i) * 0.1f;
    elev(0, 0) = NAN;
    elev(1, 1) = NAN;
    const char* tmp = std::getenv("TMPDIR");
    path = std::string(tmp ? tmp : "/tmp") + "/" + file;
  }
  ~NpzFixture() { std::remove(path.c_str()); }
};
}  // namespace
TEST(Npz, RoundTrip) {  // :43-73
  NpzFixture f;
  ASSERT_TRUE(io::saveNpz(f.path, f.map));
  ElevationMap loaded;
  ASSERT_TRUE(io::loadNpz(f.path, loaded));
  EXPECT_FLOAT_EQ(float(loaded.getResolution()), float(f.map.getResolution()));
  EXPECT_EQ(loaded.getSize()(0), f.map.getSize()(0));
  EXPECT_EQ(loaded.getSize()(1), f.map.getSize()(1));
  EXPECT_EQ(loaded.getFrameId(), std::string("map"));
  EXPECT_NEAR(loaded.getPosition()(0), f.map.getPosition()(0), 1e-4);
  EXPECT_NEAR(loaded.getPosition()(1), f.map.getPosition()(1), 1e-4);
  ASSERT_TRUE(loaded.exists(layer::elevation));
  const auto& a = f.map.get(layer::elevation);
  const auto& b = loaded.get(layer::elevation);
  ASSERT_EQ(a.rows(), b.rows());
  ASSERT_EQ(a.cols(), b.cols());
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const float x = a.data()[i], y = b.data()[i];
    if (std::isnan(x) ? !std::isnan(y) : !(x == y)) ++bad;
  }
  EXPECT_EQ(bad, size_t(0));
}
TEST(Npz, RoundTripWithStartIndex) {  // :75-87
  NpzFixture f;
  f.map.setStartIndex(nanogrid::Index(5, 3));
  ASSERT_TRUE(io::saveNpz(f.path, f.map));
  ElevationMap loaded;
  ASSERT_TRUE(io::loadNpz(f.path, loaded));
  EXPECT_EQ(loaded.getStartIndex()(0), 5);
  EXPECT_EQ(loaded.getStartIndex()(1), 3);
}
TEST(Npz, MultipleLayersSelectiveAndMissing) {  // :89-137
  NpzFixture f;
  f.map.add("variance");
  auto& var = f.map.get("variance");
  for (size_t i = 0; i < var.size(); ++i) var.data()[i] = static_cast<float>(i) * 0.01f;
  f.map.add("intensity");
  ASSERT_TRUE(io::saveNpz(f.path, f.map, {layer::elevation, "variance", "no_such_layer"}));
  ElevationMap loaded;
  ASSERT_TRUE(io::loadNpz(f.path, loaded));
  EXPECT_TRUE(loaded.exists(layer::elevation));
  ASSERT_TRUE(loaded.exists("variance"));
  EXPECT_FALSE(loaded.exists("intensity"));
  EXPECT_FALSE(loaded.exists("no_such_layer"));
  const auto& v = loaded.get("variance");
  size_t bad = 0;
  for (size_t i = 0; i < v.size(); ++i) bad += v.data()[i] == static_cast<float>(i) * 0.01f ? 0 : 1;
  EXPECT_EQ(bad, size_t(0));
}
TEST(Npz, EmptyMap) {  // :139-153
  NpzFixture f;
  ElevationMap empty(4.0f, 4.0f, 0.5f, "empty");
  ASSERT_TRUE(io::saveNpz(f.path, empty));
  ElevationMap loaded;
  ASSERT_TRUE(io::loadNpz(f.path, loaded));
  EXPECT_FLOAT_EQ(float(loaded.getResolution()), float(empty.getResolution()));
  EXPECT_EQ(loaded.getSize()(0), empty.getSize()(0));
  EXPECT_TRUE(loaded.get(layer::elevation).allNaN());
}
TEST(Npz, FutureVersionRejected) {  // :155-177
  NpzFixture f;
  ASSERT_TRUE(io::saveNpz(f.path, f.map));
  std::ifstream ifs(f.path, std::ios::binary);
  std::string content((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
  ifs.close();
  const auto pos = content.find("\"version\": 1");
  ASSERT_TRUE(pos != std::string::npos);
  content.replace(pos, 12, "\"version\":99");
  std::ofstream ofs(f.path, std::ios::binary);
  ofs.write(content.data(), std::streamsize(content.size()));
  ofs.close();
  ElevationMap loaded;
  EXPECT_FALSE(io::loadNpz(f.path, loaded));
}
TEST(Npz, BadPaths) {  // :179-187
  NpzFixture f;
  ElevationMap loaded;
  EXPECT_FALSE(io::loadNpz("/tmp/does_not_exist.npz", loaded));
  EXPECT_FALSE(io::saveNpz("/nonexistent/dir/test.npz", f.map));
}
TEST(Npz, CheckpointOfAMappedSceneForNumpy) {  // the file tests/test_cpp_host_api.py opens with numpy
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setSensorModel(SensorType::Constant);
  f.T_world_base.translation() = Eigen::Vector3d(1.0, -0.5, 0.0);
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(1.25f), f.T_base_sensor, f.T_world_base));
  const char* tmp = std::getenv("TMPDIR");
  const std::string path = std::string(tmp ? tmp : "/tmp") + "/fdm_cpp_checkpoint.npz";
  ASSERT_TRUE(io::saveNpz(path, f.map));
  ElevationMap back;
  ASSERT_TRUE(io::loadNpz(path, back));
  EXPECT_EQ(back.getLayers().size(), f.map.getLayers().size());
  EXPECT_EQ(back.getStartIndex()(0), f.map.getStartIndex()(0));
  EXPECT_EQ(back.get(layer::elevation).countFinite(), f.map.get(layer::elevation).countFinite());
}

// ------------------------------------------------------ test_config.cpp (YAML loading) ----
namespace {
std::string writeTempYaml(const std::string& content, const std::string& name) {  // test_config.cpp:24-30
  const char* tmp = std::getenv("TMPDIR");
  const std::string path = std::string(tmp ? tmp : "/tmp") + "/fdm_" + name;
  std::ofstream fs(path);
  fs << content;
  return path;
}
}  // namespace
TEST(ConfigLoad, ShippedDefaultYaml) {  // :36-43 (fastdem_amd/config/default.yaml carries the same values)
  const char* dir = std::getenv("FDM_CONFIG_DIR");
  const Config cfg = loadConfig(std::string(dir ? dir : "fastdem_amd/config") + "/default.yaml");
  EXPECT_TRUE(cfg.mapping.estimation_type == EstimationType::Kalman);
  EXPECT_TRUE(cfg.sensor_model.type == SensorType::LiDAR);
  EXPECT_TRUE(cfg.raycasting.enabled);
  EXPECT_FLOAT_EQ(cfg.point_filter.z_min, -1.0f);
  EXPECT_FLOAT_EQ(cfg.point_filter.range_max, 20.0f);
  EXPECT_FLOAT_EQ(cfg.mapping.p2.dn3, 0.84f);
  EXPECT_FLOAT_EQ(cfg.sensor_model.rgbd.normal_c, 0.4f);
  EXPECT_FLOAT_EQ(cfg.raycasting.clear_threshold, -1.0f);
}
TEST(ConfigLoad, MissingFileAndDefaults) {  // :45-58, :109-113, :149-158
  EXPECT_THROW(loadConfig("/nonexistent/path.yaml"), std::runtime_error);
  const Config cfg = loadConfig(writeTempYaml("# empty config\n", "test_empty.yaml"));
  const Config d;
  EXPECT_TRUE(cfg.mapping.mode == d.mapping.mode);
  EXPECT_TRUE(cfg.mapping.estimation_type == d.mapping.estimation_type);
  EXPECT_TRUE(cfg.sensor_model.type == d.sensor_model.type);
  EXPECT_FLOAT_EQ(cfg.point_filter.z_min, d.point_filter.z_min);
  EXPECT_FLOAT_EQ(cfg.point_filter.range_max, d.point_filter.range_max);
  EXPECT_FALSE(cfg.raycasting.enabled);
}
TEST(ConfigLoad, PartialAndEnumValues) {  // :60-107
  const Config p2 = loadConfig(writeTempYaml("mapping:\n  type: p2_quantile\n", "test_partial.yaml"));
  EXPECT_TRUE(p2.mapping.estimation_type == EstimationType::P2Quantile);
  EXPECT_TRUE(p2.mapping.mode == MappingMode::LOCAL);
  EXPECT_FLOAT_EQ(p2.sensor_model.lidar.range_noise, 0.02f);
  EXPECT_TRUE(parseConfigText("mapping:\n  type: kalman_filter\n").mapping.estimation_type == EstimationType::Kalman);
  EXPECT_TRUE(parseConfigText("sensor_model:\n  type: lidar\n").sensor_model.type == SensorType::LiDAR);
  EXPECT_TRUE(parseConfigText("sensor_model:\n  type: rgbd\n").sensor_model.type == SensorType::RGBD);
  EXPECT_TRUE(parseConfigText("sensor_model:\n  type: constant\n").sensor_model.type == SensorType::Constant);
  EXPECT_TRUE(parseConfigText("sensor_model:\n  type: \"laser\"\n").sensor_model.type == SensorType::LiDAR);
  EXPECT_TRUE(parseConfigText("sensor_model:\n  type: bogus\n").sensor_model.type == SensorType::LiDAR);
  EXPECT_TRUE(parseConfigText("mapping:\n  mode: global\n").mapping.mode == MappingMode::GLOBAL);
}
TEST(ConfigLoad, NumericBlocks) {  // :115-147
  const Config k = parseConfigText(
      "mapping:\n  type: kalman_filter\n  kalman:\n    min_variance: 0.001\n    max_variance: 0.05\n"
      "    process_noise: 0.001\n");
  EXPECT_FLOAT_EQ(k.mapping.kalman.min_variance, 0.001f);
  EXPECT_FLOAT_EQ(k.mapping.kalman.max_variance, 0.05f);
  EXPECT_FLOAT_EQ(k.mapping.kalman.process_noise, 0.001f);
  const Config f = parseConfigText(
      "point_filter:\n  z_min: -0.5   # metres\n  z_max: 2.0\n  range_min: 0.5\n  range_max: 20.0\n"
      "raycasting:\n  enabled: true\n  log_odds_ghost: 0.3\n");
  EXPECT_FLOAT_EQ(f.point_filter.z_min, -0.5f);
  EXPECT_FLOAT_EQ(f.point_filter.z_max, 2.0f);
  EXPECT_FLOAT_EQ(f.point_filter.range_min, 0.5f);
  EXPECT_FLOAT_EQ(f.point_filter.range_max, 20.0f);
  EXPECT_TRUE(f.raycasting.enabled);
  EXPECT_FLOAT_EQ(f.raycasting.log_odds_ghost, 0.3f);
}
TEST(ConfigLoad, ValidationRunsAfterParsingAndBadYamlThrows) {  // :160-183; yaml-cpp bad conversion -> runtime_error
  EXPECT_THROW(parseConfigText("mapping:\n  kalman:\n    min_variance: 0.1\n    max_variance: 0.01\n"), std::invalid_argument);
  EXPECT_THROW(parseConfigText("mapping:\n  p2:\n    dn0: 0.9\n    dn1: 0.1\n"), std::invalid_argument);
  EXPECT_FLOAT_EQ(parseConfigText("sensor_model:\n  lidar:\n    range_noise: -1.0\n").sensor_model.lidar.range_noise, 0.02f);
  EXPECT_THROW(loadConfig(writeTempYaml("point_filter:\n  z_min: not_a_number\n", "test_bad.yaml")), std::runtime_error);
  EXPECT_THROW(loadConfig(writeTempYaml("point_filter:\n  - 1\n  - 2\n", "test_seq.yaml")), std::runtime_error);
}

TEST(ConfigLoad, PostProcessYaml) {  // test_config.cpp:227-333
  const config::PostProcess all = config::loadPostProcess(writeTempYaml(
      "inpainting:\n  enabled: true\n  max_iterations: 5\n  min_valid_neighbors: 3\n"
      "uncertainty_fusion:\n  enabled: true\n  search_radius: 0.2\n  spatial_sigma: 0.1\n  quantile_lower: 0.05\n"
      "  quantile_upper: 0.95\n  min_valid_neighbors: 4\n"
      "feature_extraction:\n  enabled: true\n  analysis_radius: 0.5\n  min_valid_neighbors: 6\n",
      "test_postprocess_all.yaml"));
  EXPECT_TRUE(all.inpainting.enabled);
  EXPECT_EQ(all.inpainting.max_iterations, 5);
  EXPECT_EQ(all.inpainting.min_valid_neighbors, 3);
  EXPECT_TRUE(all.uncertainty_fusion.enabled);
  EXPECT_FLOAT_EQ(all.uncertainty_fusion.search_radius, 0.2f);
  EXPECT_FLOAT_EQ(all.uncertainty_fusion.spatial_sigma, 0.1f);
  EXPECT_FLOAT_EQ(all.uncertainty_fusion.quantile_lower, 0.05f);
  EXPECT_FLOAT_EQ(all.uncertainty_fusion.quantile_upper, 0.95f);
  EXPECT_EQ(all.uncertainty_fusion.min_valid_neighbors, 4);
  EXPECT_TRUE(all.feature_extraction.enabled);
  EXPECT_FLOAT_EQ(all.feature_extraction.analysis_radius, 0.5f);
  EXPECT_EQ(all.feature_extraction.min_valid_neighbors, 6);
  const config::PostProcess empty = config::loadPostProcess(writeTempYaml("# empty\n", "test_postprocess_empty.yaml"));
  EXPECT_FALSE(empty.inpainting.enabled);
  EXPECT_FALSE(empty.uncertainty_fusion.enabled);
  EXPECT_FALSE(empty.feature_extraction.enabled);
  const char* dir = std::getenv("FDM_CONFIG_DIR");
  const config::PostProcess shipped =
      config::loadPostProcess(std::string(dir ? dir : "fastdem_amd/config") + "/postprocess.yaml");
  EXPECT_TRUE(shipped.uncertainty_fusion.enabled);
  EXPECT_FALSE(shipped.inpainting.enabled);
  EXPECT_FALSE(shipped.feature_extraction.enabled);
  // clamping (nothing throws)
  auto pp = [](const char* text) { return config::parsePostProcess(yaml::parse(text)); };
  EXPECT_GT(pp("uncertainty_fusion:\n  search_radius: -0.5\n").uncertainty_fusion.search_radius, 0.0f);
  EXPECT_GT(pp("uncertainty_fusion:\n  spatial_sigma: 0.0\n").uncertainty_fusion.spatial_sigma, 0.0f);
  const auto inv = pp("uncertainty_fusion:\n  quantile_lower: 0.95\n  quantile_upper: 0.05\n");
  EXPECT_LT(inv.uncertainty_fusion.quantile_lower, inv.uncertainty_fusion.quantile_upper);
  EXPECT_GT(pp("feature_extraction:\n  analysis_radius: -1.0\n").feature_extraction.analysis_radius, 0.0f);
  const auto neg = pp("inpainting:\n  max_iterations: -2\n  min_valid_neighbors: -1\n");
  EXPECT_GE(neg.inpainting.max_iterations, 1);
  EXPECT_GE(neg.inpainting.min_valid_neighbors, 1);
  EXPECT_THROW(config::loadPostProcess("/nonexistent/pp.yaml"), std::runtime_error);
}

// ---------------------------------------------------------- test_config.cpp (validation) ----
TEST(Config, DefaultsAndValidation) {  // test_config.cpp:36-344, config_fastdem.cpp:128-260
  Config c;
  EXPECT_TRUE(c.mapping.mode == MappingMode::LOCAL);
  EXPECT_TRUE(c.mapping.estimation_type == EstimationType::Kalman);
  EXPECT_TRUE(c.sensor_model.type == SensorType::LiDAR);
  EXPECT_FLOAT_EQ(c.mapping.kalman.min_variance, 0.0001f);
  EXPECT_FLOAT_EQ(c.mapping.kalman.max_variance, 0.01f);
  EXPECT_FLOAT_EQ(c.mapping.p2.dn3, 0.84f);
  EXPECT_EQ(c.mapping.p2.elevation_marker, 3);
  EXPECT_FALSE(c.raycasting.enabled);
  EXPECT_NO_THROW(validated(c));
  Config bad = c;
  bad.mapping.kalman.min_variance = 0.5f;
  bad.mapping.kalman.max_variance = 0.1f;
  EXPECT_THROW(validated(bad), std::invalid_argument);
  Config unsorted = c;
  unsorted.mapping.p2.dn1 = 0.9f;
  EXPECT_THROW(validated(unsorted), std::invalid_argument);
  Config clamp = c;
  clamp.mapping.p2.elevation_marker = 9;
  clamp.sensor_model.lidar.range_noise = -1.0f;
  clamp.mapping.kalman.process_noise = -2.0f;
  const Config v = validated(clamp);
  EXPECT_EQ(v.mapping.p2.elevation_marker, 4);
  EXPECT_FLOAT_EQ(v.sensor_model.lidar.range_noise, 0.02f);
  EXPECT_FLOAT_EQ(v.mapping.kalman.process_noise, 0.0f);
}

// ------------------------------------------------------------- test_sensor_models.cpp ----
TEST(SensorModels, HostClassesAndFactory) {  // test_sensor_models.cpp:17-262
  config::SensorModel cfg;
  SensorType t;
  cfg.type = SensorType::Constant;
  EXPECT_TRUE(createSensorModel(cfg)->builtin(t) && t == SensorType::Constant);
  cfg.type = SensorType::RGBD;
  EXPECT_TRUE(createSensorModel(cfg)->builtin(t) && t == SensorType::RGBD);
  LiDARSensorModel lidar(0.02f, 0.001f);
  const auto cov = lidar.computeCovariance(Eigen::Vector3f(10.0f, 0.0f, 0.0f));
  EXPECT_NEAR(cov(0, 0), 0.02f * 0.02f, 1e-6f);
  EXPECT_NEAR(cov(1, 1), 0.01f * 0.01f, 1e-6f);
  EXPECT_NEAR(lidar.computeCovariance(Eigen::Vector3f(0, 0, 0))(2, 2), 0.01f, 1e-6f);
  RGBDSensorModel rgbd;
  EXPECT_NEAR(rgbd.computeCovariance(Eigen::Vector3f(0, 0, 0.4f))(2, 2), 0.001f * 0.001f, 1e-10f);
  EXPECT_NEAR(rgbd.computeCovariance(Eigen::Vector3f(1, 2, -0.5f))(0, 0), 0.01f, 1e-6f);
}


// ---------------------------------------------------------------- queued mode / integrateBatch (not in the reference) ----
namespace {
// a scan stream with every case the reference distinguishes: plain scans, an empty cloud, a scan whose points are all
// filtered (false, and the LOCAL map must not move: fastdem.cpp:138), a pose sequence that shifts the window
struct Stream {
  std::vector<PointCloud> clouds;
  std::vector<Eigen::Isometry3d> poses;
  Stream() {
    for (int k = 0; k < 23; ++k) {
      PointCloud c;
      if (k != 7) {  // (scan 7: an empty cloud)
        const float h = (k == 11 || k == 22) ? 50.0f : 0.2f + 0.05f * float(k % 5);  // (11, 22: above the height filter)
        for (int i = -12; i <= 12; ++i)
          for (int j = -12; j <= 12; ++j) c.add(0.21f * float(i) + 0.01f * float(k), 0.19f * float(j), h + 0.01f * float((i * 7 + j * 3) % 5));
      }
      clouds.push_back(std::move(c));
      Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
      T.translation() = Eigen::Vector3d(0.3 * k, -0.2 * k, 0.0);
      poses.push_back(T);
    }
  }
};
bool sameMaps(ElevationMap& a, ElevationMap& b) {
  if (a.getLayers() != b.getLayers()) return false;
  if (a.getStartIndex()(0) != b.getStartIndex()(0) || a.getStartIndex()(1) != b.getStartIndex()(1)) return false;
  if (a.getPosition()(0) != b.getPosition()(0) || a.getPosition()(1) != b.getPosition()(1)) return false;
  for (const auto& name : a.getLayers()) {
    const auto& A = a.get(name);
    const auto& B = b.get(name);
    for (int i = 0; i < A.rows(); ++i)
      for (int j = 0; j < A.cols(); ++j) {
        const float va = A(i, j), vb = B(i, j);
        uint32_t x, y;
        std::memcpy(&x, &va, 4);
        std::memcpy(&y, &vb, 4);
        if (x != y && !(std::isnan(va) && std::isnan(vb))) return false;
      }
  }
  return true;
}
}  // namespace
TEST(QueuedMode, SameMapAsTheSynchronousCallsAndTheLateStatus) {
  Stream st;
  ElevationMap m_sync(12.0f, 9.0f, 0.1f, "map"), m_q(12.0f, 9.0f, 0.1f, "map");
  FastDEM a(m_sync), b(m_q);
  a.setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::Constant);
  b.setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::Constant).setQueued(true);
  EXPECT_TRUE(b.queued());
  Eigen::Isometry3d Tbs = Eigen::Isometry3d::Identity();
  std::vector<bool> ra;
  for (size_t k = 0; k < st.clouds.size(); ++k) ra.push_back(a.integrate(st.clouds[k], Tbs, st.poses[k]));
  EXPECT_FALSE(ra[7]);
  EXPECT_FALSE(ra[11]);
  EXPECT_TRUE(ra[12]);
  for (size_t k = 0; k < st.clouds.size(); ++k) {
    const bool accepted = b.integrate(st.clouds[k], Tbs, st.poses[k]);
    EXPECT_EQ(accepted, !st.clouds[k].empty());  // queued: `true` = accepted; an empty cloud is decided on the host
    if (k == 11) {                               // the data-dependent `false` arrives with drain()
      EXPECT_FALSE(b.drain());
      EXPECT_EQ(b.lastStats().n_after_filter, 0u);
    }
    if (k == 12) EXPECT_TRUE(b.drain());
  }
  EXPECT_FALSE(b.drain());  // scan 22: every point filtered
  EXPECT_TRUE(b.drain());   // (nothing queued any more)
  EXPECT_TRUE(sameMaps(m_sync, m_q));
  // a configuration change between two queued scans applies to the scans queued after it
  a.setHeightFilter(-1.0f, 100.0f).setRangeFilter(0.0f, 200.0f);
  b.setHeightFilter(-1.0f, 100.0f).setRangeFilter(0.0f, 200.0f);
  EXPECT_TRUE(a.integrate(st.clouds[22], Tbs, st.poses[22]));
  EXPECT_TRUE(b.integrate(st.clouds[22], Tbs, st.poses[22]));
  EXPECT_TRUE(b.drain());
  EXPECT_TRUE(sameMaps(m_sync, m_q));
}
TEST(IntegrateBatch, SameMapAsScanByScanAndTheLastScansResult) {
  Stream st;
  ElevationMap m_one(12.0f, 9.0f, 0.1f, "map"), m_b(12.0f, 9.0f, 0.1f, "map");
  FastDEM a(m_one), b(m_b);
  a.setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::Constant);
  b.setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::Constant);
  Eigen::Isometry3d Tbs = Eigen::Isometry3d::Identity();
  bool last = false;
  for (size_t k = 0; k < st.clouds.size(); ++k) last = a.integrate(st.clouds[k], Tbs, st.poses[k]);
  std::vector<FastDEM::Scan> scans;
  for (size_t k = 0; k < st.clouds.size(); ++k) scans.push_back(FastDEM::Scan{&st.clouds[k], Tbs, st.poses[k]});
  EXPECT_EQ(b.integrateBatch(scans), last);  // scan 22 is filtered: false
  EXPECT_FALSE(last);
  EXPECT_EQ(b.lastStats().n_after_filter, a.lastStats().n_after_filter);
  EXPECT_TRUE(sameMaps(m_one, m_b));
  // a shorter batch that ends on a good scan, on the same maps
  scans.resize(9);
  for (size_t k = 0; k < 9; ++k) last = a.integrate(st.clouds[k], Tbs, st.poses[k]);
  EXPECT_EQ(b.integrateBatch(scans), last);
  EXPECT_TRUE(last);
  EXPECT_EQ(b.lastStats().n_cells_touched, a.lastStats().n_cells_touched);
  EXPECT_TRUE(sameMaps(m_one, m_b));
  EXPECT_FALSE(b.integrateBatch({}));
}

TEST(IntegrateBatch, WithRaycastingTheStageOfEveryScanRidesInTheBatch) {
  // the shipped YAML's switch (config/default.yaml:40-41): step 3 of integrateImpl behind every scan's map update —
  // scan by scan on one mapper, as one batch on the other (voxel filter, ray walks and ghost resolution of the
  // scans inside the batch launches), and through the queued mode on a third
  Stream st;
  ElevationMap m_one(12.0f, 9.0f, 0.1f, "map"), m_b(12.0f, 9.0f, 0.1f, "map"), m_q(12.0f, 9.0f, 0.1f, "map");
  FastDEM a(m_one), b(m_b), q(m_q);
  for (FastDEM* f : {&a, &b, &q})
    f->setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::Constant).enableRaycasting();
  q.setQueued(true);
  Eigen::Isometry3d Tbs = Eigen::Isometry3d::Identity();
  Tbs.translation() = Eigen::Vector3d(0.0, 0.0, 1.5);  // the sensor above the points: downward rays
  bool last = false;
  for (size_t k = 0; k < st.clouds.size(); ++k) last = a.integrate(st.clouds[k], Tbs, st.poses[k]);
  std::vector<FastDEM::Scan> scans;
  for (size_t k = 0; k < st.clouds.size(); ++k) scans.push_back(FastDEM::Scan{&st.clouds[k], Tbs, st.poses[k]});
  EXPECT_EQ(b.integrateBatch(scans), last);
  for (size_t k = 0; k < st.clouds.size(); ++k) q.integrate(st.clouds[k], Tbs, st.poses[k]);
  q.drain();
  EXPECT_TRUE(m_one.exists("raycasting"));
  EXPECT_TRUE(sameMaps(m_one, m_b));
  EXPECT_TRUE(sameMaps(m_one, m_q));
}

// ---- nanopcl::PointCloud4: the reference's own cloud layout (nanoPCL tests/test_pointcloud.cpp, test_channels.cpp,
// test_index_range.cpp re-expressed) and FastDEM::integrate on it ----
TEST(PointCloud4, ConstructorsAddAndTheTwoAccessors) {  // test_pointcloud.cpp:21-85
  nanopcl::PointCloud4 empty;
  EXPECT_EQ(empty.size(), 0u);
  EXPECT_TRUE(empty.empty());
  nanopcl::PointCloud4 sized(100);
  EXPECT_EQ(sized.size(), 100u);
  EXPECT_EQ(sized[99].w(), 1.0f);  // resize() fills with (0, 0, 0, 1)
  nanopcl::PointCloud4 c;
  c.add(1.0f, 2.0f, 3.0f);
  EXPECT_EQ(c.size(), 1u);
  EXPECT_EQ(c.point(0).x(), 1.0f);
  EXPECT_EQ(c.point(0).y(), 2.0f);
  EXPECT_EQ(c.point(0).z(), 3.0f);
  EXPECT_EQ(c[0].w(), 1.0f);
  const Eigen::Vector3f p = c.point(0);
  EXPECT_EQ(p.x(), 1.0f);
  c.point(0).x() = 10.0f;  // the 3-D view writes into the record
  EXPECT_EQ(c[0].x(), 10.0f);
  c[0] = nanopcl::Point4(c[0].x() + 100.0f, c[0].y(), c[0].z(), c[0].w());  // (the "expert" 4-D access)
  EXPECT_EQ(c.point(0).x(), 110.0f);
  for (int i = 0; i < 100; ++i) c.add(float(i), float(i), float(i));
  EXPECT_EQ(c.size(), 101u);
  EXPECT_EQ(c.point(51).x(), 50.0f);
}
TEST(PointCloud4, ResizeReserveClearReset) {  // test_pointcloud.cpp:103-142
  nanopcl::PointCloud4 c;
  c.add(1, 2, 3);
  c.add(4, 5, 6);
  c.resize(10);
  EXPECT_EQ(c.size(), 10u);
  EXPECT_EQ(c.point(0).x(), 1.0f);
  EXPECT_EQ(c.point(1).x(), 4.0f);
  nanopcl::PointCloud4 r;
  r.reserve(1000);
  EXPECT_TRUE(r.capacity() >= 1000u);
  EXPECT_EQ(r.size(), 0u);
  nanopcl::PointCloud4 k;
  k.add(1, 2, 3, nanopcl::Intensity(0.5f));
  k.clear();
  EXPECT_EQ(k.size(), 0u);
  EXPECT_TRUE(k.hasIntensity());  // clear() keeps the channel structure ...
  k.add(1, 2, 3, nanopcl::Intensity(0.5f));
  k.setFrameId("lidar");
  k.reset();
  EXPECT_FALSE(k.hasIntensity());  // ... reset() drops it, and the metadata
  EXPECT_TRUE(k.frameId().empty());
}
TEST(PointCloud4, PointsIsOneContiguousArrayOfAligned16ByteRecords) {  // test_pointcloud.cpp:144-160, point_cloud.hpp:126
  nanopcl::PointCloud4 c;
  for (int i = 0; i < 33; ++i) c.add(float(i), float(2 * i), float(3 * i));
  EXPECT_EQ(c.points().size(), 33u);
  EXPECT_EQ(reinterpret_cast<uintptr_t>(c.points().data()) % 16u, 0u);
  const float* raw = c.xyz1Data();
  for (int i = 0; i < 33; ++i) {
    EXPECT_EQ(raw[4 * i + 0], float(i));
    EXPECT_EQ(raw[4 * i + 1], float(2 * i));
    EXPECT_EQ(raw[4 * i + 2], float(3 * i));
    EXPECT_EQ(raw[4 * i + 3], 1.0f);
  }
}
TEST(PointCloud4, MetadataAndTimestampHelpers) {  // test_pointcloud.cpp:162-189
  nanopcl::PointCloud4 c;
  c.setFrameId("os_sensor");
  EXPECT_EQ(c.frameId(), std::string("os_sensor"));
  c.setTimestamp(1500000000ull);
  EXPECT_EQ(c.timestamp(), 1500000000ull);
  EXPECT_NEAR(nanopcl::toSec(1500000000ull), 1.5, 1e-9);
  EXPECT_EQ(nanopcl::fromSec(2.5), 2500000000ull);
}
TEST(PointCloud4, ChannelsFollowThePoints) {  // test_channels.cpp:21-140
  nanopcl::PointCloud4 c;
  EXPECT_FALSE(c.hasIntensity() || c.hasTime() || c.hasRing() || c.hasColor() || c.hasLabel() || c.hasNormal());
  c.add(1, 2, 3, nanopcl::Intensity(0.5f), nanopcl::Ring(7), nanopcl::Time(0.25f));
  EXPECT_TRUE(c.hasIntensity() && c.hasRing() && c.hasTime());
  EXPECT_EQ(c.intensity(0), 0.5f);
  EXPECT_EQ(c.ring(0), 7);
  EXPECT_EQ(c.time(0), 0.25f);
  c.add(4, 5, 6);  // a plain add() keeps the existing channels in step (defaults)
  EXPECT_EQ(c.intensities().size(), 2u);
  EXPECT_EQ(c.intensity(1), 0.0f);
  EXPECT_EQ(c.ring(1), 0);
  c.resize(5);
  EXPECT_EQ(c.intensities().size(), 5u);
  EXPECT_EQ(c.rings().size(), 5u);
  c.useNormal();  // a channel switched on late is sized to the cloud
  EXPECT_EQ(c.normals().size(), 5u);
  c.normal(2) = Eigen::Vector3f(0.0f, 0.0f, 1.0f);
  EXPECT_EQ(Eigen::Vector3f(c.normal(2)).z(), 1.0f);
  c.add(7, 8, 9, nanopcl::Color(10, 20, 30), nanopcl::Label(42));
  EXPECT_EQ(c.color(5).g, 20);
  EXPECT_EQ(uint32_t(c.label(5)), 42u);
  EXPECT_EQ(c.color(0).r, 0);  // the earlier points got the default colour
}
TEST(PointCloud4, ExtractEraseAndMerge) {  // test_pointcloud.cpp:191-240, test_channels.cpp:157-216
  nanopcl::PointCloud4 c;
  c.setFrameId("f");
  c.setTimestamp(9);
  for (int i = 0; i < 10; ++i) c.add(float(i), 0, 0, nanopcl::Intensity(float(i) * 0.1f));
  const nanopcl::PointCloud4 sub = c.extract({1, 3, 5});
  EXPECT_EQ(sub.size(), 3u);
  EXPECT_EQ(sub.point(1).x(), 3.0f);
  EXPECT_TRUE(sub.hasIntensity());
  EXPECT_FLOAT_EQ(sub.intensity(2), 0.5f);
  EXPECT_EQ(sub.frameId(), std::string("f"));
  EXPECT_EQ(sub.timestamp(), 9u);
  const nanopcl::PointCloud4 run = c.extract(size_t(4), size_t(3));
  EXPECT_EQ(run.size(), 3u);
  EXPECT_EQ(run.point(0).x(), 4.0f);
  c.erase({8, 2, 2, 0});  // order and duplicates do not matter
  EXPECT_EQ(c.size(), 7u);
  EXPECT_EQ(c.point(0).x(), 1.0f);
  EXPECT_EQ(c.point(1).x(), 3.0f);
  EXPECT_EQ(c.point(6).x(), 9.0f);
  EXPECT_FLOAT_EQ(c.intensity(6), 0.9f);
  nanopcl::PointCloud4 a, b;
  b.add(1, 1, 1, nanopcl::Intensity(0.7f));
  a += b;  // an empty cloud adopts the other's channels
  EXPECT_TRUE(a.hasIntensity());
  EXPECT_EQ(a.intensity(0), 0.7f);
  nanopcl::PointCloud4 plain;
  plain.add(2, 2, 2);
  a += plain;  // a cloud without the channel contributes defaults
  EXPECT_EQ(a.size(), 2u);
  EXPECT_EQ(a.intensities().size(), 2u);
  EXPECT_EQ(a.intensity(1), 0.0f);
}
TEST(PointCloud4, IndexRange) {  // test_index_range.cpp:23-175
  nanopcl::PointCloud4 c(5);
  size_t sum = 0, count = 0;
  for (size_t i : c.indices()) { sum += i; ++count; }
  EXPECT_EQ(count, 5u);
  EXPECT_EQ(sum, 10u);
  EXPECT_EQ(c.indices().size(), 5u);
  constexpr nanopcl::IndexRange r(3);
  static_assert(r.size() == 3, "constexpr");
  size_t n_empty = 0;
  for (size_t i : nanopcl::IndexRange(0)) { (void)i; ++n_empty; }
  EXPECT_EQ(n_empty, 0u);
}
TEST(PointCloud4, IntegrateGivesTheMapOfTheSoACloudBitForBit) {  // fastdem.cpp:122-190 on cloud.points().data()
  Stream st;
  ElevationMap m_soa(12.0f, 9.0f, 0.1f, "map"), m_aos(12.0f, 9.0f, 0.1f, "map");
  FastDEM a(m_soa), b(m_aos);
  a.setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::LiDAR);
  b.setHeightFilter(-1.0f, 2.0f).setRangeFilter(0.0f, 30.0f).setSensorModel(SensorType::LiDAR);
  Eigen::Isometry3d Tbs = Eigen::Isometry3d::Identity();
  Tbs.translation() = Eigen::Vector3d(0.1, 0.0, 0.4);
  for (size_t k = 0; k < st.clouds.size(); ++k) {
    PointCloud& s = st.clouds[k];
    nanopcl::PointCloud4 q;
    for (size_t i = 0; i < s.size(); ++i) {
      const Eigen::Vector3f p = static_cast<const PointCloud&>(s).point(i);
      if (k % 2) q.add(p.x(), p.y(), p.z(), nanopcl::Intensity(float(i % 17)), nanopcl::Color(uint8_t(i), uint8_t(k), 3));
      else q.add(p.x(), p.y(), p.z());
    }
    if (k % 2) {  // the same channels on the SoA side
      s.useIntensity();
      s.useColor();
      for (size_t i = 0; i < s.size(); ++i) { s.intensity(i) = float(i % 17); s.setColor(i, nanopcl::Color(uint8_t(i), uint8_t(k), 3)); }
    }
    const bool ra = a.integrate(s, Tbs, st.poses[k]);
    const bool rb = b.integrate(q, Tbs, st.poses[k]);
    EXPECT_EQ(ra, rb);
    EXPECT_EQ(a.lastStats().n_after_filter, b.lastStats().n_after_filter);
    EXPECT_EQ(a.lastStats().n_cells_touched, b.lastStats().n_cells_touched);
  }
  EXPECT_TRUE(m_soa.exists("intensity") && m_soa.exists("color"));
  EXPECT_TRUE(sameMaps(m_soa, m_aos));
  // the online overload: providers + frame id + timestamp, as FastDEM::integrate(shared_ptr) of the reference
  auto cloud = std::make_shared<nanopcl::PointCloud4>();
  cloud->add(0.5f, 0.5f, 0.3f);
  EXPECT_FALSE(b.integrate(cloud));  // no providers
}

int main(int argc, char** argv) { return mini::run(argc > 1 ? argv[1] : nullptr); }

// ---------------------------------------------------------------- the mirror's reference and copy semantics ----
// The reference hands out Matrix& that stay live across integrate() (estimators bind raw pointers,
// kalman_estimation.hpp:85-95) and its map is copyable (snapshot() by value, elevation_map.hpp:95-99; the ROS node copies
// it under a shared lock, ros1/src/fastdem_ros_node.cpp:192-199).
TEST(MirrorSemantics, AHeldReferenceKeepsItsAddressAndIsRefreshedInPlace) {
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-2.0f, 5.0f).setRangeFilter(0.0f, 20.0f).setSensorModel(SensorType::Constant);
  auto& elev = f.map.get(layer::elevation);           // obtained BEFORE the scan
  const float* const address = elev.data();
  EXPECT_EQ(elev.countFinite(), size_t(0));
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  // (debug builds: the reference is stale until the next host access of the map ...)
  EXPECT_THROW((void)elev.countFinite(), std::logic_error);
  // ... of ANY layer, by anyone: every host copy is refreshed in place
  EXPECT_TRUE(f.map.exists(layer::variance));
  (void)f.map.get(layer::variance);
  EXPECT_EQ(elev.data(), address);
  EXPECT_GT(elev.countFinite(), size_t(0));
  nanogrid::Index c;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0.0, 0.0), c));
  EXPECT_NEAR(elev(c), 1.0f, 0.1f);
  // a second scan, a write through a FRESH reference, a third scan that sees it
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(1.2f), f.T_base_sensor, f.T_world_base));
  auto& again = f.map.get(layer::elevation);
  EXPECT_EQ(&again, &elev);
  again(c) = 7.0f;
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(1.2f), f.T_base_sensor, f.T_world_base));
  EXPECT_GT(f.map.at(layer::elevation, c), 1.25f);     // (the estimator started from 7, not from ~1.1)
  // a copy of a layer's matrix is a value of its own
  nanogrid::Matrix value = f.map.get(layer::elevation);
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(1.2f), f.T_base_sensor, f.T_world_base));
  EXPECT_NO_THROW((void)value.countFinite());
}

TEST(MirrorSemantics, TheMapIsCopyableADeepCopy) {
  Fixture f;
  FastDEM mapper(f.map);
  mapper.setHeightFilter(-2.0f, 5.0f).setRangeFilter(0.0f, 20.0f).setSensorModel(SensorType::Constant);
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(1.0f), f.T_base_sensor, f.T_world_base));
  f.map.setFrameId("map");
  f.map.setTimestamp(42);
  nanogrid::Index c;
  ASSERT_TRUE(f.map.getIndex(nanogrid::Position(0.3, -0.3), c));
  f.map.at(layer::elevation, c) = 3.5f;                // (a host write that has not reached the device yet)
  ElevationMap copy = f.map;                           // copy constructor
  EXPECT_TRUE(copy.isInitialized());
  EXPECT_EQ(copy.getFrameId(), std::string("map"));
  EXPECT_EQ(copy.getTimestamp(), uint64_t(42));
  EXPECT_EQ(copy.getLayers().size(), f.map.getLayers().size());
  EXPECT_EQ(copy.getStartIndex()(0), f.map.getStartIndex()(0));
  EXPECT_FLOAT_EQ(copy.at(layer::elevation, c), 3.5f);
  for (const auto& name : f.map.getLayers()) {
    const auto& a = f.map.get(name);
    const auto& b = copy.get(name);
    bool same = true;
    for (size_t k = 0; k < a.size(); ++k) {
      const float x = a.data()[k], y = b.data()[k];
      same = same && ((std::isnan(x) && std::isnan(y)) || x == y);
    }
    EXPECT_TRUE(same);
  }
  // the two maps live their own lives
  ASSERT_TRUE(mapper.integrate(makeGroundCloud(2.0f), f.T_base_sensor, f.T_world_base));
  EXPECT_FLOAT_EQ(copy.at(layer::elevation, c), 3.5f);
  copy.clearAll();
  EXPECT_TRUE(copy.isEmpty());
  EXPECT_FALSE(f.map.isEmpty());
  ElevationMap assigned;
  assigned = f.map;                                    // copy assignment
  EXPECT_FALSE(assigned.isEmpty());
  // the ROS node's pattern (fastdem_ros_node.cpp:192-199): a snapshot by value of a few layers
  const ElevationMap snap = f.map.snapshot({layer::elevation, layer::variance});
  EXPECT_EQ(snap.getLayers().size(), size_t(4));  // (elevation_min / _max come with the constructor)
}

TEST(MirrorSemantics, MoveClearScopeSwitch) {   // which layers move()'s strips clear
  for (int basic = 0; basic < 2; ++basic) {
    ElevationMap m;
    m.setGeometry(2.0f, 2.0f, 0.1f);
    m.setMoveClearBasic(basic != 0);
    m.add("user", 1.0f);
    m.get(layer::elevation).setConstant(1.0f);
    m.move(nanogrid::Position(0.3, 0.0));              // three rows vacated
    EXPECT_EQ(m.get(layer::elevation).countFinite(), size_t(20 * 20 - 3 * 20));
    EXPECT_EQ(m.get("user").countFinite(), size_t(basic ? 20 * 20 : 20 * 20 - 3 * 20));
    m.move(nanogrid::Position(9.0, 0.0));              // beyond the map: clearAll() in both readings
    EXPECT_EQ(m.get("user").countFinite(), size_t(0));
  }
}
