// probe.cpp — conformance probe: run THIS against the real libraries of the reference and diff its output against what
// the MI355X engine's oracle assumes (scripts/conformance/expected.json, check.py).
//
// It includes the reference's own headers and the libraries they pull in — <Eigen/Geometry>, nanoGrid (fetched by the
// reference's CMake, fastdem/CMakeLists.txt:24-28), nanoPCL, fastdem — never a copy, and prints one line per probed
// fact, floats as their IEEE bit patterns.  What it probes is exactly what SURVEY.md §8(c) could not pin from inside
// the reference tree:
//   index / position   GridMap::getIndex / getPosition at cell edges, map borders, wrapped start indices
//   move               GridMap::move: start index, position, WHICH CELLS OF WHICH LAYERS turn NaN (all layers or the
//                      basic layers only? — the engine has a switch for either answer: option "move_clear_basic")
//   region / cells     the visit order and dist_sq of region(Size) / region(radius) / neighbors(), the order of cells()
//   color              nanogrid::colorVectorToValue
//   pre / map          FastDEM::integrate: the preprocessed cloud (two Eigen transforms, crops, R Sigma R^T — through
//                      onScanPreprocessed) and a hash of every layer of the map after two scans, per sensor model and
//                      estimator
//
// Build, in the reference's build tree (one command):
//   c++ -std=c++17 -O2 -I<fastdem>/include -I<fastdem>/lib/nanoPCL/include -I<nanoGrid>/include -I/usr/include/eigen3 <next line>
//       scripts/conformance/probe.cpp <fastdem build>/libfastdem.a -lyaml-cpp -lspdlog -lfmt -o probe
//   ./probe > probe.txt && python3 scripts/conformance/check.py probe.txt
// (-O2 without -march=native and without -ffast-math: the reference's Release flags, fastdem/CMakeLists.txt:4-11.)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#ifndef FDM_PROBE_MIRROR   // (FDM_PROBE_MIRROR: this repo's own check of the probe against its C++ mirror, which brings
#include <Eigen/Geometry>  //  its own small Eigen subset and has no region() / neighbors() — tests/test_cpp_host_api.py)
#endif

#include "fastdem/fastdem.hpp"
#include "fastdem/elevation_map.hpp"

#include "vectors.inc"

using fastdem::ElevationMap;

static uint32_t f32bits(float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }
static unsigned long long f64bits(double v) { unsigned long long u; std::memcpy(&u, &v, 8); return u; }
struct Fnv {
  unsigned long long h = 0xcbf29ce484222325ull;
  void word(uint32_t w) { for (int k = 0; k < 4; ++k) { h ^= (w >> (8 * k)) & 0xFFu; h *= 0x100000001b3ull; } }
};

static void probeIndex() {
  for (int gi = 0; gi < kGeoms_n; ++gi) {
    const double* g = kGeoms[gi];
    ElevationMap map;
    map.setGeometry(static_cast<float>(g[0]), static_cast<float>(g[1]), static_cast<float>(g[2]));
    map.setPosition(nanogrid::Position(g[3], g[4]));
    map.setStartIndex(nanogrid::Index(int(g[5]), int(g[6])));
    for (int qi = 0; qi < kIndexQn[gi]; ++qi) {
      nanogrid::Index idx(0, 0);
      const bool ok = map.getIndex(nanogrid::Position(kIndexQ[gi][qi][0], kIndexQ[gi][qi][1]), idx);
      std::printf("index %d.%d %d %d %d\n", gi, qi, ok ? 1 : 0, ok ? idx(0) : -1, ok ? idx(1) : -1);
    }
    const nanogrid::Size size = map.getSize();
    const int cells[4][2] = {{0, 0}, {size(0) - 1, size(1) - 1}, {int(g[5]), int(g[6])}, {size(0) / 2, 1}};
    for (auto& c : cells) {
      nanogrid::Position p(0.0, 0.0);
      map.getPosition(nanogrid::Index(c[0], c[1]), p);
      std::printf("position %d.%d.%d %016llx %016llx\n", gi, c[0], c[1], f64bits(p.x()), f64bits(p.y()));
    }
  }
}

static bool g_move_clear_basic = false;  // (mirror build only: --move-clear-basic)
static void probeMove() {
  ElevationMap map;
  map.setGeometry(2.0f, 2.0f, 0.1f);
#ifdef FDM_PROBE_MIRROR
  map.setMoveClearBasic(g_move_clear_basic);
#endif
  const int n_layers = int(sizeof(kMoveLayers) / sizeof(kMoveLayers[0]));
  for (int l = 0; l < n_layers; ++l) {
    if (!map.exists(kMoveLayers[l])) map.add(kMoveLayers[l], 1.0f);
    map.get(kMoveLayers[l]).setConstant(1.0f);
  }
  for (int mi = 0; mi < kMoves_n; ++mi) {
    map.move(nanogrid::Position(kMoves[mi][0], kMoves[mi][1]));
    const nanogrid::Index s = map.getStartIndex();
    const nanogrid::Position p = map.getPosition();
    std::printf("move %d start %d %d pos %016llx %016llx\n", mi, s(0), s(1), f64bits(p.x()), f64bits(p.y()));
    for (int l = 0; l < n_layers; ++l) {
      const auto& m = map.get(kMoveLayers[l]);
      Fnv h;
      long long count = 0;
      for (long long k = 0; k < (long long)m.size(); ++k)   // storage order: data()[col * rows + row]
        if (std::isnan(m.data()[k])) { h.word(uint32_t(k)); ++count; }
      std::printf("move %d layer %s nan %lld %016llx\n", mi, kMoveLayers[l], count, h.h);
    }
    for (int l = 0; l < n_layers; ++l) map.get(kMoveLayers[l]).setConstant(1.0f);
  }
}

#ifndef FDM_PROBE_MIRROR
static void probeRegions() {
  for (int ri = 0; ri < kRegions_n; ++ri) {
    for (int ci = 0; ci < kRegionCells_n; ++ci) {
      ElevationMap map;
      map.setGeometry(1.0f, 1.0f, 0.05f);
      map.setStartIndex(nanogrid::Index(kRegionCells[ci][2], kRegionCells[ci][3]));
      const auto reg = kRegions[ri][0] == 0.0 ? map.region(nanogrid::Size(int(kRegions[ri][1]), int(kRegions[ri][1])))
                                              : map.region(static_cast<float>(kRegions[ri][1]));
      // the cell with these LOGICAL (unwrapped) coordinates: cells() walks the map, row / col of a cell are compared
      // the way the reference's own loops do (n.row - cell.row is a world offset, feature_extraction.cpp:73-76)
      bool found = false;
      for (auto cell : map.cells()) {
        if (found) break;
        // logical coordinates of the cell = buffer index unwrapped by the start index
        const nanogrid::Size size = map.getSize();
        const nanogrid::Index s = map.getStartIndex();
        const int br = int(cell.index % size(0)), bc = int(cell.index / size(0));
        const int lr = (br - s(0) + size(0)) % size(0), lc = (bc - s(1) + size(1)) % size(1);
        if (lr != kRegionCells[ci][0] || lc != kRegionCells[ci][1]) continue;
        found = true;
        std::string line;
        int n_seen = 0;
        for (auto n : map.neighbors(cell, reg)) {
          char buf[64];
          std::snprintf(buf, sizeof(buf), " %d,%d,%08x", n.row - cell.row, n.col - cell.col, f32bits(static_cast<float>(n.dist_sq)));
          line += buf;
          ++n_seen;
        }
        std::printf("region %d.%d n %d%s\n", ri, ci, n_seen, line.c_str());
      }
    }
  }
  const int which[2] = {0, 3};
  for (int k = 0; k < 2; ++k) {
    ElevationMap map;
    map.setGeometry(1.0f, 1.0f, 0.05f);
    map.setStartIndex(nanogrid::Index(kRegionCells[which[k]][2], kRegionCells[which[k]][3]));
    std::string line;
    int seen = 0;
    for (auto cell : map.cells()) {
      if (seen++ == 5) break;
      const nanogrid::Size size = map.getSize();
      const nanogrid::Index s = map.getStartIndex();
      const int br = int(cell.index % size(0)), bc = int(cell.index / size(0));
      char buf[64];
      std::snprintf(buf, sizeof(buf), " %d,%d,%lld", (br - s(0) + size(0)) % size(0), (bc - s(1) + size(1)) % size(1),
                    static_cast<long long>(cell.index));
      line += buf;
    }
    std::printf("cells %d%s\n", k, line.c_str());
  }
}

#endif

static void probeColors() {
  for (int i = 0; i < kColors_n; ++i) {
    float packed = 0.0f;
    nanogrid::colorVectorToValue(Eigen::Vector3i(kColors[i][0], kColors[i][1], kColors[i][2]), packed);
    std::printf("color %d %08x\n", i, f32bits(packed));
  }
}

static Eigen::Isometry3d iso(const double (*m)[4]) {
  Eigen::Matrix3d Rm;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rm(r, c) = m[r][c];
  Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
#ifndef FDM_PROBE_MIRROR
  T.linear() = Rm;
#else
  T.rotate(Rm);  // (identity * R: exact)
#endif
  T.translation() = Eigen::Vector3d(m[0][3], m[1][3], m[2][3]);
  return T;
}

static void probePre() {
  const fastdem::SensorType sensors[3] = {fastdem::SensorType::Constant, fastdem::SensorType::LiDAR, fastdem::SensorType::RGBD};
  const char* names[3] = {"Constant", "LiDAR", "RGBD"};
  for (int si = 0; si < 3; ++si) {
    for (int est = 0; est < 2; ++est) {
      ElevationMap map;
      map.setGeometry(15.0f, 15.0f, 0.1f);
      fastdem::FastDEM mapper(map);
      mapper.setHeightFilter(-1.0f, 2.0f)
          .setRangeFilter(0.5f, 20.0f)
          .setSensorModel(sensors[si])
          .setMappingMode(fastdem::MappingMode::LOCAL)
          .setEstimatorType(est == 0 ? fastdem::EstimationType::Kalman : fastdem::EstimationType::P2Quantile);
      int scan = 0;
      if (est == 0)
        mapper.onScanPreprocessed([&](const fastdem::PointCloud& c) {
          std::printf("pre %s.%d n %zu\n", names[si], scan, c.size());
          for (size_t k = 0; k < c.size(); ++k) {
#ifndef FDM_PROBE_MIRROR
            const auto p = c[k];
#else
            const auto p = c.point(k);
#endif
            std::printf("pre %s.%d.%zu %08x %08x %08x cov", names[si], scan, k, f32bits(p.x()), f32bits(p.y()), f32bits(p.z()));
            const auto& C = c.covariance(k);
            for (int i = 0; i < 3; ++i)
              for (int j = 0; j < 3; ++j) std::printf(" %08x", f32bits(C(i, j)));
            std::printf("\n");
          }
        });
      for (scan = 0; scan < 2; ++scan) {
        fastdem::PointCloud cloud;
        for (int k = 0; k < kPrePoints_n; ++k)
          cloud.add(static_cast<float>(kPrePoints[k][0]), static_cast<float>(kPrePoints[k][1]), static_cast<float>(kPrePoints[k][2]));
        Eigen::Isometry3d Twb = iso(kTwb);
        Twb.translation().x() += 0.25 * scan;
        mapper.integrate(cloud, iso(kTbs), Twb);
      }
      const nanogrid::Index s = map.getStartIndex();
      const nanogrid::Position p = map.getPosition();
      std::printf("map %s.%d start %d %d pos %016llx %016llx\n", names[si], est, s(0), s(1), f64bits(p.x()), f64bits(p.y()));
      std::vector<std::string> layers = map.getLayers();
      std::sort(layers.begin(), layers.end());
      for (const auto& name : layers) {
        const auto& m = map.get(name);
        Fnv h;
        long long finite = 0;
        for (long long k = 0; k < (long long)m.size(); ++k) {
          const float v = m.data()[k];
          if (std::isfinite(v)) ++finite;
          h.word(std::isnan(v) ? 0x7FC00000u : f32bits(v));
        }
        std::printf("map %s.%d layer %s finite %lld %016llx\n", names[si], est, name.c_str(), finite, h.h);
      }
    }
  }
}

int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i)
    if (std::strcmp(argv[i], "--move-clear-basic") == 0) g_move_clear_basic = true;
  probeIndex();
  probeMove();
#ifndef FDM_PROBE_MIRROR
  probeRegions();
#endif
  probeColors();
  probePre();
  return 0;
}
