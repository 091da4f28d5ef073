// test_halo_loop.cpp — the multi-GPU loops of INTEGRATION.md §D as a real program: a C++ host with HIP and RCCL
// driving libfdm_engine.so + libfdm_halo.so.  One process = one rank; run alone it builds a ONE-rank ncclComm, so
// that every collective of the N-rank loop (scan broadcast, count all-gather, point exchange, halo exchange)
// executes through librccl on the engine's stream.  Checks: (1) the replicated-scan loop and (2) the routed-scan
// loop leave the same map as a plain engine fed the same scans, bit for bit.
//   build: make -C fastdem_amd/cpp halo_loop      run: fastdem_amd/cpp/build/fdm_halo_loop
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "fdm_halo.h"

#define CK(expr)                                                                  \
  do {                                                                            \
    const long _rc = long(expr);                                                  \
    if (_rc < 0) {                                                                \
      std::fprintf(stderr, "%s:%d: %s -> %ld (%s | %s)\n", __FILE__, __LINE__, #expr, _rc, fdm_last_error(), \
                   fdm_halo_last_error());                                        \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)
#define HCK(expr)                                                                 \
  do {                                                                            \
    if ((expr) != hipSuccess) { std::fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #expr); std::exit(1); } \
  } while (0)
#define NCK(expr)                                                                 \
  do {                                                                            \
    if ((expr) != ncclSuccess) { std::fprintf(stderr, "%s:%d: %s failed\n", __FILE__, __LINE__, #expr); std::exit(1); } \
  } while (0)

static void identity(double* T, double x, double y, double z) {
  std::memset(T, 0, 16 * sizeof(double));
  T[0] = T[5] = T[10] = T[15] = 1.0;
  T[12] = x; T[13] = y; T[14] = z;  // column-major
}

static bool same_layers(fdm_engine* a, fdm_engine* b, int rows, int cols) {
  const int n = fdm_engine_num_layers(a);
  if (n != fdm_engine_num_layers(b)) return false;
  std::vector<float> va(size_t(rows) * cols), vb(va.size());
  for (int i = 0; i < n; ++i) {
    const std::string name = fdm_engine_layer_name(a, i);
    if (fdm_engine_layer_download(a, name.c_str(), va.data(), rows, cols) != 0) return false;
    if (fdm_engine_layer_download(b, name.c_str(), vb.data(), rows, cols) != 0) return false;
    if (std::memcmp(va.data(), vb.data(), va.size() * sizeof(float)) != 0) {
      std::fprintf(stderr, "layer %s differs\n", name.c_str());
      return false;
    }
  }
  return true;
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::fprintf(stderr, "no HIP device: the engine has no CPU fallback\n");
    return 2;
  }
  HCK(hipSetDevice(0));
  const int rank = 0, world = 1;
  ncclUniqueId id;
  NCK(ncclGetUniqueId(&id));
  ncclComm_t comm;
  NCK(ncclCommInitRank(&comm, world, id, rank));

  fdm_geometry geo{};
  geo.length_x = geo.length_y = 40.0;
  geo.resolution = 0.1;
  fdm_config cfg;
  fdm_default_config(&cfg);
  cfg.mode = 1;  // GLOBAL
  const int rows = 400, cols = 400;
  fdm_tile_plan plan;
  fdm_tile tile;
  CK(fdm_tile_plan_make(rank, world, rows, cols, FDM_DEFAULT_HALO, &plan));
  fdm_tile_plan_tile(&plan, &tile);
  fdm_route_plan route;
  fdm_tile_plan_route(&plan, &route);
  fdm_engine *e_rep, *e_routed, *e_ref;
  CK(fdm_engine_create(&geo, &cfg, &tile, 0, &e_rep));
  CK(fdm_engine_create(&geo, &cfg, &tile, 0, &e_routed));
  CK(fdm_engine_create(&geo, &cfg, nullptr, 0, &e_ref));
  const char* names[] = {"elevation", "variance", "elevation_min", "elevation_max", "upper_bound",
                         "lower_bound", "n_points", "obstacle", "intensity"};
  const uint64_t ws_bytes = fdm_halo_workspace_bytes(&plan, 9) + 16;
  float* ws;
  HCK(hipMalloc(reinterpret_cast<void**>(&ws), ws_bytes));

  const uint64_t n = 120000;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> uxy(-24.f, 24.f), ui(0.f, 1.f);
  std::normal_distribution<float> nz(0.f, 0.3f);
  float *d_scan, *d_send, *d_recv;
  uint32_t *d_counts, *d_matrix;
  HCK(hipMalloc(reinterpret_cast<void**>(&d_scan), 4 * n * sizeof(float)));
  HCK(hipMalloc(reinterpret_cast<void**>(&d_send), 4 * n * sizeof(float)));
  HCK(hipMalloc(reinterpret_cast<void**>(&d_recv), 4 * n * sizeof(float)));
  HCK(hipMalloc(reinterpret_cast<void**>(&d_counts), (world + 2) * sizeof(uint32_t)));
  HCK(hipMalloc(reinterpret_cast<void**>(&d_matrix), world * (world + 2) * sizeof(uint32_t)));
  std::vector<uint32_t> h_matrix(world * (world + 2));
  std::vector<float> h(4 * n);
  double Tbs[16], Twb[16];
  identity(Tbs, 0, 0, 0.4);
  for (int k = 0; k < 4; ++k) {
    for (uint64_t i = 0; i < n; ++i) { h[i] = uxy(rng); h[n + i] = uxy(rng); h[2 * n + i] = nz(rng); h[3 * n + i] = ui(rng); }
    HCK(hipMemcpy(d_scan, h.data(), 4 * n * sizeof(float), hipMemcpyHostToDevice));
    identity(Twb, 0.7 * k, -0.3 * k, 0.0);
    // (1) replicated scan: broadcast from the ingest rank, every rank integrates the whole scan, halo exchange
    CK(fdm_halo_broadcast_scan(e_rep, comm, d_scan, 4 * n, /*root=*/0));
    CK(fdm_engine_integrate_device(e_rep, n, d_scan, d_scan + n, d_scan + 2 * n, d_scan + 3 * n, nullptr, nullptr, Tbs, Twb));
    CK(fdm_halo_exchange(e_rep, comm, &plan, names, 9, ws, ws_bytes));
    // (2) routed scan: this rank's slice (here: the whole scan) goes to the owners of its cells
    CK(fdm_engine_route_scan(e_routed, &route, n, d_scan, d_scan + n, d_scan + 2 * n, d_scan + 3 * n, Tbs, Twb, d_send, d_counts));
    CK(fdm_halo_gather_counts(e_routed, comm, d_counts, d_matrix, h_matrix.data(), world));
    uint64_t n_recv = 0;
    int32_t any_in_map = 0;
    CK(fdm_halo_route_exchange(e_routed, comm, &plan, d_send, h_matrix.data(), d_recv, n, &n_recv, &any_in_map));
    CK(fdm_engine_integrate_points4_device(e_routed, n_recv, d_recv, 1, any_in_map, Tbs, Twb));
    CK(fdm_halo_exchange(e_routed, comm, &plan, names, 9, ws, ws_bytes));
    // the reference: a plain engine, host arrays
    fdm_scan_stats st;
    CK(fdm_engine_integrate(e_ref, n, h.data(), h.data() + n, h.data() + 2 * n, h.data() + 3 * n, nullptr, nullptr, Tbs, Twb, &st));
    if (h_matrix[0] != st.n_in_map || h_matrix[1] != st.n_after_filter || n_recv != st.n_in_map) {
      std::fprintf(stderr, "scan %d: routing counters %u %u %u vs statistics %u %u\n", k, h_matrix[0], h_matrix[1],
                   h_matrix[2], st.n_in_map, st.n_after_filter);
      return 1;
    }
  }
  CK(fdm_engine_sync(e_rep));
  CK(fdm_engine_sync(e_routed));
  const bool ok = same_layers(e_rep, e_ref, rows, cols) && same_layers(e_routed, e_ref, rows, cols);
  fdm_engine_destroy(e_rep);
  fdm_engine_destroy(e_routed);
  fdm_engine_destroy(e_ref);
  ncclCommDestroy(comm);
  std::printf("halo loop: %s (replicated and routed loops vs plain engine, 4 scans of %llu points, 1-rank RCCL)\n",
              ok ? "ok" : "MISMATCH", (unsigned long long)n);
  return ok ? 0 : 1;
}
