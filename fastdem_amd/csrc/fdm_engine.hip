// fdm_engine.hip — C ABI (include/fdm_engine.h) over the HIP kernels.  gfx950 only.
//
// Host responsibilities (all O(1) per scan): cast the two Isometry3d matrices to float,
// form R = (T_wb*T_bs).rotation().cast<float>(), pick the ring slot, launch k_bin and
// k_update on the engine's stream.  No per-point or per-cell work ever runs on the CPU and
// there is NO CPU fallback: without a HIP device fdm_engine_create fails with
// FDM_ERR_NO_DEVICE.
#include "../../include/fdm_engine.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>  // stable (key, index) sort of the voxel filter

#include "fdm_kernels.hpp"
#include "fdm_raycast.hpp"
#include "fdm_egress.hpp"
#include "fdm_ingest.hpp"
#include "fdm_post.hpp"

using namespace fdm;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPCK(expr)                                                                        \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return fail(FDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

struct Layer {
  std::string name;
  float* d = nullptr;    // own array (stride 1), or nullptr when the layer is a record field
  int field = -1;        // index inside the cell record, -1 = own array
  bool pending = false;  // allocated, but not yet visible (lazy intensity / colour layers)
};

// field order inside the cell records (KalmanField / P2Field in fdm_kernels.hpp)
const char* const kKalmanFields[KF_COUNT] = {"elevation", "elevation_min", "elevation_max", "variance",
                                             "n_points", "_kalman_p", "_sample_mean", "_sample_m2",
                                             "upper_bound", "lower_bound"};
const char* const kP2Fields[PF_COUNT] = {"elevation", "elevation_min", "elevation_max", "variance",
                                         "n_points", "_p2_q0", "_p2_q1", "_p2_q2", "_p2_q3", "_p2_q4",
                                         "_p2_n0", "_p2_n1", "_p2_n2", "_p2_n3", "_p2_n4",
                                         "upper_bound", "lower_bound"};

}  // namespace

struct fdm_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  fdm_config cfg{};
  GeomConst G{};
  size_t ncell = 0;
  std::vector<Layer> layers;
  float** d_layer_ptrs = nullptr;  // device array of every layer pointer (strip clears)
  int n_layer_ptrs = 0;
  bool layer_ptrs_dirty = true;
  Scratch S{};
  DevState* d_state = nullptr;
  DevState* h_state = nullptr;  // pinned mirror for read-backs
  uint64_t scan_no = 0;
  bool have_scan = false;
  uint32_t last_n = 0;
  int last_was_integrate = 0;
  // staging for the host-pointer entry points
  float* d_stage = nullptr;
  size_t stage_cap = 0;  // in points
  int stage_rr = 0;      // rotating staging block
  int32_t* d_cell_ids = nullptr;
  size_t ids_cap = 0;
  bool want_ids = false;
  bool profile = false;
  bool wave_merge = true;
  int bin_table = 1;                 // k_bin: per-block LDS cell table (option "bin_table")
  int dbg_no_atomics = 0;
  int dbg_upd = 0;
  int bin_threads = 0;               // k_bin4 block size (0 = auto, 128 / 256 / 512): 4 points per thread
  int bin_variant = 0;  // 0 = by scan size, 4 = k_bin4 (LDS-staged), 1 = k_bin (one point/thread)
  size_t bin_part_cap = 0;   // blocks
  unsigned last_bin_blocks = 0;
  std::vector<unsigned long long> h_bin_part;
  unsigned n_tiles = 0;
  std::vector<uint32_t> h_upd_part;
  bool obst_dense_pending = false;  // host wrote the obstacle layer: next scan clears it densely
  bool estimator_ready = false;     // ElevationMapping ctor ran (ensureLayers + obstacle layer)
  bool use_records = true;          // pack the active estimator's state into cell records
  float* d_rec = nullptr;           // [ncell][rec_floats]
  int rec_kind = -1;                // -1 none, 0 Kalman, 1 P2
  int rec_floats = 0;
  float* d_tmp = nullptr;           // ncell floats: contiguous staging for strided layer transfers
  bool cap_pre = false, cap_ras = false;  // scan-callback captures
  float* d_cap = nullptr;            // 4 channels x cap_cap points
  size_t cap_cap = 0;
  float* d_ras = nullptr;            // ncell
  bool saved_want_ids = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float last_ms[3] = {0.f, 0.f, 0.f};
  // raycasting stage (fdm_raycast.hpp)
  uint32_t* rc_cnt = nullptr;        // [ncell] ray-scan points observed in the cell this frame
  uint32_t* rc_min = nullptr;        // [ncell] ord(min ray height), kRayEmpty = not traversed
  unsigned long long* vkeys[2] = {nullptr, nullptr};  // voxel keys: unsorted / sorted
  uint32_t* vidx[2] = {nullptr, nullptr};             // point indices: unsorted / sorted
  uint32_t* vsel = nullptr;          // voxel_any output staging
  size_t vcap = 0;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  hipEvent_t ev_ray[2] = {nullptr, nullptr};
  bool ray_timed = false;
  int dbg_ray = 0;
  // update(t) || bin(t+1) in ONE launch (k_update_bin): the update of the last small scan is held back
  // until the next scan arrives (or any other entry point / sync flushes it); the scratch is
  // double-buffered by scan parity.
  unsigned long long* key2[2] = {nullptr, nullptr};  // scratch of even / odd scans ([0] == the original allocation)
  uint4* aux2[2] = {nullptr, nullptr};
  bool overlap = true;          // option "overlap"
  bool chain = false;           // an update is held back: the next bin derives its geometry from the previous slot
  std::function<int()> upd_alone;   // launches the held-back update on its own
  struct BinVariant { bool bin4, has_int, has_col, wave_merge; unsigned threads; };
  std::function<int(const ScanParams&, const Scratch&, const ScanInputs&, int32_t*, unsigned, BinVariant)> upd_fused;
  bool upd_fuses_bin4 = false;  // the held-back update can ride with a k_bin4 launch (record policies only)
  int last_do_move = 0, last_gate = 0;
  // stencil post-processing (fdm_post.hpp)
  RegionEntry* d_region = nullptr;   // kMaxRegion entries
  float* d_tmp2 = nullptr;           // second ncell staging array (fusion works on two layers)
  // ingest (fdm_ingest.hpp)
  uint8_t* d_blob = nullptr;         // raw message bytes
  size_t blob_cap = 0;
  float* d_in = nullptr;             // 5 channels x in_cap: x y z intensity rgb
  size_t in_cap = 0;
  uint64_t in_n = 0;
  bool in_has_int = false, in_has_rgb = false;
  // egress (fdm_egress.hpp)
  uint32_t* pack_counts = nullptr;   // per-block valid counts / offsets (+1 for the total)
  size_t pack_counts_cap = 0;
  float* d_pack = nullptr;           // packed records
  size_t pack_cap = 0;               // in floats
};

namespace {

// Launch the held-back update kernel, if any.  Called at the top of every entry point that is not
// the next scan of the chain, and before anything that syncs or reallocates.
int join_streams(fdm_engine* e) {
  if (e->chain) {
    e->chain = false;
    std::function<int()> f = std::move(e->upd_alone);
    e->upd_alone = nullptr;
    e->upd_fused = nullptr;
    if (f) {
      if (int rc = f()) return rc;
    }
  }
  return FDM_OK;
}
int sync_all(fdm_engine* e) {
  if (int rc = join_streams(e)) return rc;
  HIPCK(hipStreamSynchronize(e->stream));
  return FDM_OK;
}

Layer* find_layer(fdm_engine* e, const char* name) {
  for (auto& l : e->layers)
    if (l.name == name) return &l;
  return nullptr;
}

// A layer as the kernels see it: base pointer + element stride (1, or the record size).
float* lptr(fdm_engine* e, const Layer& l) { return l.field >= 0 ? e->d_rec + l.field : l.d; }
int lstride(fdm_engine* e, const Layer& l) { return l.field >= 0 ? e->rec_floats : 1; }

int fill_async(fdm_engine* e, float* p, float v, size_t n, int stride = 1) {
  if (n == 0) return FDM_OK;
  const int blocks = int(std::min<size_t>((n + 255) / 256, 4096));
  hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, e->stream, p, v, n, stride);
  HIPCK(hipGetLastError());
  return FDM_OK;
}
int copy_strided(fdm_engine* e, float* dst, int ds, const float* src, int ss) {
  const int blocks = int(std::min<size_t>((e->ncell + 255) / 256, 4096));
  hipLaunchKernelGGL(k_copy_strided, dim3(blocks), dim3(256), 0, e->stream, dst, ds, src, ss, e->ncell);
  HIPCK(hipGetLastError());
  return FDM_OK;
}
int ensure_tmp(fdm_engine* e) {
  if (!e->d_tmp) HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_tmp), e->ncell * sizeof(float)));
  return FDM_OK;
}

int add_layer(fdm_engine* e, const char* name, float value, bool pending = false) {
  if (Layer* l = find_layer(e, name)) {  // GridMap::add on an existing layer overwrites it
    l->pending = l->pending && pending;
    return fill_async(e, lptr(e, *l), value, e->ncell, lstride(e, *l));
  }
  if (e->layers.size() >= size_t(kMaxLayers)) return fail(FDM_ERR_INVALID, "too many layers (max 64)");
  Layer l;
  l.name = name;
  l.pending = pending;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&l.d), e->ncell * sizeof(float)));
  e->layers.push_back(l);
  e->layer_ptrs_dirty = true;
  return fill_async(e, l.d, value, e->ncell);
}

int ensure_layer(fdm_engine* e, const char* name, float value) {
  if (Layer* l = find_layer(e, name)) {
    (void)l;
    return FDM_OK;
  }
  return add_layer(e, name, value);
}

const char* kP2Q[5] = {"_p2_q0", "_p2_q1", "_p2_q2", "_p2_q3", "_p2_q4"};
const char* kP2N[5] = {"_p2_n0", "_p2_n1", "_p2_n2", "_p2_n3", "_p2_n4"};

int activate_records(fdm_engine* e, int kind);

// ElevationMapping ctor (elevation_mapping.cpp:11-39) + Kalman/P2 ensureLayers
// (kalman_estimation.hpp:64-82, quantile_estimation.hpp:97-115): add what is missing.
int ensure_estimator_layers(fdm_engine* e) {
  int rc;
  if (e->cfg.estimation_type == 1) {
    if ((rc = ensure_layer(e, "variance", NAN))) return rc;
    if ((rc = ensure_layer(e, "n_points", 0.0f))) return rc;
    for (int k = 0; k < 5; ++k)
      if ((rc = ensure_layer(e, kP2Q[k], NAN))) return rc;
    for (int k = 0; k < 5; ++k)
      if ((rc = ensure_layer(e, kP2N[k], float(k)))) return rc;
    if ((rc = ensure_layer(e, "upper_bound", NAN))) return rc;
    if ((rc = ensure_layer(e, "lower_bound", NAN))) return rc;
  } else {
    if ((rc = ensure_layer(e, "variance", 0.0f))) return rc;
    if ((rc = ensure_layer(e, "n_points", 0.0f))) return rc;
    if ((rc = ensure_layer(e, "_kalman_p", 0.0f))) return rc;
    if ((rc = ensure_layer(e, "_sample_mean", NAN))) return rc;
    if ((rc = ensure_layer(e, "_sample_m2", 0.0f))) return rc;
    if ((rc = ensure_layer(e, "upper_bound", NAN))) return rc;
    if ((rc = ensure_layer(e, "lower_bound", NAN))) return rc;
  }
  if ((rc = ensure_layer(e, "obstacle", NAN))) return rc;
  return activate_records(e, e->cfg.estimation_type == 1 ? 1 : 0);
}

int refresh_layer_ptrs(fdm_engine* e) {
  if (!e->layer_ptrs_dirty) return FDM_OK;
  std::vector<float*> ptrs;
  for (auto& l : e->layers)
    if (l.field < 0) ptrs.push_back(l.d);  // record fields are cleared with the record
  if (ptrs.empty()) ptrs.push_back(nullptr);
  // the old array may still be referenced by an in-flight kernel: drain first
  if (int rc_sync = sync_all(e)) return rc_sync;
  if (e->d_layer_ptrs) HIPCK(hipFree(e->d_layer_ptrs));
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_layer_ptrs), ptrs.size() * sizeof(float*)));
  HIPCK(hipMemcpy(e->d_layer_ptrs, ptrs.data(), ptrs.size() * sizeof(float*), hipMemcpyHostToDevice));
  e->n_layer_ptrs = 0;
  for (auto& l : e->layers) e->n_layer_ptrs += l.field < 0 ? 1 : 0;
  e->layer_ptrs_dirty = false;
  return FDM_OK;
}

// Lazy layers become visible once a scan that carried the channel landed in the map
// (updateIntensity / updateColor, elevation_mapping.cpp:154-175).
int resolve_pending(fdm_engine* e) {
  bool any = false;
  for (auto& l : e->layers) any = any || l.pending;
  if (!any) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  HIPCK(hipMemcpy(e->h_state, e->d_state, sizeof(DevState), hipMemcpyDeviceToHost));
  const unsigned vi = e->h_state->vis_int, vc = e->h_state->vis_col, vr = e->h_state->vis_ray;
  // newly visible layers move to the END of the list in the order the reference would have created
  // them (getLayers() order is creation order and feeds the PointCloud2 field order)
  std::vector<std::pair<unsigned long long, Layer>> born;
  std::vector<Layer> keep;
  unsigned seq = 0;
  for (auto& l : e->layers) {
    unsigned stamp = 0;
    if (l.pending) {
      if (l.name == "intensity") stamp = vi;
      else if (l.name == "color") stamp = vc;
      else if (l.name == "ghost_removal" || l.name == "raycasting" || l.name == "_visibility_logodds") stamp = vr;
    }
    if (stamp) {
      l.pending = false;
      born.emplace_back((static_cast<unsigned long long>(stamp) << 8) | seq++, l);
    } else {
      keep.push_back(l);
    }
  }
  if (born.empty()) return FDM_OK;
  std::sort(born.begin(), born.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
  for (auto& b2 : born) keep.push_back(b2.second);
  e->layers.swap(keep);
  return FDM_OK;
}

// Eigen: Isometry product linear part, coeff-based 3-term dots a0b0 + (a1b1 + a2b2), then cast.
void rotation_of_product(const double* Twb, const double* Tbs, float* R) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double a0 = Twb[0 * 4 + i] * Tbs[j * 4 + 0];
      const double a1 = Twb[1 * 4 + i] * Tbs[j * 4 + 1];
      const double a2 = Twb[2 * 4 + i] * Tbs[j * 4 + 2];
      R[j * 3 + i] = static_cast<float>(a0 + (a1 + a2));
    }
}

int ensure_ids(fdm_engine* e, size_t n) {
  if (!e->want_ids) return FDM_OK;
  if (n > e->ids_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->d_cell_ids) HIPCK(hipFree(e->d_cell_ids));
    e->ids_cap = n + n / 4 + 1024;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_cell_ids), e->ids_cap * sizeof(int32_t)));
  }
  return FDM_OK;
}

int ensure_scratch_channels(fdm_engine* e, bool intensity, bool color) {
  int rc;
  if (intensity && !find_layer(e, "intensity") && (rc = add_layer(e, "intensity", NAN, true))) return rc;
  if (color && !find_layer(e, "color") && (rc = add_layer(e, "color", NAN, true))) return rc;
  return FDM_OK;
}

float* L(fdm_engine* e, const char* n) {
  Layer* l = find_layer(e, n);
  return l ? l->d : nullptr;
}

// ---- cell records: (de)activate the packed layout for the active estimator ----
// Leaving the record layout: every field goes back to its own array.
int deactivate_records(fdm_engine* e) {
  if (e->rec_kind < 0) return FDM_OK;
  for (auto& l : e->layers) {
    if (l.field < 0) continue;
    float* own = nullptr;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&own), e->ncell * sizeof(float)));
    if (int rc = copy_strided(e, own, 1, e->d_rec + l.field, e->rec_floats)) return rc;
    l.d = own;
    l.field = -1;
  }
  if (int rc_sync = sync_all(e)) return rc_sync;
  HIPCK(hipFree(e->d_rec));
  e->d_rec = nullptr;
  e->rec_kind = -1;
  e->rec_floats = 0;
  e->layer_ptrs_dirty = true;
  return FDM_OK;
}
// Entering it: the estimator's layers (which must all exist) are gathered into the records.
int activate_records(fdm_engine* e, int kind) {
  if (!e->use_records) return deactivate_records(e);
  if (e->rec_kind == kind) return FDM_OK;
  if (int rc = deactivate_records(e)) return rc;
  const int nf = kind == 1 ? int(PF_COUNT) : int(KF_COUNT);
  const char* const* names = kind == 1 ? kP2Fields : kKalmanFields;
  e->rec_floats = kind == 1 ? kP2Rec : kKalmanRec;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_rec), e->ncell * size_t(e->rec_floats) * sizeof(float)));
  if (int rc = fill_async(e, e->d_rec, NAN, e->ncell * size_t(e->rec_floats))) return rc;  // padding too
  for (int f = 0; f < nf; ++f) {
    Layer* l = find_layer(e, names[f]);
    if (!l) return fail(FDM_ERR_NO_LAYER, std::string("estimator layer missing: ") + names[f]);
    if (int rc = copy_strided(e, e->d_rec + f, e->rec_floats, l->d, 1)) return rc;
  }
  if (int rc_sync = sync_all(e)) return rc_sync;
  for (int f = 0; f < nf; ++f) {
    Layer* l = find_layer(e, names[f]);
    HIPCK(hipFree(l->d));
    l->d = nullptr;
    l->field = f;
  }
  e->rec_kind = kind;
  e->layer_ptrs_dirty = true;
  return FDM_OK;
}

// ---- raycasting stage (fdm_raycast.hpp) ----
bool voxel_size_ok(float v) { return v >= 0.001f && v <= 100.0f; }  // voxel_grid_impl.hpp:31-33

// raycasting.cpp:223-226: created on first use; invisible until a frame passed the preconditions
int ensure_ray_layers(fdm_engine* e) {
  int rc;
  for (const char* n : {"ghost_removal", "raycasting", "_visibility_logodds"})
    if (!find_layer(e, n) && (rc = add_layer(e, n, NAN, true))) return rc;
  return FDM_OK;
}

int ensure_ray_cells(fdm_engine* e) {
  if (e->rc_cnt) return FDM_OK;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->rc_cnt), e->ncell * sizeof(uint32_t)));
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->rc_min), e->ncell * sizeof(uint32_t)));
  const int blocks = int(std::min<size_t>((e->ncell + 255) / 256, 4096));
  hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, e->stream, e->rc_cnt, 0u, e->ncell);
  hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, e->stream, e->rc_min, kRayEmpty, e->ncell);
  HIPCK(hipGetLastError());
  return FDM_OK;
}

int ensure_voxel_buffers(fdm_engine* e, size_t n) {
  if (n <= e->vcap) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  for (int k = 0; k < 2; ++k) {
    if (e->vkeys[k]) HIPCK(hipFree(e->vkeys[k]));
    if (e->vidx[k]) HIPCK(hipFree(e->vidx[k]));
  }
  if (e->vsel) HIPCK(hipFree(e->vsel));
  if (e->sort_tmp) HIPCK(hipFree(e->sort_tmp));
  e->vcap = n + n / 4 + 1024;
  for (int k = 0; k < 2; ++k) {
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vkeys[k]), e->vcap * sizeof(unsigned long long)));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vidx[k]), e->vcap * sizeof(uint32_t)));
  }
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vsel), e->vcap * sizeof(uint32_t)));
  e->sort_tmp_bytes = 0;
  HIPCK(rocprim::radix_sort_pairs(nullptr, e->sort_tmp_bytes, e->vkeys[0], e->vkeys[1], e->vidx[0],
                                  e->vidx[1], e->vcap, 0, 64, e->stream));
  {  // the compact-key sort reuses the same allocations (uint32 view of the key buffers)
    size_t b32 = 0;
    HIPCK(rocprim::radix_sort_pairs(nullptr, b32, reinterpret_cast<uint32_t*>(e->vkeys[0]),
                                    reinterpret_cast<uint32_t*>(e->vkeys[1]), e->vidx[0], e->vidx[1], e->vcap, 0,
                                    32, e->stream));
    e->sort_tmp_bytes = std::max(e->sort_tmp_bytes, b32);
  }
  HIPCK(hipMalloc(&e->sort_tmp, e->sort_tmp_bytes ? e->sort_tmp_bytes : 16));
  return FDM_OK;
}

// keys -> stable sort: vkeys[1] / vidx[1] hold the voxel-ordered scan afterwards.
// `box` (nullable): centre + half extent [m] of a box that holds every finite point of the cloud;
// with it the compact 32-bit key is used when 3 * bits <= 31.  Returns through *compact which key
// type the sorted buffer holds.
int enqueue_voxel_sort(fdm_engine* e, unsigned n, float voxel_size, int flag_slot, const float* dx,
                       const float* dy, const float* dz, const double* box, bool* compact) {
  if (int rc = ensure_voxel_buffers(e, n)) return rc;
  const float inv = 1.0f / voxel_size;  // voxel_grid_impl.hpp:46
  VoxelCompact C{0, 0, 0, 0};
  if (box && std::isfinite(box[3]) && box[3] > 0.0 && box[3] * double(inv) < 4.0e6) {
    const double half = box[3] + 2.0 * double(voxel_size);  // 2-cell margin for the float transforms
    const int span = int(std::ceil(2.0 * half * double(inv))) + 4;
    int bits = 1;
    while ((1 << bits) < span) ++bits;
    if (3 * bits <= 62 && bits <= 21) {
      C.bits = bits;
      C.x0 = int(std::floor((box[0] - half) * double(inv))) - 1;
      C.y0 = int(std::floor((box[1] - half) * double(inv))) - 1;
      C.z0 = int(std::floor((box[2] - half) * double(inv))) - 1;
    }
  }
  *compact = C.bits > 0 && 3 * C.bits <= 31;  // true: the sorted buffer holds uint32 keys
  size_t bytes = e->sort_tmp_bytes;
  if (*compact) {
    uint32_t* k0 = reinterpret_cast<uint32_t*>(e->vkeys[0]);
    uint32_t* k1 = reinterpret_cast<uint32_t*>(e->vkeys[1]);
    hipLaunchKernelGGL(k_voxel_keys<uint32_t>, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, inv, flag_slot, C,
                       e->d_state, dx, dy, dz, k0, e->vidx[0], e->vsel);
    HIPCK(hipGetLastError());
    // bits 3*bits .. 31 are zero in every valid key and one in the invalid key (all ones): sorting
    // one bit past the fields is enough to keep the dropped points behind every voxel
    HIPCK(rocprim::radix_sort_pairs(e->sort_tmp, bytes, k0, k1, e->vidx[0], e->vidx[1], size_t(n), 0,
                                    unsigned(3 * C.bits + 1), e->stream));
  } else {
    hipLaunchKernelGGL(k_voxel_keys<unsigned long long>, dim3((n + 255) / 256), dim3(256), 0, e->stream, n, inv,
                       flag_slot, C, e->d_state, dx, dy, dz, e->vkeys[0], e->vidx[0], e->vsel);
    HIPCK(hipGetLastError());
    HIPCK(rocprim::radix_sort_pairs(e->sort_tmp, bytes, e->vkeys[0], e->vkeys[1], e->vidx[0], e->vidx[1],
                                    size_t(n), 0, C.bits > 0 ? unsigned(3 * C.bits + 1) : 64u, e->stream));
  }
  return FDM_OK;
}

fdm_raycast_config ray_config_of(const fdm_config& c) {
  fdm_raycast_config r;
  r.enabled = c.raycast_enabled;
  r.height_conflict_threshold = c.rc_height_conflict_threshold;
  r.log_odds_observed = c.rc_log_odds_observed;
  r.log_odds_ghost = c.rc_log_odds_ghost;
  r.log_odds_max = c.rc_log_odds_max;
  r.clear_threshold = c.rc_clear_threshold;
  return r;
}

RayParams make_ray_params(fdm_engine* e, const fdm_raycast_config& c, const float* origin, unsigned n,
                          int slot, int flag_slot) {
  RayParams Q{};
  Q.ox = origin[0]; Q.oy = origin[1]; Q.oz = origin[2];
  Q.l_obs = c.log_odds_observed;
  Q.l_ghost = c.log_odds_ghost;
  Q.l_max = c.log_odds_max;
  Q.clear_thr = c.clear_threshold;
  Q.conflict_thr = c.height_conflict_threshold;
  Q.resolution = static_cast<float>(e->G.res);
  Q.inv_voxel = 1.0f / Q.resolution;
  Q.n = n;
  Q.slot = slot;
  Q.flag_slot = flag_slot;
  Q.vis_stamp = 3u * unsigned(e->scan_no) + (flag_slot >= 0 ? 3u : 1u);
  Q.dbg = e->dbg_ray;
  return Q;
}

// processScan + resolveGhostCells on the stream.  voxel: the points are vkeys[1]/vidx[1] runs.
int enqueue_ray_stage(fdm_engine* e, const RayParams& Q, bool voxel, const float* dx, const float* dy,
                      const float* dz, bool compact_keys = false) {
  int rc;
  if ((rc = ensure_ray_cells(e))) return rc;
  Layer* elev = find_layer(e, "elevation");
  if (!elev) return FDM_OK;  // raycasting.cpp:213-216
  RayLayers L{};
  L.elevation = lptr(e, *elev);
  L.elevation_stride = lstride(e, *elev);
  L.logodds = find_layer(e, "_visibility_logodds")->d;
  L.ray_min = find_layer(e, "raycasting")->d;
  L.ghost = find_layer(e, "ghost_removal")->d;
  L.rec = e->d_rec;
  L.rec_floats = e->rec_floats;
  const unsigned blocks = (Q.n + 255u) / 256u;
  if ((rc = ensure_voxel_buffers(e, Q.n))) return rc;  // vidx[0] doubles as the ray queue
  uint32_t* ray_list = e->vidx[0];
  if (voxel) {
    if (compact_keys)
      hipLaunchKernelGGL(k_voxel_mark<uint32_t>, dim3(blocks), dim3(256), 0, e->stream, Q.n,
                         reinterpret_cast<const uint32_t*>(e->vkeys[1]), e->vidx[1], e->vsel);
    else
      hipLaunchKernelGGL(k_voxel_mark<unsigned long long>, dim3(blocks), dim3(256), 0, e->stream, Q.n,
                         e->vkeys[1], e->vidx[1], e->vsel);
    hipLaunchKernelGGL(k_ray_compact<true>, dim3(blocks), dim3(256), 0, e->stream, Q, e->G, e->d_state, dx,
                       dy, dz, e->vsel, e->rc_cnt, ray_list);
  } else {
    hipLaunchKernelGGL(k_ray_compact<false>, dim3(blocks), dim3(256), 0, e->stream, Q, e->G, e->d_state, dx,
                       dy, dz, static_cast<const uint32_t*>(nullptr), e->rc_cnt, ray_list);
  }
  HIPCK(hipGetLastError());
  const bool tiled = e->G.o_rows != e->G.rows || e->G.o_cols != e->G.cols || e->G.s_rows != e->G.rows ||
                     e->G.s_cols != e->G.cols;
  auto launch_ray = [&](auto kern, unsigned seg) {
    // upper bound of the queue: every point a ray, padded to whole wavefronts per segment
    const unsigned threads = ((Q.n + 63u) & ~63u) * seg;
    hipLaunchKernelGGL(kern, dim3((threads + 255u) / 256u), dim3(256), 0, e->stream, Q, e->G, e->d_state, dx, dy,
                       dz, ray_list, e->rc_min);
  };
  // small scans are a few hundred wavefronts of dependent round trips: 16 / 8 lanes share a ray
  // (C2: k_ray 60 -> 25 (8) -> 16 us (16)); the point count bounds the ray count from above
  if (Q.n < (1u << 16)) {
    tiled ? launch_ray(k_ray<true, 16>, 16u) : launch_ray(k_ray<false, 16>, 16u);
  } else if (Q.n < (1u << 20)) {
    tiled ? launch_ray(k_ray<true, 8>, 8u) : launch_ray(k_ray<false, 8>, 8u);
  } else {
    tiled ? launch_ray(k_ray<true, 1>, 1u) : launch_ray(k_ray<false, 1>, 1u);
  }
  HIPCK(hipGetLastError());
  hipLaunchKernelGGL(k_ray_resolve, dim3(unsigned((e->ncell + 255) / 256)), dim3(256), 0, e->stream, Q,
                     e->G, e->d_state, L, e->d_layer_ptrs, e->n_layer_ptrs, e->rc_cnt, e->rc_min,
                     unsigned(e->ncell));
  HIPCK(hipGetLastError());
  return FDM_OK;
}

// One scan = k_bin + k_update on the stream.  All pointers are device pointers.
int enqueue_scan(fdm_engine* e, ScanParams& P, uint64_t n, const float* dx, const float* dy,
                 const float* dz, const float* dint, const uint32_t* drgb, const float* dvar) {
  if (n >= 0xFFFFFFFEull) return fail(FDM_ERR_INVALID, "point count exceeds 2^32-2");
  if (dint && n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1 (intensity channel)");
  int rc;
  P.n = uint32_t(n);
  P.scan_no = uint32_t(e->scan_no);
  P.slot = int(e->scan_no & 3);
  P.has_intensity = dint != nullptr;
  P.has_color = drgb != nullptr;
  P.has_var = dvar != nullptr;
  P.sensor_type = e->cfg.sensor_type;
  if (P.sensor_type == 1) {  // LiDARSensorModel ctor takes |noise| (lidar_model.hpp:58-62)
    P.sp[0] = std::fabs(e->cfg.lidar_range_noise);
    P.sp[1] = std::fabs(e->cfg.lidar_angular_noise);
    P.sp[2] = P.sp[3] = 0.f;
  } else if (P.sensor_type == 2) {
    P.sp[0] = e->cfg.rgbd_normal_a;
    P.sp[1] = e->cfg.rgbd_normal_b;
    P.sp[2] = e->cfg.rgbd_normal_c;
    P.sp[3] = e->cfg.rgbd_lateral_factor;
  } else if (P.sensor_type == 0) {
    P.sp[0] = e->cfg.constant_uncertainty;
    P.sp[1] = P.sp[2] = P.sp[3] = 0.f;
  } else {  // unknown -> LiDAR (sensor_model.cpp:34-38)
    P.sensor_type = 1;
    P.sp[0] = std::fabs(e->cfg.lidar_range_noise);
    P.sp[1] = std::fabs(e->cfg.lidar_angular_noise);
    P.sp[2] = P.sp[3] = 0.f;
  }
  if (!e->estimator_ready) {  // a bare map: behave as if FastDEM(map) had been constructed
    if ((rc = ensure_estimator_layers(e))) return rc;
    e->estimator_ready = true;
  }
  if ((rc = ensure_scratch_channels(e, P.has_intensity, P.has_color))) return rc;
  const bool ray_on = P.integrate_mode && e->cfg.raycast_enabled && n > 0;
  if (ray_on) {
    if (!voxel_size_ok(static_cast<float>(e->G.res)))
      return fail(FDM_ERR_INVALID, "raycasting: voxel_size (= map resolution) must be in [0.001, 100]");
    if ((rc = ensure_ray_layers(e))) return rc;
  }
  if ((rc = refresh_layer_ptrs(e))) return rc;
  if ((rc = ensure_ids(e, n))) return rc;
  // scan-callback captures (off unless fdm_engine_capture enabled them)
  e->S.cap_x = e->S.cap_y = e->S.cap_z = e->S.cap_var = nullptr;
  e->S.ras_z = nullptr;
  e->S.cap_drop_nan = ray_on ? 1 : 0;
  if ((e->cap_pre || ray_on) && n) {
    if (n > e->cap_cap) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      if (e->d_cap) HIPCK(hipFree(e->d_cap));
      e->cap_cap = n + n / 4 + 1024;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_cap), e->cap_cap * 4 * sizeof(float)));
    }
    e->S.cap_x = e->d_cap;
    e->S.cap_y = e->d_cap + e->cap_cap;
    e->S.cap_z = e->d_cap + 2 * e->cap_cap;
    if (e->cap_pre) e->S.cap_var = e->d_cap + 3 * e->cap_cap;
  }
  if (e->cap_ras) {
    if (!e->d_ras) HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_ras), e->ncell * sizeof(float)));
    if ((rc = fill_async(e, e->d_ras, NAN, e->ncell))) return rc;
    e->S.ras_z = e->d_ras;
  }

  // ---- launch plan.  Small scans are launch/latency-bound (two dependent launches of ~100 blocks),
  // so the update of scan t is held back and leaves together with the bin of scan t+1 in one launch
  // (k_update_bin): they share nothing — the scratch is double-buffered by scan parity and a
  // chained bin derives its base geometry from slot t (ScanParams::chain_prev).
  const int parity = int(e->scan_no & 1);
  if (e->key2[1]) {  // the scratch set of this scan's parity
    e->S.key = e->key2[parity];
    e->S.aux = e->aux2[parity];
  }
  const bool plain = e->overlap && e->key2[1] && e->S.dense && !ray_on && !e->cap_pre &&
                     !e->cap_ras && !e->obst_dense_pending;

  // k_bin4 (4 consecutive points per thread, float4 loads) needs 16-byte aligned channels
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  // k_bin4 trades latency for fewer memory-side atomics: worth it from ~64 K points up
  const bool want4 = e->bin_variant == 4 || (e->bin_variant == 0 && n >= 65536);
  const bool use_bin4 = want4 && al16(dx) && al16(dy) && al16(dz) && al16(dint);
  // k_bin4 block size: 0 = by scan size.  2048-point blocks (512 threads) merge ~20 % more cells on
  // chip for firing-order LiDAR scans (C4: 50 -> 40 us); smaller scans keep more blocks in flight.
  const int bt = e->bin_threads ? e->bin_threads : (n >= (1u << 20) ? 512 : 256);
  const unsigned bin_threads = use_bin4 ? unsigned(bt) : 256u;
  const unsigned per_block = use_bin4 ? bin_threads * 4u : 256u;
  const unsigned bin_blocks = n ? unsigned((n + per_block - 1) / per_block) : 1u;
  if (bin_blocks > e->bin_part_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->S.bin_part) HIPCK(hipFree(e->S.bin_part));
    e->bin_part_cap = bin_blocks + bin_blocks / 4 + 64;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->S.bin_part), e->bin_part_cap * sizeof(unsigned long long)));
  }
  e->last_bin_blocks = bin_blocks;
  P.dbg_no_atomics = e->dbg_no_atomics;
  P.bin_table = e->bin_table;
  P.dbg_upd = e->dbg_upd;
  // a held-back update leaves now: fused with this bin if this scan is a plain small one, alone otherwise
  const bool bin4_fusable = use_bin4 && (bin_threads == 256 || bin_threads == 512) && e->upd_fuses_bin4;
  const bool fuse_now = e->chain && plain && (!use_bin4 || bin4_fusable) && e->upd_fused;
  if (e->chain && !fuse_now && (rc = join_streams(e))) return rc;
  P.chain_prev = 0;
  if (e->profile) HIPCK(hipEventRecord(e->ev[0], e->stream));
  int32_t* ids = e->want_ids ? e->d_cell_ids : nullptr;
  if (fuse_now) {  // the held-back update of the previous scan + this scan's bin, one launch
    P.chain_prev = 1;
    P.prev_do_move = e->last_do_move;
    P.prev_gate = e->last_gate;
    const ScanInputs in_b{dx, dy, dz, dint, drgb, dvar};
    auto fused = std::move(e->upd_fused);
    e->upd_fused = nullptr;
    e->upd_alone = nullptr;
    e->chain = false;
    const fdm_engine::BinVariant bv{use_bin4, P.has_intensity != 0, P.has_color != 0, e->wave_merge, bin_threads};
    if ((rc = fused(P, e->S, in_b, ids, bin_blocks, bv))) return rc;
  } else if (use_bin4) {
    const bool hi = P.has_intensity != 0, hc = P.has_color != 0;
    auto launch4 = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(bin_blocks), dim3(bin_threads), 0, e->stream, P, e->G, e->d_state, dx,
                         dy, dz, dint, e->S, ids);
    };
#define FDM_BIN4(T)                                         \
    if (hi && hc) launch4(k_bin4<true, true, T>);           \
    else if (hi) launch4(k_bin4<true, false, T>);           \
    else if (hc) launch4(k_bin4<false, true, T>);           \
    else launch4(k_bin4<false, false, T>);
    if (bin_threads == 128) { FDM_BIN4(128) }
    else if (bin_threads == 512) { FDM_BIN4(512) }
    else { FDM_BIN4(256) }
#undef FDM_BIN4
  } else {
    auto launch_bin = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(bin_blocks), dim3(256), 0, e->stream, P, e->G, e->d_state, dx, dy,
                         dz, dint, e->S, ids);
    };
    e->wave_merge ? launch_bin(k_bin<true>) : launch_bin(k_bin<false>);
  }
  HIPCK(hipGetLastError());
  if (e->profile) HIPCK(hipEventRecord(e->ev[1], e->stream));

  if (e->obst_dense_pending) {
    const int blocks = int(std::min<size_t>((e->ncell + 255) / 256, 4096));
    hipLaunchKernelGGL(k_obstacle_dense_clear, dim3(blocks), dim3(256), 0, e->stream, P, e->d_state,
                       L(e, "obstacle"), e->ncell);
    HIPCK(hipGetLastError());
    e->obst_dense_pending = false;
  }
  const unsigned upd_blocks = e->n_tiles;  // one block per map tile
  // P2Quantile ctor (quantile_estimation.hpp:84-95): clamp, then enforce monotone dn
  P2Params p2{};
  {
    auto clamp01 = [](float v) { return v < 0.f ? 0.f : (1.f < v ? 1.f : v); };
    for (int k = 0; k < 5; ++k) p2.dn[k] = clamp01(e->cfg.p2_dn[k]);
    for (int k = 1; k < 5; ++k) p2.dn[k] = std::max(p2.dn[k], p2.dn[k - 1]);
    p2.marker = std::min(std::max(e->cfg.p2_elevation_marker, 0), 4);
    p2.max_count = std::max(e->cfg.p2_max_sample_count, 0.0f);
  }
  // the next scan's launch (or a flush) carries this update
  const bool hold = plain;
  const ScanInputs in_u{dx, dy, dz, dint, drgb, dvar};
  auto launch_upd = [&](auto policy_tag, const auto& layers) {
    using POLICY = decltype(policy_tag);
    if (!hold) {
      if (!e->S.dense) {  // stamp-gated: 16 tiles per block, idle tiles cost one scalar load
        hipLaunchKernelGGL(k_update_stamped<POLICY>, dim3((upd_blocks + kStampTiles - 1) / kStampTiles), dim3(256),
                           0, e->stream, P, e->G, e->d_state, layers, e->d_layer_ptrs, e->n_layer_ptrs, e->S, dx,
                           dy, dz, dint, drgb, dvar, unsigned(e->ncell));
        return;
      }
      hipLaunchKernelGGL(k_update<POLICY>, dim3(upd_blocks), dim3(256), 0, e->stream, P, e->G, e->d_state,
                         layers, e->d_layer_ptrs, e->n_layer_ptrs, e->S, dx, dy, dz, dint, drgb, dvar,
                         unsigned(e->ncell));
      return;
    }
    const ScanParams Pu = P;
    const Scratch Su = e->S;
    const auto Lu = layers;
    e->upd_alone = [e, Pu, Su, Lu, in_u, upd_blocks]() -> int {
      hipLaunchKernelGGL(k_update<POLICY>, dim3(upd_blocks), dim3(256), 0, e->stream, Pu, e->G, e->d_state, Lu,
                         e->d_layer_ptrs, e->n_layer_ptrs, Su, in_u.x, in_u.y, in_u.z, in_u.intensity, in_u.rgb,
                         in_u.var, unsigned(e->ncell));
      HIPCK(hipGetLastError());
      return FDM_OK;
    };
    constexpr bool kRec = std::is_same<POLICY, KalmanRecPolicy>::value || std::is_same<POLICY, P2RecPolicy>::value;
    e->upd_fuses_bin4 = kRec;
    e->upd_fused = [e, Pu, Su, Lu, in_u, upd_blocks](const ScanParams& Pb, const Scratch& Sb, const ScanInputs& Ib,
                                                     int32_t* ids_b, unsigned bin_blocks_b,
                                                     fdm_engine::BinVariant bv) -> int {
      auto go = [&](auto kern, unsigned threads) {
        const unsigned ub = (upd_blocks + threads / 256u - 1u) / (threads / 256u);  // tiles per update block
        hipLaunchKernelGGL(kern, dim3(ub + bin_blocks_b), dim3(threads), 0, e->stream, Pu, e->G, e->d_state, Lu,
                           e->d_layer_ptrs, e->n_layer_ptrs, Su, in_u, unsigned(e->ncell), ub, Pb, Sb, Ib, ids_b);
      };
      if (!bv.bin4) {
        bv.wave_merge ? go(k_update_bin<POLICY, true>, 256u) : go(k_update_bin<POLICY, false>, 256u);
      } else if constexpr (kRec) {
#define FDM_FUSED4(T)                                                              \
        if (bv.has_int && bv.has_col) go(k_update_bin4<POLICY, true, true, T>, T);   \
        else if (bv.has_int) go(k_update_bin4<POLICY, true, false, T>, T);           \
        else if (bv.has_col) go(k_update_bin4<POLICY, false, true, T>, T);           \
        else go(k_update_bin4<POLICY, false, false, T>, T);
        if (bv.threads == 512u) { FDM_FUSED4(512) } else { FDM_FUSED4(256) }
#undef FDM_FUSED4
      } else {
        return fail(FDM_ERR_INVALID, "internal: k_bin4 fused with a per-layer policy");
      }
      HIPCK(hipGetLastError());
      return FDM_OK;
    };
    e->chain = true;
    e->last_do_move = P.do_move;
    e->last_gate = P.gate_on_filter;
  };
  const bool p2mode = e->cfg.estimation_type == 1;
  if (e->rec_kind >= 0) {  // cell records
    if (p2mode) {
      P2RecLayers Lr{};
      Lr.rec = e->d_rec; Lr.obstacle = L(e, "obstacle"); Lr.intensity = L(e, "intensity"); Lr.color = L(e, "color");
      Lr.p = p2;
      launch_upd(P2RecPolicy{}, Lr);
    } else {
      KalmanRecLayers Lr{};
      Lr.rec = e->d_rec; Lr.obstacle = L(e, "obstacle"); Lr.intensity = L(e, "intensity"); Lr.color = L(e, "color");
      Lr.min_var = e->cfg.kalman_min_variance; Lr.max_var = e->cfg.kalman_max_variance;
      Lr.q = e->cfg.kalman_process_noise;
      launch_upd(KalmanRecPolicy{}, Lr);
    }
  } else if (p2mode) {
    P2Layers Lp{};
    Lp.elevation = L(e, "elevation");
    Lp.elevation_min = L(e, "elevation_min");
    Lp.elevation_max = L(e, "elevation_max");
    Lp.variance = L(e, "variance");
    Lp.n_points = L(e, "n_points");
    Lp.upper = L(e, "upper_bound");
    Lp.lower = L(e, "lower_bound");
    Lp.obstacle = L(e, "obstacle");
    Lp.intensity = L(e, "intensity");
    Lp.color = L(e, "color");
    for (int k = 0; k < 5; ++k) {
      Lp.q[k] = L(e, kP2Q[k]);
      Lp.n[k] = L(e, kP2N[k]);
    }
    Lp.p = p2;
    launch_upd(P2Policy{}, Lp);
  } else {
    KalmanLayers Lk{};
    Lk.elevation = L(e, "elevation");
    Lk.elevation_min = L(e, "elevation_min");
    Lk.elevation_max = L(e, "elevation_max");
    Lk.variance = L(e, "variance");
    Lk.n_points = L(e, "n_points");
    Lk.kalman_p = L(e, "_kalman_p");
    Lk.sample_mean = L(e, "_sample_mean");
    Lk.sample_m2 = L(e, "_sample_m2");
    Lk.upper = L(e, "upper_bound");
    Lk.lower = L(e, "lower_bound");
    Lk.obstacle = L(e, "obstacle");
    Lk.intensity = L(e, "intensity");
    Lk.color = L(e, "color");
    Lk.min_var = e->cfg.kalman_min_variance;
    Lk.max_var = e->cfg.kalman_max_variance;
    Lk.q = e->cfg.kalman_process_noise;
    launch_upd(KalmanPolicy{}, Lk);
  }
  HIPCK(hipGetLastError());
  if (e->profile) {
    HIPCK(hipEventRecord(e->ev[2], e->stream));
    HIPCK(hipEventRecord(e->ev[3], e->stream));  // back-to-back pair: the event-to-event overhead
  }
  e->ray_timed = false;
  if (ray_on) {  // step 3 of integrateImpl (fastdem.cpp:152-159) on the map this scan just updated
    if (e->profile) HIPCK(hipEventRecord(e->ev_ray[0], e->stream));
    const float origin[3] = {P.ray_ox, P.ray_oy, P.ray_oz};
    // cropRange keeps d^2 <= range_max^2 around the BASE origin, i.e. around T_world_base's translation
    const double box[4] = {P.base_x, P.base_y, P.base_z, double(e->cfg.range_max)};
    bool compact = false;
    if ((rc = enqueue_voxel_sort(e, P.n, static_cast<float>(e->G.res), P.slot, e->S.cap_x, e->S.cap_y,
                                 e->S.cap_z, box, &compact)))
      return rc;
    const RayParams Q = make_ray_params(e, ray_config_of(e->cfg), origin, P.n, (P.slot + 1) & 3, P.slot);
    if ((rc = enqueue_ray_stage(e, Q, true, e->S.cap_x, e->S.cap_y, e->S.cap_z, compact))) return rc;
    if (e->profile) {
      HIPCK(hipEventRecord(e->ev_ray[1], e->stream));
      e->ray_timed = true;
    }
  }
  e->scan_no++;
  e->have_scan = true;
  e->last_n = uint32_t(n);
  e->last_was_integrate = P.integrate_mode;
  return FDM_OK;
}

void fill_integrate_params(fdm_engine* e, ScanParams& P, const double* Tbs, const double* Twb) {
  std::memset(&P, 0, sizeof(P));
  for (int i = 0; i < 16; ++i) {
    P.Tbs[i] = static_cast<float>(Tbs[i]);
    P.Twb[i] = static_cast<float>(Twb[i]);
  }
  rotation_of_product(Twb, Tbs, P.R);
  // cropRange (crop_impl.hpp:79-96): squares in fp32, FLT_MAX^2 = +inf
  P.min_sq = e->cfg.range_min * e->cfg.range_min;
  P.max_sq = e->cfg.range_max * e->cfg.range_max;
  P.z_min = e->cfg.z_min;
  P.z_max = e->cfg.z_max;
  P.robot_x = Twb[12];  // T_world_base.translation().head<2>() (fastdem.cpp:144)
  P.robot_y = Twb[13];
  {  // (T_world_base * T_base_sensor).translation().cast<float>() (fastdem.cpp:153-154):
     // L_wb * t_bs (3-term coeff redux a0 + (a1 + a2)) + t_wb, in double, then the cast
    float o[3];
    for (int i = 0; i < 3; ++i) {
      const double a0 = Twb[0 * 4 + i] * Tbs[12], a1 = Twb[1 * 4 + i] * Tbs[13], a2 = Twb[2 * 4 + i] * Tbs[14];
      o[i] = static_cast<float>((a0 + (a1 + a2)) + Twb[12 + i]);
    }
    P.ray_ox = o[0]; P.ray_oy = o[1]; P.ray_oz = o[2];
    P.base_x = Twb[12]; P.base_y = Twb[13]; P.base_z = Twb[14];
  }
  P.integrate_mode = 1;
  P.do_move = e->cfg.mode == 0 ? 1 : 0;
  P.gate_on_filter = 1;
}

void fill_update_params(fdm_engine* e, ScanParams& P, double rx, double ry, bool force_move) {
  std::memset(&P, 0, sizeof(P));
  P.robot_x = rx;
  P.robot_y = ry;
  P.integrate_mode = 0;
  P.do_move = (force_move || e->cfg.mode == 0) ? 1 : 0;
  P.gate_on_filter = 0;
}

// Staging for host-array entry points: kStageSlots rotating blocks of 6 channels.  A block is reused
// three scans later, when the update that gathers from it (held back by at most one scan) has long
// been launched ahead of the new copy on the same stream.
constexpr int kStageSlots = 3;

int ensure_stage(fdm_engine* e, size_t n) {
  if (n <= e->stage_cap) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  if (e->d_stage) HIPCK(hipFree(e->d_stage));
  e->stage_cap = n + n / 4 + 1024;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_stage), e->stage_cap * 6 * kStageSlots * sizeof(float)));
  return FDM_OK;
}

// H2D of the SoA channels into the staging block; returns device pointers (nullable ones stay null)
int stage_inputs(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                 const float* a, const uint32_t* rgb, const float* v, const float** dx,
                 const float** dy, const float** dz, const float** da, const uint32_t** drgb,
                 const float** dv) {
  int rc;
  if ((rc = ensure_stage(e, n))) return rc;
  e->stage_rr = (e->stage_rr + 1) % kStageSlots;
  const size_t cap = e->stage_cap;
  float* base = e->d_stage + size_t(e->stage_rr) * 6 * cap;
  auto up = [&](const void* src, int k) -> int {
    HIPCK(hipMemcpyAsync(base + cap * k, src, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return FDM_OK;
  };
  if ((rc = up(x, 0)) || (rc = up(y, 1)) || (rc = up(z, 2))) return rc;
  *dx = base;
  *dy = base + cap;
  *dz = base + cap * 2;
  *da = nullptr;
  *drgb = nullptr;
  *dv = nullptr;
  if (a) {
    if ((rc = up(a, 3))) return rc;
    *da = base + cap * 3;
  }
  if (rgb) {
    if ((rc = up(rgb, 4))) return rc;
    *drgb = reinterpret_cast<const uint32_t*>(base + cap * 4);
  }
  if (v) {
    if ((rc = up(v, 5))) return rc;
    *dv = base + cap * 5;
  }
  return FDM_OK;
}

int read_stats(fdm_engine* e, fdm_scan_stats* out, int* status) {
  if (int rc_sync = sync_all(e)) return rc_sync;
  fdm_scan_stats s{};
  *status = FDM_OK;
  if (!e->have_scan) {
    if (out) *out = s;
    return FDM_OK;
  }
  const int slot = int((e->scan_no - 1) & 3);
  HIPCK(hipMemcpy(e->h_state, e->d_state, sizeof(DevState), hipMemcpyDeviceToHost));
  const DevState& st = *e->h_state;
  uint64_t np = 0, ni = 0;
  e->h_bin_part.resize(e->last_bin_blocks);
  HIPCK(hipMemcpy(e->h_bin_part.data(), e->S.bin_part, e->last_bin_blocks * sizeof(unsigned long long),
                  hipMemcpyDeviceToHost));
  for (unsigned long long v : e->h_bin_part) {
    np += uint32_t(v);
    ni += uint32_t(v >> 32);
  }
  uint64_t nt = 0;
  e->h_upd_part.resize(e->n_tiles);
  HIPCK(hipMemcpy(e->h_upd_part.data(), e->S.upd_part, e->n_tiles * sizeof(uint32_t),
                  hipMemcpyDeviceToHost));
  for (uint32_t v : e->h_upd_part) nt += v;
  s.n_input = e->last_n;
  s.n_after_filter = uint32_t(np);
  s.n_in_map = uint32_t(ni);
  s.n_cells_touched = uint32_t(nt);
  const bool applied = e->last_was_integrate ? (np > 0) : true;
  if (applied) {
    s.shift_rows = st.cand[slot].shr;
    s.shift_cols = st.cand[slot].shc;
  }
  if (e->last_was_integrate) {
    if (e->last_n == 0) *status = FDM_SKIP_EMPTY_CLOUD;
    else if (np == 0) *status = FDM_SKIP_ALL_FILTERED;
  }
  if (out) *out = s;
  if (e->profile) {
    (void)hipEventElapsedTime(&e->last_ms[0], e->ev[0], e->ev[1]);
    (void)hipEventElapsedTime(&e->last_ms[1], e->ev[1], e->ev[2]);
  }
  return FDM_OK;
}

}  // namespace

extern "C" {

void fdm_default_config(fdm_config* c) {
  c->z_min = -std::numeric_limits<float>::max();
  c->z_max = std::numeric_limits<float>::max();
  c->range_min = 0.0f;
  c->range_max = std::numeric_limits<float>::max();
  c->sensor_type = 1;
  c->lidar_range_noise = 0.02f;
  c->lidar_angular_noise = 0.001f;
  c->rgbd_normal_a = 0.001f;
  c->rgbd_normal_b = 0.002f;
  c->rgbd_normal_c = 0.4f;
  c->rgbd_lateral_factor = 0.001f;
  c->constant_uncertainty = 0.03f;
  c->mode = 0;
  c->estimation_type = 0;
  c->kalman_min_variance = 0.0001f;
  c->kalman_max_variance = 0.01f;
  c->kalman_process_noise = 0.0f;
  const float dn[5] = {0.01f, 0.16f, 0.50f, 0.84f, 0.99f};
  for (int k = 0; k < 5; ++k) c->p2_dn[k] = dn[k];
  c->p2_elevation_marker = 3;
  c->p2_max_sample_count = 0.0f;
  c->raycast_enabled = 0;  // config/postprocess.hpp:16-23
  c->rc_height_conflict_threshold = 0.05f;
  c->rc_log_odds_observed = 0.4f;
  c->rc_log_odds_ghost = 0.2f;
  c->rc_log_odds_max = 2.0f;
  c->rc_clear_threshold = -1.0f;
}

const char* fdm_last_error(void) { return g_err.c_str(); }

static int create_impl(const fdm_geometry* g, const fdm_config* cfg, const fdm_tile* tile, int device,
                       bool with_estimator, fdm_engine** out) {
  if (!g || !cfg || !out) return fail(FDM_ERR_INVALID, "null argument");
  if (!(g->resolution > 0.0) || !(g->length_x > 0.0) || !(g->length_y > 0.0))
    return fail(FDM_ERR_INVALID, "length and resolution must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FDM_ERR_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(FDM_ERR_INVALID, "bad device ordinal");
  HIPCK(hipSetDevice(device));

  fdm_engine* e = new fdm_engine();
  e->device = device;
  e->cfg = *cfg;
  // nanogrid::GridMap::setGeometry: size = round(length / resolution); length = size * resolution
  GeomConst& G = e->G;
  G.rows = int(std::round(g->length_x / g->resolution));
  G.cols = int(std::round(g->length_y / g->resolution));
  if (G.rows <= 0 || G.cols <= 0) {
    delete e;
    return fail(FDM_ERR_INVALID, "map has no cells");
  }
  G.res = g->resolution;
  G.inv_res = 1.0 / G.res;  // fast-path only; the exact divide decides near cell edges
  G.len_x = double(G.rows) * G.res;
  G.len_y = double(G.cols) * G.res;
  G.half_x = 0.5 * G.len_x;
  G.half_y = 0.5 * G.len_y;
  if (tile) {
    if (cfg->mode != 1) {
      delete e;
      return fail(FDM_ERR_INVALID, "tiled engines require GLOBAL mode");
    }
    const bool ok = tile->rows > 0 && tile->cols > 0 && tile->row0 >= 0 && tile->col0 >= 0 &&
                    tile->row0 + tile->rows <= G.rows && tile->col0 + tile->cols <= G.cols &&
                    tile->own_row0 >= tile->row0 && tile->own_col0 >= tile->col0 &&
                    tile->own_row0 + tile->own_rows <= tile->row0 + tile->rows &&
                    tile->own_col0 + tile->own_cols <= tile->col0 + tile->cols;
    if (!ok) {
      delete e;
      return fail(FDM_ERR_INVALID, "tile window outside the map or owned window outside the tile");
    }
    G.s_r0 = tile->row0; G.s_c0 = tile->col0; G.s_rows = tile->rows; G.s_cols = tile->cols;
    G.o_r0 = tile->own_row0; G.o_c0 = tile->own_col0; G.o_rows = tile->own_rows; G.o_cols = tile->own_cols;
  } else {
    G.s_r0 = G.s_c0 = G.o_r0 = G.o_c0 = 0;
    G.s_rows = G.o_rows = G.rows;
    G.s_cols = G.o_cols = G.cols;
  }
  e->ncell = size_t(G.s_rows) * size_t(G.s_cols);
  if (e->ncell >= 0xFFFFFFFFull) {
    delete e;
    return fail(FDM_ERR_INVALID, "tile exceeds 2^32 cells");
  }

#define CK(expr)                 \
  do {                           \
    int _rc = (expr);            \
    if (_rc != FDM_OK) {         \
      fdm_engine_destroy(e);     \
      return _rc;                \
    }                            \
  } while (0)
#define HCK(expr)                                                                         \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      fdm_engine_destroy(e);                                                              \
      return fail(FDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    }                                                                                     \
  } while (0)

  HCK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  for (auto& ev : e->ev) HCK(hipEventCreate(&ev));
  for (auto& ev : e->ev_ray) HCK(hipEventCreate(&ev));
  HCK(hipMalloc(reinterpret_cast<void**>(&e->d_state), sizeof(DevState)));
  HCK(hipHostMalloc(reinterpret_cast<void**>(&e->h_state), sizeof(DevState)));
  std::memset(e->h_state, 0, sizeof(DevState));
  for (int k = 0; k < 4; ++k) {
    e->h_state->geom[k].px = g->position_x;
    e->h_state->geom[k].py = g->position_y;
    e->h_state->geom[k].sr = 0;
    e->h_state->geom[k].sc = 0;
    e->h_state->obst[k].scan = 0xFFFFFFFDu;  // "no updating scan yet"
  }
  HCK(hipMemcpy(e->d_state, e->h_state, sizeof(DevState), hipMemcpyHostToDevice));

  // Up to 4 M cells every tile is visited each scan (keys are read unconditionally, which takes
  // one dependent round trip out of the update kernel); larger maps gate tiles by scan stamps.
  e->S.dense = e->ncell <= (size_t(4) << 20) ? 1 : 0;
  e->n_tiles = unsigned((e->ncell + 255) >> 8);
  HCK(hipMalloc(reinterpret_cast<void**>(&e->S.key), e->ncell * sizeof(unsigned long long)));
  HCK(hipMalloc(reinterpret_cast<void**>(&e->S.aux), e->ncell * sizeof(uint4)));
  HCK(hipMalloc(reinterpret_cast<void**>(&e->S.upd_part), e->n_tiles * sizeof(uint32_t)));
  HCK(hipMalloc(reinterpret_cast<void**>(&e->S.tile_stamp), e->n_tiles * sizeof(uint32_t)));
  e->key2[0] = e->S.key;
  e->aux2[0] = e->S.aux;
  if (e->S.dense) {  // second scratch set: scan t+1 bins while scan t still updates (24 B/cell, <= 100 MB)
    HCK(hipMalloc(reinterpret_cast<void**>(&e->key2[1]), e->ncell * sizeof(unsigned long long)));
    HCK(hipMalloc(reinterpret_cast<void**>(&e->aux2[1]), e->ncell * sizeof(uint4)));
    const int blocks2 = int(std::min<size_t>((e->ncell + 255) / 256, 4096));
    hipLaunchKernelGGL(k_fill_u64, dim3(blocks2), dim3(256), 0, e->stream, e->key2[1], kEmptyKey, e->ncell);
    hipLaunchKernelGGL(k_fill_aux, dim3(blocks2), dim3(256), 0, e->stream, e->aux2[1], e->ncell);
  }
  {
    const int blocks = int(std::min<size_t>((e->ncell + 255) / 256, 4096));
    hipLaunchKernelGGL(k_fill_u64, dim3(blocks), dim3(256), 0, e->stream, e->S.key, kEmptyKey, e->ncell);
    hipLaunchKernelGGL(k_fill_aux, dim3(blocks), dim3(256), 0, e->stream, e->S.aux, e->ncell);
    hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, e->stream, e->S.tile_stamp, 0xFFFFFFFEu,
                       size_t(e->n_tiles));
    hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, e->stream, e->S.upd_part, 0u,
                       size_t(e->n_tiles));
    HCK(hipGetLastError());
  }
  // ElevationMap ctor: elevation, elevation_min, elevation_max = NaN (elevation_map.hpp:101-116)
  CK(add_layer(e, "elevation", NAN));
  CK(add_layer(e, "elevation_min", NAN));
  CK(add_layer(e, "elevation_max", NAN));
  if (with_estimator) {
    CK(ensure_estimator_layers(e));
    e->estimator_ready = true;
  }
  HCK(hipStreamSynchronize(e->stream));
#undef CK
#undef HCK
  *out = e;
  return FDM_OK;
}

int fdm_engine_create(const fdm_geometry* g, const fdm_config* cfg, const fdm_tile* tile, int device,
                      fdm_engine** out) {
  return create_impl(g, cfg, tile, device, true, out);
}

int fdm_engine_create_map(const fdm_geometry* g, const fdm_tile* tile, int device, fdm_engine** out) {
  fdm_config cfg;
  fdm_default_config(&cfg);
  if (tile) cfg.mode = 1;
  return create_impl(g, &cfg, tile, device, false, out);
}

void fdm_engine_destroy(fdm_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (auto& l : e->layers)
    if (l.d) (void)hipFree(l.d);
  if (e->d_rec) (void)hipFree(e->d_rec);
  if (e->d_tmp) (void)hipFree(e->d_tmp);
  if (e->d_layer_ptrs) (void)hipFree(e->d_layer_ptrs);
  e->upd_alone = nullptr;
  e->upd_fused = nullptr;
  if (e->key2[0]) (void)hipFree(e->key2[0]);
  if (e->aux2[0]) (void)hipFree(e->aux2[0]);
  if (e->key2[1]) (void)hipFree(e->key2[1]);
  if (e->aux2[1]) (void)hipFree(e->aux2[1]);

  if (e->S.bin_part) (void)hipFree(e->S.bin_part);
  if (e->S.upd_part) (void)hipFree(e->S.upd_part);
  if (e->S.tile_stamp) (void)hipFree(e->S.tile_stamp);
  if (e->d_state) (void)hipFree(e->d_state);
  if (e->h_state) (void)hipHostFree(e->h_state);
  if (e->d_stage) (void)hipFree(e->d_stage);
  if (e->d_cell_ids) (void)hipFree(e->d_cell_ids);
  if (e->d_cap) (void)hipFree(e->d_cap);
  if (e->d_ras) (void)hipFree(e->d_ras);
  for (auto& ev : e->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->ev_ray)
    if (ev) (void)hipEventDestroy(ev);
  if (e->d_region) (void)hipFree(e->d_region);
  if (e->d_tmp2) (void)hipFree(e->d_tmp2);
  if (e->d_blob) (void)hipFree(e->d_blob);
  if (e->d_in) (void)hipFree(e->d_in);
  if (e->pack_counts) (void)hipFree(e->pack_counts);
  if (e->d_pack) (void)hipFree(e->d_pack);
  if (e->rc_cnt) (void)hipFree(e->rc_cnt);
  if (e->rc_min) (void)hipFree(e->rc_min);
  for (int k = 0; k < 2; ++k) {
    if (e->vkeys[k]) (void)hipFree(e->vkeys[k]);
    if (e->vidx[k]) (void)hipFree(e->vidx[k]);
  }
  if (e->vsel) (void)hipFree(e->vsel);
  if (e->sort_tmp) (void)hipFree(e->sort_tmp);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int fdm_engine_set_config(fdm_engine* e, const fdm_config* cfg) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !cfg) return fail(FDM_ERR_INVALID, "null argument");
  if (cfg->mode != 1 && (e->G.s_rows != e->G.rows || e->G.s_cols != e->G.cols))
    return fail(FDM_ERR_INVALID, "tiled engines require GLOBAL mode");
  e->cfg = *cfg;
  e->estimator_ready = true;
  return ensure_estimator_layers(e);
}

int fdm_engine_set_stream(fdm_engine* e, void* hip_stream) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc_sync = sync_all(e)) return rc_sync;
  if (hip_stream) {
    if (e->own_stream && e->stream) HIPCK(hipStreamDestroy(e->stream));
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->own_stream = false;
  } else if (!e->own_stream) {
    HIPCK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  }
  return FDM_OK;
}

int fdm_engine_integrate_device(fdm_engine* e, uint64_t n, const float* dx, const float* dy,
                                const float* dz, const float* dint, const uint32_t* drgb,
                                const float* dvar, const double Tbs[16], const double Twb[16]) {
  if (!e || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (n && (!dx || !dy || !dz)) return fail(FDM_ERR_INVALID, "null xyz");
  HIPCK(hipSetDevice(e->device));
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  if (n == 0) {  // fastdem.cpp:125-128: nothing is touched, not even the move
    e->have_scan = true;
    e->last_n = 0;
    e->last_was_integrate = 1;
    // consume no slot; last_stats reports SKIP_EMPTY_CLOUD
    return FDM_OK;
  }
  return enqueue_scan(e, P, n, dx, dy, dz, dint, drgb, dvar);
}

int fdm_engine_integrate(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                         const float* intensity, const uint32_t* rgb, const float* sigma_z2,
                         const double Tbs[16], const double Twb[16], fdm_scan_stats* out) {
  if (e) { if (int rc_join = join_streams(e)) return rc_join; }
  if (!e || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (n == 0) {
    if (out) {
      std::memset(out, 0, sizeof(*out));
    }
    return FDM_SKIP_EMPTY_CLOUD;
  }
  if (!x || !y || !z) return fail(FDM_ERR_INVALID, "null xyz");
  HIPCK(hipSetDevice(e->device));
  const float *dx, *dy, *dz, *da, *dv;
  const uint32_t* dc;
  int rc = stage_inputs(e, n, x, y, z, intensity, rgb, sigma_z2, &dx, &dy, &dz, &da, &dc, &dv);
  if (rc) return rc;
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  if ((rc = enqueue_scan(e, P, n, dx, dy, dz, da, dc, dv))) return rc;
  int status = FDM_OK;
  if ((rc = read_stats(e, out, &status))) return rc;
  return status;
}

int fdm_engine_integrate_async(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                               const float* intensity, const uint32_t* rgb, const float* sigma_z2,
                               const double Tbs[16], const double Twb[16]) {
  if (!e || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (n == 0) return fdm_engine_integrate_device(e, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Tbs, Twb);
  if (!x || !y || !z) return fail(FDM_ERR_INVALID, "null xyz");
  HIPCK(hipSetDevice(e->device));
  const float *dx, *dy, *dz, *da, *dv;
  const uint32_t* dc;
  int rc = stage_inputs(e, n, x, y, z, intensity, rgb, sigma_z2, &dx, &dy, &dz, &da, &dc, &dv);
  if (rc) return rc;
  return fdm_engine_integrate_device(e, n, dx, dy, dz, da, dc, dv, Tbs, Twb);
}

int fdm_engine_update_device(fdm_engine* e, uint64_t n, const float* dx, const float* dy,
                             const float* dz, const float* dvar, const float* dint,
                             const uint32_t* drgb, double rx, double ry) {
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (n && (!dx || !dy || !dz)) return fail(FDM_ERR_INVALID, "null xyz");
  HIPCK(hipSetDevice(e->device));
  ScanParams P;
  fill_update_params(e, P, rx, ry, false);
  return enqueue_scan(e, P, n, dx, dy, dz, dint, drgb, dvar);
}

int fdm_engine_update(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                      const float* z_var, const float* intensity, const uint32_t* rgb, double rx,
                      double ry, fdm_scan_stats* out) {
  if (e) { if (int rc_join = join_streams(e)) return rc_join; }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (n && (!x || !y || !z)) return fail(FDM_ERR_INVALID, "null xyz");
  HIPCK(hipSetDevice(e->device));
  const float *dx = nullptr, *dy = nullptr, *dz = nullptr, *da = nullptr, *dv = nullptr;
  const uint32_t* dc = nullptr;
  int rc;
  if (n && (rc = stage_inputs(e, n, x, y, z, intensity, rgb, z_var, &dx, &dy, &dz, &da, &dc, &dv)))
    return rc;
  ScanParams P;
  fill_update_params(e, P, rx, ry, false);
  if ((rc = enqueue_scan(e, P, n, dx, dy, dz, da, dc, dv))) return rc;
  int status = FDM_OK;
  if ((rc = read_stats(e, out, &status))) return rc;
  return FDM_OK;
}

int fdm_engine_flush(fdm_engine* e) {
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  return join_streams(e);
}

void* fdm_engine_stream(fdm_engine* e) { return e ? static_cast<void*>(e->stream) : nullptr; }

int fdm_engine_sync(fdm_engine* e) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc_sync = sync_all(e)) return rc_sync;
  return FDM_OK;
}

int fdm_engine_last_stats(fdm_engine* e, fdm_scan_stats* out) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (e->have_scan && e->last_was_integrate && e->last_n == 0) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (out) std::memset(out, 0, sizeof(*out));
    return FDM_SKIP_EMPTY_CLOUD;
  }
  int status = FDM_OK;
  const int rc = read_stats(e, out, &status);
  return rc ? rc : status;
}

int fdm_engine_move(fdm_engine* e, double x, double y) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (e->G.s_rows != e->G.rows || e->G.s_cols != e->G.cols)
    return fail(FDM_ERR_INVALID, "move() is not defined for tiled engines");
  HIPCK(hipSetDevice(e->device));
  ScanParams P;
  fill_update_params(e, P, x, y, true);
  return enqueue_scan(e, P, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int fdm_engine_get_geometry(fdm_engine* e, fdm_geometry* out) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !out) return fail(FDM_ERR_INVALID, "null argument");
  if (int rc_sync = sync_all(e)) return rc_sync;
  DevGeom g;
  HIPCK(hipMemcpy(&g, &e->d_state->geom[e->scan_no & 3], sizeof(DevGeom), hipMemcpyDeviceToHost));
  out->length_x = e->G.len_x;
  out->length_y = e->G.len_y;
  out->resolution = e->G.res;
  out->position_x = g.px;
  out->position_y = g.py;
  out->rows = e->G.rows;
  out->cols = e->G.cols;
  out->start_row = g.sr;
  out->start_col = g.sc;
  return FDM_OK;
}

int fdm_engine_set_position(fdm_engine* e, double x, double y) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc_sync = sync_all(e)) return rc_sync;
  const double p[2] = {x, y};
  HIPCK(hipMemcpy(&e->d_state->geom[e->scan_no & 3].px, p, sizeof(p), hipMemcpyHostToDevice));
  return FDM_OK;
}

int fdm_engine_set_start_index(fdm_engine* e, int32_t row, int32_t col) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (row < 0 || col < 0 || row >= e->G.rows || col >= e->G.cols)
    return fail(FDM_ERR_INVALID, "start index out of range");
  if ((row || col) && (e->G.s_rows != e->G.rows || e->G.s_cols != e->G.cols))
    return fail(FDM_ERR_INVALID, "tiled engines need start index 0");
  if (int rc_sync = sync_all(e)) return rc_sync;
  const int s[2] = {row, col};
  HIPCK(hipMemcpy(&e->d_state->geom[e->scan_no & 3].sr, s, sizeof(s), hipMemcpyHostToDevice));
  return FDM_OK;
}

int fdm_engine_num_layers(fdm_engine* e) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc = resolve_pending(e)) return rc;
  int n = 0;
  for (auto& l : e->layers) n += l.pending ? 0 : 1;
  return n;
}

const char* fdm_engine_layer_name(fdm_engine* e, int i) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return nullptr; } }
  if (!e) return nullptr;
  if (resolve_pending(e)) return nullptr;
  int k = 0;
  for (auto& l : e->layers) {
    if (l.pending) continue;
    if (k++ == i) return l.name.c_str();
  }
  return nullptr;
}

int fdm_engine_layer_exists(fdm_engine* e, const char* name) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !name) return fail(FDM_ERR_INVALID, "null argument");
  if (int rc = resolve_pending(e)) return rc;
  Layer* l = find_layer(e, name);
  return (l && !l->pending) ? 1 : 0;
}

int fdm_engine_layer_add(fdm_engine* e, const char* name, float value) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !name) return fail(FDM_ERR_INVALID, "null argument");
  HIPCK(hipSetDevice(e->device));
  if (std::strcmp(name, "obstacle") == 0) e->obst_dense_pending = true;
  if (int rc = resolve_pending(e)) return rc;  // keeps getLayers() in the reference's creation order
  return add_layer(e, name, value, false);
}

int fdm_engine_layer_download(fdm_engine* e, const char* name, float* host, int32_t rows, int32_t cols) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !name || !host) return fail(FDM_ERR_INVALID, "null argument");
  if (rows != e->G.s_rows || cols != e->G.s_cols) return fail(FDM_ERR_INVALID, "shape mismatch");
  if (int rc = resolve_pending(e)) return rc;
  Layer* l = find_layer(e, name);
  if (!l || l->pending) return fail(FDM_ERR_NO_LAYER, std::string("no layer ") + name);
  const float* src = l->d;
  if (l->field >= 0) {  // record field: gather into a contiguous staging array first
    if (int rc = ensure_tmp(e)) return rc;
    if (int rc = copy_strided(e, e->d_tmp, 1, lptr(e, *l), lstride(e, *l))) return rc;
    src = e->d_tmp;
  }
  HIPCK(hipMemcpyAsync(host, src, e->ncell * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;
  return FDM_OK;
}

int fdm_engine_layer_upload(fdm_engine* e, const char* name, const float* host, int32_t rows, int32_t cols) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !name || !host) return fail(FDM_ERR_INVALID, "null argument");
  if (rows != e->G.s_rows || cols != e->G.s_cols) return fail(FDM_ERR_INVALID, "shape mismatch");
  HIPCK(hipSetDevice(e->device));
  Layer* l = find_layer(e, name);
  if (!l) {
    if (int rc = add_layer(e, name, NAN, false)) return rc;
    l = find_layer(e, name);
  }
  l->pending = false;
  if (std::strcmp(name, "obstacle") == 0) e->obst_dense_pending = true;
  if (l->field >= 0) {
    if (int rc = ensure_tmp(e)) return rc;
    HIPCK(hipMemcpyAsync(e->d_tmp, host, e->ncell * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (int rc = copy_strided(e, lptr(e, *l), lstride(e, *l), e->d_tmp, 1)) return rc;
  } else {
    HIPCK(hipMemcpyAsync(l->d, host, e->ncell * sizeof(float), hipMemcpyHostToDevice, e->stream));
  }
  if (int rc_sync = sync_all(e)) return rc_sync;
  return FDM_OK;
}

float* fdm_engine_layer_device_ptr(fdm_engine* e, const char* name) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return nullptr; } }
  if (!e || !name) return nullptr;
  Layer* l = find_layer(e, name);
  return (l && l->field < 0) ? l->d : nullptr;  // record fields have no contiguous array
}

int fdm_engine_clear(fdm_engine* e, const char* name) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  HIPCK(hipSetDevice(e->device));
  if (name) {
    Layer* l = find_layer(e, name);
    if (!l || l->pending) return fail(FDM_ERR_NO_LAYER, std::string("no layer ") + name);
    return fill_async(e, lptr(e, *l), NAN, e->ncell, lstride(e, *l));
  }
  for (auto& l : e->layers)
    if (int rc = fill_async(e, lptr(e, l), NAN, e->ncell, lstride(e, l))) return rc;
  return FDM_OK;
}

static int region_copy(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                       const char* const* names, int n_layers, float* d_buf, int to_buf) {
  if (!e || !names || !d_buf) return fail(FDM_ERR_INVALID, "null argument");
  if (nr <= 0 || nc <= 0 || r0 < 0 || c0 < 0 || r0 + nr > e->G.s_rows || c0 + nc > e->G.s_cols)
    return fail(FDM_ERR_INVALID, "region outside the stored window");
  HIPCK(hipSetDevice(e->device));
  for (int k = 0; k < n_layers; ++k) {
    Layer* l = find_layer(e, names[k]);
    if (!l) return fail(FDM_ERR_NO_LAYER, std::string("no layer ") + names[k]);
    hipLaunchKernelGGL(k_region_copy, dim3((nr + 255) / 256, nc), dim3(256), 0, e->stream, lptr(e, *l),
                       lstride(e, *l), d_buf + size_t(k) * nr * nc, e->G.s_rows, r0, c0, nr, nc, to_buf);
    HIPCK(hipGetLastError());
  }
  return FDM_OK;
}

int fdm_engine_region_pack(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                           const char* const* names, int n_layers, float* d_buf) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  return region_copy(e, r0, c0, nr, nc, names, n_layers, d_buf, 1);
}
int fdm_engine_region_unpack(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                             const char* const* names, int n_layers, const float* d_buf) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  return region_copy(e, r0, c0, nr, nc, names, n_layers, const_cast<float*>(d_buf), 0);
}

int fdm_engine_capture(fdm_engine* e, int preprocessed, int rasterized) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  e->cap_pre = preprocessed != 0;
  e->cap_ras = rasterized != 0;
  if (e->cap_pre) e->want_ids = true;  // the per-point pass flag rides on the cell-id buffer
  return FDM_OK;
}

int fdm_engine_last_preprocessed(fdm_engine* e, uint64_t cap, float* x, float* y, float* z,
                                 float* sigma_z2, uint64_t* n_out) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !n_out) return fail(FDM_ERR_INVALID, "null argument");
  *n_out = 0;
  if (!e->cap_pre) return fail(FDM_ERR_INVALID, "preprocessed-scan capture is off");
  const size_t n = e->last_n;
  if (!e->have_scan || n == 0 || !e->d_cap || !e->d_cell_ids) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  std::vector<float> h(4 * n);
  std::vector<int32_t> ids(n);
  for (int c = 0; c < 4; ++c)
    HIPCK(hipMemcpy(h.data() + c * n, e->d_cap + c * e->cap_cap, n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(ids.data(), e->d_cell_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  uint64_t w = 0;
  for (size_t i = 0; i < n; ++i) {  // order-preserving compaction = marshalling, like filterInPlace
    if (ids[i] == -1) continue;     // dropped by cropRange / cropZ
    if (w < cap) {
      if (x) x[w] = h[i];
      if (y) y[w] = h[n + i];
      if (z) z[w] = h[2 * n + i];
      if (sigma_z2) sigma_z2[w] = h[3 * n + i];
    }
    ++w;
  }
  *n_out = w;
  return FDM_OK;
}

int fdm_engine_last_rasterized(fdm_engine* e, uint64_t cap, float* x, float* y, float* z,
                               uint64_t* n_out) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !n_out) return fail(FDM_ERR_INVALID, "null argument");
  *n_out = 0;
  if (!e->cap_ras) return fail(FDM_ERR_INVALID, "rasterized-scan capture is off");
  if (!e->have_scan || !e->d_ras) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  std::vector<float> h(e->ncell);
  HIPCK(hipMemcpy(h.data(), e->d_ras, e->ncell * sizeof(float), hipMemcpyDeviceToHost));
  fdm_geometry g;
  if (int rc = fdm_engine_get_geometry(e, &g)) return rc;
  uint64_t w = 0;
  const GeomConst& G = e->G;
  for (size_t o = 0; o < e->ncell; ++o) {
    if (std::isnan(h[o])) continue;
    if (w < cap) {
      const int r = int(o % size_t(G.s_rows)) + G.s_r0, c = int(o / size_t(G.s_rows)) + G.s_c0;
      int ur = r - g.start_row, uc = c - g.start_col;  // getPositionFromIndex (grid_map_core)
      if (ur < 0) ur += G.rows;
      if (uc < 0) uc += G.cols;
      const double px = g.position_x + (0.5 * G.len_x - 0.5 * G.res) + G.res * double(-ur);
      const double py = g.position_y + (0.5 * G.len_y - 0.5 * G.res) + G.res * double(-uc);
      if (x) x[w] = float(px);
      if (y) y[w] = float(py);
      if (z) z[w] = h[o];
    }
    ++w;
  }
  *n_out = w;
  return FDM_OK;
}

int fdm_engine_enable_cell_ids(fdm_engine* e, int on) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  e->want_ids = on != 0;
  return FDM_OK;
}

int fdm_engine_last_cell_ids(fdm_engine* e, int32_t* host_out, uint64_t n) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !host_out) return fail(FDM_ERR_INVALID, "null argument");
  if (!e->want_ids || !e->d_cell_ids || n != e->last_n)
    return fail(FDM_ERR_INVALID, "cell ids not recorded for the last scan");
  HIPCK(hipMemcpyAsync(host_out, e->d_cell_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;
  return FDM_OK;
}

int fdm_engine_enable_profile(fdm_engine* e, int on) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  e->profile = on != 0;
  return FDM_OK;
}

// ---- stencil post-processing ----
namespace {
// Spatial tiles: a stencil that reaches `need` cells is exact on the owned cells iff every window side
// that is not a map side carries a halo at least that wide.
int check_halo(const fdm_engine* e, int need) {
  const GeomConst& G = e->G;
  const int top = G.o_r0 - G.s_r0, left = G.o_c0 - G.s_c0;
  const int bottom = (G.s_r0 + G.s_rows) - (G.o_r0 + G.o_rows), right = (G.s_c0 + G.s_cols) - (G.o_c0 + G.o_cols);
  const bool ok = (G.s_r0 == 0 || top >= need) && (G.s_c0 == 0 || left >= need) &&
                  (G.s_r0 + G.s_rows == G.rows || bottom >= need) && (G.s_c0 + G.s_cols == G.cols || right >= need);
  if (!ok) return fail(FDM_ERR_INVALID, "tile halo narrower than the stencil (" + std::to_string(need) + " cells needed)");
  return FDM_OK;
}
// neighbourhood offsets, dr-major / dc-minor; box = region(Size(k,k)), disc = region(radius)
int upload_region(fdm_engine* e, const std::vector<RegionEntry>& reg) {
  if (reg.size() > size_t(kMaxRegion)) return fail(FDM_ERR_INVALID, "neighbourhood larger than 256 cells");
  if (!e->d_region) HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_region), kMaxRegion * sizeof(RegionEntry)));
  HIPCK(hipMemcpyAsync(e->d_region, reg.data(), reg.size() * sizeof(RegionEntry), hipMemcpyHostToDevice, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;  // `reg` is a host temporary
  return FDM_OK;
}
void region_disc(const fdm_engine* e, float radius, std::vector<RegionEntry>& reg) {
  reg.clear();
  const float res = static_cast<float>(e->G.res);
  const int k = static_cast<int>(std::floor(radius / res + 1e-4f));
  const float r2 = radius * radius;
  for (int dr = -k; dr <= k; ++dr)
    for (int dc = -k; dc <= k; ++dc) {
      const float d2 = static_cast<float>(dr * dr + dc * dc) * (res * res);
      if (d2 <= r2 * (1.0f + 1e-5f)) reg.push_back({dr, dc, d2, 0.f});
    }
}
unsigned cell_blocks(const fdm_engine* e) { return unsigned((e->ncell + 255) / 256); }
int ensure_tmp2(fdm_engine* e) {
  if (!e->d_tmp2) HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_tmp2), e->ncell * sizeof(float)));
  return FDM_OK;
}
}  // namespace

int fdm_engine_apply_inpainting(fdm_engine* e, int max_iterations, int min_valid, int inplace) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  HIPCK(hipSetDevice(e->device));
  int rc;
  if ((rc = check_halo(e, max_iterations > 0 ? max_iterations : 0))) return rc;  // one cell per pass
  if ((rc = resolve_pending(e))) return rc;
  Layer* elev = find_layer(e, "elevation");
  if (!elev) return fail(FDM_ERR_NO_LAYER, "no layer elevation");
  const char* out_name = inplace ? "elevation" : "elevation_inpainted";
  if (!find_layer(e, out_name) && (rc = add_layer(e, out_name, NAN, false))) return rc;
  elev = find_layer(e, "elevation");
  Layer* out = find_layer(e, out_name);
  if ((rc = ensure_tmp(e))) return rc;
  float* A = lptr(e, *out);
  const int As = lstride(e, *out);
  float* B = e->d_tmp;
  const int slot = int(e->scan_no & 3);
  // `inpainted = elevation`, then up to max_iterations passes ping-ponging output layer <-> staging;
  // the reference stops after a pass that changed nothing — further passes are identities, so all
  // of them are simply run.  The copy goes to whichever side makes the LAST pass land in the layer.
  const int iters = max_iterations > 0 ? max_iterations : 0;
  const bool start_in_layer = (iters % 2) == 0;
  if (!inplace || !start_in_layer) {
    if ((rc = copy_strided(e, start_in_layer ? A : B, start_in_layer ? As : 1, lptr(e, *elev), lstride(e, *elev))))
      return rc;
  }
  bool in_layer = start_in_layer;
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL(k_inpaint_pass, dim3(cell_blocks(e)), dim3(256), 0, e->stream, e->G, e->d_state, slot,
                       in_layer ? A : B, in_layer ? As : 1, in_layer ? B : A, in_layer ? 1 : As, min_valid,
                       unsigned(e->ncell));
    in_layer = !in_layer;
  }
  HIPCK(hipGetLastError());
  return FDM_OK;
}

int fdm_engine_apply_spatial_smoothing(fdm_engine* e, const char* layer, int kernel_size, int min_valid) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !layer) return fail(FDM_ERR_INVALID, "null argument");
  if (kernel_size < 1 || kernel_size > 15 || (kernel_size & 1) == 0)
    return fail(FDM_ERR_INVALID, "kernel_size must be odd and in [1, 15]");
  HIPCK(hipSetDevice(e->device));
  int rc;
  if ((rc = check_halo(e, kernel_size / 2))) return rc;
  if ((rc = resolve_pending(e))) return rc;
  Layer* l = find_layer(e, layer);
  if (!l || l->pending) return FDM_OK;  // spatial_smoothing.hpp:42
  if ((rc = ensure_tmp(e))) return rc;
  if ((rc = copy_strided(e, e->d_tmp, 1, lptr(e, *l), lstride(e, *l)))) return rc;  // the double buffer
  hipLaunchKernelGGL(k_median, dim3(cell_blocks(e)), dim3(256), 0, e->stream, e->G, e->d_state,
                     int(e->scan_no & 3), e->d_tmp, lptr(e, *l), lstride(e, *l), kernel_size, min_valid,
                     unsigned(e->ncell));
  HIPCK(hipGetLastError());
  if (std::strcmp(layer, "obstacle") == 0) e->obst_dense_pending = true;
  return FDM_OK;
}

int fdm_engine_apply_uncertainty_fusion(fdm_engine* e, const fdm_fusion_config* cfg) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !cfg) return fail(FDM_ERR_INVALID, "null argument");
  if (!cfg->enabled) return FDM_OK;
  HIPCK(hipSetDevice(e->device));
  int rc;
  if ((rc = check_halo(e, int(std::floor(cfg->search_radius / static_cast<float>(e->G.res) + 1e-4f))))) return rc;
  if ((rc = resolve_pending(e))) return rc;
  Layer* up = find_layer(e, "upper_bound");
  Layer* lo = find_layer(e, "lower_bound");
  if (!up || !lo) return FDM_OK;  // uncertainty_fusion.cpp:108-113: warn + return
  std::vector<RegionEntry> reg;
  region_disc(e, cfg->search_radius, reg);
  {  // spatial weight of each offset (uncertainty_fusion.cpp:122-123,158): std::exp on a float
    const float inv_2s2 = 1.0f / (2.0f * cfg->spatial_sigma * cfg->spatial_sigma);
    for (auto& r : reg) r.w = std::exp(-r.dist_sq * inv_2s2);
  }
  if ((rc = upload_region(e, reg))) return rc;
  if ((rc = ensure_tmp(e)) || (rc = ensure_tmp2(e))) return rc;
  if ((rc = copy_strided(e, e->d_tmp, 1, lptr(e, *up), lstride(e, *up)))) return rc;
  if ((rc = copy_strided(e, e->d_tmp2, 1, lptr(e, *lo), lstride(e, *lo)))) return rc;
  FusionParams F{};
  F.inv_2s2 = 1.0f / (2.0f * cfg->spatial_sigma * cfg->spatial_sigma);
  F.q_lower = cfg->quantile_lower;
  F.q_upper = cfg->quantile_upper;
  F.min_valid = cfg->min_valid_neighbors;
  F.n_entries = int(reg.size());
  const unsigned fblocks = unsigned((e->ncell + kFusionThreads - 1) / kFusionThreads);
  if (int(reg.size()) <= kFusionLdsEntries) {  // sample lists in LDS
    const size_t lds = size_t(4) * reg.size() * kFusionThreads * sizeof(float);
    static bool raised = false;
    if (!raised) {
      HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fusion<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(size_t(4) * kFusionLdsEntries * kFusionThreads * sizeof(float))));
      raised = true;
    }
    hipLaunchKernelGGL(k_fusion<true>, dim3(fblocks), dim3(kFusionThreads), lds, e->stream, e->G, e->d_state,
                       int(e->scan_no & 3), e->d_region, F, e->d_tmp, e->d_tmp2, lptr(e, *up), lstride(e, *up),
                       lptr(e, *lo), lstride(e, *lo), unsigned(e->ncell));
  } else {
    hipLaunchKernelGGL(k_fusion<false>, dim3(fblocks), dim3(kFusionThreads), 0, e->stream, e->G, e->d_state,
                       int(e->scan_no & 3), e->d_region, F, e->d_tmp, e->d_tmp2, lptr(e, *up), lstride(e, *up),
                       lptr(e, *lo), lstride(e, *lo), unsigned(e->ncell));
  }
  HIPCK(hipGetLastError());
  return FDM_OK;
}

int fdm_engine_apply_feature_extraction(fdm_engine* e, float radius, int min_valid, float lo_pct, float hi_pct) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  HIPCK(hipSetDevice(e->device));
  int rc;
  if ((rc = check_halo(e, int(std::floor(radius / static_cast<float>(e->G.res) + 1e-4f))))) return rc;
  if ((rc = resolve_pending(e))) return rc;
  if (!find_layer(e, "elevation")) return FDM_OK;  // feature_extraction.cpp:33
  const char* names[7] = {"step", "slope", "roughness", "curvature", "_normal_x", "_normal_y", "_normal_z"};
  for (const char* n : names)
    if (!find_layer(e, n) && (rc = add_layer(e, n, NAN, false))) return rc;
  std::vector<RegionEntry> reg;
  region_disc(e, radius, reg);
  if ((rc = upload_region(e, reg))) return rc;
  Layer* elev = find_layer(e, "elevation");
  FeatureParams F{};
  F.resf = static_cast<float>(e->G.res);
  F.lo_pct = lo_pct;
  F.hi_pct = hi_pct;
  F.min_valid = min_valid;
  F.n_entries = int(reg.size());
  FeatureOut O{};
  float** outs[7] = {&O.step, &O.slope, &O.roughness, &O.curvature, &O.nx, &O.ny, &O.nz};
  for (int k = 0; k < 7; ++k) *outs[k] = find_layer(e, names[k])->d;
  // order statistics needed by `step`: index lo from the bottom, (count-1-hi) from the top; both grow
  // with count, so the full region bounds them
  const int nmax = int(reg.size());
  const int need_lo = nmax > 0 ? static_cast<int>(lo_pct * float(nmax - 1)) + 1 : 1;
  const int need_hi = nmax > 0 ? (nmax - 1) - static_cast<int>(hi_pct * float(nmax - 1)) + 1 : 1;
  const bool pct_ok = lo_pct >= 0.0f && hi_pct <= 1.0f && lo_pct <= 1.0f && hi_pct >= 0.0f;
  auto launch_feat = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(cell_blocks(e)), dim3(256), 0, e->stream, e->G, e->d_state, int(e->scan_no & 3),
                       e->d_region, F, lptr(e, *elev), lstride(e, *elev), O, unsigned(e->ncell));
  };
  if (pct_ok && need_lo <= 16 && need_hi <= 16) launch_feat(k_features<16>);
  else launch_feat(k_features<0>);
  HIPCK(hipGetLastError());
  return FDM_OK;
}

// ---- ingest ----
int fdm_engine_ingest_cloud2(fdm_engine* e, const void* data, int on_device, uint64_t n_points,
                             const fdm_cloud2_layout* lay, uint64_t* n_valid) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !lay) return fail(FDM_ERR_INVALID, "null argument");
  if (n_valid) *n_valid = 0;
  e->in_n = 0;
  e->in_has_int = e->in_has_rgb = false;
  if (n_points == 0) return FDM_OK;                                        // impl.hpp:178-181
  if (lay->off_x < 0 || lay->off_y < 0 || lay->off_z < 0) return FDM_OK;   // impl.hpp:183-186: no xyz
  if (!data) return fail(FDM_ERR_INVALID, "null data");
  if (n_points >= 0xFFFFFFFEull) return fail(FDM_ERR_INVALID, "point count exceeds 2^32-2");
  const uint32_t step = lay->point_step;
  auto fits = [&](int32_t off, uint32_t len) { return off < 0 || uint64_t(off) + len <= step; };
  const uint32_t ilen = lay->intensity_type == 8 ? 8 : (lay->intensity_type == 7 ? 4 : (lay->intensity_type == 4 ? 2 : 1));
  if (step == 0 || !fits(lay->off_x, 4) || !fits(lay->off_y, 4) || !fits(lay->off_z, 4) ||
      !fits(lay->off_intensity, ilen) || !fits(lay->off_rgb, 4))
    return fail(FDM_ERR_INVALID, "field offset outside the point record");
  HIPCK(hipSetDevice(e->device));
  const size_t bytes = size_t(n_points) * step;
  const uint8_t* blob = static_cast<const uint8_t*>(data);
  if (!on_device) {
    if (bytes > e->blob_cap) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      if (e->d_blob) HIPCK(hipFree(e->d_blob));
      e->blob_cap = bytes + bytes / 4 + 4096;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_blob), e->blob_cap));
    }
    HIPCK(hipMemcpyAsync(e->d_blob, data, bytes, hipMemcpyHostToDevice, e->stream));
    blob = e->d_blob;
  }
  if (n_points > e->in_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->d_in) HIPCK(hipFree(e->d_in));
    e->in_cap = ((n_points + n_points / 4 + 1024) + 3) & ~size_t(3);  // channels stay 16-byte aligned
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_in), e->in_cap * 5 * sizeof(float)));
  }
  IngestLayout L{};
  L.point_step = step;
  L.off_x = lay->off_x; L.off_y = lay->off_y; L.off_z = lay->off_z;
  L.off_intensity = lay->off_intensity; L.intensity_type = lay->intensity_type;
  L.off_rgb = lay->off_rgb;
  auto al4 = [](int32_t off) { return off < 0 || (off & 3) == 0; };
  L.aligned = (reinterpret_cast<uintptr_t>(blob) & 3u) == 0 && (step & 3u) == 0 && al4(L.off_x) && al4(L.off_y) &&
              al4(L.off_z) && al4(L.off_rgb) && (L.intensity_type < 7 || al4(L.off_intensity));
  const unsigned blocks = unsigned((n_points + 255) / 256);
  if (size_t(blocks) + 1 > e->pack_counts_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->pack_counts) HIPCK(hipFree(e->pack_counts));
    e->pack_counts_cap = size_t(blocks) + 1 + 1024;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->pack_counts), e->pack_counts_cap * sizeof(uint32_t)));
  }
  const bool hi = lay->off_intensity >= 0, hc = lay->off_rgb >= 0;
  hipLaunchKernelGGL(k_ingest_count, dim3(blocks), dim3(256), 0, e->stream, blob, L, n_points, e->pack_counts);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, e->stream, e->pack_counts, blocks);
  hipLaunchKernelGGL(k_ingest_write, dim3(blocks), dim3(256), 0, e->stream, blob, L, n_points, e->pack_counts,
                     e->d_in, e->d_in + e->in_cap, e->d_in + 2 * e->in_cap,
                     hi ? e->d_in + 3 * e->in_cap : static_cast<float*>(nullptr),
                     hc ? reinterpret_cast<uint32_t*>(e->d_in + 4 * e->in_cap) : static_cast<uint32_t*>(nullptr));
  HIPCK(hipGetLastError());
  uint32_t total = 0;
  HIPCK(hipMemcpyAsync(&total, e->pack_counts + blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;
  e->in_n = total;
  e->in_has_int = hi;
  e->in_has_rgb = hc;
  if (n_valid) *n_valid = total;
  return FDM_OK;
}

int fdm_engine_ingested(fdm_engine* e, const float** dx, const float** dy, const float** dz,
                        const float** dint, const uint32_t** drgb, uint64_t* n) {
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (dx) *dx = e->d_in;
  if (dy) *dy = e->d_in ? e->d_in + e->in_cap : nullptr;
  if (dz) *dz = e->d_in ? e->d_in + 2 * e->in_cap : nullptr;
  if (dint) *dint = e->in_has_int ? e->d_in + 3 * e->in_cap : nullptr;
  if (drgb) *drgb = e->in_has_rgb ? reinterpret_cast<const uint32_t*>(e->d_in + 4 * e->in_cap) : nullptr;
  if (n) *n = e->in_n;
  return FDM_OK;
}

int fdm_engine_integrate_cloud2(fdm_engine* e, const void* data, int on_device, uint64_t n_points,
                                const fdm_cloud2_layout* lay, const double Tbs[16], const double Twb[16],
                                fdm_scan_stats* out) {
  if (e) { if (int rc_join = join_streams(e)) return rc_join; }
  if (!e || !lay || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  uint64_t n = 0;
  int rc = fdm_engine_ingest_cloud2(e, data, on_device, n_points, lay, &n);
  if (rc) return rc;
  if (n == 0) {  // fastdem.cpp:125-128
    if (out) std::memset(out, 0, sizeof(*out));
    return FDM_SKIP_EMPTY_CLOUD;
  }
  const float *dx, *dy, *dz, *di;
  const uint32_t* dc;
  fdm_engine_ingested(e, &dx, &dy, &dz, &di, &dc, nullptr);
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  if ((rc = enqueue_scan(e, P, n, dx, dy, dz, di, dc, nullptr))) return rc;
  int status = FDM_OK;
  if ((rc = read_stats(e, out, &status))) return rc;
  return status;
}

// ---- map egress ----
namespace {
struct PackPlan {
  PackParams Q{};
  PackLayers L{};
  std::vector<std::string> fields;
  unsigned long long total = 0;
  unsigned blocks = 0;
};

int plan_pack(fdm_engine* e, const char* elevation_layer, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
              PackPlan& pl) {
  if (int rc = resolve_pending(e)) return rc;
  Layer* elev = find_layer(e, elevation_layer);
  if (!elev || elev->pending) return fail(FDM_ERR_NO_LAYER, std::string("no layer ") + elevation_layer);
  if (nr >= 0) {
    if (r0 < 0 || c0 < 0 || r0 >= e->G.rows || c0 >= e->G.cols || nr > e->G.rows || nc < 0 || nc > e->G.cols)
      return fail(FDM_ERR_INVALID, "submap outside the buffer");
  }
  pl.Q.sub_r0 = r0; pl.Q.sub_c0 = c0; pl.Q.sub_rows = nr; pl.Q.sub_cols = nc;
  pl.Q.slot = int(e->scan_no & 3);
  pl.L.elev = lptr(e, *elev);
  pl.L.elev_stride = lstride(e, *elev);
  pl.fields = {"x", "y", "z"};
  int nf = 0;
  const Layer* color = nullptr;
  for (auto& l : e->layers) {  // impl.hpp:66-77
    if (l.pending) continue;
    if (!l.name.empty() && l.name[0] == '_') continue;
    if (l.name == elevation_layer) continue;
    if (l.name == "color") { color = &l; continue; }
    if (nf >= kPackMaxFields) return fail(FDM_ERR_INVALID, "too many layers to pack");
    pl.L.ptr[nf] = lptr(e, l);
    pl.L.stride[nf] = lstride(e, l);
    pl.fields.push_back(l.name);
    ++nf;
  }
  pl.Q.n_float = nf;
  pl.Q.has_color = color ? 1 : 0;
  pl.L.color = color ? color->d : nullptr;
  if (color) pl.fields.push_back("rgb");
  pl.total = nr < 0 ? (unsigned long long)e->G.rows * e->G.cols : (unsigned long long)nr * nc;
  pl.blocks = unsigned((pl.total + 255) / 256);
  return FDM_OK;
}

// count + scan; returns the number of valid cells (host sync)
int pack_count(fdm_engine* e, const PackPlan& pl, uint64_t* n_points) {
  *n_points = 0;
  if (pl.total == 0) return FDM_OK;
  if (size_t(pl.blocks) + 1 > e->pack_counts_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->pack_counts) HIPCK(hipFree(e->pack_counts));
    e->pack_counts_cap = size_t(pl.blocks) + 1 + 1024;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->pack_counts), e->pack_counts_cap * sizeof(uint32_t)));
  }
  hipLaunchKernelGGL(k_pack_count, dim3(pl.blocks), dim3(256), 0, e->stream, pl.Q, e->G, e->d_state, pl.L,
                     e->pack_counts);
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, e->stream, e->pack_counts, pl.blocks);
  HIPCK(hipGetLastError());
  uint32_t total = 0;
  HIPCK(hipMemcpyAsync(&total, e->pack_counts + pl.blocks, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;
  *n_points = total;
  return FDM_OK;
}

int pack_write(fdm_engine* e, const PackPlan& pl, uint64_t n_points) {
  const size_t need = size_t(n_points) * pl.fields.size();
  if (need > e->pack_cap) {
    if (e->d_pack) HIPCK(hipFree(e->d_pack));
    e->pack_cap = need + need / 8 + 1024;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_pack), e->pack_cap * sizeof(float)));
  }
  if (n_points == 0) return FDM_OK;
  const size_t lds = 256 * pl.fields.size() * sizeof(float);  // <= 256 * 68 * 4 = 68 KB of the CU's 160 KB
  hipLaunchKernelGGL(k_pack_write, dim3(pl.blocks), dim3(256), lds, e->stream, pl.Q, e->G, e->d_state, pl.L,
                     e->pack_counts, e->d_pack);
  HIPCK(hipGetLastError());
  return FDM_OK;
}

void write_fields(const PackPlan& pl, char* buf, uint64_t cap) {
  if (!buf || !cap) return;
  std::string joined;
  for (size_t k = 0; k < pl.fields.size(); ++k) joined += (k ? "\n" : "") + pl.fields[k];
  std::snprintf(buf, cap, "%s", joined.c_str());
}
}  // namespace

int fdm_engine_pack_cloud_device(fdm_engine* e, const char* elevation_layer, int32_t r0, int32_t c0,
                                 int32_t nr, int32_t nc, void** d_out, uint64_t* n_points,
                                 uint32_t* point_step) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !elevation_layer || !n_points) return fail(FDM_ERR_INVALID, "null argument");
  HIPCK(hipSetDevice(e->device));
  PackPlan pl;
  int rc;
  if ((rc = plan_pack(e, elevation_layer, r0, c0, nr, nc, pl))) return rc;
  if (point_step) *point_step = uint32_t(pl.fields.size() * 4);
  if ((rc = pack_count(e, pl, n_points))) return rc;
  if ((rc = pack_write(e, pl, *n_points))) return rc;
  if (d_out) *d_out = e->d_pack;
  return FDM_OK;
}

int fdm_engine_pack_cloud(fdm_engine* e, const char* elevation_layer, int32_t r0, int32_t c0, int32_t nr,
                          int32_t nc, void* host_out, uint64_t cap_bytes, uint64_t* n_points,
                          uint32_t* point_step, char* fields_buf, uint64_t fields_cap) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !elevation_layer || !n_points) return fail(FDM_ERR_INVALID, "null argument");
  HIPCK(hipSetDevice(e->device));
  PackPlan pl;
  int rc;
  if ((rc = plan_pack(e, elevation_layer, r0, c0, nr, nc, pl))) return rc;
  if (point_step) *point_step = uint32_t(pl.fields.size() * 4);
  write_fields(pl, fields_buf, fields_cap);
  if ((rc = pack_count(e, pl, n_points))) return rc;
  const uint64_t bytes = *n_points * pl.fields.size() * 4;
  if (!host_out || cap_bytes < bytes || bytes == 0) return FDM_OK;
  if ((rc = pack_write(e, pl, *n_points))) return rc;
  HIPCK(hipMemcpyAsync(host_out, e->d_pack, bytes, hipMemcpyDeviceToHost, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;
  return FDM_OK;
}

// ---- raycasting entry points ----
int fdm_engine_apply_raycasting_device(fdm_engine* e, uint64_t n, const float* dx, const float* dy,
                                       const float* dz, const float origin[3],
                                       const fdm_raycast_config* rcfg) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !origin) return fail(FDM_ERR_INVALID, "null argument");
  const fdm_raycast_config c = rcfg ? *rcfg : ray_config_of(e->cfg);
  if (!c.enabled || n == 0) return FDM_OK;  // raycasting.cpp:207-209
  if (!dx || !dy || !dz) return fail(FDM_ERR_INVALID, "null xyz");
  if (n >= 0xFFFFFFFEull) return fail(FDM_ERR_INVALID, "point count exceeds 2^32-2");
  HIPCK(hipSetDevice(e->device));
  int rc;
  if (!find_layer(e, "elevation")) return FDM_OK;
  if ((rc = ensure_ray_layers(e))) return rc;
  if ((rc = refresh_layer_ptrs(e))) return rc;
  const RayParams Q = make_ray_params(e, c, origin, unsigned(n), int(e->scan_no & 3), -1);
  return enqueue_ray_stage(e, Q, false, dx, dy, dz);
}

int fdm_engine_apply_raycasting(fdm_engine* e, uint64_t n, const float* x, const float* y,
                                const float* z, const float origin[3], const fdm_raycast_config* rcfg) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !origin) return fail(FDM_ERR_INVALID, "null argument");
  if (!(rcfg ? rcfg->enabled : e->cfg.raycast_enabled) || n == 0) return FDM_OK;
  if (!x || !y || !z) return fail(FDM_ERR_INVALID, "null xyz");
  HIPCK(hipSetDevice(e->device));
  const float *dx, *dy, *dz, *da, *dv;
  const uint32_t* dc;
  int rc = stage_inputs(e, n, x, y, z, nullptr, nullptr, nullptr, &dx, &dy, &dz, &da, &dc, &dv);
  if (rc) return rc;
  if ((rc = fdm_engine_apply_raycasting_device(e, n, dx, dy, dz, origin, rcfg))) return rc;
  if (int rc_sync = sync_all(e)) return rc_sync;
  return FDM_OK;
}

int fdm_engine_voxel_any(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                         float voxel_size, uint32_t* out_idx, uint64_t* n_out) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !n_out) return fail(FDM_ERR_INVALID, "null argument");
  *n_out = 0;
  if (!voxel_size_ok(voxel_size)) return fail(FDM_ERR_INVALID, "voxel_size must be in [0.001, 100]");
  if (n == 0) return FDM_OK;
  if (!x || !y || !z || !out_idx) return fail(FDM_ERR_INVALID, "null argument");
  if (n >= 0xFFFFFFFEull) return fail(FDM_ERR_INVALID, "point count exceeds 2^32-2");
  HIPCK(hipSetDevice(e->device));
  const float *dx, *dy, *dz, *da, *dv;
  const uint32_t* dc;
  int rc = stage_inputs(e, n, x, y, z, nullptr, nullptr, nullptr, &dx, &dy, &dz, &da, &dc, &dv);
  if (rc) return rc;
  bool compact = false;
  if ((rc = enqueue_voxel_sort(e, unsigned(n), voxel_size, -1, dx, dy, dz, nullptr, &compact))) return rc;
  hipLaunchKernelGGL(k_voxel_select, dim3(unsigned((n + 255) / 256)), dim3(256), 0, e->stream, unsigned(n),
                     e->vkeys[1], e->vidx[1], e->vsel);
  HIPCK(hipGetLastError());
  std::vector<uint32_t> h(n);
  HIPCK(hipMemcpyAsync(h.data(), e->vsel, n * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  if (int rc_sync = sync_all(e)) return rc_sync;
  uint64_t w = 0;
  for (uint64_t i = 0; i < n; ++i)  // order-preserving compaction = marshalling
    if (h[i] != kNoIdx) out_idx[w++] = h[i];
  *n_out = w;
  return FDM_OK;
}

int fdm_engine_last_ray_ms(fdm_engine* e, float* ms) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !ms) return fail(FDM_ERR_INVALID, "null argument");
  if (!e->profile) return fail(FDM_ERR_INVALID, "profiling is off");
  *ms = 0.f;
  if (!e->ray_timed) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  HIPCK(hipEventElapsedTime(ms, e->ev_ray[0], e->ev_ray[1]));
  return FDM_OK;
}

int fdm_engine_last_kernel_ms(fdm_engine* e, float* ms2) {
  if (!e || !ms2) return fail(FDM_ERR_INVALID, "null argument");
  if (!e->profile) return fail(FDM_ERR_INVALID, "profiling is off");
  HIPCK(hipEventSynchronize(e->ev[3]));  // no flush: a held-back update stays held (the chain is what is timed)
  // an event pair around ONE short kernel also times the gap to the next command; the empty
  // pair (ev2 -> ev3) measures that gap and is subtracted, so the figures agree with rocprofv3
  float raw0 = 0.f, raw1 = 0.f, gap = 0.f;
  HIPCK(hipEventElapsedTime(&raw0, e->ev[0], e->ev[1]));
  HIPCK(hipEventElapsedTime(&raw1, e->ev[1], e->ev[2]));
  HIPCK(hipEventElapsedTime(&gap, e->ev[2], e->ev[3]));
  ms2[0] = raw0 > gap ? raw0 - gap : raw0;
  ms2[1] = e->chain ? 0.0f : (raw1 > gap ? raw1 - gap : raw1);  // held back: it rides with the next launch
  return FDM_OK;
}

/* tuning knob used by bench.py's A/B runs (not part of the reference surface) */
int fdm_engine_set_option(fdm_engine* e, const char* key, int value) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !key) return fail(FDM_ERR_INVALID, "null argument");
  if (std::strcmp(key, "wave_merge") == 0) {
    e->wave_merge = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "bin_table") == 0) {
    e->bin_table = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "overlap") == 0) {
    e->overlap = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_ray") == 0) {
    e->dbg_ray = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "bin_threads") == 0) {
    if (value != 0 && value != 128 && value != 256 && value != 512)
      return fail(FDM_ERR_INVALID, "bin_threads must be 0 (auto), 128, 256 or 512");
    e->bin_threads = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "bin_variant") == 0) {
    if (value != 0 && value != 1 && value != 4) return fail(FDM_ERR_INVALID, "bin_variant must be 0, 1 or 4");
    e->bin_variant = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "records") == 0) {  // cell-record layout (1, default) or one array per layer (0)
    e->use_records = value != 0;
    if (e->estimator_ready) return activate_records(e, e->cfg.estimation_type == 1 ? 1 : 0);
    return FDM_OK;
  }
  if (std::strcmp(key, "dense") == 0) {  // force stamp-gated (0) or dense (1) update sweeps
    e->S.dense = value != 0;
    e->obst_dense_pending = true;  // stamps were not maintained while dense
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_upd") == 0) {
    e->dbg_upd = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_no_atomics") == 0) {  // measurement only: results are wrong when set
    e->dbg_no_atomics = value;
    return FDM_OK;
  }
  return fail(FDM_ERR_INVALID, std::string("unknown option ") + key);
}

}  // extern "C"
