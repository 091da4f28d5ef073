// fdm_engine_ray.inl — host side of the raycasting stage (kernels: fdm_raycast.hpp): voxel sort, ray
// queue, resolve; entry points fdm_engine_apply_raycasting*, fdm_engine_voxel_any, fdm_engine_last_ray_ms.
// The body of fdm_engine_ray.hip (one of the library's three translation units, fdm_engine_host.hpp).

namespace fdmh {

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
  // bucket counts (kept at zero between scans by k_ray_bin_scan) | bucket offsets | per-block sums
  const size_t bin_words = 2u * size_t(kRayBins) + kRayBins / kRayBinBlock;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->ray_bins), bin_words * sizeof(uint32_t)));
  HIPCK(hipMemsetAsync(e->ray_bins, 0, bin_words * sizeof(uint32_t), e->stream));
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
  if (e->ray_blk) HIPCK(hipFree(e->ray_blk));
  if (e->sort_tmp) HIPCK(hipFree(e->sort_tmp));
  e->vcap = n + n / 4 + 1024;
  for (int k = 0; k < 2; ++k) {
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vkeys[k]), e->vcap * sizeof(unsigned long long)));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vidx[k]), e->vcap * sizeof(uint32_t)));
  }
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vsel), e->vcap * sizeof(uint32_t)));
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->ray_blk), (e->vcap / 512u + 2u) * sizeof(uint32_t)));  // (blocks of >= 512 points)
  // the radix sort's histogram: 256 bins x tiles, + the 256 totals (fdm_rsort.hpp)
  const size_t tiles = std::max((e->vcap + kRsTile - 1) / kRsTile, (size_t(kRsSmallMax) + kRsTileSmall - 1) / kRsTileSmall);
  e->sort_tmp_bytes = (256u * tiles + 256u) * sizeof(uint32_t);
  HIPCK(hipMalloc(&e->sort_tmp, e->sort_tmp_bytes));
  return FDM_OK;
}

// Stable sort of the n pairs in (vkeys[src], vidx[src]) by the low `bits` bits of the key; the result lands in
// (vkeys[1], vidx[1]).  `src` must be voxel_sort_source(bits): the buffers alternate once per pass.
int voxel_sort_passes(unsigned bits) { return int((bits + 7u) / 8u); }
int voxel_sort_source(unsigned bits) { return (voxel_sort_passes(bits) & 1) ? 0 : 1; }
template <typename KEY, unsigned TILE>
int enqueue_radix_sort_t(fdm_engine* e, unsigned n, unsigned bits) {
  const unsigned tiles = (n + TILE - 1u) / TILE;
  uint32_t* const hist = static_cast<uint32_t*>(e->sort_tmp);
  uint32_t* const total = hist + size_t(256) * tiles;
  int src = voxel_sort_source(bits);
  for (int pass = 0; pass < voxel_sort_passes(bits); ++pass, src ^= 1) {
    const KEY* kin = reinterpret_cast<const KEY*>(e->vkeys[src]);
    KEY* kout = reinterpret_cast<KEY*>(e->vkeys[src ^ 1]);
    const unsigned shift = unsigned(pass) * 8u;
    // (the first pass's histogram is k_voxel_keys' and its indices are the positions)
    if (pass > 0)
      hipLaunchKernelGGL((k_rs_hist<KEY, TILE>), dim3(tiles), dim3(256), 0, e->stream, n, kin, shift, tiles, hist);
    hipLaunchKernelGGL(k_rs_scan, dim3(256), dim3(256), 0, e->stream, tiles, hist, total);
    if (pass > 0)
      hipLaunchKernelGGL((k_rs_scatter<KEY, true, int(TILE / 256u)>), dim3(tiles), dim3(256), 0, e->stream, n, kin,
                         e->vidx[src], kout, e->vidx[src ^ 1], shift, tiles, hist, total);
    else
      hipLaunchKernelGGL((k_rs_scatter<KEY, false, int(TILE / 256u)>), dim3(tiles), dim3(256), 0, e->stream, n, kin,
                         static_cast<const uint32_t*>(nullptr), kout, e->vidx[src ^ 1], shift, tiles, hist, total);
  }
  HIPCK(hipGetLastError());
  return FDM_OK;
}
template <typename KEY>
int enqueue_radix_sort(fdm_engine* e, unsigned n, unsigned bits) {
  return rs_tile(n) == kRsTileSmall ? enqueue_radix_sort_t<KEY, kRsTileSmall>(e, n, bits)
                                    : enqueue_radix_sort_t<KEY, kRsTile>(e, n, bits);
}
template <typename KEY>
void launch_voxel_keys(fdm_engine* e, unsigned n, float inv, int flag_slot, const VoxelCompact& C, const float* dx,
                       const float* dy, const float* dz, KEY* keys) {
  const unsigned tile = rs_tile(n), tiles = (n + tile - 1u) / tile;
  uint32_t* const hist = static_cast<uint32_t*>(e->sort_tmp);
  if (tile == kRsTileSmall)
    hipLaunchKernelGGL((k_voxel_keys<KEY, kRsTileSmall>), dim3(tiles), dim3(256), 0, e->stream, n, inv, flag_slot, C,
                       e->d_state, dx, dy, dz, keys, e->vsel, tiles, hist);
  else
    hipLaunchKernelGGL((k_voxel_keys<KEY, kRsTile>), dim3(tiles), dim3(256), 0, e->stream, n, inv, flag_slot, C,
                       e->d_state, dx, dy, dz, keys, e->vsel, tiles, hist);
}

// keys -> stable sort: vkeys[1] / vidx[1] hold the voxel-ordered scan afterwards.
// `box` (nullable): centre (3) + half extent [m] of a box that holds every finite point of the cloud, then the
// map-frame z interval [lo, hi] they lie in (NaN, NaN if unknown);
// with it the compact 32-bit key is used when 3 * bits <= 31.  *key_mode tells what the buffers hold:
// 0 = sorted uint64 keys, 1 = sorted uint32 compact keys, 2 = points grouped by key bucket, unsorted (k_vs_*).
// The compact voxel key of a scan whose points lie in `box` (see enqueue_voxel_sort): bits == 0 if the box is unknown
// or too large for it.
VoxelCompact voxel_compact_of(float voxel_size, const double* box) {
  const float inv = 1.0f / voxel_size;  // voxel_grid_impl.hpp:46
  VoxelCompact C{0, 0, 0, 0, 0};
  if (box && std::isfinite(box[3]) && box[3] > 0.0 && box[3] * double(inv) < 4.0e6) {
    const double half = box[3] + 2.0 * double(voxel_size);  // 2-cell margin for the float transforms
    const int span = int(std::ceil(2.0 * half * double(inv))) + 4;
    int bits = 1;
    while ((1 << bits) < span) ++bits;
    if (3 * bits <= 62 && bits <= 21) {
      C.bits = C.zbits = bits;
      C.x0 = int(std::floor((box[0] - half) * double(inv))) - 1;
      C.y0 = int(std::floor((box[1] - half) * double(inv))) - 1;
      C.z0 = int(std::floor((box[2] - half) * double(inv))) - 1;
      // box[4..5] (optional): map-frame z interval the crops leave (cropZ slab tilted by T_world_base)
      if (std::isfinite(box[4]) && std::isfinite(box[5]) && box[5] > box[4]) {
        const double zlo = std::max(box[4], box[2] - half) - 2.0 * double(voxel_size);
        const double zhi = std::min(box[5], box[2] + half) + 2.0 * double(voxel_size);
        const int zspan = int(std::ceil((zhi - zlo) * double(inv))) + 4;
        int zb = 1;
        while ((1 << zb) < zspan) ++zb;
        if (zb < bits) {
          C.zbits = zb;
          C.z0 = int(std::floor(zlo * double(inv))) - 1;
        }
      }
    }
  }
  return C;
}

// The box an integrate() scan's preprocessed points lie in: cropRange keeps d^2 <= range_max^2 around the BASE origin,
// i.e. around T_world_base's translation, and cropZ keeps z_base in [z_min, z_max]: z_map = R20 x + R21 y + R22 z + t_z
// with |R20 x + R21 y| <= hypot(R20, R21) * range_max (column-major Twb: R2j = Twb[4 j + 2])
void ray_box_of(const fdm_engine* e, const ScanParams& P, double box[6]) {
  double zlo = NAN, zhi = NAN;
  if (std::isfinite(double(e->cfg.z_min)) && std::isfinite(double(e->cfg.z_max)) &&
      std::fabs(double(e->cfg.z_min)) < 1e6 && std::fabs(double(e->cfg.z_max)) < 1e6 &&
      std::isfinite(double(e->cfg.range_max)) && double(e->cfg.range_max) < 1e6) {
    const double r20 = double(P.Twb[2]), r21 = double(P.Twb[6]), r22 = double(P.Twb[10]);
    const double tilt = std::hypot(r20, r21) * double(e->cfg.range_max);
    const double a = r22 * double(e->cfg.z_min), b = r22 * double(e->cfg.z_max);
    zlo = P.base_z + std::min(a, b) - tilt;
    zhi = P.base_z + std::max(a, b) + tilt;
  }
  box[0] = P.base_x; box[1] = P.base_y; box[2] = P.base_z; box[3] = double(e->cfg.range_max); box[4] = zlo; box[5] = zhi;
}

int enqueue_voxel_sort(fdm_engine* e, unsigned n, float voxel_size, int flag_slot, const float* dx,
                       const float* dy, const float* dz, const double* box, int* key_mode) {
  if (int rc = ensure_voxel_buffers(e, n)) return rc;
  const float inv = 1.0f / voxel_size;  // voxel_grid_impl.hpp:46
  const VoxelCompact C = voxel_compact_of(voxel_size, box);
  const int key_bits = 2 * C.bits + C.zbits;
  const bool compact = C.bits > 0 && key_bits <= 31;  // true: the sorted buffer holds uint32 keys
  *key_mode = compact ? 1 : 0;
  if (compact && e->voxel_small && n <= unsigned(e->voxel_small_max)) {
    // small scans: no sort at all (k_vs_*: fdm_raycast.hpp).  vkeys[0] = keys by point | places by point, vs_rec =
    // {key, point, bucket start, bucket size} by position; k_vs_mark runs from enqueue_ray_stage (key_mode 2)
    if (!e->vs_cnt) {
      const size_t words = (size_t(1) << kVsFineBits) + kVsCoarse + 1u;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vs_cnt), words * sizeof(uint32_t)));
      HIPCK(hipMemsetAsync(e->vs_cnt, 0, words * sizeof(uint32_t), e->stream));
    }
    VoxelSmall& V = e->vs;
    // fine buckets = one (z, y) row of voxels when that fits 2^18 counters, else the key's top 18 bits
    V.shift = unsigned(std::max(C.bits, key_bits - int(kVsFineBits)));
    V.fine = e->vs_cnt;
    V.coarse = e->vs_cnt + (size_t(1) << kVsFineBits);
    V.total = V.coarse + kVsCoarse;
    uint32_t* k0 = reinterpret_cast<uint32_t*>(e->vkeys[0]);
    V.place = k0 + e->vcap;                                  // (vkeys[0] holds 2 x vcap uint32)
    if (e->vs_rec_cap < n) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      if (e->vs_rec) HIPCK(hipFree(e->vs_rec));
      e->vs_rec = nullptr;
      e->vs_rec_cap = std::max<size_t>(size_t(1) << 16, size_t(n) + n / 4);
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->vs_rec), e->vs_rec_cap * sizeof(uint4)));
    }
    V.rec = e->vs_rec;
    V.cap = unsigned(e->vs_rec_cap);
    V.ibits = 1u;
    V.dbg = (e->dbg_ray >> 8) & 3;
    while ((1u << V.ibits) < n) ++V.ibits;
    const unsigned blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_vs_count, dim3(blocks), dim3(256), 0, e->stream, n, inv, flag_slot, C, V, e->d_state, dx, dy,
                       dz, k0, e->vsel);
    hipLaunchKernelGGL(k_vs_scatter, dim3(blocks), dim3(256), 0, e->stream, n, V, k0);
    HIPCK(hipGetLastError());
    *key_mode = 2;
    return FDM_OK;
  }
  if (compact) {
    // bits 3*bits .. 31 are zero in every valid key and one in the invalid key (all ones): sorting
    // one bit past the fields is enough to keep the dropped points behind every voxel
    const unsigned sort_bits = unsigned(key_bits + 1);
    const int src = voxel_sort_source(sort_bits);
    launch_voxel_keys<uint32_t>(e, n, inv, flag_slot, C, dx, dy, dz, reinterpret_cast<uint32_t*>(e->vkeys[src]));
    HIPCK(hipGetLastError());
    return enqueue_radix_sort<uint32_t>(e, n, sort_bits);
  }
  const unsigned sort_bits = C.bits > 0 ? unsigned(key_bits + 1) : 64u;
  const int src = voxel_sort_source(sort_bits);
  launch_voxel_keys<unsigned long long>(e, n, inv, flag_slot, C, dx, dy, dz, e->vkeys[src]);
  HIPCK(hipGetLastError());
  return enqueue_radix_sort<unsigned long long>(e, n, sort_bits);
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
  Q.by_sector = 0;
  Q.ctx = 0;
  Q.pre_slot = -1;
  Q.pre_do_move = Q.pre_gate = 0;
  return Q;
}

// ---- two raycasting stages in flight (option "ray_overlap", large scans held back with their update) ----
// The stage's buffers live in the engine's members; a second set is swapped in and out around the launches of a stage
// of the other context (bank 1), so that every helper keeps working on "the" members.
void ray_bank_swap(fdm_engine* e) {
  fdm_engine::RayBank& b = e->ray_bank1;
  std::swap(e->rc_cnt, b.rc_cnt); std::swap(e->rc_min, b.rc_min); std::swap(e->ray_bins, b.ray_bins);
  std::swap(e->vkeys[0], b.vkeys[0]); std::swap(e->vkeys[1], b.vkeys[1]);
  std::swap(e->vidx[0], b.vidx[0]); std::swap(e->vidx[1], b.vidx[1]);
  std::swap(e->vsel, b.vsel); std::swap(e->ray_blk, b.ray_blk);
  std::swap(e->sort_tmp, b.sort_tmp); std::swap(e->sort_tmp_bytes, b.sort_tmp_bytes);
  std::swap(e->vcap, b.vcap);
}
struct RayBankScope {  // bank `ctx` is the live one inside the scope
  fdm_engine* e;
  bool swapped;
  RayBankScope(fdm_engine* e_, int ctx) : e(e_), swapped(ctx == 1) { if (swapped) ray_bank_swap(e); }
  ~RayBankScope() { if (swapped) ray_bank_swap(e); }
};
int ensure_ray_streams(fdm_engine* e) {
  if (e->ray_stream[0]) return FDM_OK;
  for (int k = 0; k < 2; ++k) {
    HIPCK(hipStreamCreateWithFlags(&e->ray_stream[k], hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&e->ev_ray_pre[k], hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&e->ev_ray_res[k], hipEventDisableTiming));
  }
  HIPCK(hipEventCreateWithFlags(&e->ev_ray_bin, hipEventDisableTiming));
  return FDM_OK;
}

// processScan + resolveGhostCells on the stream.  voxel: the points are vkeys[1]/vidx[1] runs.
int enqueue_ray_stage(fdm_engine* e, const RayParams& Q_in, bool voxel, const float* dx, const float* dy,
                      const float* dz, int key_mode, int phase) {
  int rc;
  RayParams Q = Q_in;
  if ((rc = ensure_ray_cells(e))) return rc;
  Layer* elev = find_layer(e, "elevation");
  if (!elev) return FDM_OK;  // raycasting.cpp:213-216
  if (phase == 2) {  // (the first part left earlier, on a ray stream: start_ray_stage_early)
    RayLayers L{};
    L.elevation = lptr(e, *elev);
    L.elevation_stride = lstride(e, *elev);
    L.logodds = find_layer(e, "_visibility_logodds")->d;
    L.ray_min = find_layer(e, "raycasting")->d;
    L.ghost = find_layer(e, "ghost_removal")->d;
    L.rec = e->d_rec;
    L.rec_floats = e->rec_floats;
    hipLaunchKernelGGL(k_ray_resolve, dim3(unsigned((e->ncell + 255) / 256)), dim3(256), 0, e->stream, Q,
                       e->G, e->d_state, L, e->d_layer_ptrs, e->n_layer_ptrs, e->rc_cnt, e->rc_min,
                       unsigned(e->ncell));
    HIPCK(hipGetLastError());
    return FDM_OK;
  }
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
  if (voxel && key_mode == 2) {
    hipLaunchKernelGGL(k_vs_mark, dim3(blocks), dim3(256), 0, e->stream, e->vs, e->vsel);
  } else if (voxel) {
    if (key_mode == 1)
      hipLaunchKernelGGL(k_voxel_mark<uint32_t>, dim3(blocks), dim3(256), 0, e->stream, Q.n,
                         reinterpret_cast<const uint32_t*>(e->vkeys[1]), e->vidx[1], e->vsel);
    else
      hipLaunchKernelGGL(k_voxel_mark<unsigned long long>, dim3(blocks), dim3(256), 0, e->stream, Q.n,
                         e->vkeys[1], e->vidx[1], e->vsel);
  }
  // large scans: queue bucketed by (wedge, length class) before the walk (see k_ray_compact)
  const bool large = Q.n >= unsigned(e->ray_large_min);  // one lane per ray, queue bucketed by (wedge, length)
  const bool sort_queue = large && !(e->dbg_ray & 2048);
  // large scans walk with an angular sector's minimum-height image in LDS (fdm_raywedge.hpp): the queue is ordered
  // (sector, length class) for it
  const bool wedge = sort_queue && e->ray_wedge != 0;
  if (wedge) Q.by_sector = 1;
  uint32_t* ray_key = sort_queue ? reinterpret_cast<uint32_t*>(e->vkeys[0]) : nullptr;       // vkeys hold 2 x vcap uint32
  uint32_t* ray_rank = sort_queue ? reinterpret_cast<uint32_t*>(e->vkeys[0]) + e->vcap : nullptr;
  uint32_t* bin_cnt = sort_queue ? e->ray_bins : nullptr;
  if (sort_queue) {
    // points per thread of the queue builder: 8 on multi-million-point scans (the queue tail is one same-address
    // returning atomic per block), 2 below (a 272 K-point scan is 133 blocks of 2 048 points: half the chip)
    unsigned queue_blocks = 0u, queue_block_points = 0u;  // the queue builder's grid: the scatter walks the same block regions
    auto compact = [&](auto PTS) {
      constexpr unsigned kPts = decltype(PTS)::value;
      const unsigned cblocks = (Q.n + 256u * kPts - 1u) / (256u * kPts);
      if (voxel)
        hipLaunchKernelGGL((k_ray_compact<true, int(kPts)>), dim3(cblocks), dim3(256), 0, e->stream, Q, e->G, e->d_state,
                           dx, dy, dz, e->vsel, e->rc_cnt, ray_list, ray_key, ray_rank, bin_cnt, e->ray_blk);
      else
        hipLaunchKernelGGL((k_ray_compact<false, int(kPts)>), dim3(cblocks), dim3(256), 0, e->stream, Q, e->G, e->d_state,
                           dx, dy, dz, static_cast<const uint32_t*>(nullptr), e->rc_cnt, ray_list, ray_key, ray_rank,
                           bin_cnt, e->ray_blk);
      queue_blocks = cblocks;
      queue_block_points = 256u * kPts;
    };
    if (Q.n <= kRsSmallMax) compact(std::integral_constant<unsigned, 2>{});
    else compact(std::integral_constant<unsigned, 8>{});
    uint32_t* bin_start = e->ray_bins + kRayBins;
    uint32_t* bin_part = e->ray_bins + 2u * kRayBins;
    static_assert(kRaySectors * kRaySectorClasses == kRayScan1Threads * kRayScan1Per,
                  "k_ray_bin_scan1 scans every (sector, length class) bucket");
    if (wedge) {
      hipLaunchKernelGGL(k_ray_bin_scan1, dim3(1), dim3(kRayScan1Threads), 0, e->stream, Q, e->G, e->d_state, bin_cnt,
                         bin_start);
    } else {
      hipLaunchKernelGGL(k_ray_bin_sum, dim3(kRayBins / kRayBinBlock), dim3(256), 0, e->stream, Q, e->G, e->d_state,
                         bin_cnt, bin_part);
      hipLaunchKernelGGL(k_ray_bin_scan, dim3(kRayBins / kRayBinBlock), dim3(256), 0, e->stream, Q, e->G, e->d_state,
                         bin_cnt, bin_part, bin_start);
    }
    hipLaunchKernelGGL(k_ray_scatter, dim3(queue_blocks), dim3(256), 0, e->stream, Q, e->G, e->d_state, ray_list, ray_key,
                       ray_rank, bin_start, e->ray_blk, queue_block_points, e->vidx[1]);
    ray_list = e->vidx[1];
  } else if (voxel) {
    hipLaunchKernelGGL((k_ray_compact<true, 1>), dim3(blocks), dim3(256), 0, e->stream, Q, e->G, e->d_state, dx,
                       dy, dz, e->vsel, e->rc_cnt, ray_list, ray_key, ray_rank, bin_cnt, static_cast<uint32_t*>(nullptr));
  } else {
    hipLaunchKernelGGL((k_ray_compact<false, 1>), dim3(blocks), dim3(256), 0, e->stream, Q, e->G, e->d_state, dx,
                       dy, dz, static_cast<const uint32_t*>(nullptr), e->rc_cnt, ray_list, ray_key, ray_rank, bin_cnt,
                       static_cast<uint32_t*>(nullptr));
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
  if (wedge) {
    const unsigned H = std::min(unsigned(std::max(e->G.rows, e->G.cols)) + 2u, kRwRowsMax);
    const unsigned lds = H * kRwCols * unsigned(sizeof(uint32_t));
    // one workgroup per sector; a sector of a very dense scan is shared by several (each flushes its own window)
    const unsigned sectors = kRaySectors;
    // (a camera's 60-degree field of view puts its rays into 43 of the 256 sectors; four workgroups per sector for
    // scans of that size measured 22 us against 18: every one initialises and flushes a window of its own)
    const unsigned parts = e->ray_wedge_parts > 0 ? unsigned(e->ray_wedge_parts)
                                                  : std::max(1u, std::min(8u, Q.n / (sectors * 8u * kRwThreads)));
    const uint32_t* bin_start = e->ray_bins + kRayBins;
    auto launch_wedge = [&](auto kern) -> int {
      if (int rc_lds = allow_lds(kern, lds)) return rc_lds;
      hipLaunchKernelGGL(kern, dim3(sectors * parts), dim3(kRwThreads), lds, e->stream, Q, e->G, e->d_state, dx, dy, dz,
                         ray_list, bin_start, e->rc_min, H, parts);
      return FDM_OK;
    };
    const bool fwin = !(e->dbg_ray & (1 << 20));  // (dbg_ray 1048576, measurement only: the integer window of round 5)
    if (tiled) rc = fwin ? launch_wedge(k_ray_wedge<true, true>) : launch_wedge(k_ray_wedge<true, false>);
    else rc = fwin ? launch_wedge(k_ray_wedge<false, true>) : launch_wedge(k_ray_wedge<false, false>);
    if (rc) return rc;
  } else if (Q.n < (1u << 16) && !large) {
    tiled ? launch_ray(k_ray<true, 16>, 16u) : launch_ray(k_ray<false, 16>, 16u);
  } else if (!large) {
    tiled ? launch_ray(k_ray<true, 8>, 8u) : launch_ray(k_ray<false, 8>, 8u);
  } else {
    tiled ? launch_ray(k_ray<true, 1>, 1u) : launch_ray(k_ray<false, 1>, 1u);
  }
  HIPCK(hipGetLastError());
  if (phase == 1) return FDM_OK;
  hipLaunchKernelGGL(k_ray_resolve, dim3(unsigned((e->ncell + 255) / 256)), dim3(256), 0, e->stream, Q,
                     e->G, e->d_state, L, e->d_layer_ptrs, e->n_layer_ptrs, e->rc_cnt, e->rc_min,
                     unsigned(e->ncell));
  HIPCK(hipGetLastError());
  return FDM_OK;
}

// The raycasting stage of a scan of integrate(), on the map its update — launched just before — leaves: voxel filter
// of the scan's preprocessed cloud, processScan, resolveGhostCells.  Everything it needs was fixed when the scan was
// enqueued (PendingUpdate::RQ, the cloud of the scan's parity): it may run after the NEXT scan's bin half.
// Option "ray_overlap": the part of a large scan's stage that needs the scan and the map GEOMETRY only — voxel filter,
// ray queue, walk — is launched when the scan's bin half has been, on the ray stream of the scan's parity, with the
// geometry derived as the update will commit it (RayParams::pre_slot).  Stages of consecutive scans then run beside
// each other (and beside the fused launches of the main stream); k_ray_resolve stays where it was: behind the scan's
// update, ahead of the next one, on the main stream, which waits for the early part there.
int start_ray_stage_early(fdm_engine* e, fdm_engine::PendingUpdate& u, const ScanParams& P) {
  u.ray_pre = 0;
  const bool want = e->ray_overlap > 0 || (e->ray_overlap < 0 && (e->sync_call || u.RQ.n >= 1000000u));
  if (!want || !u.ray || u.RQ.n < unsigned(e->ray_large_min) || e->profile || !e->ray_wedge) return FDM_OK;
  if (e->voxel_small && u.RQ.n <= unsigned(e->voxel_small_max)) return FDM_OK;  // (the sort-free filter keeps state of its own)
  if (!find_layer(e, "elevation")) return FDM_OK;
  int rc;
  if ((rc = ensure_ray_streams(e))) return rc;
  const int ctx = int(P.scan_no & 1u);
  bool fresh = false;
  {  // allocations (they may drain the streams and, with them, flush this very scan: then the stage has run) before anything is enqueued
    RayBankScope bank(e, ctx);
    fresh = e->rc_cnt == nullptr || u.RQ.n > e->vcap;
    if ((rc = ensure_ray_cells(e)) || (rc = ensure_voxel_buffers(e, u.RQ.n))) return rc;
  }
  if (!u.ray || !e->chain) return FDM_OK;
  hipStream_t rs = e->ray_stream[ctx];
  // the scan's bin half (and everything before it): marked right behind that launch (enqueue_scan) — a mark taken here
  // would also wait for the previous scan's stage, whose resolve has been put on the main stream since
  if (fresh || !e->ray_bin_marked) HIPCK(hipEventRecord(e->ev_ray_bin, e->stream));
  HIPCK(hipStreamWaitEvent(rs, e->ev_ray_bin, 0));
  if (e->ray_res_pending[ctx]) HIPCK(hipStreamWaitEvent(rs, e->ev_ray_res[ctx], 0));  // the bank's previous stage has been resolved
  RayParams Q = u.RQ;
  Q.ctx = ctx;
  Q.pre_slot = P.slot;
  Q.pre_do_move = P.do_move;
  Q.pre_gate = P.gate_on_filter;
  hipStream_t main_stream = e->stream;
  int key_mode = 0;
  {
    RayBankScope bank(e, ctx);
    e->stream = rs;
    rc = enqueue_voxel_sort(e, Q.n, static_cast<float>(e->G.res), Q.flag_slot, u.ray_x, u.ray_y, u.ray_z, u.ray_box, &key_mode);
    if (!rc) rc = enqueue_ray_stage(e, Q, true, u.ray_x, u.ray_y, u.ray_z, key_mode, 1);
    e->stream = main_stream;
  }
  if (rc) return rc;
  HIPCK(hipEventRecord(e->ev_ray_pre[ctx], rs));
  u.RQ = Q;
  u.ray_pre = 1 + ctx;
  u.ray_key_mode = key_mode;
  return FDM_OK;
}

int run_held_ray_stage(fdm_engine* e, fdm_engine::PendingUpdate& u) {
  if (!u.ray) return FDM_OK;
  u.ray = false;
  int rc;
  if (u.ray_pre) {  // the first part is on its way (start_ray_stage_early): wait for it here, resolve
    const int ctx = u.ray_pre - 1;
    u.ray_pre = 0;
    HIPCK(hipStreamWaitEvent(e->stream, e->ev_ray_pre[ctx], 0));
    {
      RayBankScope bank(e, ctx);
      rc = enqueue_ray_stage(e, u.RQ, true, u.ray_x, u.ray_y, u.ray_z, u.ray_key_mode, 2);
    }
    if (rc) return rc;
    HIPCK(hipEventRecord(e->ev_ray_res[ctx], e->stream));
    e->ray_res_pending[ctx] = true;
    e->ray_timed = false;
    return FDM_OK;
  }
  if (e->profile) HIPCK(hipEventRecord(e->ev_ray[0], e->stream));
  int key_mode = 0;
  if ((rc = enqueue_voxel_sort(e, u.RQ.n, static_cast<float>(e->G.res), u.RQ.flag_slot, u.ray_x, u.ray_y, u.ray_z,
                               u.ray_box, &key_mode)))
    return rc;
  if ((rc = enqueue_ray_stage(e, u.RQ, true, u.ray_x, u.ray_y, u.ray_z, key_mode))) return rc;
  if (e->profile) {
    HIPCK(hipEventRecord(e->ev_ray[1], e->stream));
    e->ray_timed = true;
  }
  return FDM_OK;
}

}  // namespace fdmh

extern "C" {

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
  int key_mode = 0;  // no box: full 63-bit keys
  if ((rc = enqueue_voxel_sort(e, unsigned(n), voxel_size, -1, dx, dy, dz, nullptr, &key_mode))) return rc;
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


}  // extern "C"
