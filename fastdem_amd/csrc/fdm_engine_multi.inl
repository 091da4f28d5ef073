// fdm_engine_multi.inl — host side of the batch pipeline (fdm_multi.hpp): which scans of a
// fdm_engine_integrate_device_batch call may leave as one batch, the per-batch scratch sets, and the launches.
// The body of fdm_engine_multi.hip (inside namespace fdmh): do not compile on its own.

// Maps up to this many cells take batches: the kMaxBatch scratch sets of both parities cost 2 KB per cell.
constexpr size_t kBatchMaxCells = size_t(1) << 18;

// A 4x4 whose last row is exactly (0 0 0 1) — what Isometry3d::matrix() always is.
bool affine_last_row(const double* T) { return T[3] == 0.0 && T[7] == 0.0 && T[11] == 0.0 && T[15] == 1.0; }

// How many of the leading `count` scans can leave as ONE batch (0 or 1: take the single-scan path).
// A batch is a run of plain small scans — exactly the scans the single-scan path would hold back and fuse
// (enqueue_scan's `plain`), on the per-cell scratch pipeline with the one-point-per-thread bin kernel, from ONE
// sensor (same T_base_sensor) and with the same optional channels.
uint32_t multi_run(fdm_engine* e, uint32_t count, const fdm_device_scan* scans) {
  if (!e->batch || count < 2u || !e->overlap || !e->wave_merge || e->bin_variant == 4) return 0u;
  if (!e->estimator_ready || e->rec_kind < 0 || !e->S.dense || e->ncell > kBatchMaxCells) return 0u;
  if (e->cap_pre || e->cap_ras || e->want_ids || e->profile) return 0u;
  // raycasting on (fastdem.cpp:152-159): the stage rides in the batch (fdm_rbatch.hpp) if every scan can take the
  // sort-free voxel filter on compact keys and the engine holds the whole map
  const bool ray = e->cfg.raycast_enabled != 0;
  if (ray) {
    if (!e->batch_ray || !e->voxel_small || !voxel_size_ok(static_cast<float>(e->G.res))) return 0u;
    if (e->G.o_rows != e->G.rows || e->G.o_cols != e->G.cols || e->G.s_rows != e->G.rows || e->G.s_cols != e->G.cols) return 0u;
  }
  poll_dense_paid(e);
  if (e->obst_dense_pending || e->last_kind == 1 || e->next_drop_nonfinite) return 0u;
  if (e->dbg_no_atomics || e->dbg_upd || e->move_clear_basic) return 0u;
  const fdm_device_scan& f = scans[0];
  const size_t kt = e->ncell / 1024u;  // (the thresholds below were measured in units of 1 024 cells, round 2)
  uint32_t run = 0;
  // (automatic: 32 with the quantile estimator, and with raycasting on — a batch is then seven launches, five of them the
  //  ray stage's, whose fixed costs halve per scan: configs[1] 9.25 -> 7.78 us per scan, 24 per launch 8.3; 16 for Kalman alone)
  const uint32_t batch_max = e->batch_max > 0 ? uint32_t(e->batch_max)
                                              : ((e->cfg.estimation_type == 1 || ray) ? uint32_t(kMaxBatch) : 16u);
  const uint32_t cap = std::min<uint32_t>(count, batch_max);
  for (; run < cap; ++run) {
    const fdm_device_scan& s = scans[run];
    if (s.n == 0 || !s.x || !s.y || !s.z) break;
    if (s.n >= 393216u) break;  // the 4-points-per-thread kernels take over there (enqueue_scan)
    // ... and the record-pool pipeline on maps with enough tiles
    const bool enough_tiles = e->tiled_forced || kt >= 240 || (kt >= 160 && s.n >= 100000);
    if (e->tiled && s.n >= e->tiled_min && enough_tiles) break;
    if ((s.intensity != nullptr) != (f.intensity != nullptr) || (s.rgb != nullptr) != (f.rgb != nullptr) ||
        (s.sigma_z2 != nullptr) != (f.sigma_z2 != nullptr))
      break;
    if (!affine_last_row(s.T_base_sensor) || !affine_last_row(s.T_world_base)) break;
    if (std::memcmp(s.T_base_sensor, f.T_base_sensor, 16 * sizeof(double)) != 0) break;
    if (ray) {
      if (s.n > uint64_t(e->voxel_small_max)) break;
      ScanParams P;
      double box[6];
      fill_integrate_params(e, P, s.T_base_sensor, s.T_world_base);
      ray_box_of(e, P, box);
      const VoxelCompact C = voxel_compact_of(static_cast<float>(e->G.res), box);
      if (C.bits <= 0 || 2 * C.bits + C.zbits > 31) break;
    }
  }
  return run >= 2u ? run : 0u;
}

// The ring of batch states: batches are numbered with e->mseq, a launch re-arms the state of the batch after next.
int ensure_mstate(fdm_engine* e) {
  if (e->mstate) return FDM_OK;
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mstate), kMStates * sizeof(MState)));
  HIPCK(hipMemsetAsync(e->mstate, 0, kMStates * sizeof(MState), e->stream));
  return FDM_OK;
}

int ensure_multi(fdm_engine* e, size_t max_n, size_t blocks) {
  const size_t slots = size_t(kMaxBatch) * e->ncell;
  if (int rc = ensure_mstate(e)) return rc;
  if (!e->mkey[0] || !e->mupd_part) {  // (all or nothing: a partial set after a failed hipMalloc is given back)
    auto release = [&]() {
      for (int k = 0; k < 2; ++k) {
        if (e->mkey[k]) (void)hipFree(e->mkey[k]);
        if (e->maux[k]) (void)hipFree(e->maux[k]);
        if (e->mzs[k]) (void)hipFree(e->mzs[k]);
        e->mkey[k] = nullptr; e->maux[k] = nullptr; e->mzs[k] = nullptr;
      }
      if (e->mupd_part) (void)hipFree(e->mupd_part);
      e->mupd_part = nullptr;
    };
    release();
    auto alloc = [&](auto*& ptr, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(&ptr), bytes) == hipSuccess; };
    bool ok = true;
    for (int k = 0; k < 2 && ok; ++k)
      ok = alloc(e->mkey[k], slots * sizeof(unsigned long long)) && alloc(e->maux[k], slots * sizeof(uint4)) &&
           alloc(e->mzs[k], slots * sizeof(uint2));
    ok = ok && alloc(e->mupd_part, ((e->ncell + 31u) / 32u) * sizeof(uint32_t));  // (32 cells per update block at least: upd_cells_per_block)
    if (!ok) {
      (void)hipGetLastError();
      release();
      return fail(FDM_ERR_HIP, "batch pipeline: out of device memory for the per-scan scratch sets");
    }
    for (int k = 0; k < 2; ++k) {
      const int blocks_f = int(std::min<size_t>((slots + 255) / 256, 4096));
      hipLaunchKernelGGL(k_fill_u64, dim3(blocks_f), dim3(256), 0, e->stream, e->mkey[k], kEmptyKey, slots);
      hipLaunchKernelGGL(k_fill_aux, dim3(blocks_f), dim3(256), 0, e->stream, e->maux[k], slots);
      HIPCK(hipGetLastError());
      HIPCK(hipMemsetAsync(e->mzs[k], 0xFF, slots * sizeof(uint2), e->stream));
    }
  }
  if (max_n > e->mobs_stride) {
    if (int rc_sync = sync_all(e)) return rc_sync;  // (a held-back batch update reads the old arrays)
    e->mobs_stride = ((max_n + max_n / 4 + 1024) + 3) & ~size_t(3);
    for (int k = 0; k < 2; ++k) {
      if (e->mobs[k]) HIPCK(hipFree(e->mobs[k]));
      if (e->mcobs[k]) HIPCK(hipFree(e->mcobs[k]));
      e->mobs[k] = nullptr;
      e->mcobs[k] = nullptr;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mobs[k]), size_t(kMaxBatch) * e->mobs_stride * sizeof(float2)));
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mcobs[k]), size_t(kMaxBatch) * e->mobs_stride * sizeof(uint32_t)));
    }
  }
  if (blocks > e->mbin_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    e->mbin_cap = blocks + blocks / 4 + 64;
    for (int k = 0; k < 2; ++k) {
      if (e->mbin_part[k]) HIPCK(hipFree(e->mbin_part[k]));
      e->mbin_part[k] = nullptr;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mbin_part[k]), e->mbin_cap * sizeof(unsigned long long)));
    }
  }
  return FDM_OK;
}

// Buffers of the batch's raycasting launches (fdm_rbatch.hpp), slots as wide as the observation arrays' (mobs_stride).
int ensure_rbatch(fdm_engine* e) {
  auto alloc = [&](auto*& ptr, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(&ptr), bytes) == hipSuccess; };
  if (!e->rb_state) {
    const size_t counters = size_t(kMaxBatch) * ((size_t(1) << kVsFineBits) + kVsCoarse);
    const size_t img = size_t(kMaxBatch) * e->ncell;
    bool ok = alloc(e->rb_state, sizeof(RState)) && alloc(e->rb_counters, counters * sizeof(uint32_t)) &&
              alloc(e->rb_img, 2u * img * sizeof(uint32_t));
    if (!ok) {
      (void)hipGetLastError();
      if (e->rb_state) (void)hipFree(e->rb_state);
      if (e->rb_counters) (void)hipFree(e->rb_counters);
      if (e->rb_img) (void)hipFree(e->rb_img);
      e->rb_state = nullptr; e->rb_counters = nullptr; e->rb_img = nullptr;
      return fail(FDM_ERR_HIP, "batch pipeline: out of device memory for the raycasting images");
    }
    HIPCK(hipMemsetAsync(e->rb_state, 0, sizeof(RState), e->stream));
    HIPCK(hipMemsetAsync(e->rb_counters, 0, counters * sizeof(uint32_t), e->stream));
    HIPCK(hipMemsetAsync(e->rb_img, 0, img * sizeof(uint32_t), e->stream));                      // evidence counts
    HIPCK(hipMemsetAsync(e->rb_img + img, 0xFF, img * sizeof(uint32_t), e->stream));            // kRayEmpty
  }
  if (e->rb_stride != e->mobs_stride) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->rb_cap) HIPCK(hipFree(e->rb_cap));
    if (e->rb_u32) HIPCK(hipFree(e->rb_u32));
    if (e->rb_rec) HIPCK(hipFree(e->rb_rec));
    e->rb_cap = nullptr; e->rb_u32 = nullptr; e->rb_rec = nullptr;
    e->rb_stride = 0;
    const size_t slots = size_t(kMaxBatch) * e->mobs_stride;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->rb_cap), 3u * slots * sizeof(float)));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->rb_u32), 7u * slots * sizeof(uint32_t)));  // keys | place | sel | 4 ray queues
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->rb_rec), slots * sizeof(uint4)));
    e->rb_stride = e->mobs_stride;
  }
  return FDM_OK;
}

// The five raycasting launches of the batch the launch before binned (its preprocessed clouds are in rb_cap).
int launch_rbatch(fdm_engine* e, const RBatch& R) {
  unsigned max_n = 0u;
  for (unsigned k = 0; k < R.count; ++k) max_n = std::max(max_n, R.n[k]);
  const dim3 grid((max_n + 255u) / 256u, R.count);
  hipLaunchKernelGGL(k_rb_count, grid, dim3(256), 0, e->stream, R, e->G);
  hipLaunchKernelGGL(k_rb_scatter, grid, dim3(256), 0, e->stream, R);
  hipLaunchKernelGGL(k_rb_mark, grid, dim3(256), 0, e->stream, R);
  hipLaunchKernelGGL(k_rb_compact, grid, dim3(256), 0, e->stream, R, e->G);
  // The walks.  LDS images (k_rb_ray_lds) when a quadrant of a centred sensor fits the workgroup's LDS with room to
  // spare for a sensor off the centre; else one lane (or SEG lanes) per ray on global atomics.
  if (e->rb_lds_words == 0u) {  // (once per engine: the largest dynamic LDS a workgroup of this kernel may ask for)
    e->rb_lds_words = 16384u;   // 64 KB without asking
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_rb_ray_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            128 * 1024) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rb_ray_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            128 * 1024) == hipSuccess)
      e->rb_lds_words = 32768u;
    else
      (void)hipGetLastError();
  }
  const size_t quadrant = (size_t(e->G.rows) / 2 + 2) * (size_t(e->G.cols) / 2 + 2);
  if (e->batch_ray_lds && quadrant * 2 <= e->rb_lds_words) {
    unsigned parts = e->batch_ray_parts > 0 ? unsigned(e->batch_ray_parts) : std::max(1u, 256u / (4u * R.count));
    parts = std::min(parts, std::max(1u, (max_n + kRbRayThreads - 1u) / kRbRayThreads));
    // (LDS for twice a centred sensor's quadrant — a LOCAL map follows the robot, a GLOBAL map's sensor may wander: a
    // larger quadrant takes the kernel's global-atomic loop — or for the whole map when that is less)
    const unsigned words = unsigned(std::min<size_t>({size_t(e->rb_lds_words), size_t(e->G.rows) * size_t(e->G.cols),
                                                      e->batch_ray_words > 0 ? size_t(e->batch_ray_words) : quadrant * 2}));
    if (e->dbg_ray & (1 << 20))  // (dbg_ray 1048576, measurement only: the integer image of round 4)
      hipLaunchKernelGGL(k_rb_ray_lds<false>, dim3(4u * parts, R.count), dim3(kRbRayThreads), words * sizeof(uint32_t),
                         e->stream, R, e->G, parts, words);
    else
      hipLaunchKernelGGL(k_rb_ray_lds<true>, dim3(4u * parts, R.count), dim3(kRbRayThreads), words * sizeof(uint32_t),
                         e->stream, R, e->G, parts, words);
  } else {
    // upper bound of a scan's queue: every point a ray, padded to whole wavefronts per segment
    auto rays = [&](auto seg) {
      constexpr int SEG = decltype(seg)::value;
      const unsigned threads = ((max_n + 63u) & ~63u) * unsigned(SEG);
      hipLaunchKernelGGL((k_rb_ray<SEG>), dim3((threads + 255u) / 256u, R.count, 4), dim3(256), 0, e->stream, R, e->G);
    };
    switch (e->batch_ray_seg) {
      case 1: rays(std::integral_constant<int, 1>{}); break;
      case 8: rays(std::integral_constant<int, 8>{}); break;
      case 16: rays(std::integral_constant<int, 16>{}); break;
      default: rays(std::integral_constant<int, 4>{}); break;
    }
  }
  HIPCK(hipGetLastError());
  return FDM_OK;
}

template <typename F>
int with_channels(int ch, F&& f) {
  switch (ch & 3) {
    case 0: return f(std::integral_constant<int, 0>{});
    case 1: return f(std::integral_constant<int, 1>{});
    case 2: return f(std::integral_constant<int, 2>{});
    default: return f(std::integral_constant<int, 3>{});
  }
}

// One k_mbatch launch: [ update U | bin B | crop Cn ], any of which may be empty (count == 0).
int launch_mbatch(fdm_engine* e, int ch, const MUpd& U, const MBin& B, const MCrop& Cn, const MCommon& K) {
  // (7.2 KB of kernel arguments with 32 scans per batch: the AQL kernarg segment has no 4 KB limit — scripts/ubench/
  // kernarg_big.hip passes 8 KB by value on this stack, checked on the box in round 6)
  static_assert(sizeof(MUpd) + sizeof(MBin) + sizeof(MCrop) + sizeof(MCommon) + sizeof(GeomConst) + 160 <= 8192,
                "k_mbatch: kernel arguments beyond 8 KB");
  const unsigned ub = U.count ? unsigned((e->ncell + upd_cells_per_block(U.count) - 1u) / upd_cells_per_block(U.count)) : 0u;
  // rows as wide as the widest scan; the update's blocks fill as many leading rows as they need
  unsigned gx = 0u;
  for (unsigned k = 0; k < B.count; ++k) gx = std::max(gx, (B.n[k] + kMBlock - 1u) / kMBlock);
  if (gx == 0u) gx = std::min(std::max(ub, Cn.count * kMScout + 1u), 64u);
  if (gx == 0u) return FDM_OK;
  const unsigned crows = Cn.count ? (Cn.count * kMScout + 1u + gx - 1u) / gx : 0u;  // the scout blocks of the next batch + its walker block, in rows of their own
  const unsigned urows = (ub + gx - 1u) / gx, rows = urows + B.count + crows;
  MCommon Kt = K;
  Kt.timeline = (e->d_timeline && gx * rows <= e->timeline_cap && B.count) ? e->d_timeline : nullptr;
  if (Kt.timeline) { e->timeline_blocks = gx * rows; e->timeline_upd = gx; e->timeline_bin = urows; }
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    if constexpr (is_rec_policy<POLICY>) {
      return with_channels(ch, [&](auto chc) -> int {
        constexpr int CH = decltype(chc)::value;
        if (U.ray.stamp != 0u || B.cap != nullptr)  // (a half with raycasting: the variant whose update resolves ray events)
          hipLaunchKernelGGL((k_mbatch<POLICY, CH, true>), dim3(gx, rows), dim3(256), 0, e->stream, U, B, Cn, Kt, e->G,
                             e->d_state, layers, e->d_layer_ptrs, e->n_layer_ptrs, unsigned(e->ncell), ub, urows);
        else
          hipLaunchKernelGGL((k_mbatch<POLICY, CH, false>), dim3(gx, rows), dim3(256), 0, e->stream, U, B, Cn, Kt, e->G,
                             e->d_state, layers, e->d_layer_ptrs, e->n_layer_ptrs, unsigned(e->ncell), ub, urows);
        HIPCK(hipGetLastError());
        return FDM_OK;
      });
    } else {
      return fail(FDM_ERR_INVALID, "internal: batch launch with a per-layer policy");
    }
  });
}

// The held-back update of a batch on its own (launch_update_alone forwards here).
int launch_multi_update(fdm_engine* e, const fdm_engine::PendingUpdate& u) {
  MBin B;
  MCrop Cn;
  MCommon K;
  std::memset(&B, 0, sizeof(B));
  std::memset(&Cn, 0, sizeof(Cn));
  std::memset(&K, 0, sizeof(K));
  return launch_mbatch(e, u.ch, u.MU, B, Cn, K);
}

// `count` (2 .. kMaxBatch) scans that multi_run() accepted leave as ONE launch: the held-back update of the previous
// batch (when there is one), this batch's bin, and — `next_count` > 0 — the crop pass of the batch the caller will
// enqueue next (scans[count .. count + next_count)).  This batch's update is held back.
int enqueue_multi(fdm_engine* e, uint32_t count, const fdm_device_scan* scans, uint32_t next_count) {
  int rc;
  const fdm_device_scan& f = scans[0];
  const bool hi = f.intensity != nullptr, hc = f.rgb != nullptr, hv = f.sigma_z2 != nullptr;
  const int ch = (hi ? 1 : 0) | (hc ? 2 : 0);
  const bool ray = e->cfg.raycast_enabled != 0;  // (multi_run checked that the batch can carry the stage)
  if ((rc = ensure_scratch_channels(e, hi, hc))) return rc;
  if (ray && (rc = ensure_ray_layers(e))) return rc;
  if ((rc = refresh_layer_ptrs(e))) return rc;
  // a held-back update of another kind (single scan, other channels) leaves first
  if (e->chain && !(e->pend.multi && e->pend.ch == ch) && (rc = join_streams(e))) return rc;

  MBin B;
  MCrop Cn;
  MCommon K;
  MUpd U;
  RBatch R;
  std::memset(&B, 0, sizeof(B));
  std::memset(&Cn, 0, sizeof(Cn));
  std::memset(&K, 0, sizeof(K));
  std::memset(&U, 0, sizeof(U));
  std::memset(&R, 0, sizeof(R));
  size_t max_n = 0;
  unsigned blocks = 0;
  ScanParams P;
  for (uint32_t k = 0; k < count; ++k) {
    const fdm_device_scan& s = scans[k];
    fill_integrate_params(e, P, s.T_base_sensor, s.T_world_base);
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r) B.t[k].Twb[c * 3 + r] = P.Twb[c * 4 + r];
    std::memcpy(B.t[k].R, P.R, sizeof(P.R));
    B.n[k] = uint32_t(s.n);
    B.px[k] = s.x; B.py[k] = s.y; B.pz[k] = s.z; B.pint[k] = s.intensity; B.prgb[k] = s.rgb; B.pvar[k] = s.sigma_z2;
    B.first_block[k] = blocks;
    B.robot_x[k] = P.robot_x;
    B.robot_y[k] = P.robot_y;
    blocks += unsigned((s.n + kMBlock - 1u) / kMBlock);
    max_n = std::max<size_t>(max_n, s.n);
    if (ray) {  // the stage's per-scan parameters, as enqueue_scan derives them for one scan
      double box[6];
      ray_box_of(e, P, box);
      R.C[k] = voxel_compact_of(static_cast<float>(e->G.res), box);
      const int key_bits = 2 * R.C[k].bits + R.C[k].zbits;
      R.shift[k] = unsigned(std::max(R.C[k].bits, key_bits - int(kVsFineBits)));
      R.ibits[k] = 1u;
      while ((1u << R.ibits[k]) < unsigned(s.n)) ++R.ibits[k];
      R.n[k] = uint32_t(s.n);
      R.ox[k] = P.ray_ox; R.oy[k] = P.ray_oy; R.oz[k] = P.ray_oz;
    }
  }
  for (uint32_t k = count; k <= uint32_t(kMaxBatch); ++k) B.first_block[k] = blocks;
  if ((rc = ensure_multi(e, max_n, blocks))) return rc;
  if (ray && (rc = ensure_rbatch(e))) return rc;

  K.min_sq = P.min_sq; K.max_sq = P.max_sq; K.z_min = P.z_min; K.z_max = P.z_max;
  sensor_params(e->cfg, K.sensor_type, K.sp);
  std::memcpy(K.Tbs, P.Tbs, sizeof(K.Tbs));
  K.integrate_mode = 1;
  K.do_move = P.do_move;
  K.gate_on_filter = P.gate_on_filter;
  K.has_var = hv ? 1 : 0;
  K.bin_table = e->bin_table;
  K.dbg = e->dbg_batch;
  K.walk = e->batch_walk < 0 ? (e->cfg.estimation_type == 1 ? 1 : 0) : e->batch_walk;

  const unsigned seq = e->mseq++;
  const int slot = int(seq % unsigned(kMStates)), par = int(seq & 1u);
  // the crop pass that ran one launch ahead left this batch's pass bits in its state word — if it was for THIS batch
  const bool pre = e->pre_valid && e->pre_call == e->batch_call && e->pre_scans == scans && e->pre_count == count && e->pre_seq == seq;
  if (e->pre_valid && !pre)  // (bits of a batch that never came: not expected inside one call)
    HIPCK(hipMemsetAsync(e->mstate[slot].flags, 0, sizeof(unsigned) * kLineWords, e->stream));
  e->pre_valid = false;
  B.count = count;
  B.scan_no0 = uint32_t(e->scan_no);
  B.ms = e->mstate + slot;
  const bool fuse = e->chain && e->pend.multi;  // (same channels: checked above)
  B.prev = fuse ? e->pend.MU.ms : nullptr;
  B.prev_count = fuse ? e->pend.MU.count : 0u;
  B.obs_stride = unsigned(e->mobs_stride);
  B.key = e->mkey[par];
  B.aux = e->maux[par];
  B.zs = e->mzs[par];
  B.obs = e->mobs[par];
  B.cobs = e->mcobs[par];
  B.bin_part = e->mbin_part[par];
  B.cap = ray ? e->rb_cap : nullptr;
  if (fuse) U = e->pend.MU;

  // "scan k has a point that survives the crops" (it moves a LOCAL map, fastdem.cpp:138-145) must be in the batch's
  // state word when its bin blocks start: the scouts of the previous launch left it there if they scouted THIS batch;
  // otherwise (the first batch of a call) a small launch of scouts runs ahead of it
  auto fill_scouts = [&](MCrop& C, uint32_t n, const fdm_device_scan* ss, MState* ms) {
    std::memset(&C, 0, sizeof(C));
    for (uint32_t k = 0; k < n; ++k) {
      C.n[k] = uint32_t(ss[k].n);
      C.px[k] = ss[k].x; C.py[k] = ss[k].y; C.pz[k] = ss[k].z;
      C.robot_x[k] = ss[k].T_world_base[12];  // T_world_base.translation().head<2>() (fastdem.cpp:144), as MBin::robot_x
      C.robot_y[k] = ss[k].T_world_base[13];
    }
    C.count = n;
    C.ms = ms;
  };
  const bool gated = K.do_move && K.gate_on_filter;
  if (gated && !pre) {
    MUpd U0;
    MBin B0;
    MCrop C0;
    std::memset(&U0, 0, sizeof(U0));
    std::memset(&B0, 0, sizeof(B0));
    fill_scouts(C0, count, scans, e->mstate + slot);
    // (no batch is binned by this launch: the walker block starts the chain where this batch's bin blocks will)
    C0.prev = fuse ? e->pend.MU.ms : nullptr;
    C0.prev_count = fuse ? e->pend.MU.count : 0u;
    C0.scan_no0 = uint32_t(e->scan_no);
    if ((rc = launch_mbatch(e, ch, U0, B0, C0, K))) return rc;
  }
  B.pre = 1u;
  // ... and the scouts of the batch after this one ride in this launch (option "batch_crop" 0: they never do)
  if (next_count >= 2u && gated && e->batch_crop &&
      std::memcmp(scans[count].T_base_sensor, f.T_base_sensor, 16 * sizeof(double)) == 0) {
    fill_scouts(Cn, next_count, scans + count, e->mstate + int((seq + 1u) % unsigned(kMStates)));
    e->pre_valid = true;
    e->pre_call = e->batch_call;
    e->pre_scans = scans + count;
    e->pre_count = next_count;
    e->pre_seq = seq + 1u;
  }
  e->chain = false;
  if ((rc = launch_mbatch(e, ch, U, B, Cn, K))) return rc;
  if (ray) {  // this batch's raycasting images, between the launch that binned it and the one that will update it
    if (++e->rb_seq == 0u) ++e->rb_seq;
    const size_t slots = size_t(kMaxBatch) * e->rb_stride, img = size_t(kMaxBatch) * e->ncell;
    R.count = count;
    R.stride = unsigned(e->rb_stride);
    R.ncell = unsigned(e->ncell);
    R.stamp = e->rb_seq;
    R.do_move = K.do_move;
    R.gate_on_filter = K.gate_on_filter;
    R.dbg = e->dbg_ray;
    R.ms = B.ms;
    R.rs = e->rb_state;
    R.resolution = static_cast<float>(e->G.res);
    R.inv_voxel = 1.0f / R.resolution;  // (voxel size = map resolution; RayParams::inv_voxel, VoxelCompact's 1 / voxel_size)
    R.cap = e->rb_cap;
    R.keys = e->rb_u32; R.place = e->rb_u32 + slots; R.sel = e->rb_u32 + 2u * slots; R.ray_list = e->rb_u32 + 3u * slots;  // (4 x slots)
    R.rec = e->rb_rec;
    R.fine = e->rb_counters;
    R.coarse = e->rb_counters + (size_t(kMaxBatch) << kVsFineBits);
    R.rc_cnt = e->rb_img;
    R.rc_min = e->rb_img + img;
    if ((rc = launch_rbatch(e, R))) return rc;
  }

  // this batch's update is held back (option "batch_fuse" 0: launched at once, for per-kernel measurements)
  MUpd& N = e->pend.MU;
  std::memset(&N, 0, sizeof(N));
  N.count = count;
  N.scan_no0 = B.scan_no0;
  N.obs_stride = B.obs_stride;
  N.do_move = K.do_move;
  N.gate_on_filter = K.gate_on_filter;
  N.ms = B.ms;
  N.rearm = e->mstate + int((seq + 3u) % unsigned(kMStates));
  N.key = B.key; N.aux = B.aux; N.zs = B.zs; N.obs = B.obs; N.cobs = B.cobs;
  N.upd_part = e->mupd_part;
  if (ray) {
    const fdm_raycast_config c = ray_config_of(e->cfg);
    N.ray.rs = e->rb_state;
    N.ray.stamp = e->rb_seq;
    N.ray.rc_cnt = R.rc_cnt;
    N.ray.rc_min = R.rc_min;
    N.ray.logodds = find_layer(e, "_visibility_logodds")->d;
    N.ray.ray_min = find_layer(e, "raycasting")->d;
    N.ray.ghost = find_layer(e, "ghost_removal")->d;
    N.ray.l_obs = c.log_odds_observed; N.ray.l_ghost = c.log_odds_ghost; N.ray.l_max = c.log_odds_max;
    N.ray.clear_thr = c.clear_threshold; N.ray.conflict_thr = c.height_conflict_threshold;
  }
  e->pend.multi = true;
  e->pend.ray = false;
  e->pend.ch = ch;
  e->pend.tiled = false;
  e->chain = true;
  e->last_do_move = P.do_move;
  e->last_gate = P.gate_on_filter;
  if (!e->batch_fuse && (rc = join_streams(e))) return rc;
  // bookkeeping as enqueue_scan leaves it after the batch's last scan
  const fdm_device_scan& l = scans[count - 1u];
  e->last_kind = 0;
  e->last_bin_blocks = blocks - B.first_block[count - 1u];
  e->last_bin_part = B.bin_part + B.first_block[count - 1u];
  e->last_upd_tiles = unsigned((e->ncell + upd_cells_per_block(count) - 1u) / upd_cells_per_block(count));
  e->last_upd_part = e->mupd_part;
  e->ray_timed = false;
  e->scan_no += count;
  e->last_batch_n = int(count);
  ++e->n_mbatch;
  e->have_scan = true;
  e->last_n = uint32_t(l.n);
  e->last_n_input = uint32_t(l.n);
  e->ingest_blocks = 0;
  e->last_was_integrate = 1;
  return FDM_OK;
}
