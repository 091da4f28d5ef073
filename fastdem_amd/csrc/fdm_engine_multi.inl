// fdm_engine_multi.inl — host side of the batch pipeline (fdm_multi.hpp): which scans of a
// fdm_engine_integrate_device_batch call may leave as one batch, the per-batch scratch sets, and the launches.
// Part of fdm_engine.hip's translation unit (inside its anonymous namespace): do not compile on its own.

// Maps up to this many cells take batches: the kMaxBatch scratch sets of both parities cost 1 KB per cell.
constexpr size_t kBatchMaxCells = size_t(1) << 18;

// A 4x4 whose last row is exactly (0 0 0 1) — what Isometry3d::matrix() always is.
bool affine_last_row(const double* T) { return T[3] == 0.0 && T[7] == 0.0 && T[11] == 0.0 && T[15] == 1.0; }

// How many of the leading `count` scans can leave as ONE batch (0 or 1: take the single-scan path).
// A batch is a run of plain small scans — exactly the scans the single-scan path would hold back and fuse
// (enqueue_scan's `plain`), on the per-cell scratch pipeline with the one-point-per-thread bin kernel.
uint32_t multi_run(fdm_engine* e, uint32_t count, const fdm_device_scan* scans) {
  if (!e->batch || count < 2u || !e->overlap || !e->wave_merge || e->bin_variant == 4) return 0u;
  if (!e->estimator_ready || e->rec_kind < 0 || !e->S.dense || e->ncell > kBatchMaxCells) return 0u;
  if (e->cfg.raycast_enabled || e->cap_pre || e->cap_ras || e->want_ids || e->profile) return 0u;
  if (e->obst_dense_pending || e->last_kind == 1 || e->next_drop_nonfinite) return 0u;
  if (e->dbg_no_atomics || e->dbg_upd) return 0u;
  const fdm_device_scan& f = scans[0];
  const size_t kt = e->ncell / kTileCells;
  uint32_t run = 0;
  const uint32_t cap = std::min<uint32_t>(count, uint32_t(e->batch_max));
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
  }
  return run >= 2u ? run : 0u;
}

int ensure_multi(fdm_engine* e, size_t max_n, size_t blocks) {
  const size_t slots = size_t(kMaxBatch) * e->ncell;
  if (!e->mkey[0]) {
    for (int k = 0; k < 2; ++k) {
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mkey[k]), slots * sizeof(unsigned long long)));
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->maux[k]), slots * sizeof(uint4)));
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mzs[k]), slots * sizeof(uint2)));
      const int blocks_f = int(std::min<size_t>((slots + 255) / 256, 4096));
      hipLaunchKernelGGL(k_fill_u64, dim3(blocks_f), dim3(256), 0, e->stream, e->mkey[k], kEmptyKey, slots);
      hipLaunchKernelGGL(k_fill_aux, dim3(blocks_f), dim3(256), 0, e->stream, e->maux[k], slots);
      HIPCK(hipGetLastError());
      HIPCK(hipMemsetAsync(e->mzs[k], 0xFF, slots * sizeof(uint2), e->stream));
    }
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mstate), 2 * sizeof(MState)));
    HIPCK(hipMemsetAsync(e->mstate, 0, 2 * sizeof(MState), e->stream));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->mscans), 2 * sizeof(MScanBlock)));
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

template <typename F>
int with_channels(int ch, F&& f) {
  switch (ch & 3) {
    case 0: return f(std::integral_constant<int, 0>{});
    case 1: return f(std::integral_constant<int, 1>{});
    case 2: return f(std::integral_constant<int, 2>{});
    default: return f(std::integral_constant<int, 3>{});
  }
}

// The held-back update of a batch on its own (launch_update_alone forwards here).
int launch_multi_update(fdm_engine* e, const fdm_engine::PendingUpdate& u) {
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    if constexpr (is_rec_policy<POLICY>) {
      return with_channels(u.ch, [&](auto chc) -> int {
        constexpr int CH = decltype(chc)::value;
        hipLaunchKernelGGL((k_mupdate<POLICY, CH>), dim3(e->n_tiles), dim3(256), 0, e->stream, u.MB, e->G, e->d_state,
                           layers, e->d_layer_ptrs, e->n_layer_ptrs, unsigned(e->ncell));
        HIPCK(hipGetLastError());
        return FDM_OK;
      });
    } else {
      return fail(FDM_ERR_INVALID, "internal: batch update with a per-layer policy");
    }
  });
}

// `count` (2 .. kMaxBatch) scans that multi_run() accepted: parameter upload, then the bin launch of this batch —
// fused with the held-back update of the previous batch when there is one.  This batch's update is held back.
int enqueue_multi(fdm_engine* e, uint32_t count, const fdm_device_scan* scans) {
  int rc;
  const fdm_device_scan& f = scans[0];
  const bool hi = f.intensity != nullptr, hc = f.rgb != nullptr, hv = f.sigma_z2 != nullptr;
  const int ch = (hi ? 1 : 0) | (hc ? 2 : 0);
  if ((rc = ensure_scratch_channels(e, hi, hc))) return rc;
  if ((rc = refresh_layer_ptrs(e))) return rc;
  // a held-back update of another kind (single scan, other channels) leaves first
  if (e->chain && !(e->pend.multi && e->pend.ch == ch) && (rc = join_streams(e))) return rc;

  MScanBlock blk;
  MBatch B;
  std::memset(&blk, 0, sizeof(blk));
  std::memset(&B, 0, sizeof(B));
  size_t max_n = 0;
  unsigned blocks = 0;
  ScanParams P;
  for (uint32_t k = 0; k < count; ++k) {
    const fdm_device_scan& s = scans[k];
    fill_integrate_params(e, P, s.T_base_sensor, s.T_world_base);
    MScan& m = blk.s[k];
    std::memcpy(m.Tbs, P.Tbs, sizeof(m.Tbs));
    std::memcpy(m.Twb, P.Twb, sizeof(m.Twb));
    std::memcpy(m.R, P.R, sizeof(m.R));
    m.n = uint32_t(s.n);
    m.robot_x = P.robot_x;
    m.robot_y = P.robot_y;
    m.x = s.x; m.y = s.y; m.z = s.z; m.intensity = s.intensity; m.rgb = s.rgb; m.var = s.sigma_z2;
    m.scan_no = uint32_t(e->scan_no + k);
    B.first_block[k] = blocks;
    B.robot_x[k] = P.robot_x;
    B.robot_y[k] = P.robot_y;
    blocks += unsigned((s.n + 255u) / 256u);
    max_n = std::max<size_t>(max_n, s.n);
  }
  for (uint32_t k = count; k <= uint32_t(kMaxBatch); ++k) B.first_block[k] = blocks;
  if ((rc = ensure_multi(e, max_n, blocks))) return rc;

  const int par = e->mparity;
  e->mparity ^= 1;
  B.count = count;
  B.scan_no0 = uint32_t(e->scan_no);
  B.scans = e->mscans + size_t(par) * kMaxBatch;
  B.ms = e->mstate + par;
  const bool fuse = e->chain && e->pend.multi;  // (same channels: checked above)
  B.prev = fuse ? e->pend.MB.ms : nullptr;
  B.prev_count = fuse ? e->pend.MB.count : 0u;
  B.obs_stride = unsigned(e->mobs_stride);
  B.key = e->mkey[par];
  B.aux = e->maux[par];
  B.zs = e->mzs[par];
  B.obs = e->mobs[par];
  B.cobs = e->mcobs[par];
  B.bin_part = e->mbin_part[par];
  B.upd_part = e->S.upd_part;
  B.min_sq = P.min_sq; B.max_sq = P.max_sq; B.z_min = P.z_min; B.z_max = P.z_max;
  sensor_params(e->cfg, B.sensor_type, B.sp);
  B.integrate_mode = 1;
  B.do_move = P.do_move;
  B.gate_on_filter = P.gate_on_filter;
  B.has_var = hv ? 1 : 0;
  B.bin_table = e->bin_table;

  hipLaunchKernelGGL(k_mput, dim3(1), dim3(256), 0, e->stream, blk, const_cast<MScan*>(B.scans), count, B.ms);
  HIPCK(hipGetLastError());
  if (fuse) {
    const fdm_engine::PendingUpdate& u = e->pend;
    e->chain = false;
    rc = with_policy(e, [&](auto tag, const auto& layers) -> int {
      using POLICY = decltype(tag);
      if constexpr (is_rec_policy<POLICY>) {
        return with_channels(ch, [&](auto chc) -> int {
          constexpr int CH = decltype(chc)::value;
          hipLaunchKernelGGL((k_mupdate_mbin<POLICY, CH>), dim3(e->n_tiles + blocks), dim3(256), 0, e->stream, u.MB,
                             e->G, e->d_state, layers, e->d_layer_ptrs, e->n_layer_ptrs, unsigned(e->ncell),
                             e->n_tiles, B);
          HIPCK(hipGetLastError());
          return FDM_OK;
        });
      } else {
        return fail(FDM_ERR_INVALID, "internal: batch update with a per-layer policy");
      }
    });
    if (rc) return rc;
  } else {
    rc = with_channels(ch, [&](auto chc) -> int {
      constexpr int CH = decltype(chc)::value;
      hipLaunchKernelGGL((k_mbin<CH>), dim3(blocks), dim3(256), 0, e->stream, B, e->G, e->d_state, unsigned(e->ncell));
      HIPCK(hipGetLastError());
      return FDM_OK;
    });
    if (rc) return rc;
  }
  // this batch's update is held back
  e->pend.multi = true;
  e->pend.MB = B;
  e->pend.ch = ch;
  e->pend.tiled = false;
  e->chain = true;
  e->last_do_move = P.do_move;
  e->last_gate = P.gate_on_filter;
  // bookkeeping as enqueue_scan leaves it after the batch's last scan
  const fdm_device_scan& l = scans[count - 1u];
  e->last_kind = 0;
  e->last_bin_blocks = blocks - B.first_block[count - 1u];
  e->last_bin_part = B.bin_part + B.first_block[count - 1u];
  e->last_upd_tiles = e->n_tiles;
  e->last_upd_part = e->S.upd_part;
  e->ray_timed = false;
  e->scan_no += count;
  e->have_scan = true;
  e->last_n = uint32_t(l.n);
  e->last_n_input = uint32_t(l.n);
  e->ingest_blocks = 0;
  e->last_was_integrate = 1;
  return FDM_OK;
}
