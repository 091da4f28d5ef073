// fdm_engine_tbatch.inl — host side of the tile-batch pipeline (fdm_tbatch.hpp): which scans of a
// fdm_engine_integrate_device_batch call may leave as one tile batch, the per-scan record pools, the launches.
// Part of fdm_engine.hip's translation unit (inside its anonymous namespace): do not compile on its own.

// How many of the leading `count` scans can leave as ONE tile batch (0 or 1: take the single-scan path).  A tile batch
// is a run of plain LARGE scans — exactly the scans enqueue_scan would send through the record pools, hold back and
// fuse — from ONE sensor (same T_base_sensor) and with the same optional channels.
uint32_t tbatch_run(fdm_engine* e, uint32_t count, const fdm_device_scan* scans) {
  if (!e->tbatch || count < 2u || !e->overlap || !e->tiled || e->bin_variant == 1) return 0u;
  if (!e->estimator_ready || e->rec_kind < 0 || !e->key2[1]) return 0u;
  if (e->cfg.raycast_enabled || e->cap_pre || e->cap_ras || e->want_ids || e->profile) return 0u;
  if (e->obst_dense_pending || e->last_kind == 0 || e->next_drop_nonfinite) return 0u;  // (0: the scratch pipeline keeps its own obstacle books)
  if (e->dbg_no_atomics || e->dbg_upd) return 0u;
  const fdm_device_scan& f = scans[0];
  const size_t kt = e->ncell / kTileCells;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  uint32_t run = 0;
  const uint32_t cap = std::min<uint32_t>(count, uint32_t(e->tbatch_max));
  for (; run < cap; ++run) {
    const fdm_device_scan& s = scans[run];
    if (s.n == 0 || !s.x || !s.y || !s.z) break;
    if (s.n < e->tbatch_min || s.n < e->tiled_min || s.n >= 0xF00000ull) break;  // (4 wavefronts x 15 360 bin blocks: 16-bit counters)
    const bool enough_tiles = e->tiled_forced || kt >= 240 || (kt >= 160 && s.n >= 100000);
    if (!enough_tiles) break;
    if (!al16(s.x) || !al16(s.y) || !al16(s.z) || !al16(s.intensity)) break;
    if ((s.intensity != nullptr) != (f.intensity != nullptr) || (s.rgb != nullptr) != (f.rgb != nullptr) ||
        (s.sigma_z2 != nullptr) != (f.sigma_z2 != nullptr))
      break;
    if (!affine_last_row(s.T_base_sensor) || !affine_last_row(s.T_world_base)) break;
    if (std::memcmp(s.T_base_sensor, f.T_base_sensor, 16 * sizeof(double)) != 0) break;
  }
  return run >= 2u ? run : 0u;
}

// Tiles whose chunk counts one wavefront of an update group reads per queue pop: 1 while the tile count keeps the
// groups busy by itself, kTBSpanMax on very large maps (nearly every tile idle).
unsigned tbatch_span(const fdm_engine* e) { return e->TG.n_tiles <= 16384u ? 1u : kTBSpanMax; }

int ensure_tbatch(fdm_engine* e, size_t max_n, unsigned max_blocks, unsigned total_blocks, uint32_t count) {
  int rc;
  if ((rc = ensure_mstate(e))) return rc;
  if ((rc = ensure_tile_aux(e))) return rc;
  const size_t need_rec = size_t(max_blocks) * 1024u;
  (void)max_n;
  if (int(count) > e->tb_slots || need_rec > e->tb_rec_stride || max_blocks + 1u > e->tb_stride) {
    if ((rc = sync_all(e))) return rc;  // (every chunk list is consumed: the row counts are all zero)
    e->tb_slots = std::max(e->tb_slots, std::max(int(count), e->tbatch_max));
    if (need_rec > e->tb_rec_stride) e->tb_rec_stride = need_rec + need_rec / 4 + 8192;
    if (e->tb_rec_stride >= 0x7FFFFFF0ull) return fail(FDM_ERR_INVALID, "tile batch: scan too large");
    if (max_blocks + 1u > e->tb_stride) e->tb_stride = max_blocks + max_blocks / 4 + 16;
    const size_t desc_words = size_t(e->tb_slots) * e->TG.n_tiles * e->tb_stride;
    for (int k = 0; k < 2; ++k) {
      if (e->tb_rec[k]) (void)hipFree(e->tb_rec[k]);
      if (e->tb_desc[k]) (void)hipFree(e->tb_desc[k]);
      e->tb_rec[k] = nullptr;
      e->tb_desc[k] = nullptr;
    }
    for (int k = 0; k < 2; ++k) {
      if (hipMalloc(reinterpret_cast<void**>(&e->tb_rec[k]), size_t(e->tb_slots) * e->tb_rec_stride * sizeof(TileRec)) != hipSuccess ||
          hipMalloc(reinterpret_cast<void**>(&e->tb_desc[k]), desc_words * 8) != hipSuccess) {
        (void)hipGetLastError();
        for (int q = 0; q < 2; ++q) {
          if (e->tb_rec[q]) (void)hipFree(e->tb_rec[q]);
          if (e->tb_desc[q]) (void)hipFree(e->tb_desc[q]);
          e->tb_rec[q] = nullptr;
          e->tb_desc[q] = nullptr;
        }
        e->tb_slots = 0; e->tb_rec_stride = 0; e->tb_stride = 0;
        return fail(FDM_ERR_HIP, "tile batch: out of device memory for the record pools");
      }
      HIPCK(hipMemsetAsync(e->tb_desc[k], 0, desc_words * 8, e->stream));
    }
  }
  if (total_blocks > e->tb_bin_cap) {
    if ((rc = sync_all(e))) return rc;
    e->tb_bin_cap = size_t(total_blocks) + total_blocks / 4 + 64;
    for (int k = 0; k < 2; ++k) {
      if (e->tb_bin_part[k]) HIPCK(hipFree(e->tb_bin_part[k]));
      e->tb_bin_part[k] = nullptr;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->tb_bin_part[k]), e->tb_bin_cap * sizeof(unsigned long long)));
    }
  }
  return FDM_OK;
}

// One k_tbatch launch: [ update U | bin B | scouts C ], any of which may be empty (count == 0).
int launch_tbatch(fdm_engine* e, bool hi, bool hc, const TBUpd& U, const TBBin& B, const TBScout& C, const TBCommon& K) {
  static_assert(sizeof(TBUpd) + sizeof(TBBin) + sizeof(TBScout) + sizeof(TBCommon) + sizeof(GeomConst) + sizeof(TileGrid) + sizeof(TileAux) + 200 <= 4096,
                "k_tbatch: kernel arguments beyond 4 KB");
  const unsigned ug = U.count ? U.n_groups : 0u;
  const unsigned bb = B.count ? B.first_block[B.count] : 0u;
  const unsigned sb = C.count * kScoutBlocks;
  if (ug + bb + sb == 0u) return FDM_OK;
  const unsigned lds = std::max(tile_lds_bytes(hi, hc), tbin_lds_bytes(hi, hc, 256u));
  TBCommon Kt = K;
  Kt.timeline = (e->d_timeline && ug + bb <= e->timeline_cap && B.count) ? e->d_timeline : nullptr;
  if (Kt.timeline) { e->timeline_blocks = ug + bb; e->timeline_upd = ug; e->timeline_bin = bb; }
  const TileAux A{e->tile_stamp32, e->upd_part32, nullptr, nullptr};
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    if constexpr (is_rec_policy<POLICY>) {
      int rc = FDM_OK;
      auto go = [&](auto kern) {
        if ((rc = allow_lds(kern, lds))) return;
        hipLaunchKernelGGL(kern, dim3(ug + bb + sb), dim3(256), lds, e->stream, U, B, C, Kt, e->G, e->TG, e->d_state, layers,
                           e->d_layer_ptrs, e->n_layer_ptrs, A, ug);
      };
      if (hi && hc) go(k_tbatch<POLICY, true, true>);
      else if (hi) go(k_tbatch<POLICY, true, false>);
      else if (hc) go(k_tbatch<POLICY, false, true>);
      else go(k_tbatch<POLICY, false, false>);
      if (rc) return rc;
      HIPCK(hipGetLastError());
      return FDM_OK;
    } else {
      return fail(FDM_ERR_INVALID, "internal: tile batch with a per-layer policy");
    }
  });
}

// The held-back update of a tile batch on its own (launch_update_alone forwards here).
int launch_tbatch_update(fdm_engine* e, const fdm_engine::PendingUpdate& u) {
  TBBin B;
  TBScout C;
  TBCommon K;
  std::memset(&B, 0, sizeof(B));
  std::memset(&C, 0, sizeof(C));
  std::memset(&K, 0, sizeof(K));
  return launch_tbatch(e, u.tb_hi, u.tb_hc, u.TU, B, C, K);
}

// Scout flags written for a batch that did not come (the caller's next scans took another path): the state they were
// written into is cleaned before anybody reads it.
int drop_scouted(fdm_engine* e) {
  if (!e->tpre_valid) return FDM_OK;
  e->tpre_valid = false;
  MState* const ms = e->mstate + int(e->tpre_seq % unsigned(kMStates));
  HIPCK(hipMemsetAsync(ms->done, 0, sizeof(ms->done), e->stream));
  return FDM_OK;
}

void fill_scout(TBScout& C, MState* ms, uint32_t count, const fdm_device_scan* scans) {
  std::memset(&C, 0, sizeof(C));
  C.count = count;
  C.ms = ms;
  for (uint32_t k = 0; k < count; ++k) {
    C.n[k] = uint32_t(scans[k].n);
    C.px[k] = scans[k].x; C.py[k] = scans[k].y; C.pz[k] = scans[k].z;
  }
}

// `count` (2 .. tbatch_max) scans that tbatch_run() accepted leave as ONE launch: the held-back update of the previous
// tile batch (when there is one), this batch's bin and — `next_count` >= 2 — the scouts of the batch the caller will
// enqueue next (scans[count .. count + next_count)).  This batch's update is held back.
int enqueue_tbatch(fdm_engine* e, uint32_t count, const fdm_device_scan* scans, uint32_t next_count) {
  int rc;
  const fdm_device_scan& f = scans[0];
  const bool hi = f.intensity != nullptr, hc = f.rgb != nullptr, hv = f.sigma_z2 != nullptr;
  if ((rc = ensure_scratch_channels(e, hi, hc))) return rc;
  if ((rc = refresh_layer_ptrs(e))) return rc;
  // a held-back update of another kind (single scan, small-scan batch, other channels) leaves first
  if (e->chain && !(e->pend.tb && e->pend.tb_hi == hi && e->pend.tb_hc == hc) && (rc = join_streams(e))) return rc;

  TBBin B;
  TBCommon K;
  TBUpd U;
  std::memset(&B, 0, sizeof(B));
  std::memset(&K, 0, sizeof(K));
  std::memset(&U, 0, sizeof(U));
  size_t max_n = 0;
  unsigned blocks = 0, max_blocks = 0;
  ScanParams P;
  for (uint32_t k = 0; k < count; ++k) {
    const fdm_device_scan& s = scans[k];
    fill_integrate_params(e, P, s.T_base_sensor, s.T_world_base);
    std::memcpy(B.Twb[k], P.Twb, sizeof(P.Twb));
    std::memcpy(B.R[k], P.R, sizeof(P.R));
    B.n[k] = uint32_t(s.n);
    B.px[k] = s.x; B.py[k] = s.y; B.pz[k] = s.z; B.pint[k] = s.intensity; B.prgb[k] = s.rgb; B.pvar[k] = s.sigma_z2;
    B.first_block[k] = blocks;
    B.robot_x[k] = P.robot_x;
    B.robot_y[k] = P.robot_y;
    const unsigned nb = unsigned((s.n + 1023u) / 1024u);
    blocks += nb;
    max_blocks = std::max(max_blocks, nb);
    max_n = std::max<size_t>(max_n, s.n);
  }
  for (uint32_t k = count; k <= uint32_t(kTBMax); ++k) B.first_block[k] = blocks;
  if ((rc = ensure_tbatch(e, max_n, max_blocks, blocks, count))) return rc;

  std::memcpy(K.Tbs, P.Tbs, sizeof(K.Tbs));
  sensor_params(e->cfg, K.sensor_type, K.sp);
  K.min_sq = P.min_sq; K.max_sq = P.max_sq; K.z_min = P.z_min; K.z_max = P.z_max;
  K.do_move = P.do_move;
  K.gate_on_filter = P.gate_on_filter;
  K.has_var = hv ? 1 : 0;
  K.dbg = e->dbg_batch;

  const unsigned seq = e->mseq++;
  const int slot = int(seq % unsigned(kMStates)), par = int(seq & 1u);
  if (e->pre_valid)  // (crop bits a small-scan batch launch left for a batch that never came)
    HIPCK(hipMemsetAsync(e->mstate[slot].flags, 0, sizeof(unsigned) * kLineWords, e->stream));
  e->pre_valid = false;
  // "scan k has a point that survives the crops" (it moves a LOCAL map, fastdem.cpp:138-145) must be known when the bin
  // blocks start: the scouts of the previous launch left it in this batch's state — if they scouted THIS batch;
  // otherwise a small launch of scouts runs ahead of the batch (the first batch of a chain: once per call)
  const bool gated = K.do_move && K.gate_on_filter;
  const bool scouted = e->tpre_valid && e->tpre_scans == scans && e->tpre_count == count && e->tpre_seq == seq;
  if (e->tpre_valid && !scouted && (rc = drop_scouted(e))) return rc;
  e->tpre_valid = false;
  TBScout Cs;
  std::memset(&Cs, 0, sizeof(Cs));
  if (gated && !scouted) {
    TBUpd U0;
    TBBin B0;
    std::memset(&U0, 0, sizeof(U0));
    std::memset(&B0, 0, sizeof(B0));
    fill_scout(Cs, e->mstate + slot, count, scans);
    if ((rc = launch_tbatch(e, hi, hc, U0, B0, Cs, K))) return rc;
    std::memset(&Cs, 0, sizeof(Cs));
  }
  if (gated && next_count >= 2u && std::memcmp(scans[count].T_base_sensor, f.T_base_sensor, 16 * sizeof(double)) == 0) {
    fill_scout(Cs, e->mstate + int((seq + 1u) % unsigned(kMStates)), next_count, scans + count);
    e->tpre_valid = true;
    e->tpre_scans = scans + count;
    e->tpre_count = next_count;
    e->tpre_seq = seq + 1u;
  }
  const bool fuse = e->chain && e->pend.tb;  // (same channels: checked above)
  B.count = count;
  B.scan_no0 = uint32_t(e->scan_no);
  B.ms = e->mstate + slot;
  B.prev = fuse ? e->pend.TU.ms : nullptr;
  B.prev_count = fuse ? e->pend.TU.count : 0u;
  B.stride = e->tb_stride;
  B.rec0 = e->tb_rec[par];
  B.desc0 = e->tb_desc[par];
  B.rec_stride = e->tb_rec_stride;
  B.desc_stride = size_t(e->TG.n_tiles) * e->tb_stride;
  B.rare = e->tile_rare;
  B.bin_part = e->tb_bin_part[par];
  if (fuse) U = e->pend.TU;
  e->chain = false;
  if ((rc = launch_tbatch(e, hi, hc, U, B, Cs, K))) return rc;

  // this batch's update is held back (option "batch_fuse" 0: launched at once, for per-kernel measurements)
  TBUpd& N = e->pend.TU;
  std::memset(&N, 0, sizeof(N));
  N.count = count;
  N.scan_no0 = B.scan_no0;
  N.do_move = K.do_move;
  N.gate_on_filter = K.gate_on_filter;
  N.span = tbatch_span(e);
  N.n_pops = (e->TG.n_tiles + N.span - 1u) / N.span;
  // (the rare-path scratch is sized per update group of the one-scan launches: ensure_tile_aux)
  N.n_groups = std::min<unsigned>(std::min<unsigned>(N.n_pops, unsigned(e->tb_groups)), e->TG.n_tiles / tile_span(e) + 2u);
  N.stride = B.stride;
  N.ms = B.ms;
  N.rearm = e->mstate + int((seq + 3u) % unsigned(kMStates));
  N.rec0 = B.rec0; N.desc0 = B.desc0; N.rec_stride = B.rec_stride; N.desc_stride = B.desc_stride; N.rare = B.rare;
  e->pend.multi = false;
  e->pend.tb = true;
  e->pend.tb_hi = hi;
  e->pend.tb_hc = hc;
  e->pend.tiled = true;
  e->chain = true;
  e->last_do_move = P.do_move;
  e->last_gate = P.gate_on_filter;
  if (!e->batch_fuse && (rc = join_streams(e))) return rc;
  // bookkeeping as enqueue_scan leaves it after the batch's last scan
  const fdm_device_scan& l = scans[count - 1u];
  e->last_kind = 1;
  e->last_bin_blocks = blocks - B.first_block[count - 1u];
  e->last_bin_part = B.bin_part + B.first_block[count - 1u];
  e->last_upd_tiles = e->TG.n_tiles;
  e->last_upd_part = e->upd_part32;
  e->ray_timed = false;
  e->scan_no += count;
  e->last_batch_n = int(count);
  ++e->n_tbatch;
  e->have_scan = true;
  e->last_n = uint32_t(l.n);
  e->last_n_input = uint32_t(l.n);
  e->ingest_blocks = 0;
  e->last_was_integrate = 1;
  return FDM_OK;
}
