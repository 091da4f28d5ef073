// fdm_engine_opts.inl — scan-callback captures, per-point cell ids, the per-launch profile and fdm_engine_set_option.
// Part of fdm_engine.hip's translation unit (inside its extern "C" block): do not compile on its own.

int fdm_engine_capture(fdm_engine* e, int preprocessed, int rasterized) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  e->cap_pre = preprocessed != 0;
  e->cap_cov = preprocessed == 2;
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

int fdm_engine_last_preprocessed_cov(fdm_engine* e, uint64_t cap, float* cov9, uint64_t* n_out) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e || !n_out || !cov9) return fail(FDM_ERR_INVALID, "null argument");
  *n_out = 0;
  if (!e->cap_pre || !e->cap_cov) return fail(FDM_ERR_INVALID, "covariance capture is off (fdm_engine_capture(e, 2, ..))");
  const size_t n = e->last_n;
  if (!e->have_scan || n == 0 || !e->d_cap || !e->d_cell_ids) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  std::vector<float> h(9 * n);
  std::vector<int32_t> ids(n);
  for (int c = 0; c < 9; ++c)
    HIPCK(hipMemcpy(h.data() + c * n, e->d_cap + (4 + c) * e->cap_cap, n * sizeof(float), hipMemcpyDeviceToHost));
  HIPCK(hipMemcpy(ids.data(), e->d_cell_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  uint64_t w = 0;
  for (size_t i = 0; i < n; ++i) {  // the same order-preserving compaction as fdm_engine_last_preprocessed
    if (ids[i] == -1) continue;
    if (w < cap)
      for (int c = 0; c < 9; ++c) cov9[w * 9 + c] = h[c * n + i];
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
  if (std::strcmp(key, "batch") == 0) {  // fdm_engine_integrate_device_batch: group small scans into batch launches
    e->batch = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "voxel_small") == 0) {  // 0: every scan through the library sort
    e->voxel_small = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "voxel_small_max") == 0) {
    if (value < 1 || value > (1 << 20)) return fail(FDM_ERR_INVALID, "voxel_small_max: 1 .. 2^20 points");
    e->voxel_small_max = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_batch") == 0) {
    e->dbg_batch = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_walk") == 0) {
    if (value < -1 || value > 1) return fail(FDM_ERR_INVALID, "batch_walk: -1 (automatic), 0, 1");
    e->batch_walk = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_crop") == 0) {
    e->batch_crop = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_ray") == 0) {  // 0: scans of an engine with raycasting on leave one by one
    e->batch_ray = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_ray_lds") == 0) {
    e->batch_ray_lds = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_ray_words") == 0) {
    if (value < 0) return fail(FDM_ERR_INVALID, "batch_ray_words: >= 0");
    e->batch_ray_words = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_ray_parts") == 0) {
    if (value < 0 || value > 64) return fail(FDM_ERR_INVALID, "batch_ray_parts: 0 (automatic) .. 64");
    e->batch_ray_parts = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_ray_seg") == 0) {
    if (value != 1 && value != 4 && value != 8 && value != 16) return fail(FDM_ERR_INVALID, "batch_ray_seg: 1, 4, 8 or 16 lanes per ray");
    e->batch_ray_seg = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_fuse") == 0) {
    e->batch_fuse = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "batch_max") == 0) {
    if (value != 0 && (value < 2 || value > kMaxBatch)) return fail(FDM_ERR_INVALID, "batch_max: 0 (automatic) or 2 .. 32 scans per launch");
    e->batch_max = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "bin_delay") == 0 || std::strcmp(key, "bin_delay_blocks") == 0) {
    if (value < 0 || value > 65535) return fail(FDM_ERR_INVALID, "bin_delay: 0 .. 65535");
    if (key[9] == '_') e->bin_delay_blocks = value; else e->bin_delay = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "bin_stagger") == 0) {
    if (value < 0 || value > 64) return fail(FDM_ERR_INVALID, "bin_stagger: 0 .. 64");
    e->bin_stagger = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "move_clear_basic") == 0) {  // which layers GridMap::move()'s strips clear (see fdm_engine::move_clear_basic)
    if (int rc_sync = sync_all(e)) return rc_sync;
    e->move_clear_basic = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "cnt_shift") == 0) {
    if (value < 0 || value > 5) return fail(FDM_ERR_INVALID, "cnt_shift: 0 .. 5");
    if (e->pool[0].cnt) return fail(FDM_ERR_INVALID, "cnt_shift: the record pools exist already");
    e->cnt_shift = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "tiled_lds_pad") == 0) {
    if (value < -1 || value > 120 * 1024) return fail(FDM_ERR_INVALID, "tiled_lds_pad: -1 (automatic) or 0 .. 122880 bytes");
    e->tiled_lds_pad = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "upd_prio") == 0) {
    e->upd_prio = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "upd_blocks") == 0 || std::strcmp(key, "upd_blocks_alone") == 0) {  // update blocks of a large-scan launch
    if (value < 1 || value > 65535) return fail(FDM_ERR_INVALID, "upd_blocks: 1 .. 65535");
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (key[10] == '_') e->upd_blocks_alone = value; else e->upd_blocks = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "zero_copy") == 0) {
    if (value < 0) return fail(FDM_ERR_INVALID, "zero_copy: a point count (0 = off)");
    e->zero_copy = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_timeline") == 0) {  // measurement only: block start / end ticks of the fused tiled launches
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->d_timeline) { (void)hipFree(e->d_timeline); e->d_timeline = nullptr; }
    e->timeline_cap = 0;
    if (value > 0) {
      e->timeline_cap = 1u << 16;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_timeline), size_t(e->timeline_cap) * 16));
      HIPCK(hipMemset(e->d_timeline, 0, size_t(e->timeline_cap) * 16));
    }
    return FDM_OK;
  }
  if (std::strcmp(key, "sync_spin_us") == 0) {
    e->sync_spin_us = value < 0 ? 0 : value;
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_post") == 0) {
    e->dbg_post = value;
    return FDM_OK;
  }
  // (the ray options are read when a scan's raycasting stage is LAUNCHED; under ray_hold the previous scan's stage may
  // still be pending: it leaves first, so that a switch never lands in the scan before it — results are the same either
  // way, A/B timings are attributed to the right scan: ADVICE r05)
  if (std::strcmp(key, "ray_large_min") == 0) {
    if (int rc = join_streams(e)) return rc;
    e->ray_large_min = value < 1 ? 1 : value;
    return FDM_OK;
  }
  if (std::strcmp(key, "ray_hold") == 0) {
    if (int rc = join_streams(e)) return rc;
    e->ray_hold = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "ray_wedge") == 0) {
    if (int rc = join_streams(e)) return rc;
    e->ray_wedge = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "ray_overlap") == 0) {  // two raycasting stages of large scans in flight (fdm_engine_ray.inl)
    if (value < -1 || value > 1) return fail(FDM_ERR_INVALID, "ray_overlap: -1 (automatic), 0, 1");
    if (int rc = sync_all(e)) return rc;
    e->ray_overlap = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "ray_wedge_parts") == 0) {  // workgroups per sector of k_ray_wedge (0 = by the scan's size)
    if (value < 0 || value > 16) return fail(FDM_ERR_INVALID, "ray_wedge_parts: 0 .. 16");
    if (int rc = join_streams(e)) return rc;
    e->ray_wedge_parts = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "dbg_ray") == 0) {
    if (int rc = join_streams(e)) return rc;
    e->dbg_ray = value;
    return FDM_OK;
  }
  if (std::strcmp(key, "tiled") == 0) {  // large scans through per-tile record pools (1, default) or the per-cell scratch (0)
    e->tiled = value != 0;
    return FDM_OK;
  }
  if (std::strcmp(key, "borrow_inputs") == 0) {  // 1: device arrays of enqueue-only scans stay untouched by the caller
    e->borrow_inputs = value != 0;              //    until the NEXT-BUT-ONE scan is enqueued (or a flush): no staging copy
    return FDM_OK;
  }
  if (std::strcmp(key, "tiled_min") == 0) {
    if (value < 0) return fail(FDM_ERR_INVALID, "tiled_min: a point count");
    e->tiled_min = unsigned(value);
    e->tiled_forced = true;
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
    e->obst_owe_armed = false;
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
