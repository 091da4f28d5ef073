// fdm_engine.hip — C ABI (include/fdm_engine.h) over the HIP kernels.  gfx950 only.
//
// Host responsibilities (all O(1) per scan): cast the two Isometry3d matrices to float,
// form R = (T_wb*T_bs).rotation().cast<float>(), pick the ring slot, launch k_bin and
// k_update on the engine's stream.  No per-point or per-cell work ever runs on the CPU and
// there is NO CPU fallback: without a HIP device fdm_engine_create fails with
// FDM_ERR_NO_DEVICE.
// (One of the library's three translation units: fdm_engine_host.hpp.)
#include "fdm_engine_host.hpp"

namespace fdmh {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace fdmh

namespace fdmh {

// Launch the held-back update kernel, if any.  Called at the top of every entry point that is not
// the next scan of the chain, and before anything that syncs or reallocates.
// The raycasting stage of the scan whose update was just launched (fdm_engine_ray.inl).
int join_streams(fdm_engine* e) {
  if (e->chain) {
    e->chain = false;
    if (int rc = launch_update_alone(e, e->pend)) return rc;
    if (int rc = run_held_ray_stage(e, e->pend)) return rc;
  }
  return FDM_OK;
}
// Has a scan that observed a cell paid the obstacle layer's debt?  k_obstacle_dense_paid leaves the number it paid for
// in pinned host memory: no stream wait — an enqueue-only caller gets its fused / batch launches back a scan or two
// after the switch, not at its next sync.  (A number read early only means one more scan tries; a number read is
// final: a new debt takes a new number.)
void poll_dense_paid(fdm_engine* e) {
  if (!e->obst_dense_pending || !e->obst_owe_armed || !e->h_stats) return;
  if (__atomic_load_n(&e->h_stats->dense_paid, __ATOMIC_ACQUIRE) == e->obst_owe_seq) e->obst_dense_pending = false;
}
int sync_all(fdm_engine* e) {
  if (int rc = join_streams(e)) return rc;
  HIPCK(hipStreamSynchronize(e->stream));
  for (hipStream_t rs : e->ray_stream)  // (every early stage part has its resolve on the main stream behind it by now: idle)
    if (rs) HIPCK(hipStreamSynchronize(rs));
  e->bstage_busy = false;  // (the stream has drained: nothing reads the host-batch staging block any more)
  poll_dense_paid(e);
  return FDM_OK;
}
// DevState::fault after the stream has drained: a batch launch whose in-kernel wait for the scans ahead ran out of
// polls (fdm_multi.hpp).  Sticky on the device until it has been reported ONCE — the maps of that
// batch are undefined, reset() and go on.  Only looked at when such a launch was enqueued since the last look.
int report_fault(fdm_engine* e) {
  if (!e->fault_watch) return FDM_OK;
  e->fault_watch = false;
  unsigned f = 0u;
  HIPCK(hipMemcpy(&f, &e->d_state->fault, sizeof(f), hipMemcpyDeviceToHost));
  if (!f) return FDM_OK;
  HIPCK(hipMemset(&e->d_state->fault, 0, sizeof(f)));
  return fail(FDM_ERR_HIP, "device-side fault: a batch's geometry-chain wait ran out of polls (the batch's map update is undefined)");
}

Layer* find_layer(fdm_engine* e, const char* name) {
  for (auto& l : e->layers)
    if (l.name == name) return &l;
  return nullptr;
}

// A layer as the kernels see it: base pointer + element stride (1, or the record size).
float* lptr(fdm_engine* e, const Layer& l) { return l.field >= 0 ? e->d_rec + l.field : l.d; }
int lstride(fdm_engine* e, const Layer& l) { return l.field >= 0 ? e->rec_floats : 1; }

int fill_async(fdm_engine* e, float* p, float v, size_t n, int stride) {
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

int add_layer(fdm_engine* e, const char* name, float value, bool pending) {
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

// With cell records the intensity layer lives IN the record (kKalmanIntSlot / kP2IntSlot): a plain array that exists
// when the records are (re)activated, or is created later, moves there.
int adopt_intensity(fdm_engine* e) {
  if (e->rec_kind < 0) return FDM_OK;
  Layer* l = find_layer(e, "intensity");
  if (!l || l->field >= 0) return FDM_OK;
  const int slot = e->rec_kind == 1 ? kP2IntSlot : kKalmanIntSlot;
  if (int rc = copy_strided(e, e->d_rec + slot, e->rec_floats, l->d, 1)) return rc;
  if (int rc_sync = sync_all(e)) return rc_sync;
  HIPCK(hipFree(l->d));
  l->d = nullptr;
  l->field = slot;
  e->layer_ptrs_dirty = true;
  return FDM_OK;
}

int ensure_scratch_channels(fdm_engine* e, bool intensity, bool color) {
  int rc;
  if (intensity && !find_layer(e, "intensity")) {
    if ((rc = add_layer(e, "intensity", NAN, true))) return rc;
    if ((rc = adopt_intensity(e))) return rc;
  }
  if (color && !find_layer(e, "color") && (rc = add_layer(e, "color", NAN, true))) return rc;
  return FDM_OK;
}

float* L(fdm_engine* e, const char* n) {
  Layer* l = find_layer(e, n);
  return l ? l->d : nullptr;
}
// a layer that may be a record field: pointer to its element 0 and the distance between elements
float* Lany(fdm_engine* e, const char* n, int* stride) {
  Layer* l = find_layer(e, n);
  *stride = l ? lstride(e, *l) : 1;
  return l ? lptr(e, *l) : nullptr;
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
  return adopt_intensity(e);
}

// ---- launch helpers -------------------------------------------------------------------------
P2Params p2_params(const fdm_config& c) {  // P2Quantile ctor (quantile_estimation.hpp:84-95): clamp, then monotone dn
  P2Params p2{};
  auto clamp01 = [](float v) { return v < 0.f ? 0.f : (1.f < v ? 1.f : v); };
  for (int k = 0; k < 5; ++k) p2.dn[k] = clamp01(c.p2_dn[k]);
  for (int k = 1; k < 5; ++k) p2.dn[k] = std::max(p2.dn[k], p2.dn[k - 1]);
  p2.marker = std::min(std::max(c.p2_elevation_marker, 0), 4);
  p2.max_count = std::max(c.p2_max_sample_count, 0.0f);
  return p2;
}

#include "fdm_engine_launch.inl"  // with_policy, record pools, launch_tbin / launch_update_alone / launch_update_fused

// SensorModel parameters as the kernels take them (sensor_type 0 Constant, 1 LiDAR, 2 RGB-D)
void sensor_params(const fdm_config& cfg, int& type, float* sp) {
  type = cfg.sensor_type;
  if (type == 2) {
    sp[0] = cfg.rgbd_normal_a;
    sp[1] = cfg.rgbd_normal_b;
    sp[2] = cfg.rgbd_normal_c;
    sp[3] = cfg.rgbd_lateral_factor;
  } else if (type == 0) {
    sp[0] = cfg.constant_uncertainty;
    sp[1] = sp[2] = sp[3] = 0.f;
  } else {  // LiDARSensorModel ctor takes |noise| (lidar_model.hpp:58-62); unknown -> LiDAR (sensor_model.cpp:34-38)
    type = 1;
    sp[0] = std::fabs(cfg.lidar_range_noise);
    sp[1] = std::fabs(cfg.lidar_angular_noise);
    sp[2] = sp[3] = 0.f;
  }
}

int ensure_stage(fdm_engine* e, size_t n);

// One scan = k_bin + k_update on the stream.  All pointers are device pointers.
// `gather` (nullable): where the UPDATE kernel reads the winning points from.  Set when dx..dvar are
// pinned host arrays seen through PCIe: the bin kernel then writes the scan through to these HBM
// arrays as it reads it (Scratch::wt_x), so the scan crosses the link exactly once and no copy is
// queued.  Null: the update gathers from dx..dvar themselves.
int enqueue_scan(fdm_engine* e, ScanParams& P, uint64_t n, const float* dx, const float* dy,
                 const float* dz, const float* dint, const uint32_t* drgb, const float* dvar,
                 const ScanInputs* gather) {
  if (n >= 0xFFFFFFFEull) return fail(FDM_ERR_INVALID, "point count exceeds 2^32-2");
  if (n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1");
  int rc;
  poll_dense_paid(e);
  P.n = uint32_t(n);
  P.scan_no = uint32_t(e->scan_no);
  P.slot = int(e->scan_no & 3);
  P.has_intensity = dint != nullptr;
  P.has_color = drgb != nullptr;
  P.has_var = dvar != nullptr;
  sensor_params(e->cfg, P.sensor_type, P.sp);
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
  e->S.cap_x = e->S.cap_y = e->S.cap_z = e->S.cap_var = e->S.cap_cov = nullptr;
  e->S.cap_stride = 0;
  e->S.ras_z = nullptr;
  e->S.cap_drop_nan = ray_on ? 1 : 0;
  e->S.wt_x = e->S.wt_y = e->S.wt_z = e->S.wt_var = nullptr;
  e->S.wt_rgb = nullptr;
  e->S.wt_src_var = nullptr;
  e->S.wt_src_rgb = nullptr;
  if (gather && n) {
    e->S.wt_x = const_cast<float*>(gather->x);
    e->S.wt_y = const_cast<float*>(gather->y);
    e->S.wt_z = const_cast<float*>(gather->z);
    if (dvar) { e->S.wt_var = const_cast<float*>(gather->var); e->S.wt_src_var = dvar; }
    if (drgb) { e->S.wt_rgb = const_cast<uint32_t*>(gather->rgb); e->S.wt_src_rgb = drgb; }
  }
  // (a scan with raycasting and nothing else optional is held back like a plain one: its stage runs behind its update)
  const bool ray_held = ray_on && e->ray_hold && e->overlap && e->key2[1] && !e->cap_pre && !e->cap_ras;
  if (ray_held && n) {
    if (n > e->rcap_cap) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      for (float*& p : e->d_rcap) { if (p) HIPCK(hipFree(p)); p = nullptr; }
      e->rcap_cap = (n + n / 4 + 1024 + 63) & ~size_t(63);  // (a multiple of 64 points: every channel starts 16-byte aligned)
      for (float*& p : e->d_rcap) HIPCK(hipMalloc(reinterpret_cast<void**>(&p), e->rcap_cap * 3 * sizeof(float)));
    }
    float* const base = e->d_rcap[e->scan_no & 1];
    e->S.cap_x = base;
    e->S.cap_y = base + e->rcap_cap;
    e->S.cap_z = base + 2 * e->rcap_cap;
  } else if ((e->cap_pre || ray_on) && n) {
    if (n > e->cap_cap) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      if (e->d_cap) HIPCK(hipFree(e->d_cap));
      e->cap_cap = (n + n / 4 + 1024 + 63) & ~size_t(63);  // (a multiple of 64 points: every channel starts 16-byte aligned)
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_cap), e->cap_cap * 13 * sizeof(float)));  // x y z var + 9 cov
    }
    e->S.cap_x = e->d_cap;
    e->S.cap_y = e->d_cap + e->cap_cap;
    e->S.cap_z = e->d_cap + 2 * e->cap_cap;
    if (e->cap_pre) e->S.cap_var = e->d_cap + 3 * e->cap_cap;
    if (e->cap_pre && e->cap_cov) { e->S.cap_cov = e->d_cap + 4 * e->cap_cap; e->S.cap_stride = e->cap_cap; }
  }
  if (e->cap_ras) {
    if (!e->d_ras) HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_ras), e->ncell * sizeof(float)));
    if ((rc = fill_async(e, e->d_ras, NAN, e->ncell))) return rc;
    e->S.ras_z = e->d_ras;
  }

  // ---- launch plan.  One scan = bin + update.  The update of a plain scan is held back and leaves
  // together with the bin of the next scan in ONE launch (k_update_bin / k_tupdate_tbin): the two
  // halves share nothing — the per-scan scratch (or record pool) is double-buffered by scan parity
  // and a chained bin derives its base geometry from slot t (ScanParams::chain_prev).
  const int parity = int(e->scan_no & 1);
  // float4 loads need 16-byte aligned channels
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const bool aligned = al16(dx) && al16(dy) && al16(dz) && al16(dint);
  // large scans: per-tile record pools (fdm_tiled.hpp); needs the cell-record layout
  // ... and a map with enough 32x32 tiles to keep the chip busy with one block per tile (configs[2], 49 tiles:
  // 31 us against 13.8 us through the per-cell scratch; configs[3], 1444 tiles: 40.7 against 51.3 us).  Measured
  // crossover (128-beam scans of 32 K / 262 K points): 100 tiles 11.9 vs 7.4 / 15.3 vs 13.2 us (scratch wins),
  // 169 tiles 11.8 vs 7.7 / 15.0 vs 19.3, 256 tiles 12.6 vs 15.5 / 15.3 vs 24.4, 625 tiles 13.7 vs 14.4 / 16.6 vs 25.5
  const size_t kt = e->ncell / 1024u;  // (the thresholds below were measured in units of 1 024 cells, round 2)
  const bool enough_tiles = e->tiled_forced || kt >= 240 || (kt >= 160 && n >= 100000);
  const bool tiled = e->tiled && e->rec_kind >= 0 && n >= e->tiled_min && aligned && n < 0x7FFF0000ull &&
                     e->bin_variant != 1 && enough_tiles;
  if (e->last_kind >= 0 && e->last_kind != int(tiled)) {  // the pipelines keep separate books on which tiles hold
    e->obst_dense_pending = true;                         // obstacle cells
    e->obst_owe_armed = false;
  }
  e->last_kind = int(tiled);
  if (e->obst_dense_pending && !e->obst_owe_armed) {  // (a new debt: also one a host write of the layer left)
    hipLaunchKernelGGL(k_obstacle_dense_owe, dim3(1), dim3(1), 0, e->stream, e->d_state, ++e->obst_owe_seq);
    HIPCK(hipGetLastError());
    e->obst_owe_armed = true;
  }
  if (e->key2[1]) {  // the scratch set of this scan's parity
    e->S.key = e->key2[parity];
    e->S.aux = e->aux2[parity];
    e->S.zs = e->zs2[parity];
  }
  const bool plain = e->overlap && e->key2[1] && (!ray_on || ray_held) && !e->cap_pre &&
                     !e->cap_ras && !e->obst_dense_pending;
  // A held-back update of the scratch pipeline gathers the winning points AFTER this call has returned and
  // the next scan has been enqueued.  Device arrays handed to the enqueue-only entry points are therefore
  // written through to the engine's rotating staging block by the bin kernel (12 B/point), and the update
  // gathers from that copy: the caller's arrays are free as soon as the bin kernel has run, which is the
  // ordinary stream contract.  (The tiled pipeline never looks at a scan twice.)
  ScanInputs auto_gather{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (!tiled && plain && !gather && n && !e->borrow_inputs) {
    if ((rc = ensure_stage(e, n))) return rc;
    e->stage_rr = (e->stage_rr + 1) % kStageSlots;
    float* base = e->d_stage + size_t(e->stage_rr) * 6 * e->stage_cap;
    const size_t cap = e->stage_cap;
    auto_gather = ScanInputs{base, base + cap, base + cap * 2, nullptr,
                             reinterpret_cast<const uint32_t*>(base + cap * 4), base + cap * 5};
    gather = &auto_gather;
    e->S.wt_x = base;
    e->S.wt_y = base + cap;
    e->S.wt_z = base + cap * 2;
    if (dvar) { e->S.wt_var = base + cap * 5; e->S.wt_src_var = dvar; }
    if (drgb) { e->S.wt_rgb = reinterpret_cast<uint32_t*>(base + cap * 4); e->S.wt_src_rgb = drgb; }
  }

  // k_bin4 (1024-point blocks, 16 B loads) needs enough blocks to hide a block's latency chain: since k_bin folds
  // its runs through the same per-block LDS table, it wins up to ~400 K points (4 x the blocks; RGB-D 272 K: 13.0
  // vs 15.9 us, 128-beam 262 K: 33.8 vs 39.8; 524 K: 51.1 vs 39.6 the other way)
  const bool want4 = e->bin_variant == 4 || (e->bin_variant == 0 && n >= 393216);
  const bool use_bin4 = !tiled && want4 && aligned;
  // 4-points-per-thread kernels: 256-thread blocks (1024 points).  (2048-point blocks merged ~20 % more cells on
  // chip when every merged cell still cost memory-side atomics; with the record pools four resident blocks per
  // CU beat two: C4 fused launch 44.9 against 46.4 us.)
  const unsigned bin_threads = 256u;
  const unsigned per_block = (use_bin4 || tiled) ? bin_threads * 4u : 256u;
  const unsigned bin_blocks = n ? unsigned((n + per_block - 1) / per_block) : 1u;
  if (bin_blocks > e->bin_part_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->S.bin_part) HIPCK(hipFree(e->S.bin_part));
    e->bin_part_cap = bin_blocks + bin_blocks / 4 + 64;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->S.bin_part), e->bin_part_cap * sizeof(unsigned long long)));
  }
  if (tiled && (rc = ensure_tile_pool(e, size_t(bin_blocks) * per_block, bin_blocks,
                                      P.has_intensity != 0,
                                      P.has_color != 0)))
    return rc;
  e->last_bin_blocks = bin_blocks;
  e->last_bin_part = e->S.bin_part;
  P.dbg_no_atomics = e->dbg_no_atomics;
  P.bin_table = e->bin_table;
  P.dbg_upd = e->dbg_upd;
  P.move_basic = e->move_clear_basic;
  P.drop_nonfinite = e->next_drop_nonfinite;
  e->next_drop_nonfinite = 0;
  int32_t* ids = e->want_ids ? e->d_cell_ids : nullptr;
  const ScanInputs in_b{dx, dy, dz, dint, drgb, dvar};
  // a scan that asks for nothing optional takes the LEAN bin body (fdm_kernels.hpp)
  // (2: lean, but x / y / z written through for the held-back update's gather)
  const int lean = (ids || e->S.cap_x || P.drop_nonfinite || P.dbg_no_atomics) ? 0 : (e->S.wt_x ? 2 : 1);
  const fdm_engine::BinVariant bv{use_bin4, P.has_intensity != 0, P.has_color != 0, e->wave_merge, bin_threads, lean};
  // a held-back update leaves now: fused with this bin if the two belong to the same pipeline and
  // this scan is a plain one, alone otherwise
  const bool fusable = tiled || ((!use_bin4 || (e->rec_kind >= 0 && e->S.dense)) && e->wave_merge);
  // (the fused tiled launch compiles the channels in once, for both halves)
  const bool same_channels = !tiled || (e->pend.P.has_intensity == P.has_intensity && e->pend.P.has_color == P.has_color);
  const bool fuse_now = e->chain && plain && fusable && !e->pend.multi && e->pend.tiled == tiled && same_channels;
  if (e->chain && !fuse_now && (rc = join_streams(e))) return rc;
  P.chain_prev = 0;
  if (e->profile) HIPCK(hipEventRecord(e->ev[0], e->stream));
  if (fuse_now) {  // the held-back update of the previous scan + this scan's bin, one launch
    P.chain_prev = 1;
    P.prev_do_move = e->last_do_move;
    P.prev_gate = e->last_gate;
    e->chain = false;
    if ((rc = launch_update_fused(e, e->pend, P, in_b, tiled ? e->pool[parity] : TilePool{}, ids, bin_blocks, bv)))
      return rc;
  } else if (tiled) {
    if ((rc = launch_tbin(e, P, in_b, e->pool[parity], ids, bin_blocks, bv))) return rc;
  } else if (use_bin4) {
    const bool hi = P.has_intensity != 0, hc = P.has_color != 0;
    auto launch4 = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(bin_blocks), dim3(bin_threads), 0, e->stream, P, e->G, e->d_state, dx,
                         dy, dz, dint, e->S, ids);
    };
    if (hi && hc) launch4(k_bin4<true, true, 256>);
    else if (hi) launch4(k_bin4<true, false, 256>);
    else if (hc) launch4(k_bin4<false, true, 256>);
    else launch4(k_bin4<false, false, 256>);
  } else {
    auto launch_bin = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(bin_blocks), dim3(256), 0, e->stream, P, e->G, e->d_state, dx, dy,
                         dz, dint, e->S, ids);
    };
    if (!e->wave_merge) launch_bin(k_bin<false>);
    else if (P.has_intensity && P.has_color) launch_bin(k_bin<true, 3>);
    else if (P.has_color) launch_bin(k_bin<true, 2>);
    else if (P.has_intensity) launch_bin(k_bin<true, 1>);
    else launch_bin(k_bin<true, 0>);
  }
  HIPCK(hipGetLastError());
  if (e->profile) HIPCK(hipEventRecord(e->ev[1], e->stream));
  // (option "ray_overlap": this scan's stage may start as soon as its bin half has run — marked HERE, ahead of the wait for
  //  the previous scan's stage that run_held_ray_stage is about to put on this stream)
  e->ray_bin_marked = false;
  if (ray_held && e->ray_overlap && e->ray_stream[0]) {
    HIPCK(hipEventRecord(e->ev_ray_bin, e->stream));
    e->ray_bin_marked = true;
  }
  // the previous scan's raycasting stage, if it was held back with the update that has just left in the fused launch:
  // behind that update, ahead of everything of this scan but its bin half (which reads no layer)
  if (fuse_now && (rc = run_held_ray_stage(e, e->pend))) return rc;

  if (e->obst_dense_pending) {
    // (every scan enqueued while the flag stands tries: the first one that observed a cell clears and pays)
    const int blocks = int(std::min<size_t>((e->ncell + 255) / 256, 4096));
    hipLaunchKernelGGL(k_obstacle_dense_clear, dim3(blocks), dim3(256), 0, e->stream, P, e->d_state,
                       L(e, "obstacle"), e->ncell);
    hipLaunchKernelGGL(k_obstacle_dense_paid, dim3(1), dim3(1), 0, e->stream, P, e->d_state, e->h_stats_dev);
    HIPCK(hipGetLastError());
  }
  // this scan's update: held back (the next scan's launch or a flush carries it) or launched now
  fdm_engine::PendingUpdate& u = e->pend;
  u.multi = false;
  u.tiled = tiled;
  u.P = P;
  u.S = e->S;
  u.in = (gather && n) ? ScanInputs{gather->x, gather->y, gather->z, nullptr, drgb ? gather->rgb : nullptr,
                                    dvar ? gather->var : nullptr}
                       : in_b;
  u.upd_blocks = e->n_tiles;
  if (tiled) {
    u.Q = e->pool[parity];
    u.A = TileAux{e->tile_stamp32, e->upd_part32, e->S.ras_z, e->d_timeline};
    e->last_upd_tiles = e->TG.n_tiles;
    e->last_upd_part = e->upd_part32;
  } else {
    e->last_upd_tiles = e->n_tiles;
    e->last_upd_part = e->S.upd_part;
  }
  if (plain) {
    e->chain = true;
    e->last_do_move = P.do_move;
    e->last_gate = P.gate_on_filter;
  } else if ((rc = launch_update_alone(e, u))) {
    return rc;
  }
  if (e->profile) {
    HIPCK(hipEventRecord(e->ev[2], e->stream));
    HIPCK(hipEventRecord(e->ev[3], e->stream));  // back-to-back pair: the event-to-event overhead
  }
  e->ray_timed = false;
  u.ray = false;
  if (ray_on) {  // step 3 of integrateImpl (fastdem.cpp:152-159) on the map this scan just updated
    const float origin[3] = {P.ray_ox, P.ray_oy, P.ray_oz};
    u.RQ = make_ray_params(e, ray_config_of(e->cfg), origin, P.n, (P.slot + 1) & 3, P.slot);
    u.ray_x = e->S.cap_x; u.ray_y = e->S.cap_y; u.ray_z = e->S.cap_z;
    ray_box_of(e, P, u.ray_box);
    u.ray = true;
    u.ray_pre = 0;
    // held back with the update (plain): it runs behind it — in the next scan's launch sequence or at the next flush;
    // otherwise now, the update has just been launched
    if (!plain && (rc = run_held_ray_stage(e, u))) return rc;
    // (option "ray_overlap": what the stage can do before the update leaves at once, on a stream of its own)
    if (plain && (rc = start_ray_stage_early(e, u, P))) return rc;
  }
  e->scan_no++;
  e->last_batch_n = 0;
  e->have_scan = true;
  e->last_n = uint32_t(n);
  e->last_n_input = uint32_t(n);
  e->ingest_blocks = 0;
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
int ensure_stage(fdm_engine* e, size_t n) {
  if (n <= e->stage_cap) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;
  if (e->d_stage) HIPCK(hipFree(e->d_stage));
  e->stage_cap = ((n + n / 4 + 1024) + 3) & ~size_t(3);  // channels stay 16-byte aligned (k_bin4's float4 loads)
  HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_stage), e->stage_cap * 6 * kStageSlots * sizeof(float)));
  return FDM_OK;
}

// Device-visible alias of a pinned (hipHostMalloc / hipHostRegister) host pointer; null for anything else.
// Blocks of the engine's own pinned pool (fdm_host_alloc: every nanopcl::PointCloud channel of the C++ mirror) are
// answered from the pool's table — hipPointerGetAttributes costs ~2.5 us a call, four to six calls per scan.
const void* host_pool_alias(const void* p);  // (below, with the pool)
const void* pinned_alias(const void* p) {
  if (const void* known = host_pool_alias(p)) return known;
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is reported as an error by older runtimes: not ours to keep
    return nullptr;
  }
  return a.type == hipMemoryTypeHost ? a.devicePointer : nullptr;
}

// Host SoA channels -> what the kernels read.  PINNED arrays (up to `zero_copy` points) are not copied:
// the bin kernel reads them in place through their device-visible alias and writes them through to
// the staging block, which is where the update kernel gathers from (*gather; see enqueue_scan) — no
// copy commands, the scan crosses PCIe exactly once.  Anything else is copied into the staging block
// with hipMemcpyAsync and *gather stays unset (x = null); so is everything when gather == nullptr (callers
// whose kernels have no write-through).  Nullable channels stay null.
int stage_inputs(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                 const float* a, const uint32_t* rgb, const float* v, const float** dx,
                 const float** dy, const float** dz, const float** da, const uint32_t** drgb,
                 const float** dv, ScanInputs* gather) {
  int rc;
  if ((rc = ensure_stage(e, n))) return rc;
  e->stage_rr = (e->stage_rr + 1) % kStageSlots;
  const size_t cap = e->stage_cap;
  float* base = e->d_stage + size_t(e->stage_rr) * 6 * cap;
  if (gather) *gather = ScanInputs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (gather && e->zero_copy && n <= uint64_t(e->zero_copy)) {
    const float* mx = static_cast<const float*>(pinned_alias(x));
    const float* my = mx ? static_cast<const float*>(pinned_alias(y)) : nullptr;
    const float* mz = my ? static_cast<const float*>(pinned_alias(z)) : nullptr;
    const float* ma = (mz && a) ? static_cast<const float*>(pinned_alias(a)) : nullptr;
    const uint32_t* mc = (mz && rgb) ? static_cast<const uint32_t*>(pinned_alias(rgb)) : nullptr;
    const float* mv = (mz && v) ? static_cast<const float*>(pinned_alias(v)) : nullptr;
    if (mz && (!a || ma) && (!rgb || mc) && (!v || mv)) {
      *dx = mx; *dy = my; *dz = mz; *da = ma; *drgb = mc; *dv = mv;
      *gather = ScanInputs{base, base + cap, base + cap * 2, nullptr,
                           reinterpret_cast<const uint32_t*>(base + cap * 4), base + cap * 5};
      return FDM_OK;
    }
  }
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
  if (int rc = join_streams(e)) return rc;
  fdm_scan_stats s{};
  *status = FDM_OK;
  if (!e->have_scan) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (out) *out = s;
    return FDM_OK;
  }
  {  // sum the partial counts behind the scan's kernels; the result lands in pinned host memory
    const unsigned work = std::max<unsigned>(e->last_bin_blocks, e->last_upd_tiles);
    const unsigned blocks = std::min(64u, std::max(1u, (work + 4095u) / 4096u));
    hipLaunchKernelGGL(k_collect_stats, dim3(blocks), dim3(256), 0, e->stream, e->last_bin_part, e->last_bin_blocks,
                       e->last_upd_part, e->last_upd_tiles, e->pack_counts, e->ingest_blocks, e->d_state,
                       int((e->scan_no - 1) & 3), e->d_stats_acc, e->h_stats_dev, ++e->stats_seq);
    HIPCK(hipGetLastError());
    // The statistics kernel is the last thing on the stream and ends with a system-scope store of its sequence
    // number into the pinned block: the host polls that word for a while before it falls back to a stream wait
    // (a thread sleeping in hipStreamSynchronize is woken some 10-20 us after the stream has drained — a third of
    // what a synchronous integrate() of a VLP-16 scan takes end to end).
    bool seen = false;
    // (a scan of more than ~100 K points keeps the device busy for longer than the poll window: sleep at once)
    if (e->sync_spin_us > 0 && e->last_n <= 131072u && e->last_kind != 1) {
      const auto t0 = std::chrono::steady_clock::now();
      const volatile unsigned long long* const w = &e->h_stats->seq;
      for (;;) {
        if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == e->stats_seq) { seen = true; break; }
        if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >
            e->sync_spin_us)
          break;
      }
    }
    if (!seen) {
      HIPCK(hipStreamSynchronize(e->stream));
    } else {
      // the sequence word says the stream has drained; a kernel fault is asynchronous and would otherwise go
      // unnoticed until the next blocking call
      const hipError_t q = hipStreamQuery(e->stream);
      if (q != hipSuccess && q != hipErrorNotReady)
        return fail(FDM_ERR_HIP, std::string("stream fault after the scan: ") + hipGetErrorString(q));
      HIPCK(hipGetLastError());
    }
  }
  if (e->h_stats->fault) {
    e->fault_watch = false;
    HIPCK(hipMemset(&e->d_state->fault, 0, sizeof(unsigned)));  // (reported once)
    return fail(FDM_ERR_HIP, "device-side fault: a batch's geometry-chain wait ran out of polls (the batch's map update is undefined)");
  }
  const uint64_t np = e->h_stats->n_pass, ni = e->h_stats->n_in, nt = e->h_stats->n_touched;
  if (e->ingest_blocks) {  // PointCloud2 scan: cloud.size() is the number of finite points (from_impl)
    e->last_n_input = uint32_t(e->h_stats->n_finite);
    e->ingest_blocks = 0;
  }
  s.n_input = e->last_n_input;
  s.n_after_filter = uint32_t(np);
  s.n_in_map = uint32_t(ni);
  s.n_cells_touched = uint32_t(nt);
  const bool applied = e->last_was_integrate ? (np > 0) : true;
  if (applied) {
    s.shift_rows = e->h_stats->shr;
    s.shift_cols = e->h_stats->shc;
  }
  if (e->last_was_integrate) {
    if (e->last_n_input == 0) *status = FDM_SKIP_EMPTY_CLOUD;
    else if (np == 0) *status = FDM_SKIP_ALL_FILTERED;
  }
  if (out) *out = s;
  poll_dense_paid(e);  // (the stream has drained)
  if (e->profile) {
    (void)hipEventElapsedTime(&e->last_ms[0], e->ev[0], e->ev[1]);
    (void)hipEventElapsedTime(&e->last_ms[1], e->ev[1], e->ev[2]);
  }
  return FDM_OK;
}

}  // namespace fdmh

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
  {  // fraction bits of the fixed-point index estimate: whatever an int32 has left beside the map size
    int bits = 1;
    while ((1 << bits) < std::max(G.rows, G.cols) + 2 && bits < 30) ++bits;
    G.idx_shift = std::max(1, std::min(20, 30 - bits));
    G.idx_pad = 0;
    G.inv_res_k = std::ldexp(1.0, G.idx_shift) / G.res;
  }
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
  for (auto& ev : e->ev_timer) HCK(hipEventCreate(&ev));
  HCK(hipMalloc(reinterpret_cast<void**>(&e->d_state), sizeof(DevState)));
  HCK(hipHostMalloc(reinterpret_cast<void**>(&e->h_state), sizeof(DevState)));
  std::memset(e->h_state, 0, sizeof(DevState));
  HCK(hipHostMalloc(reinterpret_cast<void**>(&e->h_stats), sizeof(StatsOut), hipHostMallocMapped));
  std::memset(e->h_stats, 0, sizeof(StatsOut));
  HCK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->h_stats_dev), e->h_stats, 0));
  HCK(hipMalloc(reinterpret_cast<void**>(&e->d_stats_acc), sizeof(StatsAcc)));
  HCK(hipMemset(e->d_stats_acc, 0, sizeof(StatsAcc)));
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
  for (int k = 0; k < 2; ++k) {
    HCK(hipMalloc(reinterpret_cast<void**>(&e->zs2[k]), e->ncell * sizeof(uint2)));
    HCK(hipMemsetAsync(e->zs2[k], 0xFF, e->ncell * sizeof(uint2), e->stream));
  }
  e->S.zs = e->zs2[0];
  {  // second scratch set: scan t+1 bins while scan t still updates (24 B/cell: 100 MB at 4 M cells, 1.5 GB at 64 M)
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
  if (e->key2[0]) (void)hipFree(e->key2[0]);
  if (e->aux2[0]) (void)hipFree(e->aux2[0]);
  if (e->key2[1]) (void)hipFree(e->key2[1]);
  if (e->aux2[1]) (void)hipFree(e->aux2[1]);
  for (auto* z : e->zs2) if (z) (void)hipFree(z);

  for (auto& q : e->pool) {
    if (q.hot) (void)hipFree(q.hot);
    if (q.cold) (void)hipFree(q.cold);
    if (q.cnt) (void)hipFree(q.cnt);
    if (q.desc) (void)hipFree(q.desc);
  }
  for (int k = 0; k < 2; ++k) {
    if (e->mkey[k]) (void)hipFree(e->mkey[k]);
    if (e->maux[k]) (void)hipFree(e->maux[k]);
    if (e->mzs[k]) (void)hipFree(e->mzs[k]);
    if (e->mobs[k]) (void)hipFree(e->mobs[k]);
    if (e->mcobs[k]) (void)hipFree(e->mcobs[k]);
    if (e->mbin_part[k]) (void)hipFree(e->mbin_part[k]);
  }
  if (e->mstate) (void)hipFree(e->mstate);
  if (e->mupd_part) (void)hipFree(e->mupd_part);
  if (e->rb_state) (void)hipFree(e->rb_state);
  if (e->rb_cap) (void)hipFree(e->rb_cap);
  if (e->rb_u32) (void)hipFree(e->rb_u32);
  if (e->rb_rec) (void)hipFree(e->rb_rec);
  if (e->rb_counters) (void)hipFree(e->rb_counters);
  if (e->rb_img) (void)hipFree(e->rb_img);
  if (e->d_bstage) (void)hipFree(e->d_bstage);
  if (e->d_route_owner) (void)hipFree(e->d_route_owner);
  if (e->d_route_cnt) (void)hipFree(e->d_route_cnt);
  if (e->tile_stamp32) (void)hipFree(e->tile_stamp32);
  if (e->upd_part32) (void)hipFree(e->upd_part32);
  if (e->tile_rare) (void)hipFree(e->tile_rare);
  if (e->S.bin_part) (void)hipFree(e->S.bin_part);
  if (e->S.upd_part) (void)hipFree(e->S.upd_part);
  if (e->S.tile_stamp) (void)hipFree(e->S.tile_stamp);
  if (e->d_state) (void)hipFree(e->d_state);
  if (e->h_state) (void)hipHostFree(e->h_state);
  if (e->h_stats) (void)hipHostFree(e->h_stats);
  if (e->d_stats_acc) (void)hipFree(e->d_stats_acc);
  if (e->d_stage) (void)hipFree(e->d_stage);
  if (e->d_aos) (void)hipFree(e->d_aos);
  if (e->d_cell_ids) (void)hipFree(e->d_cell_ids);
  if (e->d_cap) (void)hipFree(e->d_cap);
  for (float* p : e->d_rcap) if (p) (void)hipFree(p);
  if (e->d_ras) (void)hipFree(e->d_ras);
  for (auto& ev : e->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->ev_ray)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->ev_timer)
    if (ev) (void)hipEventDestroy(ev);
  if (e->d_region) (void)hipFree(e->d_region);
  if (e->d_post_pool) (void)hipFree(e->d_post_pool);
  if (e->d_feat_tab) (void)hipFree(e->d_feat_tab);
  if (e->d_timeline) (void)hipFree(e->d_timeline);
  if (e->d_tmp2) (void)hipFree(e->d_tmp2);
  if (e->d_blob) (void)hipFree(e->d_blob);
  if (e->d_in) (void)hipFree(e->d_in);
  if (e->pack_counts) (void)hipFree(e->pack_counts);
  if (e->d_pack) (void)hipFree(e->d_pack);
  if (e->rc_cnt) (void)hipFree(e->rc_cnt);
  if (e->rc_min) (void)hipFree(e->rc_min);
  if (e->ray_bins) (void)hipFree(e->ray_bins);
  for (int k = 0; k < 2; ++k) {
    if (e->vkeys[k]) (void)hipFree(e->vkeys[k]);
    if (e->vidx[k]) (void)hipFree(e->vidx[k]);
  }
  if (e->vsel) (void)hipFree(e->vsel);
  if (e->ray_blk) (void)hipFree(e->ray_blk);
  if (e->sort_tmp) (void)hipFree(e->sort_tmp);
  if (e->vs_cnt) (void)hipFree(e->vs_cnt);
  if (e->vs_rec) (void)hipFree(e->vs_rec);
  {  // the second set of the raycasting stage's buffers, its streams and events (option "ray_overlap")
    fdm_engine::RayBank& b = e->ray_bank1;
    for (void* p : {static_cast<void*>(b.rc_cnt), static_cast<void*>(b.rc_min), static_cast<void*>(b.ray_bins),
                    static_cast<void*>(b.vkeys[0]), static_cast<void*>(b.vkeys[1]), static_cast<void*>(b.vidx[0]),
                    static_cast<void*>(b.vidx[1]), static_cast<void*>(b.vsel), static_cast<void*>(b.ray_blk), b.sort_tmp})
      if (p) (void)hipFree(p);
    for (hipStream_t rs : e->ray_stream) if (rs) (void)hipStreamDestroy(rs);
    for (hipEvent_t ev : {e->ev_ray_pre[0], e->ev_ray_pre[1], e->ev_ray_res[0], e->ev_ray_res[1], e->ev_ray_bin})
      if (ev) (void)hipEventDestroy(ev);
  }
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
    e->last_n_input = 0;
    e->ingest_blocks = 0;
    e->last_was_integrate = 1;
    // consume no slot; last_stats reports SKIP_EMPTY_CLOUD
    return FDM_OK;
  }
  return enqueue_scan(e, P, n, dx, dy, dz, dint, drgb, dvar);
}

int fdm_engine_integrate_device_batch(fdm_engine* e, uint32_t count, const fdm_device_scan* scans) {
  if (!e || (count && !scans)) return fail(FDM_ERR_INVALID, "null argument");
  ++e->batch_call;  // (the look-ahead of an earlier call — keyed on the caller's array ADDRESS — must never match a later call's array)
  for (uint32_t k = 0; k < count; ++k) {
    // runs of small plain scans leave as batches: one bin launch + one update launch per kMaxBatch scans
    if (const uint32_t run = multi_run(e, count - k, scans + k)) {
      HIPCK(hipSetDevice(e->device));
      // (look-ahead: the batch after this one, whose crops ride in this launch)
      const uint32_t next = k + run < count ? multi_run(e, count - k - run, scans + k + run) : 0u;
      if (int rc = enqueue_multi(e, run, scans + k, next)) return rc;
      k += run - 1u;
      continue;
    }
    const fdm_device_scan& s = scans[k];
    const int rc = fdm_engine_integrate_device(e, s.n, s.x, s.y, s.z, s.intensity, s.rgb, s.sigma_z2,
                                               s.T_base_sensor, s.T_world_base);
    if (rc < 0) return rc;  // (an empty cloud is skipped like the reference does, the batch goes on)
  }
  return FDM_OK;
}

// N consecutive FastDEM::integrate calls on HOST clouds.  Pinned channels are handed to the batch launches through
// their device-visible alias (read in place, once, over PCIe: no batch kernel looks at a scan twice); pageable ones are
// copied into an engine-owned block first (hipMemcpyAsync, which the runtime stages).
int fdm_engine_integrate_host_batch(fdm_engine* e, uint32_t count, const fdm_device_scan* host_scans, fdm_scan_stats* out_last) {
  if (!e || (count && !host_scans)) return fail(FDM_ERR_INVALID, "null argument");
  HIPCK(hipSetDevice(e->device));
  std::vector<fdm_device_scan> d(host_scans, host_scans + count);
  // what has to be staged: floats per scan (x y z [i] [rgb] [var]) of the scans whose channels are not all pinned
  std::vector<char> stage(count, 0);
  size_t need = 0;
  for (uint32_t k = 0; k < count; ++k) {
    fdm_device_scan& s = d[k];
    if (s.n == 0) continue;
    if (!s.x || !s.y || !s.z) return fail(FDM_ERR_INVALID, "null xyz");
    if (s.n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1");
    // (option zero_copy is a point-count BOUND, as in stage_inputs: larger clouds are staged also from pinned memory)
    const float* mx = (e->zero_copy && s.n <= uint64_t(e->zero_copy)) ? static_cast<const float*>(pinned_alias(s.x)) : nullptr;
    const float* my = mx ? static_cast<const float*>(pinned_alias(s.y)) : nullptr;
    const float* mz = my ? static_cast<const float*>(pinned_alias(s.z)) : nullptr;
    const float* ma = (mz && s.intensity) ? static_cast<const float*>(pinned_alias(s.intensity)) : nullptr;
    const uint32_t* mc = (mz && s.rgb) ? static_cast<const uint32_t*>(pinned_alias(s.rgb)) : nullptr;
    const float* mv = (mz && s.sigma_z2) ? static_cast<const float*>(pinned_alias(s.sigma_z2)) : nullptr;
    if (mz && (!s.intensity || ma) && (!s.rgb || mc) && (!s.sigma_z2 || mv)) {
      s.x = mx; s.y = my; s.z = mz; s.intensity = ma; s.rgb = mc; s.sigma_z2 = mv;
    } else {
      stage[k] = 1;
      const size_t pad = (size_t(s.n) + 3u) & ~size_t(3);  // (every channel 16-byte aligned)
      need += pad * (3u + (s.intensity ? 1u : 0u) + (s.rgb ? 1u : 0u) + (s.sigma_z2 ? 1u : 0u));
    }
  }
  if (need > e->bstage_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;  // (a launch still in flight may read the old block)
    if (e->d_bstage) HIPCK(hipFree(e->d_bstage));
    e->d_bstage = nullptr;
    e->bstage_cap = need + need / 4 + 4096;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_bstage), e->bstage_cap * sizeof(float)));
  } else if (need && e->bstage_busy) {
    // the previous call's launches may still be reading the block this call is about to overwrite
    if (int rc_sync = sync_all(e)) return rc_sync;
  }
  float* cur = e->d_bstage;
  for (uint32_t k = 0; k < count; ++k) {
    if (!stage[k]) continue;
    fdm_device_scan& s = d[k];
    const size_t pad = (size_t(s.n) + 3u) & ~size_t(3);
    auto up = [&](const void* src) -> const float* {
      float* dst = cur;
      cur += pad;
      return hipMemcpyAsync(dst, src, size_t(s.n) * 4u, hipMemcpyHostToDevice, e->stream) == hipSuccess ? dst : nullptr;
    };
    const float *ux = up(s.x), *uy = up(s.y), *uz = up(s.z);
    const float* ua = s.intensity ? up(s.intensity) : nullptr;
    const float* uc = s.rgb ? up(s.rgb) : nullptr;
    const float* uv = s.sigma_z2 ? up(s.sigma_z2) : nullptr;
    if (!ux || !uy || !uz || (s.intensity && !ua) || (s.rgb && !uc) || (s.sigma_z2 && !uv))
      return fail(FDM_ERR_HIP, "staging a host cloud");
    s.x = ux; s.y = uy; s.z = uz; s.intensity = ua; s.rgb = reinterpret_cast<const uint32_t*>(uc); s.sigma_z2 = uv;
  }
  e->bstage_busy = e->bstage_busy || need != 0;  // (only a drained stream clears it: an earlier call's launches may still read the block)
  if (int rc_batch = fdm_engine_integrate_device_batch(e, count, d.data())) return rc_batch;
  if (!out_last) return FDM_OK;
  int status = FDM_OK;
  if (int rc = read_stats(e, out_last, &status)) return rc;
  e->bstage_busy = false;  // (the stream has drained)
  return status;
}

int fdm_engine_integrate_device_batch_timed(fdm_engine* e, uint32_t count, const fdm_device_scan* scans) {
  if (int rc = fdm_engine_timer_start(e)) return rc;
  const int rc_batch = fdm_engine_integrate_device_batch(e, count, scans);
  const int rc_stop = fdm_engine_timer_stop(e);  // (also on the error path: fdm_engine_timer_ms must never see a stale mark)
  return rc_batch ? rc_batch : rc_stop;
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
  ScanInputs gather;
  int rc = stage_inputs(e, n, x, y, z, intensity, rgb, sigma_z2, &dx, &dy, &dz, &da, &dc, &dv, &gather);
  if (rc) return rc;
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  e->sync_call = true;  // (the flush follows at once: the scan's raycasting stage may start beside its update, "ray_overlap")
  rc = enqueue_scan(e, P, n, dx, dy, dz, da, dc, dv, gather.x ? &gather : nullptr);
  e->sync_call = false;
  if (rc) return rc;
  int status = FDM_OK;
  if ((rc = read_stats(e, out, &status))) return rc;
  return status;
}

// FastDEM::integrate on the reference's OWN cloud layout: nanopcl::PointCloud::points() is a contiguous
// AlignedVector<Vector4f> of {x, y, z, 1} (nanopcl/core/point_cloud.hpp:126-134, types.hpp:19-22).  The 16-byte
// records are read where they lie when the memory is pinned (once, over PCIe, by the de-interleave launch) and copied
// to the device first when it is pageable; the optional channels are separate arrays as in the reference.
int fdm_engine_integrate_points4(fdm_engine* e, uint64_t n, const float* xyz1, const float* intensity,
                                 const uint32_t* rgb, const float* sigma_z2, const double Tbs[16],
                                 const double Twb[16], fdm_scan_stats* out) {
  if (e) { if (int rc_join = join_streams(e)) return rc_join; }
  if (!e || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (n == 0) {
    if (out) std::memset(out, 0, sizeof(*out));
    return FDM_SKIP_EMPTY_CLOUD;
  }
  if (!xyz1) return fail(FDM_ERR_INVALID, "null points");
  if (n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1");
  if (reinterpret_cast<uintptr_t>(xyz1) & 15u) return fail(FDM_ERR_INVALID, "points not 16-byte aligned");
  HIPCK(hipSetDevice(e->device));
  int rc;
  if ((rc = ensure_stage(e, n))) return rc;
  e->stage_rr = (e->stage_rr + 1) % kStageSlots;
  const size_t cap = e->stage_cap;
  float* base = e->d_stage + size_t(e->stage_rr) * 6 * cap;
  const float4* src = static_cast<const float4*>(pinned_alias(xyz1));
  if (!src) {  // pageable: one copy of the records, then the de-interleave reads HBM
    if (n > e->aos_cap) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      if (e->d_aos) HIPCK(hipFree(e->d_aos));
      e->aos_cap = n + n / 4 + 1024;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_aos), e->aos_cap * sizeof(float4)));
    }
    HIPCK(hipMemcpyAsync(e->d_aos, xyz1, n * sizeof(float4), hipMemcpyHostToDevice, e->stream));
    src = e->d_aos;
  }
  const int blocks = int(std::min<uint64_t>((n + 255) / 256, 8192));
  hipLaunchKernelGGL(k_points4_to_soa, dim3(blocks), dim3(256), 0, e->stream, src, size_t(n), base, base + cap, base + cap * 2,
                     static_cast<float*>(nullptr));  // (the fourth component is the homogeneous 1: not a channel)
  HIPCK(hipGetLastError());
  auto up = [&](const void* h, int k) -> int {
    HIPCK(hipMemcpyAsync(base + cap * k, h, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    return FDM_OK;
  };
  const float* da = nullptr;
  const uint32_t* dc = nullptr;
  const float* dv = nullptr;
  if (intensity) { if ((rc = up(intensity, 3))) return rc; da = base + cap * 3; }
  if (rgb) { if ((rc = up(rgb, 4))) return rc; dc = reinterpret_cast<const uint32_t*>(base + cap * 4); }
  if (sigma_z2) { if ((rc = up(sigma_z2, 5))) return rc; dv = base + cap * 5; }
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  // (the channels already sit in the engine's own staging slot, which stays valid until read_stats below has flushed the
  // held-back update: the update gathers from the slot itself — without this a plain non-tiled scan took a SECOND slot
  // and had the bin kernel write x / y / z through to it, 12 B per point for nothing: ADVICE r05)
  const bool borrow = e->borrow_inputs;
  e->borrow_inputs = true;
  e->sync_call = true;
  rc = enqueue_scan(e, P, n, base, base + cap, base + cap * 2, da, dc, dv);
  e->sync_call = false;
  e->borrow_inputs = borrow;
  if (rc) return rc;
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
  ScanInputs gather;
  int rc = stage_inputs(e, n, x, y, z, intensity, rgb, sigma_z2, &dx, &dy, &dz, &da, &dc, &dv, &gather);
  if (rc) return rc;
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  return enqueue_scan(e, P, n, dx, dy, dz, da, dc, dv, gather.x ? &gather : nullptr);
}

// ---- scan routing for spatially tiled global maps (include/fdm_engine.h) ----
static int route_scan_impl(fdm_engine* e, const fdm_route_plan* plan, uint64_t n, const float* dx, const float* dy,
                           const float* dz, const float* dint, const double Tbs[16], const double Twb[16],
                           float* d_send, uint32_t* d_counts, bool soa) {
  if (!e || !plan || !Tbs || !Twb || !d_counts) return fail(FDM_ERR_INVALID, "null argument");
  if (plan->world < 1 || plan->world > kMaxRanks || plan->grid_rows * plan->grid_cols != plan->world)
    return fail(FDM_ERR_INVALID, "route plan: 1 .. 16 ranks in a grid_rows x grid_cols grid");
  if (n && (!dx || !dy || !dz || !d_send)) return fail(FDM_ERR_INVALID, "null xyz / send buffer");
  if (n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1");
  if (e->cfg.mode != 1) return fail(FDM_ERR_INVALID, "scan routing is defined for GLOBAL maps");
  HIPCK(hipSetDevice(e->device));
  RoutePlan R{};
  R.world = plan->world; R.pr = plan->grid_rows; R.pc = plan->grid_cols;
  for (int k = 0; k <= kMaxRanks; ++k) { R.row_edge[k] = plan->row_edge[k]; R.col_edge[k] = plan->col_edge[k]; }
  const unsigned cols = unsigned(R.world + 2);
  const unsigned blocks = unsigned((n + 255) / 256);
  // (also on the FIRST call when its slice is empty: k_route_scan / k_route_base write the totals and the base offsets
  // whatever n is — a null table there was a GPU memory fault that took the rank down and left its peers waiting)
  if (!e->d_route_cnt || !e->d_route_owner || n > e->route_cap || blocks > e->route_blocks_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->d_route_owner) HIPCK(hipFree(e->d_route_owner));
    if (e->d_route_cnt) HIPCK(hipFree(e->d_route_cnt));
    e->d_route_owner = nullptr; e->d_route_cnt = nullptr;
    e->route_cap = std::max<size_t>(e->route_cap, size_t(n) + size_t(n) / 4 + 1024);
    e->route_blocks_cap = (e->route_cap + 255) / 256;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_route_owner), e->route_cap));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_route_cnt),
                    (e->route_blocks_cap * size_t(kMaxRanks + 2) + kMaxRanks) * sizeof(uint32_t)));
  }
  uint32_t* const base = e->d_route_cnt + e->route_blocks_cap * size_t(kMaxRanks + 2);
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  sensor_params(e->cfg, P.sensor_type, P.sp);
  P.n = uint32_t(n);
  // the geometry of a GLOBAL map never moves: while the last scan's update is still held back (it will commit slot
  // scan_no & 3) the slot that scan read holds the same values
  P.slot = int((e->chain ? e->scan_no + 3 : e->scan_no) & 3);
  if (blocks) {
    hipLaunchKernelGGL(k_route_count, dim3(blocks), dim3(256), 0, e->stream, P, e->G, R, e->d_state, dx, dy, dz,
                       e->d_route_owner, e->d_route_cnt);
    HIPCK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_route_scan, dim3(cols), dim3(1024), 0, e->stream, e->d_route_cnt, blocks, R.world, d_counts);
  hipLaunchKernelGGL(k_route_base, dim3(1), dim3(64), 0, e->stream, d_counts, R.world, base, soa ? 1 : 0);
  HIPCK(hipGetLastError());
  if (blocks) {
    if (soa)
      hipLaunchKernelGGL(k_route_scatter<true>, dim3(blocks), dim3(256), 0, e->stream, unsigned(n), R.world, e->d_route_owner,
                         e->d_route_cnt, base, d_counts, dx, dy, dz, dint, reinterpret_cast<float4*>(d_send));
    else
      hipLaunchKernelGGL(k_route_scatter<false>, dim3(blocks), dim3(256), 0, e->stream, unsigned(n), R.world, e->d_route_owner,
                         e->d_route_cnt, base, d_counts, dx, dy, dz, dint, reinterpret_cast<float4*>(d_send));
    HIPCK(hipGetLastError());
  }
  return FDM_OK;
}

int fdm_engine_route_scan(fdm_engine* e, const fdm_route_plan* plan, uint64_t n, const float* dx, const float* dy,
                          const float* dz, const float* dint, const double Tbs[16], const double Twb[16],
                          float* d_send, uint32_t* d_counts) {
  return route_scan_impl(e, plan, n, dx, dy, dz, dint, Tbs, Twb, d_send, d_counts, false);
}
int fdm_engine_route_scan_soa(fdm_engine* e, const fdm_route_plan* plan, uint64_t n, const float* dx, const float* dy,
                              const float* dz, const float* dint, const double Tbs[16], const double Twb[16],
                              float* d_send, uint32_t* d_counts) {
  return route_scan_impl(e, plan, n, dx, dy, dz, dint, Tbs, Twb, d_send, d_counts, true);
}

int fdm_engine_integrate_soa4_device(fdm_engine* e, uint64_t n, const float* d_block, int has_intensity, int any_in_map,
                                     const double Tbs[16], const double Twb[16]) {
  if (!e || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (n && !d_block) return fail(FDM_ERR_INVALID, "null points");
  if (n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1");
  if (n && (reinterpret_cast<uintptr_t>(d_block) & 15u)) return fail(FDM_ERR_INVALID, "share not 16-byte aligned");
  HIPCK(hipSetDevice(e->device));
  if (n == 0) return fdm_engine_integrate_points4_device(e, 0, nullptr, has_intensity, any_in_map, Tbs, Twb);
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  P.force_inside = any_in_map ? 1 : 0;
  const size_t stride = (size_t(n) + 3u) & ~size_t(3);  // the share's channel blocks: x | y | z | intensity
  return enqueue_scan(e, P, n, d_block, d_block + stride, d_block + 2 * stride,
                      has_intensity ? d_block + 3 * stride : nullptr, nullptr, nullptr);
}

int fdm_engine_integrate_points4_device(fdm_engine* e, uint64_t n, const float* d_points4, int has_intensity,
                                        int any_in_map, const double Tbs[16], const double Twb[16]) {
  if (!e || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (n && !d_points4) return fail(FDM_ERR_INVALID, "null points");
  if (n >= 0x7FFFFFFFull) return fail(FDM_ERR_INVALID, "point count exceeds 2^31-1");
  HIPCK(hipSetDevice(e->device));
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  P.force_inside = any_in_map ? 1 : 0;
  if (n == 0) {
    if (!any_in_map) {  // nothing of the scan reached any map tile: fastdem.cpp:145-161 leaves the map untouched
      e->have_scan = true;
      e->last_n = 0;
      e->last_n_input = 0;
      e->ingest_blocks = 0;
      e->last_was_integrate = 1;
      return FDM_OK;
    }
    // this tile saw none of the scan's points, but the scan did observe cells elsewhere: update() still runs
    // (the whole-layer obstacle clear, elevation_mapping.cpp:144-146)
    return enqueue_scan(e, P, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  }
  int rc;
  if ((rc = ensure_stage(e, n))) return rc;
  e->stage_rr = (e->stage_rr + 1) % kStageSlots;  // (a held-back update never reads this block: see ensure_stage)
  const size_t cap = e->stage_cap;
  float* base = e->d_stage + size_t(e->stage_rr) * 6 * cap;
  float* di = has_intensity ? base + cap * 3 : nullptr;
  const int blocks = int(std::min<uint64_t>((n + 255) / 256, 8192));
  hipLaunchKernelGGL(k_points4_to_soa, dim3(blocks), dim3(256), 0, e->stream, reinterpret_cast<const float4*>(d_points4),
                     size_t(n), base, base + cap, base + cap * 2, di);
  HIPCK(hipGetLastError());
  return enqueue_scan(e, P, n, base, base + cap, base + cap * 2, di, nullptr, nullptr);
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
  ScanInputs gather{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (n && (rc = stage_inputs(e, n, x, y, z, intensity, rgb, z_var, &dx, &dy, &dz, &da, &dc, &dv, &gather)))
    return rc;
  ScanParams P;
  fill_update_params(e, P, rx, ry, false);
  if ((rc = enqueue_scan(e, P, n, dx, dy, dz, da, dc, dv, gather.x ? &gather : nullptr))) return rc;
  int status = FDM_OK;
  if ((rc = read_stats(e, out, &status))) return rc;
  return FDM_OK;
}

// ---- pinned host pool (fdm_host_alloc) ----
extern "C++" {
namespace fdmh {
struct HostPool {
  static constexpr int kMinShift = 12, kMaxShift = 30;  // 4 KiB .. 1 GiB classes; larger blocks are not pooled
  std::mutex mu;
  std::unordered_map<const void*, int> live;            // block -> class (>= 0 pinned, -1 pinned unpooled, -2 pageable)
  std::unordered_map<const void*, const void*> alias;   // pinned block -> its device-visible address
  std::vector<void*> idle[kMaxShift + 1];
};
HostPool* host_pool() {
  static HostPool* pool = new HostPool;  // never destroyed: clouds with static storage may be freed after main()
  return pool;
}
const void* host_pool_alias(const void* p) {
  HostPool& hp = *host_pool();
  std::lock_guard<std::mutex> lock(hp.mu);
  auto it = hp.live.find(p);
  if (it == hp.live.end() || it->second == -2) return nullptr;
  auto al = hp.alias.find(p);
  return al == hp.alias.end() ? nullptr : al->second;
}
}  // namespace fdmh
}  // extern "C++"

void* fdm_host_alloc(uint64_t bytes) {
  HostPool& hp = *host_pool();
  int cls = HostPool::kMinShift;
  while (cls <= HostPool::kMaxShift && (1ull << cls) < bytes) ++cls;
  const bool pooled = cls <= HostPool::kMaxShift;
  const size_t size = pooled ? (size_t(1) << cls) : size_t(bytes);
  if (pooled) {
    std::lock_guard<std::mutex> lock(hp.mu);
    if (!hp.idle[cls].empty()) {
      void* p = hp.idle[cls].back();
      hp.idle[cls].pop_back();
      hp.live[p] = cls;
      return p;
    }
  }
  void* p = nullptr;
  int tag = pooled ? cls : -1;
  if (hipHostMalloc(&p, size, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess || !p) {
    (void)hipGetLastError();
    p = std::aligned_alloc(64, (size + 63) & ~size_t(63));  // no usable GPU: pageable, the engine will copy
    tag = -2;
    if (!p) return nullptr;
  }
  void* dev = nullptr;
  if (tag != -2 && hipHostGetDevicePointer(&dev, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    dev = nullptr;  // (pinned_alias falls back to hipPointerGetAttributes)
  }
  std::lock_guard<std::mutex> lock(hp.mu);
  hp.live[p] = tag;
  if (dev) hp.alias[p] = dev;
  return p;
}

void fdm_host_free(void* p) {
  if (!p) return;
  HostPool& hp = *host_pool();
  int tag;
  {
    std::lock_guard<std::mutex> lock(hp.mu);
    auto it = hp.live.find(p);
    if (it == hp.live.end()) return;  // not ours
    tag = it->second;
    hp.live.erase(it);
    if (tag >= 0) {
      hp.idle[tag].push_back(p);
      return;
    }
  }
  if (tag == -1) {
    {
      std::lock_guard<std::mutex> lock(hp.mu);
      hp.alias.erase(p);
    }
    (void)hipHostFree(p);
  } else {
    std::free(p);
  }
}

void fdm_host_trim(void) {
  HostPool& hp = *host_pool();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lock(hp.mu);
    for (auto& v : hp.idle) {
      drop.insert(drop.end(), v.begin(), v.end());
      v.clear();
    }
    for (void* p : drop) hp.alias.erase(p);
  }
  for (void* p : drop) (void)hipHostFree(p);
}

int fdm_host_is_pinned(const void* p) {
  HostPool& hp = *host_pool();
  std::lock_guard<std::mutex> lock(hp.mu);
  auto it = hp.live.find(p);
  return (it != hp.live.end() && it->second != -2) ? 1 : 0;
}

int fdm_engine_flush(fdm_engine* e) {
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  return join_streams(e);
}

void* fdm_engine_stream(fdm_engine* e) { return e ? static_cast<void*>(e->stream) : nullptr; }

int fdm_engine_last_pipeline(fdm_engine* e) { return e ? e->last_kind : -1; }

int fdm_engine_last_batch(fdm_engine* e) { return e ? e->last_batch_n : 0; }

int fdm_engine_timer_start(fdm_engine* e) {
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc = join_streams(e)) return rc;  // a held-back update belongs to what came before the mark
  HIPCK(hipEventRecord(e->ev_timer[0], e->stream));
  return FDM_OK;
}

int fdm_engine_timer_stop(fdm_engine* e) {
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc = join_streams(e)) return rc;  // the last scan's update is part of the timed work
  HIPCK(hipEventRecord(e->ev_timer[1], e->stream));
  return FDM_OK;
}

int fdm_engine_timer_ms(fdm_engine* e, float* ms) {
  if (!e || !ms) return fail(FDM_ERR_INVALID, "null argument");
  if (int rc = sync_all(e)) return rc;
  HIPCK(hipEventElapsedTime(ms, e->ev_timer[0], e->ev_timer[1]));
  return report_fault(e);  // (the timed work has run: a batch whose chain wait gave up must not pass as a measurement)
}

int fdm_engine_debug_timeline(fdm_engine* e, uint64_t* ticks, uint64_t cap_blocks, uint32_t* n_blocks,
                              uint32_t* n_update_blocks) {
  if (!e || !ticks || !n_blocks || !n_update_blocks) return fail(FDM_ERR_INVALID, "null argument");
  if (!e->d_timeline) return fail(FDM_ERR_INVALID, "option dbg_timeline is off");
  HIPCK(hipSetDevice(e->device));
  HIPCK(hipStreamSynchronize(e->stream));  // (no flush: a held-back update stays held back)
  *n_blocks = e->timeline_blocks;
  *n_update_blocks = e->timeline_upd;
  const uint64_t n = std::min<uint64_t>(cap_blocks, e->timeline_blocks);
  if (n) HIPCK(hipMemcpy(ticks, e->d_timeline, n * 16, hipMemcpyDeviceToHost));
  return FDM_OK;
}

// measurement / debugging only: how many entries of the batch pipeline's per-scan scratch sets are NOT in their
// clean state once everything enqueued has run (the update half leaves every entry it consumed clean; anything else
// would leak into a later batch).  out[0..2] = keys, aux words, zero-sign words.
int fdm_engine_debug_batch_dirty(fdm_engine* e, uint64_t out[3]) {
  if (!e || !out) return fail(FDM_ERR_INVALID, "null argument");
  out[0] = out[1] = out[2] = 0;
  if (int rc = sync_all(e)) return rc;
  if (!e->mkey[0]) return FDM_OK;
  const size_t slots = size_t(kMaxBatch) * e->ncell;
  std::vector<unsigned long long> hk(slots);
  std::vector<uint4> ha(slots);
  std::vector<uint2> hz(slots);
  for (int k = 0; k < 2; ++k) {
    HIPCK(hipMemcpy(hk.data(), e->mkey[k], slots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(ha.data(), e->maux[k], slots * sizeof(uint4), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(hz.data(), e->mzs[k], slots * sizeof(uint2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < slots; ++i) {
      out[0] += hk[i] != kEmptyKey;
      out[1] += (ha[i].x != 0u) + (ha[i].y != 0u) + (ha[i].z != kNoIdx) + (ha[i].w != 0u);
      out[2] += (hz[i].x != 0xFFFFFFFFu) + (hz[i].y != 0xFFFFFFFFu);
    }
  }
  return FDM_OK;
}

int fdm_engine_debug_batch_launches(fdm_engine* e, uint64_t out[2]) {
  if (!e || !out) return fail(FDM_ERR_INVALID, "null argument");
  out[0] = e->n_mbatch;
  out[1] = 0;  // (tile batches: removed in round 5, the slot stays for the callers of the array)
  return FDM_OK;
}

int fdm_engine_record_event(fdm_engine* e, void* hip_event) {
  if (!e || !hip_event) return fail(FDM_ERR_INVALID, "null argument");
  if (int rc = join_streams(e)) return rc;  // the map is current behind this event
  HIPCK(hipEventRecord(static_cast<hipEvent_t>(hip_event), e->stream));
  return FDM_OK;
}

int fdm_engine_wait_event(fdm_engine* e, void* hip_event) {
  if (!e || !hip_event) return fail(FDM_ERR_INVALID, "null argument");
  HIPCK(hipStreamWaitEvent(e->stream, static_cast<hipEvent_t>(hip_event), 0));
  return FDM_OK;
}

int fdm_engine_sync(fdm_engine* e) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  if (!e) return fail(FDM_ERR_INVALID, "null engine");
  if (int rc_sync = sync_all(e)) return rc_sync;
  return report_fault(e);
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

#include "fdm_engine_layers.inl"  // geometry, named layers, halo regions
#include "fdm_engine_opts.inl"    // captures, cell ids, profile, options

}  // extern "C"

// The stages either side of the hot path are translation units of their own: fdm_engine_ray.hip (raycasting),
// fdm_engine_post.hip (stencil post-processing, PointCloud2 ingest / map egress).
