// fdm_engine_layers.inl — the GridMap calls a FastDEM caller makes around integrate(): geometry (move, position, start
// index), named layers (list / add / download / upload / clear) and the packed halo regions of the multi-GPU tiling.
// Part of fdm_engine.hip's translation unit (inside its extern "C" block): do not compile on its own.

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
  if (std::strcmp(name, "obstacle") == 0) { e->obst_dense_pending = true; e->obst_owe_armed = false; }
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
  if (std::strcmp(name, "obstacle") == 0) { e->obst_dense_pending = true; e->obst_owe_armed = false; }
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

// A copy of a map (ElevationMap's copy constructor, snapshot(): elevation_map.hpp:95-99; the ROS node copies the map under
// a shared lock, ros1/src/fastdem_ros_node.cpp:192-199): layer `name` of `src` into the layer of the same name of `dst`
// (created if missing), device to device — no host staging.  Same device, same stored window.
int fdm_engine_layer_copy(fdm_engine* dst, fdm_engine* src, const char* name) {
  if (!dst || !src || !name) return fail(FDM_ERR_INVALID, "null argument");
  if (dst == src) return FDM_OK;
  if (int rc = join_streams(src)) return rc;
  if (int rc = join_streams(dst)) return rc;
  if (dst->device != src->device) return fail(FDM_ERR_INVALID, "layer_copy: engines on different devices");
  if (dst->G.s_rows != src->G.s_rows || dst->G.s_cols != src->G.s_cols) return fail(FDM_ERR_INVALID, "layer_copy: shape mismatch");
  HIPCK(hipSetDevice(dst->device));
  if (int rc = resolve_pending(src)) return rc;
  Layer* ls = find_layer(src, name);
  if (!ls || ls->pending) return fail(FDM_ERR_NO_LAYER, std::string("no layer ") + name);
  if (int rc = sync_all(src)) return rc;  // (the source map is current; its stream has drained)
  Layer* ld = find_layer(dst, name);
  if (!ld) {
    if (int rc = add_layer(dst, name, NAN, false)) return rc;
    ld = find_layer(dst, name);
    ls = find_layer(src, name);
  }
  ld->pending = false;
  if (std::strcmp(name, "obstacle") == 0) { dst->obst_dense_pending = true; dst->obst_owe_armed = false; }
  if (int rc = copy_strided(dst, lptr(dst, *ld), lstride(dst, *ld), lptr(src, *ls), lstride(src, *ls))) return rc;
  return sync_all(dst);
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

// One launch for `n_rects` rectangles x `n_layers` layers (k_regions_copy); more than kRegionRects / kRegionLayers of
// either: several launches.
static int regions_copy(fdm_engine* e, int n_rects, const fdm_region* rects, const char* const* names, int n_layers,
                        float* d_buf, int to_buf) {
  if (!e || !names || !d_buf || (n_rects && !rects)) return fail(FDM_ERR_INVALID, "null argument");
  if (n_rects < 0 || n_layers < 0) return fail(FDM_ERR_INVALID, "negative count");
  HIPCK(hipSetDevice(e->device));
  std::vector<Layer*> L(size_t(std::max(n_layers, 0)));
  for (int k = 0; k < n_layers; ++k) {
    L[k] = find_layer(e, names[k]);
    if (!L[k]) return fail(FDM_ERR_NO_LAYER, std::string("no layer ") + names[k]);
  }
  for (int q = 0; q < n_rects; ++q) {
    const fdm_region& r = rects[q];
    if (r.nr <= 0 || r.nc <= 0 || r.r0 < 0 || r.c0 < 0 || r.r0 + r.nr > e->G.s_rows || r.c0 + r.nc > e->G.s_cols)
      return fail(FDM_ERR_INVALID, "region outside the stored window");
  }
  for (int q0 = 0; q0 < n_rects; q0 += kRegionRects) {
    const int nq = std::min(kRegionRects, n_rects - q0);
    for (int l0 = 0; l0 < n_layers; l0 += kRegionLayers) {
      const int nl = std::min(kRegionLayers, n_layers - l0);
      RegionArgs A{};
      size_t max_cells = 0;
      for (int q = 0; q < nq; ++q) {
        const fdm_region& r = rects[q0 + q];
        const size_t cells = size_t(r.nr) * size_t(r.nc);
        A.r0[q] = r.r0; A.c0[q] = r.c0; A.nr[q] = r.nr; A.nc[q] = r.nc;
        A.off[q] = r.offset + size_t(l0) * cells;  // (the rectangle's block is layer-major over ALL n_layers)
        max_cells = std::max(max_cells, cells);
      }
      for (int l = 0; l < nl; ++l) { A.layer[l] = lptr(e, *L[l0 + l]); A.es[l] = lstride(e, *L[l0 + l]); }
      A.n_rects = nq; A.n_layers = nl; A.s_rows = e->G.s_rows; A.to_buf = to_buf;
      const unsigned gx = unsigned(std::min<size_t>((max_cells + 255) / 256, 1024));
      hipLaunchKernelGGL(k_regions_copy, dim3(gx, unsigned(nq * nl)), dim3(256), 0, e->stream, A, d_buf);
      HIPCK(hipGetLastError());
    }
  }
  return FDM_OK;
}

int fdm_engine_regions_pack(fdm_engine* e, int32_t n_rects, const fdm_region* rects, const char* const* names,
                            int n_layers, float* d_buf) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  return regions_copy(e, n_rects, rects, names, n_layers, d_buf, 1);
}
int fdm_engine_regions_unpack(fdm_engine* e, int32_t n_rects, const fdm_region* rects, const char* const* names,
                              int n_layers, const float* d_buf) {
  if (e) { if (int rc_join = join_streams(e)) { (void)rc_join; return rc_join; } }
  return regions_copy(e, n_rects, rects, names, n_layers, const_cast<float*>(d_buf), 0);
}
int fdm_engine_region_pack(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                           const char* const* names, int n_layers, float* d_buf) {
  const fdm_region r{r0, c0, nr, nc, 0};
  return fdm_engine_regions_pack(e, 1, &r, names, n_layers, d_buf);
}
int fdm_engine_region_unpack(fdm_engine* e, int32_t r0, int32_t c0, int32_t nr, int32_t nc,
                             const char* const* names, int n_layers, const float* d_buf) {
  const fdm_region r{r0, c0, nr, nc, 0};
  return fdm_engine_regions_unpack(e, 1, &r, names, n_layers, d_buf);
}
