// fdm_engine_io.inl — host side of PointCloud2 ingest (fdm_ingest.hpp) and map egress (fdm_egress.hpp).
// Part of fdm_engine_post.hip (one of the library's three translation units, fdm_engine_host.hpp).

namespace {
// fdm_cloud2_layout -> IngestLayout; L.aligned covers the record layout only (the caller adds the blob's address)
int check_cloud2_layout(const fdm_cloud2_layout* lay, IngestLayout& L) {
  const uint32_t step = lay->point_step;
  auto fits = [&](int32_t off, uint32_t len) { return off < 0 || uint64_t(off) + len <= step; };
  const uint32_t ilen = lay->intensity_type == 8 ? 8 : (lay->intensity_type == 7 ? 4 : (lay->intensity_type == 4 ? 2 : 1));
  if (step == 0 || !fits(lay->off_x, 4) || !fits(lay->off_y, 4) || !fits(lay->off_z, 4) ||
      !fits(lay->off_intensity, ilen) || !fits(lay->off_rgb, 4))
    return fail(FDM_ERR_INVALID, "field offset outside the point record");
  L.point_step = step;
  L.off_x = lay->off_x; L.off_y = lay->off_y; L.off_z = lay->off_z;
  L.off_intensity = lay->off_intensity; L.intensity_type = lay->intensity_type;
  L.off_rgb = lay->off_rgb;
  auto al4 = [](int32_t off) { return off < 0 || (off & 3) == 0; };
  L.aligned = (step & 3u) == 0 && al4(L.off_x) && al4(L.off_y) && al4(L.off_z) && al4(L.off_rgb) &&
              (L.intensity_type < 7 || al4(L.off_intensity));
  return FDM_OK;
}
}  // namespace

extern "C" {

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
  IngestLayout L{};
  if (int rc_lay = check_cloud2_layout(lay, L)) return rc_lay;
  const uint32_t step = L.point_step;
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
  L.aligned = L.aligned && (reinterpret_cast<uintptr_t>(blob) & 3u) == 0;
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

// from_impl + integrate without a host round trip in between: one decode kernel writes the SoA
// channels at the MESSAGE's indices (no compaction), the bin kernel drops the non-finite points
// itself, and cloud.size() (n_input, the empty-cloud decision) is summed on the device with the
// other statistics.  A pinned message (fdm_host_alloc) is decoded in place over PCIe.
int fdm_engine_integrate_cloud2(fdm_engine* e, const void* data, int on_device, uint64_t n_points,
                                const fdm_cloud2_layout* lay, const double Tbs[16], const double Twb[16],
                                fdm_scan_stats* out) {
  if (e) { if (int rc_join = join_streams(e)) return rc_join; }
  if (!e || !lay || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  auto empty = [&]() {  // fastdem.cpp:125-128
    if (out) std::memset(out, 0, sizeof(*out));
    return int(FDM_SKIP_EMPTY_CLOUD);
  };
  if (n_points == 0) return empty();                                              // impl.hpp:178-181
  if (lay->off_x < 0 || lay->off_y < 0 || lay->off_z < 0) return empty();          // impl.hpp:183-186: no xyz
  if (!data) return fail(FDM_ERR_INVALID, "null data");
  if (n_points >= 0xFFFFFFFEull) return fail(FDM_ERR_INVALID, "point count exceeds 2^32-2");
  IngestLayout L{};
  int rc;
  if ((rc = check_cloud2_layout(lay, L))) return rc;
  HIPCK(hipSetDevice(e->device));
  const size_t bytes = size_t(n_points) * L.point_step;
  const uint8_t* blob = static_cast<const uint8_t*>(data);
  if (!on_device) {
    const void* alias = e->zero_copy ? pinned_alias(data) : nullptr;
    if (alias) {
      blob = static_cast<const uint8_t*>(alias);
    } else {
      if (bytes > e->blob_cap) {
        if (int rc_sync = sync_all(e)) return rc_sync;
        if (e->d_blob) HIPCK(hipFree(e->d_blob));
        e->blob_cap = bytes + bytes / 4 + 4096;
        HIPCK(hipMalloc(reinterpret_cast<void**>(&e->d_blob), e->blob_cap));
      }
      HIPCK(hipMemcpyAsync(e->d_blob, data, bytes, hipMemcpyHostToDevice, e->stream));
      blob = e->d_blob;
    }
  }
  L.aligned = L.aligned && (reinterpret_cast<uintptr_t>(blob) & 3u) == 0;
  if ((rc = ensure_stage(e, n_points))) return rc;
  e->stage_rr = (e->stage_rr + 1) % kStageSlots;
  const size_t cap = e->stage_cap;
  float* base = e->d_stage + size_t(e->stage_rr) * 6 * cap;
  const unsigned blocks = unsigned((n_points + 255) / 256);
  if (size_t(blocks) + 1 > e->pack_counts_cap) {
    if (int rc_sync = sync_all(e)) return rc_sync;
    if (e->pack_counts) HIPCK(hipFree(e->pack_counts));
    e->pack_counts_cap = size_t(blocks) + 1 + 1024;
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->pack_counts), e->pack_counts_cap * sizeof(uint32_t)));
  }
  const bool hi = lay->off_intensity >= 0, hc = lay->off_rgb >= 0;
  float* di = hi ? base + cap * 3 : nullptr;
  uint32_t* dc = hc ? reinterpret_cast<uint32_t*>(base + cap * 4) : nullptr;
  hipLaunchKernelGGL(k_ingest_soa, dim3(blocks), dim3(256), 0, e->stream, blob, L, n_points, e->pack_counts, base,
                     base + cap, base + cap * 2, di, dc);
  HIPCK(hipGetLastError());
  ScanParams P;
  fill_integrate_params(e, P, Tbs, Twb);
  e->next_drop_nonfinite = 1;
  e->sync_call = true;
  rc = enqueue_scan(e, P, n_points, base, base + cap, base + cap * 2, di, dc, nullptr);
  e->sync_call = false;
  if (rc) return rc;
  e->ingest_blocks = blocks;  // read_stats sums the finite counts with the scan's other statistics
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


}  // extern "C"
