// fdm_halo.cpp — libfdm_halo.so: tile plan + RCCL scan broadcast / halo exchange on the engine's stream
// (include/fdm_halo.h).  Host code only; the HIP pack / unpack kernels are libfdm_engine.so's
// fdm_engine_region_pack / _unpack.
#include "../../include/fdm_halo.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& m) {
  g_err = m;
  return code;
}

void grid_for(int world, int& pr, int& pc) {
  pr = int(std::floor(std::sqrt(double(world))));
  while (world % pr) --pr;
  pc = world / pr;
}
// Python's round(): half to even (fastdem_amd/tiling.py::_split uses it)
long round_half_even(double v) {
  const double f = std::floor(v), d = v - f;
  if (d > 0.5) return long(f) + 1;
  if (d < 0.5) return long(f);
  return (long(f) % 2 == 0) ? long(f) : long(f) + 1;
}
void split(int n, int parts, int i, int& start, int& len) {
  const long a = round_half_even(double(i) * n / parts), b = round_half_even(double(i + 1) * n / parts);
  start = int(a);
  len = int(b - a);
}
fdm_rect owned_rect(int rank, int world, int rows, int cols) {
  int pr, pc;
  grid_for(world, pr, pc);
  const int i = rank / pc, j = rank % pc;
  fdm_rect r;
  split(rows, pr, i, r.r0, r.nr);
  split(cols, pc, j, r.c0, r.nc);
  return r;
}
fdm_rect stored_rect(const fdm_rect& o, int rows, int cols, int halo) {
  const int r0 = std::max(0, o.r0 - halo), c0 = std::max(0, o.c0 - halo);
  const int r1 = std::min(rows, o.r0 + o.nr + halo), c1 = std::min(cols, o.c0 + o.nc + halo);
  return fdm_rect{r0, c0, r1 - r0, c1 - c0};
}
fdm_rect intersect(const fdm_rect& a, const fdm_rect& b) {
  const int r0 = std::max(a.r0, b.r0), c0 = std::max(a.c0, b.c0);
  const int r1 = std::min(a.r0 + a.nr, b.r0 + b.nr), c1 = std::min(a.c0 + a.nc, b.c0 + b.nc);
  return fdm_rect{r0, c0, std::max(0, r1 - r0), std::max(0, c1 - c0)};
}
// ---- transport: RCCL unless the host installed its own (include/fdm_halo.h) ----
fdm_halo_transport g_transport{};
bool g_custom = false;
int comm_all_gather(void* comm, const void* d_send, void* d_recv, size_t bytes, hipStream_t stream) {
  if (g_custom) return g_transport.all_gather(comm, d_send, d_recv, bytes, stream) ? fail(FDM_ERR_HIP, "transport: all_gather") : FDM_OK;
  const ncclResult_t r = ncclAllGather(d_send, d_recv, bytes, ncclUint8, static_cast<ncclComm_t>(comm), stream);
  return r == ncclSuccess ? FDM_OK : fail(FDM_ERR_HIP, std::string("ncclAllGather: ") + ncclGetErrorString(r));
}
int comm_group_start(void* comm) {
  if (g_custom) return g_transport.group_start(comm) ? fail(FDM_ERR_HIP, "transport: group_start") : FDM_OK;
  const ncclResult_t r = ncclGroupStart();
  return r == ncclSuccess ? FDM_OK : fail(FDM_ERR_HIP, std::string("ncclGroupStart: ") + ncclGetErrorString(r));
}
int comm_send(void* comm, const void* d_buf, size_t bytes, int peer, hipStream_t stream) {
  if (g_custom) return g_transport.send(comm, d_buf, bytes, peer, stream) ? fail(FDM_ERR_HIP, "transport: send") : FDM_OK;
  const ncclResult_t r = ncclSend(d_buf, bytes, ncclUint8, peer, static_cast<ncclComm_t>(comm), stream);
  return r == ncclSuccess ? FDM_OK : fail(FDM_ERR_HIP, std::string("ncclSend: ") + ncclGetErrorString(r));
}
int comm_recv(void* comm, void* d_buf, size_t bytes, int peer, hipStream_t stream) {
  if (g_custom) return g_transport.recv(comm, d_buf, bytes, peer, stream) ? fail(FDM_ERR_HIP, "transport: recv") : FDM_OK;
  const ncclResult_t r = ncclRecv(d_buf, bytes, ncclUint8, peer, static_cast<ncclComm_t>(comm), stream);
  return r == ncclSuccess ? FDM_OK : fail(FDM_ERR_HIP, std::string("ncclRecv: ") + ncclGetErrorString(r));
}
int comm_group_end(void* comm, hipStream_t stream) {
  if (g_custom) return g_transport.group_end(comm, stream) ? fail(FDM_ERR_HIP, "transport: group_end") : FDM_OK;
  const ncclResult_t r = ncclGroupEnd();
  return r == ncclSuccess ? FDM_OK : fail(FDM_ERR_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(r));
}
bool have_comm(void* comm) { return g_custom || comm != nullptr; }
uint64_t pad4(uint64_t v) { return (v + 3u) & ~uint64_t(3); }
}  // namespace

extern "C" {

const char* fdm_halo_last_error(void) { return g_err.c_str(); }

void fdm_halo_set_transport(const fdm_halo_transport* t) {
  g_custom = t != nullptr;
  g_transport = t ? *t : fdm_halo_transport{};
}

int fdm_tile_plan_make(int32_t rank, int32_t world, int32_t rows, int32_t cols, int32_t halo, fdm_tile_plan* out) {
  if (!out) return fail(FDM_ERR_INVALID, "null plan");
  if (world <= 0 || rank < 0 || rank >= world || rows <= 0 || cols <= 0 || halo < 0)
    return fail(FDM_ERR_INVALID, "bad rank / world / size / halo");
  fdm_tile_plan p{};
  p.rank = rank; p.world = world; p.rows = rows; p.cols = cols; p.halo = halo;
  grid_for(world, p.grid_rows, p.grid_cols);
  p.owned = owned_rect(rank, world, rows, cols);
  p.stored = stored_rect(p.owned, rows, cols, halo);
  if (p.owned.nr <= 0 || p.owned.nc <= 0) return fail(FDM_ERR_INVALID, "more tiles than cells along an axis");
  for (int other = 0; other < world; ++other) {
    if (other == rank) continue;
    const fdm_rect theirs = owned_rect(other, world, rows, cols);
    const fdm_rect theirs_st = stored_rect(theirs, rows, cols, halo);
    const fdm_rect s = intersect(p.owned, theirs_st), r = intersect(theirs, p.stored);
    if (s.nr > 0 && s.nc > 0) {
      if (p.n_sends >= FDM_MAX_NEIGHBOURS) return fail(FDM_ERR_INVALID, "halo wider than a tile: more than 8 neighbours");
      p.send_rank[p.n_sends] = other;
      p.send_rect[p.n_sends++] = s;
    }
    if (r.nr > 0 && r.nc > 0) {
      if (p.n_recvs >= FDM_MAX_NEIGHBOURS) return fail(FDM_ERR_INVALID, "halo wider than a tile: more than 8 neighbours");
      p.recv_rank[p.n_recvs] = other;
      p.recv_rect[p.n_recvs++] = r;
    }
  }
  *out = p;
  return FDM_OK;
}

void fdm_tile_plan_tile(const fdm_tile_plan* p, fdm_tile* t) {
  if (!p || !t) return;
  t->row0 = p->stored.r0; t->col0 = p->stored.c0; t->rows = p->stored.nr; t->cols = p->stored.nc;
  t->own_row0 = p->owned.r0; t->own_col0 = p->owned.c0; t->own_rows = p->owned.nr; t->own_cols = p->owned.nc;
}

uint64_t fdm_halo_workspace_bytes(const fdm_tile_plan* p, int32_t n_layers) {
  if (!p || n_layers <= 0) return 0;
  uint64_t cells = 0;
  for (int k = 0; k < p->n_sends; ++k) cells += uint64_t(p->send_rect[k].nr) * uint64_t(p->send_rect[k].nc);
  for (int k = 0; k < p->n_recvs; ++k) cells += uint64_t(p->recv_rect[k].nr) * uint64_t(p->recv_rect[k].nc);
  return cells * uint64_t(n_layers) * sizeof(float);
}

int fdm_halo_broadcast_scan(fdm_engine* e, void* nccl_comm, float* d_packed, uint64_t count, int32_t root) {
  if (!e || !nccl_comm || !d_packed) return fail(FDM_ERR_INVALID, "null argument");
  hipStream_t stream = static_cast<hipStream_t>(fdm_engine_stream(e));
  const ncclResult_t r = ncclBroadcast(d_packed, d_packed, size_t(count), ncclFloat, root,
                                       static_cast<ncclComm_t>(nccl_comm), stream);
  if (r != ncclSuccess) return fail(FDM_ERR_HIP, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
  return FDM_OK;
}

int64_t fdm_halo_exchange(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const char* const* names,
                          int32_t n_names, float* d_ws, uint64_t ws_bytes) {
  if (!e || !p || !names) return fail(FDM_ERR_INVALID, "null argument");
  if ((p->n_sends == 0 && p->n_recvs == 0) || n_names <= 0) return 0;  // (a plan may name the rank itself as a neighbour)
  if (!have_comm(nccl_comm) || !d_ws) return fail(FDM_ERR_INVALID, "null communicator / workspace");
  if (ws_bytes < fdm_halo_workspace_bytes(p, n_names)) return fail(FDM_ERR_INVALID, "workspace too small");
  hipStream_t stream = static_cast<hipStream_t>(fdm_engine_stream(e));
  // pack: ONE launch for every strip and layer (the call launches a held-back update first: the map is current)
  float* cur = d_ws;
  float* send_buf[FDM_MAX_NEIGHBOURS];
  float* recv_buf[FDM_MAX_NEIGHBOURS];
  fdm_region sreg[FDM_MAX_NEIGHBOURS], rreg[FDM_MAX_NEIGHBOURS];
  int64_t sent = 0;
  for (int k = 0; k < p->n_sends; ++k) {
    const fdm_rect& r = p->send_rect[k];
    send_buf[k] = cur;
    sreg[k] = fdm_region{r.r0 - p->stored.r0, r.c0 - p->stored.c0, r.nr, r.nc, uint64_t(cur - d_ws)};
    cur += size_t(r.nr) * size_t(r.nc) * size_t(n_names);
    sent += int64_t(r.nr) * r.nc * n_names * int64_t(sizeof(float));
  }
  if (p->n_sends)
    if (int rc = fdm_engine_regions_pack(e, p->n_sends, sreg, names, n_names, d_ws))
      return fail(rc, std::string("regions_pack: ") + fdm_last_error());
  for (int k = 0; k < p->n_recvs; ++k) {
    const fdm_rect& r = p->recv_rect[k];
    recv_buf[k] = cur;
    rreg[k] = fdm_region{r.r0 - p->stored.r0, r.c0 - p->stored.c0, r.nr, r.nc, uint64_t(cur - d_ws)};
    cur += size_t(r.nr) * size_t(r.nc) * size_t(n_names);
  }
  // one group: xGMI is point-to-point, a strip is ~10^2 KB — latency-bound, so everything leaves together
  if (int rc = comm_group_start(nccl_comm)) return rc;
  int rc_x = FDM_OK;
  for (int k = 0; rc_x == FDM_OK && k < p->n_recvs; ++k) {
    const fdm_rect& q = p->recv_rect[k];
    rc_x = comm_recv(nccl_comm, recv_buf[k], size_t(q.nr) * size_t(q.nc) * size_t(n_names) * 4u, p->recv_rank[k], stream);
  }
  for (int k = 0; rc_x == FDM_OK && k < p->n_sends; ++k) {
    const fdm_rect& q = p->send_rect[k];
    rc_x = comm_send(nccl_comm, send_buf[k], size_t(q.nr) * size_t(q.nc) * size_t(n_names) * 4u, p->send_rank[k], stream);
  }
  const int rc_e = comm_group_end(nccl_comm, stream);
  if (rc_x || rc_e) return rc_x ? rc_x : rc_e;
  if (p->n_recvs)
    if (int rc = fdm_engine_regions_unpack(e, p->n_recvs, rreg, names, n_names, d_ws))
      return fail(rc, std::string("regions_unpack: ") + fdm_last_error());
  return sent;
}

void fdm_tile_plan_route(const fdm_tile_plan* p, fdm_route_plan* out) {
  if (!p || !out) return;
  *out = fdm_route_plan{};
  out->world = p->world;
  out->grid_rows = p->grid_rows;
  out->grid_cols = p->grid_cols;
  for (int i = 0; i <= p->grid_rows && i <= 16; ++i) {
    int start, len;
    split(p->rows, p->grid_rows, std::min(i, p->grid_rows - 1), start, len);
    out->row_edge[i] = i < p->grid_rows ? start : p->rows;
  }
  for (int j = 0; j <= p->grid_cols && j <= 16; ++j) {
    int start, len;
    split(p->cols, p->grid_cols, std::min(j, p->grid_cols - 1), start, len);
    out->col_edge[j] = j < p->grid_cols ? start : p->cols;
  }
}

int fdm_halo_gather_counts(fdm_engine* e, void* nccl_comm, const uint32_t* d_counts, uint32_t* d_matrix,
                           uint32_t* h_matrix, int32_t world) {
  if (!e || !have_comm(nccl_comm) || !d_counts || !d_matrix || !h_matrix || world < 1) return fail(FDM_ERR_INVALID, "null argument");
  hipStream_t stream = static_cast<hipStream_t>(fdm_engine_stream(e));
  const size_t per = size_t(world) + 2;
  if (int rc = comm_all_gather(nccl_comm, d_counts, d_matrix, per * 4u, stream)) return rc;
  if (hipMemcpyAsync(h_matrix, d_matrix, per * size_t(world) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return fail(FDM_ERR_HIP, "reading the routing counts back");
  return FDM_OK;
}

namespace {
// soa: shares are four channel blocks of pad4(count) floats (fdm_engine_route_scan_soa) and the rank's own share
// stays in the send buffer; else 16-byte records, the own share copied into its place among the sources.
int route_exchange_impl(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const float* d_send,
                        const uint32_t* h_matrix, float* d_recv, uint64_t recv_capacity, uint64_t* n_recv,
                        int32_t* any_in_map, bool soa) {
  if (!e || !p || !h_matrix || !n_recv || !any_in_map) return fail(FDM_ERR_INVALID, "null argument");
  const int W = p->world, me = p->rank;
  const size_t per = size_t(W) + 2;
  auto rows = [&](uint64_t v) { return soa ? pad4(v) : v; };  // 16-byte rows a share of v points takes
  uint64_t total = 0, inside = 0;
  for (int src = 0; src < W; ++src) {
    if (!(soa && src == me)) total += rows(h_matrix[size_t(src) * per + size_t(me)]);
    inside += h_matrix[size_t(src) * per + size_t(W) + 1];
  }
  *n_recv = total;
  *any_in_map = inside ? 1 : 0;
  if (total > recv_capacity) return fail(FDM_ERR_INVALID, "receive buffer too small for the routed points");
  if (W > 1 && !have_comm(nccl_comm)) return fail(FDM_ERR_INVALID, "null communicator");
  if ((total && !d_recv) || !d_send) return fail(FDM_ERR_INVALID, "null point buffer");
  hipStream_t stream = static_cast<hipStream_t>(fdm_engine_stream(e));
  // where each peer's share starts: in my send buffer (owner-major) / in my receive buffer (source-major)
  uint64_t send_off = 0, recv_off = 0;
  int rc = FDM_OK;
  bool grouped = false;
  for (int peer = 0; peer < W && rc == FDM_OK; ++peer) {
    const uint64_t ns = rows(h_matrix[size_t(me) * per + size_t(peer)]), nr = rows(h_matrix[size_t(peer) * per + size_t(me)]);
    if (peer == me) {
      if (!soa && ns && hipMemcpyAsync(d_recv + 4 * recv_off, d_send + 4 * send_off, ns * 16, hipMemcpyDeviceToDevice, stream) !=
                            hipSuccess)
        return fail(FDM_ERR_HIP, "copying the rank's own share");
      send_off += ns;
      if (!soa) recv_off += nr;
      continue;
    }
    if (ns || nr) {
      if (!grouped) { rc = comm_group_start(nccl_comm); grouped = true; }
      if (rc == FDM_OK && nr) rc = comm_recv(nccl_comm, d_recv + 4 * recv_off, size_t(nr) * 16u, peer, stream);
      if (rc == FDM_OK && ns) rc = comm_send(nccl_comm, d_send + 4 * send_off, size_t(ns) * 16u, peer, stream);
    }
    send_off += ns;
    recv_off += nr;
  }
  if (grouped) {
    const int rc2 = comm_group_end(nccl_comm, stream);
    if (rc || rc2) return rc ? rc : rc2;
  }
  return rc;
}
}  // namespace

int fdm_halo_route_exchange(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const float* d_send,
                            const uint32_t* h_matrix, float* d_recv, uint64_t recv_capacity, uint64_t* n_recv,
                            int32_t* any_in_map) {
  return route_exchange_impl(e, nccl_comm, p, d_send, h_matrix, d_recv, recv_capacity, n_recv, any_in_map, false);
}

// ---- the routed step as ONE call (the host loop of tiling.RoutedScan in C: no interpreter between the launches) ----
// Two halves: the FRONT (route the slice, gather the table, copy it to the host — enqueue only, an event behind it) and
// the BACK (wait for the table, exchange the points, integrate).  fdm_halo_routed_step runs both for one scan;
// fdm_halo_routed_submit runs the front of scan k+1 and THEN the back of scan k, so that the device works on scan k's
// points while the host waits for scan k+1's table (two sets of send / table buffers).
struct RoutedSlot {
  float* d_send = nullptr;      // [max_points] x 16 B, owner-major
  uint32_t* d_row = nullptr;    // world + 2 counters | the rank's two transforms as raw fp64 bits (64 words)
  uint32_t* d_table = nullptr;  // [world] rows
  uint32_t* h_table = nullptr;  // pinned copy
  double* h_pose = nullptr;     // pinned: T_base_sensor | T_world_base
  hipEvent_t ready = nullptr;   // the table is on the host
  double T[32];                 // the caller's transforms of the scan in this slot
  int has_i = 0, sensors = 0;
  bool pending = false;
};
struct fdm_routed_ws {
  int world = 0;
  uint64_t max_points = 0, recv_cap = 0;
  RoutedSlot slot[2];
  unsigned seq = 0;
  float* d_recv = nullptr;      // [recv_cap] x 16 B, source-major
  uint32_t* h_matrix = nullptr; // [world][world + 2] of the last finished scan (plain host memory)
  bool obstacle_dirty = true;   // (sensors mode) the tile's obstacle layer may hold non-NaN cells
  bool single_pending = false;  // world == 1: a submitted scan whose matrix has not been handed out yet
};

namespace {
int routed_front(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const fdm_route_plan* route, fdm_routed_ws* w,
                 RoutedSlot& s, uint64_t n, const float* d_x, const float* d_y, const float* d_z, const float* d_i,
                 const double* Tbs, const double* Twb, int sensors) {
  const int W = p->world;
  const size_t per = size_t(W) + 2, row = per + 64;
  hipStream_t stream = static_cast<hipStream_t>(fdm_engine_stream(e));
  // 1. this rank's slice, partitioned by owner (three small kernels); its counters land in the row
  // (N scans, one per rank: shares as channel blocks the owners read in place; one scan cut into slices: records)
  if (int rc = sensors ? fdm_engine_route_scan_soa(e, route, n, d_x, d_y, d_z, d_i, Tbs, Twb, s.d_send, s.d_row)
                       : fdm_engine_route_scan(e, route, n, d_x, d_y, d_z, d_i, Tbs, Twb, s.d_send, s.d_row))
    return fail(rc, std::string("fdm_engine_route_scan: ") + fdm_last_error());
  // 2. the row travels with the rank's transforms (N-sensor mode: the owners need every source's)
  for (int k = 0; k < 16; ++k) { s.h_pose[k] = s.T[k] = Tbs[k]; s.h_pose[16 + k] = s.T[16 + k] = Twb[k]; }
  s.has_i = d_i ? 1 : 0;
  s.sensors = sensors;
  if (hipMemcpyAsync(s.d_row + per, s.h_pose, 32 * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess)
    return fail(FDM_ERR_HIP, "uploading the transforms");
  // 3. one small all-gather, ONE host read-back (the sizes of the exchange are host-side arguments)
  if (W > 1)
    if (int rc = comm_all_gather(nccl_comm, s.d_row, s.d_table, row * 4u, stream)) return rc;
  if (hipMemcpyAsync(s.h_table, W > 1 ? s.d_table : s.d_row, row * 4 * size_t(W), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipEventRecord(s.ready, stream) != hipSuccess)
    return fail(FDM_ERR_HIP, "reading the routing table back");
  s.pending = true;
  (void)w;
  return FDM_OK;
}

int routed_back(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, fdm_routed_ws* w, RoutedSlot& s,
                uint32_t* h_matrix_out) {
  const int W = p->world, me = p->rank;
  const size_t per = size_t(W) + 2, row = per + 64;
  if (hipEventSynchronize(s.ready) != hipSuccess) return fail(FDM_ERR_HIP, "waiting for the routing table");
  s.pending = false;
  for (int src = 0; src < W; ++src)
    for (size_t j = 0; j < per; ++j) w->h_matrix[size_t(src) * per + j] = s.h_table[size_t(src) * row + j];
  if (h_matrix_out) std::copy(w->h_matrix, w->h_matrix + size_t(W) * per, h_matrix_out);
  // 4. the points travel to their owners
  const bool soa = s.sensors != 0;
  uint64_t total = 0;
  for (int src = 0; src < W; ++src)
    if (!(soa && src == me)) total += soa ? pad4(w->h_matrix[size_t(src) * per + size_t(me)]) : w->h_matrix[size_t(src) * per + size_t(me)];
  if (total > w->recv_cap) {
    if (w->d_recv) {
      if (hipStreamSynchronize(static_cast<hipStream_t>(fdm_engine_stream(e))) != hipSuccess)  // (an earlier integrate may read it)
        return fail(FDM_ERR_HIP, "stream");
      (void)hipFree(w->d_recv);
    }
    w->d_recv = nullptr;
    w->recv_cap = total + total / 4 + 1024;
    if (hipMalloc(reinterpret_cast<void**>(&w->d_recv), w->recv_cap * 16) != hipSuccess)
      return fail(FDM_ERR_HIP, "allocating the receive buffer");
  }
  uint64_t n_recv = 0;
  int32_t any_in_map = 0;
  if (int rc = route_exchange_impl(e, nccl_comm, p, s.d_send, w->h_matrix, w->d_recv, w->recv_cap, &n_recv, &any_in_map, soa))
    return rc;
  // 5. the owners integrate: the logical scan as one, or — N sensors — every source with its own transforms, in rank order
  if (!s.sensors) {
    if (int rc = fdm_engine_integrate_points4_device(e, n_recv, w->d_recv, s.has_i, any_in_map, s.T, s.T + 16))
      return fail(rc, std::string("fdm_engine_integrate_points4_device: ") + fdm_last_error());
    return FDM_OK;
  }
  // (shares: the sources' in the receive buffer in rank order, this rank's own where the routing kernels left it)
  uint64_t off = 0, own_off = 0;
  for (int d = 0; d < me; ++d) own_off += pad4(w->h_matrix[size_t(me) * per + size_t(d)]);
  for (int src = 0; src < W; ++src) {
    const uint64_t ns = w->h_matrix[size_t(src) * per + size_t(me)];
    const bool seen = w->h_matrix[size_t(src) * per + size_t(W) + 1] > 0;  // that scan observed a cell somewhere
    if (ns || (seen && w->obstacle_dirty)) {  // (else: nothing for this tile and its obstacle layer is clear already)
      double T[32];
      std::memcpy(T, s.h_table + size_t(src) * row + per, sizeof(T));
      const float* share = src == me ? s.d_send + 4 * own_off : w->d_recv + 4 * off;
      if (int rc = fdm_engine_integrate_soa4_device(e, ns, ns ? share : nullptr, s.has_i, seen ? 1 : 0, T, T + 16))
        return fail(rc, std::string("fdm_engine_integrate_soa4_device: ") + fdm_last_error());
      w->obstacle_dirty = ns > 0;
    }
    if (src != me) off += pad4(ns);
  }
  return FDM_OK;
}

int routed_check(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const fdm_route_plan* route, fdm_routed_ws* w,
                 uint64_t n, const double* Tbs, const double* Twb) {
  if (!e || !p || !route || !w || !Tbs || !Twb) return fail(FDM_ERR_INVALID, "null argument");
  if (w->world != p->world) return fail(FDM_ERR_INVALID, "workspace of another plan");
  if (n > w->max_points) return fail(FDM_ERR_INVALID, "more points than the workspace holds");
  if (p->world > 1 && !have_comm(nccl_comm)) return fail(FDM_ERR_INVALID, "null communicator");
  return FDM_OK;
}

// world == 1: every cell is this rank's, nothing has to be routed — the step IS fdm_engine_integrate_device (the bin
// kernel drops what the crops and the map reject, as it does for any scan).  The counter matrix is the engine's
// statistics, read (a wait for the scan) only when the caller asked for it.
int routed_single(fdm_engine* e, uint64_t n, const float* d_x, const float* d_y, const float* d_z, const float* d_i,
                  const double* Tbs, const double* Twb, uint32_t* h_matrix_out) {
  if (int rc = fdm_engine_integrate_device(e, n, d_x, d_y, d_z, d_i, nullptr, nullptr, Tbs, Twb); rc < 0)
    return fail(rc, std::string("fdm_engine_integrate_device: ") + fdm_last_error());
  if (h_matrix_out) {
    fdm_scan_stats st{};
    const int rc = fdm_engine_last_stats(e, &st);
    if (rc < 0) return fail(rc, std::string("fdm_engine_last_stats: ") + fdm_last_error());
    h_matrix_out[0] = st.n_in_map; h_matrix_out[1] = st.n_after_filter; h_matrix_out[2] = st.n_in_map;
  }
  return FDM_OK;
}
}  // namespace

int fdm_halo_routed_ws_create(const fdm_tile_plan* p, uint64_t max_points, fdm_routed_ws** out) {
  if (!p || !out || p->world < 1) return fail(FDM_ERR_INVALID, "null argument");
  fdm_routed_ws* w = new fdm_routed_ws;
  w->world = p->world;
  w->max_points = max_points ? max_points : 1;
  const size_t row = size_t(p->world) + 2 + 64;
  bool ok = true;
  for (RoutedSlot& s : w->slot) {
    ok = ok && hipMalloc(reinterpret_cast<void**>(&s.d_send), (w->max_points + 3u * size_t(p->world) + 4u) * 16) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s.d_row), row * 4) == hipSuccess &&
         hipMalloc(reinterpret_cast<void**>(&s.d_table), row * 4 * size_t(p->world)) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s.h_table), row * 4 * size_t(p->world), hipHostMallocDefault) == hipSuccess &&
         hipHostMalloc(reinterpret_cast<void**>(&s.h_pose), 32 * sizeof(double), hipHostMallocDefault) == hipSuccess &&
         hipMemset(s.d_row, 0, row * 4) == hipSuccess &&
         hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess;
  }
  w->h_matrix = new uint32_t[size_t(p->world) * (size_t(p->world) + 2)];
  if (!ok) {
    fdm_halo_routed_ws_destroy(w);
    return fail(FDM_ERR_HIP, "allocating the routed-step workspace");
  }
  *out = w;
  return FDM_OK;
}

void fdm_halo_routed_ws_destroy(fdm_routed_ws* w) {
  if (!w) return;
  for (RoutedSlot& s : w->slot) {
    if (s.d_send) (void)hipFree(s.d_send);
    if (s.d_row) (void)hipFree(s.d_row);
    if (s.d_table) (void)hipFree(s.d_table);
    if (s.h_table) (void)hipHostFree(s.h_table);
    if (s.h_pose) (void)hipHostFree(s.h_pose);
    if (s.ready) (void)hipEventDestroy(s.ready);
  }
  if (w->d_recv) (void)hipFree(w->d_recv);
  delete[] w->h_matrix;
  delete w;
}

int fdm_halo_routed_step(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const fdm_route_plan* route,
                         fdm_routed_ws* w, uint64_t n, const float* d_x, const float* d_y, const float* d_z,
                         const float* d_intensity, const double T_base_sensor[16], const double T_world_base[16],
                         int32_t sensors, uint32_t* h_matrix_out) {
  if (int rc = routed_check(e, nccl_comm, p, route, w, n, T_base_sensor, T_world_base)) return rc;
  if (int rc = fdm_halo_routed_flush(e, nccl_comm, p, w, nullptr)) return rc;  // (a submitted scan comes first)
  if (p->world == 1) return routed_single(e, n, d_x, d_y, d_z, d_intensity, T_base_sensor, T_world_base, h_matrix_out);
  RoutedSlot& s = w->slot[w->seq++ & 1u];
  if (int rc = routed_front(e, nccl_comm, p, route, w, s, n, d_x, d_y, d_z, d_intensity, T_base_sensor, T_world_base, sensors))
    return rc;
  return routed_back(e, nccl_comm, p, w, s, h_matrix_out);
}

int fdm_halo_routed_submit(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, const fdm_route_plan* route,
                           fdm_routed_ws* w, uint64_t n, const float* d_x, const float* d_y, const float* d_z,
                           const float* d_intensity, const double T_base_sensor[16], const double T_world_base[16],
                           int32_t sensors, uint32_t* h_matrix_prev) {
  if (int rc = routed_check(e, nccl_comm, p, route, w, n, T_base_sensor, T_world_base)) return rc;
  if (p->world == 1) {  // (nothing to overlap: the scan is enqueued at once; its matrix is the engine's statistics)
    if (h_matrix_prev && w->single_pending) {
      fdm_scan_stats st{};
      if (int rc = fdm_engine_last_stats(e, &st); rc < 0) return fail(rc, std::string("fdm_engine_last_stats: ") + fdm_last_error());
      h_matrix_prev[0] = st.n_in_map; h_matrix_prev[1] = st.n_after_filter; h_matrix_prev[2] = st.n_in_map;
    }
    w->single_pending = true;
    return routed_single(e, n, d_x, d_y, d_z, d_intensity, T_base_sensor, T_world_base, nullptr);
  }
  RoutedSlot& s = w->slot[w->seq & 1u];
  RoutedSlot& prev = w->slot[(w->seq & 1u) ^ 1u];
  ++w->seq;
  if (s.pending) return fail(FDM_ERR_INVALID, "internal: routed slot still pending");
  // the front of this scan first: its kernels and its table copy run while the host is busy with the previous scan ...
  if (int rc = routed_front(e, nccl_comm, p, route, w, s, n, d_x, d_y, d_z, d_intensity, T_base_sensor, T_world_base, sensors))
    return rc;
  // ... whose table arrived during the previous call: exchange + integrate are enqueued behind this scan's front
  if (prev.pending) return routed_back(e, nccl_comm, p, w, prev, h_matrix_prev);
  return FDM_OK;
}

int fdm_halo_routed_flush(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* p, fdm_routed_ws* w, uint32_t* h_matrix_out) {
  if (!e || !p || !w) return fail(FDM_ERR_INVALID, "null argument");
  if (p->world == 1 && w->single_pending) {
    w->single_pending = false;
    if (h_matrix_out) {
      fdm_scan_stats st{};
      if (int rc = fdm_engine_last_stats(e, &st); rc < 0) return fail(rc, std::string("fdm_engine_last_stats: ") + fdm_last_error());
      h_matrix_out[0] = st.n_in_map; h_matrix_out[1] = st.n_after_filter; h_matrix_out[2] = st.n_in_map;
    }
    return FDM_OK;
  }
  // (oldest first: with submit() at most one slot is pending)
  for (unsigned k = 0; k < 2u; ++k) {
    RoutedSlot& s = w->slot[(w->seq + k) & 1u];
    if (s.pending)
      if (int rc = routed_back(e, nccl_comm, p, w, s, h_matrix_out)) return rc;
  }
  return FDM_OK;
}

}  // extern "C"
