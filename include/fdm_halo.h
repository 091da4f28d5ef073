/* fdm_halo.h — C ABI for driving the spatial tiling of ONE global map from a C++ host with RCCL.
 *
 * The reference has no multi-GPU path; its callers that would need one keep a single very large GLOBAL map
 * (ros1/config/global_mapping.yaml:14-17, driven by ros1/src/fastdem_ros_node.cpp:171-182).  This library
 * (libfdm_halo.so, links librccl; libfdm_engine.so itself has no RCCL dependency) gives such a host the
 * three things it needs: the tile plan, the scan distribution and the halo exchange — all enqueued on the
 * ENGINE's stream, so there is no host synchronisation and no cross-stream event anywhere:
 *
 *     fdm_tile_plan plan;  fdm_tile tile;
 *     fdm_tile_plan_make(rank, world, rows, cols, FDM_DEFAULT_HALO, &plan);
 *     fdm_tile_plan_tile(&plan, &tile);
 *     fdm_engine_create(&geometry, &config, &tile, device, &engine);          // GLOBAL mode
 *     per scan:
 *       fdm_halo_broadcast_scan(engine, comm, d_packed, channels * n, root);  // [channels][n] float32
 *       fdm_engine_integrate_device(engine, n, d_packed, d_packed + n, d_packed + 2 * n, ...);
 *       fdm_halo_exchange(engine, comm, &plan, names, n_names, d_workspace, workspace_bytes);
 *
 * fastdem_amd/tiling.py is the same plan / exchange over torch.distributed; tests/test_halo_capi.py checks
 * the two plans against each other.
 */
#ifndef FDM_HALO_H
#define FDM_HALO_H

#include "fdm_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FDM_DEFAULT_HALO 6 /* cells: feature extraction reaches ceil(0.3 m / 0.05 m) (config/postprocess.hpp:45) */
#define FDM_MAX_NEIGHBOURS 8

typedef struct fdm_rect {
  int32_t r0, c0, nr, nc; /* GLOBAL cell coordinates */
} fdm_rect;

typedef struct fdm_tile_plan {
  int32_t rank, world, rows, cols, halo;
  int32_t grid_rows, grid_cols; /* pr x pc tiles, pr * pc == world, as square as possible, pr <= pc */
  fdm_rect owned, stored;       /* stored = owned + halo ring, clamped at the map border */
  int32_t n_sends, n_recvs;
  int32_t send_rank[FDM_MAX_NEIGHBOURS];
  fdm_rect send_rect[FDM_MAX_NEIGHBOURS]; /* my owned cells inside that neighbour's halo ring */
  int32_t recv_rank[FDM_MAX_NEIGHBOURS];
  fdm_rect recv_rect[FDM_MAX_NEIGHBOURS]; /* that neighbour's owned cells inside my halo ring */
} fdm_tile_plan;

/* The plan of `rank` (needs halo < every tile's extent, i.e. at most 8 neighbours). */
int fdm_tile_plan_make(int32_t rank, int32_t world, int32_t rows, int32_t cols, int32_t halo, fdm_tile_plan* out);
void fdm_tile_plan_tile(const fdm_tile_plan* plan, fdm_tile* out);
/* Bytes of device workspace fdm_halo_exchange needs for `n_layers` layers (send + receive buffers). */
uint64_t fdm_halo_workspace_bytes(const fdm_tile_plan* plan, int32_t n_layers);

/* Scan distribution: ncclBroadcast of `count` floats (the packed SoA channels) from `root`, on the engine's
 * stream.  `nccl_comm` is the host application's ncclComm_t. */
int fdm_halo_broadcast_scan(fdm_engine* e, void* nccl_comm, float* d_packed, uint64_t count, int32_t root);

/* One halo exchange on the engine's stream: HIP pack of every send rectangle, ONE ncclGroup of
 * ncclSend / ncclRecv with the (at most 8) neighbours, HIP unpack.  Launches a held-back map update first.
 * Returns the number of bytes sent, or a negative FDM_ERR_*. */
int64_t fdm_halo_exchange(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* plan, const char* const* names,
                          int32_t n_names, float* d_workspace, uint64_t workspace_bytes);

/* ---- Scan routing: every rank holds a SLICE of the logical scan, points travel to the rank that owns their cell ----
 *     fdm_route_plan route;  fdm_tile_plan_route(&plan, &route);
 *     per scan, every rank:
 *       fdm_engine_route_scan(engine, &route, n_slice, d_x, d_y, d_z, d_i, Tbs, Twb, d_send, d_counts);
 *       fdm_halo_gather_counts(engine, comm, d_counts, d_matrix, h_matrix);      // ncclAllGather + one host read-back
 *       fdm_halo_route_exchange(engine, comm, &plan, d_send, h_matrix, d_recv, recv_capacity, &n_recv, &any_in_map);
 *       fdm_engine_integrate_points4_device(engine, n_recv, d_recv, has_intensity, any_in_map, Tbs, Twb);
 *       fdm_halo_exchange(...);
 * h_matrix[src * (world + 2) + dst] = points rank src sends to rank dst; columns world / world + 1 = the slice's
 * n_after_filter / n_in_map.  Received points arrive in rank order = scan order (fdm_engine.h, fdm_engine_route_scan). */
void fdm_tile_plan_route(const fdm_tile_plan* plan, fdm_route_plan* out);
/* ncclAllGather of every rank's world + 2 counters into d_matrix (device, world * (world + 2) words), copied to
 * h_matrix; waits for the stream (the sizes of the exchange are host-side arguments of ncclSend / ncclRecv). */
int fdm_halo_gather_counts(fdm_engine* e, void* nccl_comm, const uint32_t* d_counts, uint32_t* d_matrix,
                           uint32_t* h_matrix, int32_t world);
/* One ncclGroup of ncclSend / ncclRecv of 16-byte point records with every other rank (the rank's own share is a
 * device copy), on the engine's stream.  d_recv must hold recv_capacity records; *n_recv = records received (rank order),
 * *any_in_map = some slice had a point inside the map. */
int fdm_halo_route_exchange(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* plan, const float* d_send,
                            const uint32_t* h_matrix, float* d_recv, uint64_t recv_capacity, uint64_t* n_recv,
                            int32_t* any_in_map);

/* The whole routed step as ONE call (what fastdem_amd/tiling.py::RoutedScan.integrate does, without an interpreter
 * between the launches): route this rank's slice -> all-gather of the counters and the rank's transforms, one host
 * read-back -> point exchange -> the owner integrates.  sensors = 0: the slices are ONE logical scan (every rank passes
 * the same transforms); sensors = 1: one scan per rank with its own transforms, integrated by the owners in rank
 * order (an owner skips a source that sent it nothing unless that source's scan must clear its obstacle layer).
 * sensors = 1 moves every share as four channel blocks (fdm_engine_route_scan_soa): the owner's bin kernel reads a
 * share in place — no de-interleave pass, the rank's own share is not copied.  world == 1: nothing is routed at all,
 * the step IS fdm_engine_integrate_device (the bin kernel drops what the crops / the map reject itself); the counter
 * matrix is then read from the engine's statistics only when h_matrix_out asks for it (that read waits for the scan).
 * The workspace holds the send / receive / table buffers (max_points = largest slice); h_matrix_out (nullable,
 * world * (world + 2) words) receives the counter matrix of the step.  nccl_comm may be null when world == 1.
 * Followed, as before, by fdm_halo_exchange for the halo rings. */
typedef struct fdm_routed_ws fdm_routed_ws;
int fdm_halo_routed_ws_create(const fdm_tile_plan* plan, uint64_t max_points, fdm_routed_ws** out);
void fdm_halo_routed_ws_destroy(fdm_routed_ws* ws);
int fdm_halo_routed_step(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* plan, const fdm_route_plan* route,
                         fdm_routed_ws* ws, uint64_t n, const float* d_x, const float* d_y, const float* d_z,
                         const float* d_intensity, const double T_base_sensor[16], const double T_world_base[16],
                         int32_t sensors, uint32_t* h_matrix_out);
/* The same step, software-pipelined over consecutive scans: submit() routes scan k+1 and enqueues the copy of its
 * table, THEN exchanges and integrates scan k (whose table arrived meanwhile) — the device works on scan k's points while
 * the host waits for scan k+1's table.  Scans are integrated in submission order, one call late; flush() integrates
 * the last one (fdm_halo_routed_step flushes first).  h_matrix_prev / h_matrix_out: the counter matrix of the scan
 * that was integrated by the call (nullable).  The caller's arrays of scan k+1 are read by submit(k+1)'s kernels only. */
int fdm_halo_routed_submit(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* plan, const fdm_route_plan* route,
                           fdm_routed_ws* ws, uint64_t n, const float* d_x, const float* d_y, const float* d_z,
                           const float* d_intensity, const double T_base_sensor[16], const double T_world_base[16],
                           int32_t sensors, uint32_t* h_matrix_prev);
int fdm_halo_routed_flush(fdm_engine* e, void* nccl_comm, const fdm_tile_plan* plan, fdm_routed_ws* ws,
                          uint32_t* h_matrix_out);

/* ---- Transport ----
 * Every collective above goes through a small table of operations; the default one is RCCL on the engine's stream
 * (`nccl_comm` = an ncclComm_t).  A host may install another transport — the tests install one that stages through
 * host memory over torch.distributed/gloo, so that the multi-rank code of this library (per-peer offsets, grouping,
 * source order) runs with REAL peers on a box with one GPU, where RCCL refuses two ranks on one device.  `nccl_comm`
 * is then handed to the callbacks untouched as `comm`.  Byte counts; device pointers; every operation is issued on
 * the engine's stream (`hip_stream`) and may complete asynchronously on it.  Process-wide; NULL restores RCCL. */
typedef struct fdm_halo_transport {
  int (*all_gather)(void* comm, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);
  int (*group_start)(void* comm);
  int (*send)(void* comm, const void* d_buf, uint64_t bytes, int32_t peer, void* hip_stream);
  int (*recv)(void* comm, void* d_buf, uint64_t bytes, int32_t peer, void* hip_stream);
  int (*group_end)(void* comm, void* hip_stream);
} fdm_halo_transport;
void fdm_halo_set_transport(const fdm_halo_transport* t);

const char* fdm_halo_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
