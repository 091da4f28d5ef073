// fdm_engine_host.hpp — what the translation units of libfdm_engine.so share: the engine object behind the C ABI and the
// host helpers one unit defines for the others.  The library is THREE objects (round 6; one 2 000-line unit + 8 .inl +
// 13 kernel headers used to be one hipcc invocation, every A/B a full rebuild): fdm_engine.hip (the scan path: bin /
// update / batch kernels, layers, options), fdm_engine_ray.hip (the raycasting stage of one scan: voxel filter, radix
// sort, ray queue, walks, resolve), fdm_engine_post.hip (stencils, egress, ingest).  Non-template kernels in the shared
// headers are `inline` (weak symbols: one definition per library), a unit emits only the kernels it launches.
#pragma once
#include "../../include/fdm_engine.h"
#include "../../include/fdm_engine_debug.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>


#include "fdm_kernels.hpp"
#include "fdm_tiled.hpp"
#include "fdm_multi.hpp"
#include "fdm_route.hpp"
#include "fdm_raycast.hpp"
#include "fdm_raywedge.hpp"
#include "fdm_rbatch.hpp"
#include "fdm_rsort.hpp"
#include "fdm_egress.hpp"
#include "fdm_ingest.hpp"
#include "fdm_post.hpp"

using namespace fdm;

// (the helpers every translation unit of the library shares: defined in fdm_engine.hip)
namespace fdmh {
int fail(int code, const std::string& msg);  // remembers the message for fdm_last_error(), returns `code`
}
using namespace fdmh;

#define HIPCK(expr)                                                                        \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return fail(FDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

namespace fdmh {
struct Layer {
  std::string name;
  float* d = nullptr;    // own array (stride 1), or nullptr when the layer is a record field
  int field = -1;        // index inside the cell record, -1 = own array
  bool pending = false;  // allocated, but not yet visible (lazy intensity / colour layers)
};
}  // namespace fdmh

namespace {
// field order inside the cell records (KalmanField / P2Field in fdm_kernels.hpp)
const char* const kKalmanFields[KF_COUNT] = {"elevation", "elevation_min", "elevation_max", "variance",
                                             "n_points", "_kalman_p", "_sample_mean", "_sample_m2",
                                             "upper_bound", "lower_bound"};
const char* const kP2Fields[PF_COUNT] = {"elevation", "elevation_min", "elevation_max", "variance",
                                         "n_points", "_p2_q0", "_p2_q1", "_p2_q2", "_p2_q3", "_p2_q4",
                                         "_p2_n0", "_p2_n1", "_p2_n2", "_p2_n3", "_p2_n4",
                                         "upper_bound", "lower_bound"};

const char* const kP2Q[5] = {"_p2_q0", "_p2_q1", "_p2_q2", "_p2_q3", "_p2_q4"};
const char* const kP2N[5] = {"_p2_n0", "_p2_n1", "_p2_n2", "_p2_n3", "_p2_n4"};
}  // namespace

struct fdm_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  fdm_config cfg{};
  GeomConst G{};
  size_t ncell = 0;
  std::vector<Layer> layers;
  float** d_layer_ptrs = nullptr;  // device array of every layer pointer (strip clears)
  int n_layer_ptrs = 0;
  bool layer_ptrs_dirty = true;
  Scratch S{};
  DevState* d_state = nullptr;
  DevState* h_state = nullptr;  // pinned mirror for read-backs
  StatsOut* h_stats = nullptr;  // pinned + device-mapped: k_collect_stats writes here
  StatsOut* h_stats_dev = nullptr;  // the device's alias of h_stats
  unsigned long long stats_seq = 0; // sequence number of the last statistics launch (StatsOut::seq)
  int sync_spin_us = 150;           // read_stats polls the pinned block this long before a stream wait (option)
  StatsAcc* d_stats_acc = nullptr;
  uint64_t scan_no = 0;
  bool have_scan = false;
  uint32_t last_n = 0;        // points of the last scan as enqueued (array length of the captures / cell ids)
  uint32_t last_n_input = 0;  // ... as the reference counts them (cloud.size(): finite points of a PointCloud2)
  int next_drop_nonfinite = 0;  // set by fdm_engine_integrate_cloud2 for the scan it enqueues
  unsigned ingest_blocks = 0; // > 0: the last scan came through k_ingest_soa; its finite count is still on the device
  int last_was_integrate = 0;
  // staging for the host-pointer entry points
  float* d_stage = nullptr;
  float4* d_aos = nullptr;  // fdm_engine_integrate_points4 on pageable memory: the cloud's {x, y, z, 1} records
  size_t aos_cap = 0;
  size_t stage_cap = 0;  // in points
  int stage_rr = 0;      // rotating staging block
  int32_t* d_cell_ids = nullptr;
  size_t ids_cap = 0;
  bool want_ids = false;
  bool profile = false;
  bool wave_merge = true;
  int bin_table = 1;                 // k_bin: per-block LDS cell table (option "bin_table")
  // fdm_engine_integrate_async: scans of up to this many points whose arrays are PINNED host memory
  // are read in place by the bin kernel (0 = always stage with copy commands; option "zero_copy")
  int zero_copy = 1 << 30;
  int dbg_no_atomics = 0;
  int dbg_upd = 0;
  int bin_variant = 0;  // 0 = by scan size, 4 = k_bin4 (LDS-staged), 1 = k_bin (one point/thread)
  size_t bin_part_cap = 0;   // blocks
  unsigned last_bin_blocks = 0;
  std::vector<unsigned long long> h_bin_part;
  unsigned n_tiles = 0;
  std::vector<uint32_t> h_upd_part;
  bool obst_dense_pending = false;  // host wrote the obstacle layer / the pipeline changed: the next scan that observes a
                                    // cell clears it densely — which scan that is only the device knows (DevState::
                                    // dense_owed / dense_paid); the flag falls at the first sync behind it
  bool obst_owe_armed = false;      // ... the device has been told about the current debt
  unsigned obst_owe_seq = 0;
  bool estimator_ready = false;     // ElevationMapping ctor ran (ensureLayers + obstacle layer)
  bool use_records = true;          // pack the active estimator's state into cell records
  float* d_rec = nullptr;           // [ncell][rec_floats]
  int rec_kind = -1;                // -1 none, 0 Kalman, 1 P2
  int rec_floats = 0;
  float* d_tmp = nullptr;           // ncell floats: contiguous staging for strided layer transfers
  bool cap_pre = false, cap_ras = false;  // scan-callback captures
  bool cap_cov = false;                   // ... the preprocessed cloud with its 3x3 covariance channel
  float* d_cap = nullptr;            // 4 channels x cap_cap points
  size_t cap_cap = 0;
  // the preprocessed cloud of a scan whose raycasting stage is HELD BACK with its update (option "ray_hold"): by scan
  // parity — the next scan's bin half writes its own while the stage of this one has not run yet; 3 channels x rcap_cap
  float* d_rcap[2] = {nullptr, nullptr};
  size_t rcap_cap = 0;
  int ray_hold = 1;
  float* d_ras = nullptr;            // ncell
  bool saved_want_ids = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float last_ms[3] = {0.f, 0.f, 0.f};
  // raycasting stage (fdm_raycast.hpp)
  uint32_t* rc_cnt = nullptr;        // [ncell] ray-scan points observed in the cell this frame
  uint32_t* rc_min = nullptr;        // [ncell] ord(min ray height), kRayEmpty = not traversed
  uint32_t* ray_bins = nullptr;      // large scans: ray-queue bucket counts | offsets | block sums (fdm_raycast.hpp)
  unsigned long long* vkeys[2] = {nullptr, nullptr};  // voxel keys: unsorted / sorted
  uint32_t* vidx[2] = {nullptr, nullptr};             // point indices: unsorted / sorted
  uint32_t* vsel = nullptr;          // voxel_any output staging
  uint32_t* ray_blk = nullptr;       // rays queued per block of k_ray_compact (large scans: block-local queue regions)
  size_t vcap = 0;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int voxel_small = 1;               // option "voxel_small": scans of <= 64 K points take the sort-free voxel filter
  uint32_t* vs_cnt = nullptr;        // fine | coarse bucket counters | valid points of k_vs_*
  uint4* vs_rec = nullptr;           // {key, point, bucket start, bucket size} by position
  size_t vs_rec_cap = 0;
  int voxel_small_max = 1 << 16;     // option "voxel_small_max": largest scan that takes it
  VoxelSmall vs{};                   // the last small-scan filter's parameters (k_vs_mark runs from enqueue_ray_stage)
  hipEvent_t ev_ray[2] = {nullptr, nullptr};
  hipEvent_t ev_timer[2] = {nullptr, nullptr};  // fdm_engine_timer_start / _stop
  bool ray_timed = false;
  int dbg_ray = 0;
  // scans from this many points up: bucketed ray queue, one lane per ray (option "ray_large_min").  Stage time, shared-
  // ray segments vs this path: 131 K points 0.48 vs 0.52 ms, 262 K 0.78 vs 0.57, 524 K 1.17 vs 0.67, RGB-D 272 K 0.23 vs 0.19
  int ray_large_min = 196608;
  // large scans: the walk keeps an angular sector's minimum-height image in LDS (option "ray_wedge", fdm_raywedge.hpp;
  // 0 = one lane per ray on memory-side atomics, k_ray<., 1>)
  int ray_wedge = 1;
  int ray_wedge_parts = 0;           // option "ray_wedge_parts": workgroups per sector (0 = by the scan's size)
  // two stages in flight (option "ray_overlap", fdm_engine_ray.inl): the map-independent part of a large scan's stage
  // (voxel filter, queue, walk) leaves on a stream of its own as soon as the scan's bin half has been launched, by scan
  // parity; k_ray_resolve stays behind the scan's update on the main stream
  int ray_overlap = 0;               // option "ray_overlap": 0 = off (default), 1 = whenever possible, -1 = for synchronous calls (whose update
                                     // then runs beside the stage's first part) and scans of >= 1 M points.  Measured in a process of its
                                     // own (scripts/ray_overlap_ab.py): configs[3] streamed 336 -> 290 us per scan, synchronous integrate()
                                     // 0.344 -> 0.311 ms, configs[2] synchronous 0.106 -> 0.092 ms; a STREAM of 272 K-point scans is bound
                                     // by the host's launches and pays for the events: 95 -> 125 us.  OFF by default because the gain
                                     // depends on the process: HIP maps streams onto four hardware queues, and in a process that has used
                                     // more than that (bench.py by the time of its large raycasting leg) the three streams of a stage
                                     // share queues and the events serialise them: 339 -> 365 us (profiles/r06/ray_overlap_ab.txt)
  bool sync_call = false;            // a synchronous entry point is running (its flush follows the enqueue at once)
  struct RayBank {                   // the second set of the stage's buffers (ray_bank_swap)
    uint32_t *rc_cnt = nullptr, *rc_min = nullptr, *ray_bins = nullptr;
    unsigned long long* vkeys[2] = {nullptr, nullptr};
    uint32_t* vidx[2] = {nullptr, nullptr};
    uint32_t *vsel = nullptr, *ray_blk = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0, vcap = 0;
  } ray_bank1;
  hipStream_t ray_stream[2] = {nullptr, nullptr};
  hipEvent_t ev_ray_pre[2] = {nullptr, nullptr}, ev_ray_res[2] = {nullptr, nullptr}, ev_ray_bin = nullptr;
  bool ray_res_pending[2] = {false, false};  // ev_ray_res[k] has been recorded since the bank was last used
  bool ray_bin_marked = false;       // ev_ray_bin was recorded behind the current scan's bin half (enqueue_scan)
  // update(t) || bin(t+1) in ONE launch (k_update_bin): the update of the last small scan is held back
  // until the next scan arrives (or any other entry point / sync flushes it); the scratch is
  // double-buffered by scan parity.
  unsigned long long* key2[2] = {nullptr, nullptr};  // scratch of even / odd scans ([0] == the original allocation)
  uint4* aux2[2] = {nullptr, nullptr};
  uint2* zs2[2] = {nullptr, nullptr};
  bool overlap = true;          // option "overlap"
  bool chain = false;           // an update is held back: the next bin derives its geometry from the previous slot
  struct BinVariant { bool bin4, has_int, has_col, wave_merge; unsigned threads; int lean; };  // lean: see bin4_body
  // the held-back update (plain data: the layer set cannot change while it is pending, every entry
  // point that could change it flushes first)
  struct PendingUpdate {
    bool multi = false;     // a whole batch (fdm_multi.hpp): MU / ch are what matters
    MUpd MU;
    int ch = 0;
    bool tiled = false;     // large-scan pipeline (fdm_tiled.hpp) or the per-cell scratch one
    ScanParams P;
    Scratch S;              // scratch pipeline: the key / aux set of the scan's parity, captures
    ScanInputs in;          // scratch pipeline: where the winning points are gathered from
    TilePool Q;             // tiled pipeline: the record pool of the scan's parity
    TileAux A;
    unsigned upd_blocks = 0;
    // the scan's raycasting stage (fastdem.cpp:152-159), which runs right behind this update wherever that is launched
    bool ray = false;
    int ray_pre = 0;        // 1 + context if the stage's first part already left on a ray stream (ray_overlap)
    int ray_key_mode = 0;   // ... and what enqueue_voxel_sort left in the buffers
    RayParams RQ;
    const float *ray_x = nullptr, *ray_y = nullptr, *ray_z = nullptr;  // the scan's preprocessed cloud (d_rcap[parity])
    double ray_box[6] = {0, 0, 0, 0, 0, 0};
  } pend;
  // ---- tiled pipeline state (allocated when the first large scan arrives) ----
  bool borrow_inputs = false;       // option "borrow_inputs": a held-back update gathers from the CALLER's device arrays
  int tiled = 1;                    // option "tiled": large scans go through per-tile record pools
  unsigned tiled_min = 2048;        // ... from this many points up (on a map of >= 512 tiles the pipeline wins at every
                                    // size measured: 2 K points 13.1 vs 14.6 us, 32 K 16.6 vs 20.5, 262 K 18.9 vs 34.3)
  bool tiled_forced = false;        // tiled_min was set by hand (option "tiled_min"): no map-size condition
  int upd_blocks = 768;             // option "upd_blocks": update blocks (four tile wavefronts each) of a FUSED launch
  int upd_blocks_alone = 2048;      // option "upd_blocks_alone": ... of an update launch of its own
  int move_clear_basic = 0;         // option "move_clear_basic": GridMap::move()'s strips clear {elevation, elevation_min, elevation_max} only (the other reading of nanoGrid: DESIGN.md §6); such an engine takes no batch launches
  int upd_prio = 1;                 // option "upd_prio": update wavefronts run at raised issue priority
  int tiled_lds_pad = -1;           // option "tiled_lds_pad": extra dynamic LDS per block of the large-scan bin / fused launches; -1 = as much as
                                    // makes it SIX blocks per CU (seven: configs[3] 32.3 -> 31.8 us at six; five — what the fixed 4 KB of round 5
                                    // came to for a scan with an intensity channel, 29.7 KB per block — 31.4 -> 30.3 us at six, profiles/r06/probe_l.json)
  int cnt_shift = 5;                // option "cnt_shift": one tile counter per 2^cnt_shift words (TilePool::cnt_shift); takes effect before the pools exist
  int bin_delay = 0;                // option "bin_delay": see TileWork::delay (measurement: profiles/r06/probe_delay.json)
  int bin_delay_blocks = 1024;      // option "bin_delay_blocks"
  int bin_stagger = 0;              // option "bin_stagger": start stagger of the fused launch's first-round bin blocks (TileWork::stagger)
  size_t tile_rare_waves = 0;       // update wavefronts the rare-path scratch is sized for
  TileGrid TG{};
  TilePool pool[2] = {};            // by scan parity
  size_t pool_cap = 0;              // records per pool
  unsigned desc_stride = 0;
  uint32_t* tile_stamp32 = nullptr;
  uint32_t* upd_part32 = nullptr;
  uint32_t* tile_rare = nullptr;    // the update's rare-path scratch, 3 KB per update wavefront
  unsigned last_upd_tiles = 0;      // length of the per-tile statistics of the last scan
  uint32_t* last_upd_part = nullptr;
  int last_kind = -1;               // pipeline of the last scan (0 scratch, 1 tiled)
  int last_do_move = 0, last_gate = 0;
  // ---- batch pipeline (fdm_multi.hpp): up to kMaxBatch small scans per launch, allocated by the first batch ----
  int batch = 1;                     // option "batch": fdm_engine_integrate_device_batch groups eligible scans
  int batch_max = 0;                 // option "batch_max": scans per launch (2 .. kMaxBatch = 32); 0 = automatic: 32 with the quantile
                                     // estimator or with raycasting on (configs[1] with it: 9.25 -> 7.78 us per scan), 16 with Kalman
                                     // alone — measured (profiles/r06/batch_max.txt): configs[2] (P2, 272 K-point
                                     // scans) 52.8 -> 56.9 G pts/s at 32, configs[1] (Kalman, 28.8 K-point scans) 27.0 -> 26.2: that
                                     // launch is within ~2 x of its instruction-issue floor, a second round of blocks only adds its time
  int batch_fuse = 1;                // option "batch_fuse": hold a batch's update back for the next batch's bin launch
  unsigned long long* mkey[2] = {nullptr, nullptr};  // [kMaxBatch][ncell] per batch parity
  uint4* maux[2] = {nullptr, nullptr};
  uint2* mzs[2] = {nullptr, nullptr};
  float2* mobs[2] = {nullptr, nullptr};              // [kMaxBatch][mobs_stride]
  uint32_t* mcobs[2] = {nullptr, nullptr};
  size_t mobs_stride = 0;
  unsigned long long* mbin_part[2] = {nullptr, nullptr};
  size_t mbin_cap = 0;
  uint32_t* mupd_part = nullptr;     // [update blocks] touched cells of a batch's last scan
  MState* mstate = nullptr;          // [kMStates] ring, slot = batch number % kMStates
  unsigned mseq = 0;                 // batches enqueued so far
  int last_batch_n = 0;              // scans of the batch launch the last scan left in (0: it took the single-scan path)
  uint64_t n_mbatch = 0;             // batch launches since creation (fdm_engine_debug_batch_launches)
  bool fault_watch = false;          // a launch that can raise DevState::fault was enqueued since it was last read (none can since round 4)
  int dbg_batch = 0;                 // measurement only (option "dbg_batch")
  int batch_crop = 1;                // option "batch_crop": evaluate the next batch's crops one launch ahead
  int batch_walk = -1;               // option "batch_walk": the chain of moves walked one launch ahead (fdm_multi.hpp mwalk_body): -1 = for the quantile estimator only, 0 off, 1 on
  unsigned long long batch_call = 0; // calls of fdm_engine_integrate_device_batch so far: a look-ahead is only ever honoured inside the call that made it
  unsigned long long pre_call = 0;
  bool pre_valid = false;            // the last launch carried the crop pass of the batch (pre_scans, pre_count) = number pre_seq
  const fdm_device_scan* pre_scans = nullptr;
  uint32_t pre_count = 0;
  unsigned pre_seq = 0;
  const unsigned long long* last_bin_part = nullptr;  // per-block statistics of the last scan (either pipeline)
  // ---- raycasting inside the small-scan batches (fdm_rbatch.hpp) ----
  int batch_ray = 1;                 // option "batch_ray": 0 = an engine with raycasting on takes the single-scan path
  int batch_ray_seg = 4;             // option "batch_ray_seg": lanes per ray of k_rb_ray (1, 4, 8, 16)
  int batch_ray_lds = 1;             // option "batch_ray_lds": 0 = always the global-atomic walk (k_rb_ray)
  int batch_ray_parts = 0;           // option "batch_ray_parts": workgroups per quadrant and scan of k_rb_ray_lds (0 = fill the chip)
  int batch_ray_words = 0;           // option "batch_ray_words": LDS image words of k_rb_ray_lds (0 = twice a centred sensor's quadrant)
  unsigned rb_lds_words = 0;         // dynamic LDS k_rb_ray_lds may use, in 32-bit words (0: not asked yet)
  RState* rb_state = nullptr;
  float* rb_cap = nullptr;           // [3][kMaxBatch][rb_stride] preprocessed clouds of the batch being binned
  uint32_t* rb_u32 = nullptr;        // keys | place | sel | ray_list, [kMaxBatch][rb_stride] each
  uint4* rb_rec = nullptr;           // [kMaxBatch][rb_stride]
  uint32_t* rb_counters = nullptr;   // fine [kMaxBatch][2^18] | coarse [kMaxBatch][kVsCoarse]
  uint32_t* rb_img = nullptr;        // rc_cnt [kMaxBatch][ncell] | rc_min [kMaxBatch][ncell]
  size_t rb_stride = 0;
  unsigned rb_seq = 0;               // stamp of the last batch's ray launches (RState::any)
  float* d_bstage = nullptr;         // fdm_engine_integrate_host_batch: pageable clouds of a call, staged back to back
  size_t bstage_cap = 0;             // floats
  bool bstage_busy = false;          // launches of the previous call may still be reading it
  // scan routing (fdm_route.hpp)
  uint8_t* d_route_owner = nullptr;  // [route_cap] owner rank of every point of the slice
  uint32_t* d_route_cnt = nullptr;   // [route_blocks_cap][world + 2] block counts -> offsets | [kMaxRanks] bases at the end
  size_t route_cap = 0, route_blocks_cap = 0;
  // stencil post-processing (fdm_post.hpp)
  RegionEntry* d_region = nullptr;   // region_cap entries
  size_t region_cap = 0;
  float* d_post_pool = nullptr;      // per-thread lists of the big-neighbourhood stencil kernels
  size_t post_pool_bytes = 0;
  FeatEntry* d_feat_tab = nullptr;   // kMaxRegion entries: the region as k_features_tiled reads it
  std::vector<RegionEntry> h_region; // what d_region holds (upload_region skips an identical table)
  std::vector<FeatEntry> h_feat_tab; // what d_feat_tab holds
  int dbg_post = 0;                  // measurement only: 1 = untiled feature kernel, 32 = fusion with integer samples, 64 = features with min / max chains
  unsigned long long* d_timeline = nullptr;  // measurement only: {start, end} ticks per block of the last fused launch
  unsigned timeline_cap = 0;         // blocks the buffer holds
  unsigned timeline_blocks = 0, timeline_upd = 0;  // grid of the last fused launch, its update blocks
  unsigned timeline_bin = 0;         // ... its bin blocks (batch launches: the rest are crop blocks)
  float* d_tmp2 = nullptr;           // second ncell staging array (fusion works on two layers)
  // ingest (fdm_ingest.hpp)
  uint8_t* d_blob = nullptr;         // raw message bytes
  size_t blob_cap = 0;
  float* d_in = nullptr;             // 5 channels x in_cap: x y z intensity rgb
  size_t in_cap = 0;
  uint64_t in_n = 0;
  bool in_has_int = false, in_has_rgb = false;
  // egress (fdm_egress.hpp)
  uint32_t* pack_counts = nullptr;   // per-block valid counts / offsets (+1 for the total)
  size_t pack_counts_cap = 0;
  float* d_pack = nullptr;           // packed records
  size_t pack_cap = 0;               // in floats
};

// ---- helpers shared by the library's translation units (fdm_engine.hip | fdm_engine_ray.hip | fdm_engine_post.hip) ----
namespace fdmh {
int join_streams(fdm_engine* e);
void poll_dense_paid(fdm_engine* e);
int sync_all(fdm_engine* e);
Layer* find_layer(fdm_engine* e, const char* name);
float* lptr(fdm_engine* e, const Layer& l);
int lstride(fdm_engine* e, const Layer& l);
int fill_async(fdm_engine* e, float* p, float v, size_t n, int stride = 1);
int copy_strided(fdm_engine* e, float* dst, int ds, const float* src, int ss);
int ensure_tmp(fdm_engine* e);
int add_layer(fdm_engine* e, const char* name, float value, bool pending = false);
int ensure_layer(fdm_engine* e, const char* name, float value);
int refresh_layer_ptrs(fdm_engine* e);
int resolve_pending(fdm_engine* e);
float* L(fdm_engine* e, const char* n);
float* Lany(fdm_engine* e, const char* n, int* stride);
void fill_integrate_params(fdm_engine* e, ScanParams& P, const double* Tbs, const double* Twb);
int enqueue_scan(fdm_engine* e, ScanParams& P, uint64_t n, const float* dx, const float* dy, const float* dz,
                 const float* dint, const uint32_t* drgb, const float* dvar, const ScanInputs* gather = nullptr);
int read_stats(fdm_engine* e, fdm_scan_stats* out, int* status);
int ensure_stage(fdm_engine* e, size_t n);
const void* pinned_alias(const void* p);
constexpr int kStageSlots = 3;  // rotating staging blocks (see ensure_stage)
int stage_inputs(fdm_engine* e, uint64_t n, const float* x, const float* y, const float* z,
                 const float* a, const uint32_t* rgb, const float* v, const float** dx,
                 const float** dy, const float** dz, const float** da, const uint32_t** drgb,
                 const float** dv, ScanInputs* gather = nullptr);
// the raycasting stage (fdm_engine_ray.hip), called from the scan path
bool voxel_size_ok(float v);
int ensure_ray_layers(fdm_engine* e);
VoxelCompact voxel_compact_of(float voxel_size, const double* box);
void ray_box_of(const fdm_engine* e, const ScanParams& P, double box[6]);
int enqueue_voxel_sort(fdm_engine* e, unsigned n, float voxel_size, int flag_slot, const float* dx,
                       const float* dy, const float* dz, const double* box, int* key_mode);
fdm_raycast_config ray_config_of(const fdm_config& c);
RayParams make_ray_params(fdm_engine* e, const fdm_raycast_config& c, const float* origin, unsigned n,
                          int slot, int flag_slot);
int enqueue_ray_stage(fdm_engine* e, const RayParams& Q, bool voxel, const float* dx, const float* dy,
                      const float* dz, int key_mode = 0, int phase = 3);  // phase: 1 = everything but k_ray_resolve, 2 = k_ray_resolve, 3 = both
int run_held_ray_stage(fdm_engine* e, fdm_engine::PendingUpdate& u);
int start_ray_stage_early(fdm_engine* e, fdm_engine::PendingUpdate& u, const ScanParams& P);
// the large-scan pipeline (fdm_engine_tiled.hip)
int ensure_tile_aux(fdm_engine* e);
int ensure_tile_pool(fdm_engine* e, size_t records, unsigned blocks, bool has_int, bool has_col);
unsigned update_blocks(const fdm_engine* e, bool fused);
int launch_tbin(fdm_engine* e, const ScanParams& P, const ScanInputs& in, const TilePool& Q, int32_t* ids,
                unsigned bin_blocks, fdm_engine::BinVariant bv);
int launch_tiled_update_alone(fdm_engine* e, const fdm_engine::PendingUpdate& u);
int launch_tiled_update_fused(fdm_engine* e, const fdm_engine::PendingUpdate& u, const ScanParams& Pb,
                              const ScanInputs& Ib, const TilePool& Qb, int32_t* ids_b, unsigned bin_blocks_b,
                              fdm_engine::BinVariant bv);
// the batch pipeline (fdm_engine_multi.hip)
uint32_t multi_run(fdm_engine* e, uint32_t count, const fdm_device_scan* scans);
int enqueue_multi(fdm_engine* e, uint32_t count, const fdm_device_scan* scans, uint32_t next_count);
int launch_multi_update(fdm_engine* e, const fdm_engine::PendingUpdate& u);
int launch_update_alone(fdm_engine* e, const fdm_engine::PendingUpdate& u);
P2Params p2_params(const fdm_config& c);
void sensor_params(const fdm_config& cfg, int& type, float* sp);
int ensure_estimator_layers(fdm_engine* e);
int ensure_scratch_channels(fdm_engine* e, bool intensity, bool color);
void rotation_of_product(const double* Twb, const double* Tbs, float* R);

// f(policy tag, layer set) for the engine's estimator and layer layout
template <typename F>
int with_policy(fdm_engine* e, F&& f) {
  const bool p2mode = e->cfg.estimation_type == 1;
  if (e->rec_kind >= 0) {  // cell records
    if (p2mode) {
      P2RecLayers Lr{};
      Lr.rec = e->d_rec; Lr.obstacle = L(e, "obstacle"); Lr.color = L(e, "color");
      Lr.intensity = Lany(e, "intensity", &Lr.istride);
      Lr.p = p2_params(e->cfg);
      return f(P2RecPolicy{}, Lr);
    }
    KalmanRecLayers Lr{};
    Lr.rec = e->d_rec; Lr.obstacle = L(e, "obstacle"); Lr.color = L(e, "color");
    Lr.intensity = Lany(e, "intensity", &Lr.istride);
    Lr.min_var = e->cfg.kalman_min_variance; Lr.max_var = e->cfg.kalman_max_variance;
    Lr.q = e->cfg.kalman_process_noise;
    return f(KalmanRecPolicy{}, Lr);
  }
  if (p2mode) {
    P2Layers Lp{};
    Lp.elevation = L(e, "elevation");
    Lp.elevation_min = L(e, "elevation_min");
    Lp.elevation_max = L(e, "elevation_max");
    Lp.variance = L(e, "variance");
    Lp.n_points = L(e, "n_points");
    Lp.upper = L(e, "upper_bound");
    Lp.lower = L(e, "lower_bound");
    Lp.obstacle = L(e, "obstacle");
    Lp.intensity = L(e, "intensity");
    Lp.color = L(e, "color");
    for (int k = 0; k < 5; ++k) {
      Lp.q[k] = L(e, kP2Q[k]);
      Lp.n[k] = L(e, kP2N[k]);
    }
    Lp.p = p2_params(e->cfg);
    return f(P2Policy{}, Lp);
  }
  KalmanLayers Lk{};
  Lk.elevation = L(e, "elevation");
  Lk.elevation_min = L(e, "elevation_min");
  Lk.elevation_max = L(e, "elevation_max");
  Lk.variance = L(e, "variance");
  Lk.n_points = L(e, "n_points");
  Lk.kalman_p = L(e, "_kalman_p");
  Lk.sample_mean = L(e, "_sample_mean");
  Lk.sample_m2 = L(e, "_sample_m2");
  Lk.upper = L(e, "upper_bound");
  Lk.lower = L(e, "lower_bound");
  Lk.obstacle = L(e, "obstacle");
  Lk.intensity = L(e, "intensity");
  Lk.color = L(e, "color");
  Lk.min_var = e->cfg.kalman_min_variance;
  Lk.max_var = e->cfg.kalman_max_variance;
  Lk.q = e->cfg.kalman_process_noise;
  return f(KalmanPolicy{}, Lk);
}
template <typename POLICY>
constexpr bool is_rec_policy = std::is_same<POLICY, KalmanRecPolicy>::value || std::is_same<POLICY, P2RecPolicy>::value;


// kernels with more than 64 KB of dynamic LDS need the attribute once
template <typename K>
int allow_lds(K kern, unsigned bytes) {
  static std::mutex mu;
  static std::unordered_map<const void*, unsigned> seen;
  if (bytes <= 65536u) return FDM_OK;
  const void* f = reinterpret_cast<const void*>(kern);
  std::lock_guard<std::mutex> lock(mu);
  auto it = seen.find(f);
  if (it != seen.end() && it->second >= bytes) return FDM_OK;
  HIPCK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
  seen[f] = bytes;
  return FDM_OK;
}

}  // namespace fdmh
