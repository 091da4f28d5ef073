// fdm_multi.hpp — a BATCH of small scans in ONE launch instead of one launch per scan.
//
// Why: a VLP-16 scan (28.8 K points, 113 bin blocks, 88 update blocks) is launch- and
// latency-bound: one fused launch per scan costs ~6 us — ~3 us of it the queue's own per-kernel floor, the rest
// two or three dependent memory round trips of ~3 us each under load.  fdm_engine_integrate_device_batch sees its
// scans up front, so up to kMaxBatch of them are binned by one launch into per-scan scratch sets and every cell
// thread of the update applies the batch's observations in scan order.  Per cell the reference only fixes the
// order of the scans (elevation_mapping.cpp:94-125: one independent update per observed cell and scan;
// GridMap::move strips and the obstacle-layer clear happen between scans), and that order is kept exactly — the
// map after the batch is bit-identical to integrating the scans one by one.
//
// One launch per batch, k_mbatch = [ update of batch b-1 | bin of batch b | scouts (+ walker) of batch b+1 ]:
//
//   scout rows   Scan k is binned against the map geometry AFTER the LOCAL-mode moves of scans 0..k, and whether
//                scan j moved the map depends on whether any of its points survived the crops (fastdem.cpp:138) —
//                device-side data.  The crops need no geometry, so the question is answered ONE LAUNCH AHEAD: a few
//                scout blocks per scan of the next batch (first transform + cropRange + cropZ, the same float
//                operations the bin half will run; first survivor, early exit) OR "scan k has a surviving point"
//                into the next batch's state word; the kernel boundary is the barrier.  The first batch of a call has
//                no launch ahead of it: its scouts run as a small launch of their own.  (Rounds 3: an in-launch protocol
//                for that case — every block adds itself to its scan's counter, later scans' blocks poll a state word,
//                bounded spin, a sticky fault when it ran out.  Gone: no block of this kernel ever waits for another.)
//                One more block of these rows, the WALKER (mwalk_body), walks the next batch's chain of moves ahead of
//                its launch, assuming every scan passes (on for the quantile estimator: option "batch_walk").
//   bin half     512 points per block (two per thread), scan k of the batch; the first wavefront walks the chain of k
//                moves (every lane rounds its own scan's pose once; only the position is accumulated in scan order;
//                the reference's divide only for poses on a rounding tie) while the point loads are in flight.
//                Points fold straight into a per-block LDS cell table (axis_fast indexing), the table's occupied
//                entries are compacted and flushed with memory-side atomics on the block's unique cells.
//                No second look at the scan: the thread that merges a block-local minimum into the scratch also
//                evaluates that point's sigma_z^2 (the block keeps its points' sensor-frame coordinates in LDS) and
//                stores {map-frame z, sigma_z^2} at the POINT's index in an observation array; likewise the colour
//                of a block-local last point.  The update finds the winner's entry through the index in the
//                reduced key, and the caller's arrays are dead as soon as the launch has run.
//   update half  64 cells per block (memory order), four threads per cell.  Round trip 1: the cell's keys of
//                all scans (four scans per thread), its estimator record, the per-scan geometry.  The block's
//                (cell, scan) events are then compacted into an LDS list (kEvCap entries) and spread evenly over
//                the threads: round trip 2 fetches {aux, zs, obs} of ALL events at once (a thread holds 2-3 of them
//                whatever their distribution over the cells), the observations go back through LDS to the cell's
//                lead thread, which walks its events in scan order: strips vacated since the previous event,
//                estimator step in registers, one record store.  With raycast_enabled (template RAY) the walk also
//                resolves every scan's raycasting stage behind that scan's observation — the images come from the
//                launches of fdm_rbatch.hpp between the launch that binned the batch and this one.
//
// All per-batch parameters travel as kernel arguments (< 4 KB): no copy command, no upload kernel.
// Algorithmic bytes are per scan what they were: 12 B/point (+4 intensity, +4 colour), 72 / 124 B per
// touched cell, 4 B per map cell per scan for the obstacle clear.
#pragma once

#include "fdm_kernels.hpp"

namespace fdm {

constexpr int kMaxBatch = 32;            // scans per launch (one bit per scan in 32-bit state words).  Round 6: 16 -> 32 — a
                                         // 16-scan VLP-16 batch is LESS than one round of blocks on the chip and pays the launch's
                                         // fixed costs (3.4 us queue floor, the first round trip of 1 264 blocks at once) in full
constexpr int kScansPerThread = 4;   // update half: a thread looks after four scans of its cell, so a cell takes four threads in
                                     // a batch of up to 16 scans (64 cells per block) and eight beyond (32 cells per block):
                                     // MUpd::tpc — at most 1 024 (cell, scan) events per block = two rounds of the exchange either way
// the n lowest bits (n <= 32: a shift by 32 is not a shift)
__host__ __device__ constexpr unsigned lowbits(unsigned n) { return n >= 32u ? 0xFFFFFFFFu : (1u << n) - 1u; }
constexpr int kLineWords = 32;           // a 128-byte line of 32-bit words
#ifndef FDM_MB_WAVES
#define FDM_MB_WAVES 6  // waves per SIMD k_mbatch is compiled for (<= 80 VGPRs; the LDS allows 6 blocks per CU): every block of a 16-scan VLP-16 batch resident at once
#endif
constexpr int kMStates = 4;              // ring of batch states: update b-1 | bin b | crop b+1 | being re-armed

// Device-resident bookkeeping of one batch.  Every word other blocks poll or add to sits on its own 128-byte
// line: pollers of `flags` must not queue up in front of the adds to `done`.
struct MState {
  DevGeom E[kMaxBatch];        // geometry before scan k (written by the scan's first bin block)
  DevCand C[kMaxBatch];        // geometry after its move + the index shift
  unsigned flags[kLineWords];  // [0]: bit k = scan k has a surviving point (scouts); [1]: the pass bits PE / PC below assume; [2]: != 0 = PE / PC are valid
  unsigned inside[kMaxBatch];  // some point of scan k landed in the map (elevation_mapping.cpp:118)
  unsigned pad[16];
  DevGeom PE[kMaxBatch];       // the chain of moves walked ONE LAUNCH AHEAD by the walker block (mwalk_body): geometry before
  DevCand PC[kMaxBatch];       // scan k / after its move, assuming the pass bits in flags[1]
};

struct MCommon {  // what all scans of a batch share
  float min_sq, max_sq, z_min, z_max;
  float sp[4];
  float Tbs[16];  // T_base_sensor (one sensor per batch: a scan with another extrinsic closes the batch)
  int sensor_type, integrate_mode, do_move, gate_on_filter, has_var, bin_table;
  int dbg, walk;                 // walk: 1 = the chain of moves is walked one launch ahead (mwalk_body).  dbg, measurement only: 1 = no scratch atomics, 2 = no chain walk (both: wrong results); 3 = every move by the reference's divide; 4 = a chain wait reports MState::err as if it had run out of polls (tests)
  unsigned long long* timeline;  // measurement only (nullable): {start, end} of every block in 100 MHz ticks
};
struct MScanT {   // per scan: T_world_base without its constant last row (0 0 0 1), column-major 3 x 4 | rotation of the product
  float Twb[12], R[9];
};
struct MBin {     // bin half: batch b
  unsigned count, scan_no0;
  unsigned first_block[kMaxBatch + 1];   // bin blocks before scan k
  unsigned n[kMaxBatch];                 // points of scan k
  MState* ms;
  const MState* prev;                    // the previous batch's state while its update shares this launch (else null)
  unsigned prev_count;
  unsigned obs_stride;                   // points per scan slot of obs / cobs
  unsigned pre;                          // (always 1: scouts decided the batch's pass bits before this launch)
  unsigned pad;
  // per-scan scratch, slot k at + k * ncell (key, aux, zs as in Scratch)
  unsigned long long* key;
  uint4* aux;
  uint2* zs;
  float2* obs;                           // [count][obs_stride] {map-frame z, sigma_z^2} of block-local minima, by point index
  uint32_t* cobs;                        // [count][obs_stride] colour of block-local last points, by point index
  unsigned long long* bin_part;          // [bin blocks]
  float* cap;                            // raycasting: [3][count][obs_stride] map-frame x (NaN: dropped by the crops), y, z by point index (else null)
  double robot_x[kMaxBatch], robot_y[kMaxBatch];  // T_world_base translation of every scan (the chain of moves)
  const float* px[kMaxBatch];
  const float* py[kMaxBatch];
  const float* pz[kMaxBatch];
  const float* pint[kMaxBatch];
  const uint32_t* prgb[kMaxBatch];
  const float* pvar[kMaxBatch];
  MScanT t[kMaxBatch];
};
// Raycasting stage of the batch's scans (fastdem.cpp:152-159, once per scan BEHIND that scan's map update).  The ray
// launches of fdm_rbatch.hpp leave, per scan, the two order-free images of processScan — observed-evidence counts and the
// minimum ray height per cell — and the update half resolves them cell by cell, each scan's behind that scan's
// observation (resolveGhostCells, raycasting.cpp:175-202).
struct RState {   // device bookkeeping of a batch's ray launches (one per engine: they run between the launch that bins the batch and the one that updates it)
  unsigned any[kMaxBatch];        // == the batch's stamp: the voxel-filtered scan k is not empty (raycasting.cpp:207-209)
  unsigned origin_in[kMaxBatch];  // the sensor origin lies in the map after scan k's move (raycasting.cpp:217-220)
  unsigned qcount[kMaxBatch][4];  // queued downward rays of scan k, by the quadrant they leave the sensor's cell into
  unsigned total[kMaxBatch];      // valid points of scan k (VoxelSmall::total)
};
struct MRay {
  const RState* rs;
  unsigned stamp;          // 0: no raycasting in this batch
  unsigned pad;
  uint32_t* rc_cnt;        // [count][ncell] observed-evidence counts; zero between batches
  uint32_t* rc_min;        // [count][ncell] ord(min ray height); kRayEmpty between batches
  float* logodds;          // _visibility_logodds
  float* ray_min;          // raycasting
  float* ghost;            // ghost_removal
  float l_obs, l_ghost, l_max, clear_thr, conflict_thr;
  float pad2;
};
struct MUpd {     // update half: batch b-1
  unsigned count, scan_no0, obs_stride;
  int do_move, gate_on_filter;
  unsigned pad;
  MRay ray;
  MState* ms;
  MState* rearm;                         // the state of the batch after next: zeroed by the committing block
  unsigned long long* key;
  uint4* aux;
  uint2* zs;
  const float2* obs;
  const uint32_t* cobs;
  uint32_t* upd_part;                    // [update blocks] cells touched by the batch's LAST scan
};
struct MCrop {    // crop half: batch b+1 (count == 0: none)
  unsigned count, pad;
  unsigned first_block[kMaxBatch + 1];
  unsigned n[kMaxBatch];
  MState* ms;
  // the walker block: the scouted batch's poses, and — when no batch is binned by this launch (the scout-only launch in
  // front of a call's first batch) — where its chain starts: the previous batch's state (complete), or the geometry ring
  double robot_x[kMaxBatch], robot_y[kMaxBatch];
  const MState* prev;
  unsigned prev_count, scan_no0;
  const float* px[kMaxBatch];
  const float* py[kMaxBatch];
  const float* pz[kMaxBatch];
};

// The members the arithmetic reads, by name (pointers only: an array member copied in device code is spilled,
// see DevObst).
struct MView {
  const float* Tbs;
  const float* Twb12;
  const float* R;
  const float* sp;
  float min_sq, max_sq, z_min, z_max;
  int sensor_type, integrate_mode;
};

// p <- T * p for a T whose last row is (0 0 0 1), stored without it: the same products and sums, in the same
// order, as transform4 on the full matrix (the row's coefficients are the literals 0 and 1).
__device__ __forceinline__ void transform4_affine(const float* __restrict__ t12, float& x, float& y, float& z, float& w) {
  float o[4];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = t12[0 + r] * x;
    acc = t12[3 + r] * y + acc;
    acc = t12[6 + r] * z + acc;
    acc = t12[9 + r] * w + acc;
    o[r] = acc;
  }
  {
    float acc = 0.0f * x;
    acc = 0.0f * y + acc;
    acc = 0.0f * z + acc;
    acc = 1.0f * w + acc;
    o[3] = acc;
  }
  x = o[0]; y = o[1]; z = o[2]; w = o[3];
}
// first half of preprocess_point (T_base_sensor, cropRange, cropZ): does the point survive?  w rides along.
__device__ __forceinline__ bool mcrops(const MView& V, float& x, float& y, float& z, float& w) {
  w = 1.0f;
  transform4(V.Tbs, x, y, z, w);
  const float d2 = sum3(x * x, y * y, z * z);
  bool pass = (d2 >= V.min_sq) && (d2 <= V.max_sq);
  pass = pass && (z >= V.z_min) && (z <= V.z_max);
  return pass;
}

// LDS of one block: the bin half's cell table + point staging, or the update half's event exchange.
constexpr int kEvCap = 512;  // (cell, scan) events of a 256-cell tile exchanged per round (two per thread)
#ifndef FDM_M_PTS
#define FDM_M_PTS 2
#endif
constexpr int kMPts = FDM_M_PTS;                 // points per thread of the bin half: 512-point blocks — half the blocks (every block
constexpr unsigned kMBlock = 256u * kMPts;  // of a 16-scan batch resident beside the update half) and ~20 % fewer (block, cell) pairs
struct MBinLds {
  unsigned long long t_key[kMBlock];
  uint32_t t_cell[kMBlock], t_zmx[kMBlock], t_imx[kMBlock], t_fst[kMBlock], t_lst[kMBlock];
  float4 s_pt[kMBlock];  // sensor-frame x, y, z | map-frame z of the block's points
  uint16_t s_list[kMBlock];  // the occupied table slots, compacted for the flush
  DevCand s_cand;
  unsigned s_pass[4], s_in[4], s_occ[4 * kMPts];
};
struct MEvent { uint32_t idx; uint16_t cell, k; };  // winner's point index | cell in tile | scan
struct MObs { float min_z, var, max_z, iobs; };
constexpr unsigned kUpdCells = 64u;   // cells per update block at most (MUpd::tpc == 4; 32 with eight threads per cell)
__host__ __device__ constexpr unsigned upd_threads_per_cell(unsigned count) { return count > 16u ? 8u : 4u; }
__host__ __device__ constexpr unsigned upd_cells_per_block(unsigned count) { return 256u / upd_threads_per_cell(count); }
constexpr uint32_t kMRayEmpty = 0xFFFFFFFFu;  // (= kRayEmpty, fdm_raycast.hpp)
template <bool COL, bool RAY>
struct MUpdLds {
  MEvent ev[kEvCap];
  MObs ob[kEvCap];
  uint32_t rgb[COL ? kEvCap : 1];
  uint32_t rcnt[RAY ? kMaxBatch * kUpdCells : 1], rmin[RAY ? kMaxBatch * kUpdCells : 1];  // [scan][cell of the block]
  unsigned s_w[4], s_t[4];
};
template <bool COL, bool RAY>
constexpr unsigned kMLdsBytes = sizeof(MUpdLds<COL, RAY>) > sizeof(MBinLds) ? sizeof(MUpdLds<COL, RAY>) : sizeof(MBinLds);

// ---------------------------------------------------------------------------------------------
// crop half: SCOUT blocks of the NEXT batch's scans — does scan k hold a point that survives the crops (then it moves a
// LOCAL map, fastdem.cpp:138-145)?  kMScout blocks per scan walk it with a grid stride, kMBlock points per block and
// step, and leave at the first step in which any thread of the block sees a survivor: for a real scan that is the
// first step, only a scan without survivors (a covered sensor) is read to its end.  (Round 3 ran the crops over EVERY
// point of the next batch here — as many blocks again as the bin half, a third of the launch's blocks and of its input
// traffic, for sixteen bits: configs[2] 10.5 us per scan, VERDICT r03 #6.)
constexpr unsigned kMScout = 4u;
__device__ __forceinline__ void mcrop_body(const MCrop& Cn, const MCommon& K, const unsigned k, const unsigned sb) {
  MView V;
  V.Tbs = K.Tbs; V.Twb12 = nullptr; V.R = nullptr; V.sp = K.sp;
  V.min_sq = K.min_sq; V.max_sq = K.max_sq; V.z_min = K.z_min; V.z_max = K.z_max;
  V.sensor_type = K.sensor_type; V.integrate_mode = K.integrate_mode;
  const unsigned n = Cn.n[k];
  const float* __restrict__ const px = Cn.px[k];
  const float* __restrict__ const py = Cn.py[k];
  const float* __restrict__ const pz = Cn.pz[k];
#pragma unroll 1
  for (unsigned i0 = sb * kMBlock; i0 < n; i0 += kMScout * kMBlock) {  // (block-uniform)
    bool pass = false;
#pragma unroll
    for (int h = 0; h < kMPts; ++h) {
      const unsigned i = i0 + unsigned(h) * 256u + threadIdx.x;
      if (i < n) {
        float x = px[i], y = py[i], z = pz[i], w;
        pass = mcrops(V, x, y, z, w) || pass;
      }
    }
    if (__syncthreads_or(pass ? 1 : 0)) {
      if (threadIdx.x == 0) atomicOr(&Cn.ms->flags[0], 1u << k);  // (<= kMScout per scan)
      return;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The chain of LOCAL-mode moves of a batch, walked ONCE and one launch ahead.  Scan k is binned against the geometry
// after the moves of scans 0 .. k - 1, and every bin block used to walk that chain itself behind its point loads — up to
// fifteen dependent moves (~0.3 us each on a full chip: every second pose of a robot advancing half a cell per scan sits
// on a rounding tie and takes the reference's subtract + divide), 4.5 us for the blocks of a batch's sixteenth scan, which
// is what ended the launch.  One extra block of the scout rows (the WALKER) walks the NEXT batch's chain while this
// launch runs — geometry before / after every scan's move into MState::PE / PC — assuming every scan of it passes the
// crops (the scouts' answer is not complete before the launch ends; a scan with no surviving point is a covered sensor).
// A bin block whose scan's predecessors did pass takes PE[k] / PC[k] (two loads that leave with its point loads); if
// one did not, it walks as before.  Measured (16 scans per launch): configs[2] (P2, colour; 87 us launches) 5.40 -> 4.79 us
// per scan; configs[1] (Kalman; 16 us launches) unchanged on streamed inputs and SLOWER on cache-resident ones (1.02 ->
// 1.30 us per scan): that launch is bound by its memory-side atomics, the chains staggered the sixteen rows of bin blocks
// by 0.3 us each and left the update half the first microseconds — all rows at once make the update half the last to
// finish (phase stamps: 13.6 -> 17.1 us).  So the walker is ON for the quantile estimator and OFF for Kalman by default
// (option "batch_walk": -1 automatic, 0 off, 1 on).  The walk itself is the reference's sequence, scan by scan: geometry before scan
// k + 1 = scan k's candidate if scan k moved the map, else unchanged — the same numbers the per-block walk produces
// (it defers the index wrap of short moves, which yields the candidate's wrapped index).
//
// Geometry before the first scan of a batch, from the state of the batch before it (complete) or the geometry ring.
__device__ __forceinline__ DevGeom mbatch_start(const MState* __restrict__ prev, unsigned prev_count, unsigned scan_no0,
                                                const MCommon& K, const DevState* __restrict__ st) {
  if (!prev) return st->geom[scan_no0 & 3u];
  const unsigned pk = prev_count - 1u;
  DevGeom g = prev->E[pk];
  if (K.do_move && (!K.gate_on_filter || ((prev->flags[0] >> pk) & 1u) != 0u)) {
    const DevCand c = prev->C[pk];
    g.px = c.px; g.py = c.py; g.sr = c.sr; g.sc = c.sc;
  }
  return g;
}
// One batch's chain from `g` on (every lane of one wavefront, uniformly; lane j holds scan j's pose): E / C of every scan
// into `out` (nullable), the last scan's returned.  The arithmetic is mbin_body's chain walk (see there for `nearest`,
// the margins and the divide).
__device__ __forceinline__ void mwalk_batch(DevGeom g, const unsigned count, const double* __restrict__ rx,
                                            const double* __restrict__ ry, const unsigned passmask, const MCommon& K,
                                            const GeomConst& G, MState* __restrict__ out, DevGeom& e_last, DevCand& c_last) {
  const unsigned lane = threadIdx.x & 63u;
  double pose_x = 0.0, pose_y = 0.0;
  if (lane < count) { pose_x = rx[lane]; pose_y = ry[lane]; }
  auto lane_f64 = [](double v, unsigned j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), int(j));
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), int(j));
    return __hiloint2double(hi, lo);
  };
  auto nearest = [&](double a, double p0, int& t) -> bool {
    t = 0;
    if (!(fabs(p0) < 1.0e7)) return false;
    if (fabs(a) <= 1e-4) return true;
    const double ue = a + 0.5 * (a > 0 ? 1 : -1);
    t = static_cast<int>(ue);
    const double f = fabs(ue - double(t));
    return f > 1e-4 && f < 1.0 - 1e-4 && fabs(ue) < 1.0e6;
  };
  int tx = 0, ty = 0;
  const bool sx = nearest((pose_x - g.px) * G.inv_res, g.px, tx), sy = nearest((pose_y - g.py) * G.inv_res, g.py, ty);
  const unsigned sure_x = K.dbg == 3 ? 0u : uni(unsigned(__ballot(sx))), sure_y = K.dbg == 3 ? 0u : uni(unsigned(__ballot(sy)));
  int vx = 0, vy = 0;
  DevCand c;
  c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
#pragma unroll 1
  for (unsigned k = 0; k < count; ++k) {
    int dx, dy;
    if ((sure_x >> k) & 1u) {
      dx = __builtin_amdgcn_readlane(tx, int(k)) - vx;
    } else {
      const double t = div_by_res(lane_f64(pose_x, k) - g.px, G.res, G.inv_res);
      dx = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
    }
    if ((sure_y >> k) & 1u) {
      dy = __builtin_amdgcn_readlane(ty, int(k)) - vy;
    } else {
      const double t = div_by_res(lane_f64(pose_y, k) - g.py, G.res, G.inv_res);
      dy = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
    }
    c.sr = g.sr - dx; c.sc = g.sc - dy;
    wrap_index(c.sr, G.rows);
    wrap_index(c.sc, G.cols);
    c.px = g.px + double(dx) * G.res;
    c.py = g.py + double(dy) * G.res;
    c.shr = -dx; c.shc = -dy;
    if (out && lane == 0u) { out->PE[k] = g; out->PC[k] = c; }
    e_last = g;
    if ((passmask >> k) & 1u) {  // GridMap::move: the map is where the candidate says
      g.px = c.px; g.py = c.py; g.sr = c.sr; g.sc = c.sc;
      vx += dx; vy += dy;
    }
  }
  c_last = c;
}
// The walker block (first wavefront): chain of the scouted batch Cn into Cn.ms->PE / PC.
__device__ __forceinline__ void mwalk_body(const MBin& B, const MCrop& Cn, const MCommon& K, const GeomConst& G,
                                           const DevState* __restrict__ st) {
  if (threadIdx.x >= 64u || !K.do_move || !K.walk || K.dbg == 2) return;
  const unsigned lane = threadIdx.x & 63u;
  DevGeom g;
  if (B.count) {  // behind the batch this launch bins: its own chain as pre-walked if its scouts agreed, else walked here
    const unsigned act = K.gate_on_filter ? uni(B.ms->flags[0]) : 0xFFFFFFFFu, pre = uni(B.ms->flags[1]);
    const bool pre_valid = uni(B.ms->flags[2]) != 0u;
    const unsigned last = B.count - 1u;
    DevGeom e_l;
    DevCand c_l;
    if (pre_valid && ((pre ^ act) & lowbits(last)) == 0u) {
      e_l = B.ms->PE[last];
      c_l = B.ms->PC[last];
    } else {
      mwalk_batch(mbatch_start(B.prev, B.prev_count, B.scan_no0, K, st), B.count, B.robot_x, B.robot_y, act, K, G, nullptr,
                  e_l, c_l);
    }
    g = e_l;
    if ((act >> last) & 1u) { g.px = c_l.px; g.py = c_l.py; g.sr = c_l.sr; g.sc = c_l.sc; }
  } else {
    g = mbatch_start(Cn.prev, Cn.prev_count, Cn.scan_no0, K, st);
  }
  DevGeom e_l;
  DevCand c_l;
  mwalk_batch(g, Cn.count, Cn.robot_x, Cn.robot_y, 0xFFFFFFFFu, K, G, Cn.ms, e_l, c_l);
  if (lane == 0u) { Cn.ms->flags[1] = lowbits(Cn.count); Cn.ms->flags[2] = 1u; }
}

// ---------------------------------------------------------------------------------------------
// bin half: block `bid` of the batch's bin grid.  CH: bit 0 intensity, bit 1 colour (compile-time, as k_bin's).
template <int CH, bool RAY>
__device__ __forceinline__ void mbin_body(const MBin& B, const MCommon& K, const GeomConst& G,
                                          DevState* __restrict__ st, const unsigned ncell, const unsigned k,
                                          const unsigned lb, MBinLds& S) {
  // (scan k of the batch and the block's number inside the scan come straight from the grid: nothing to look up
  // before the point loads can leave)
  constexpr bool has_int = (CH & 1) != 0, has_col = (CH & 2) != 0;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  MState* const ms = B.ms;
  MView V;
  V.Tbs = K.Tbs; V.Twb12 = B.t[k].Twb; V.R = B.t[k].R; V.sp = K.sp;
  V.min_sq = K.min_sq; V.max_sq = K.max_sq; V.z_min = K.z_min; V.z_max = K.z_max;
  V.sensor_type = K.sensor_type; V.integrate_mode = K.integrate_mode;
  const float* __restrict__ const px = B.px[k];
  const float* __restrict__ const py = B.py[k];
  const float* __restrict__ const pz = B.pz[k];

  // (the poses and the geometry the chain starts from leave before the point loads: loads return in order, and the
  // walk below waits for these; the previous batch's candidate is fetched whether or not it will apply — one round
  // trip instead of flag -> candidate)
  double pose_x = 0.0, pose_y = 0.0;
  DevGeom g_start, g_pre;
  DevCand c_prev, c_pre;
  unsigned f_prev = 0u, w_pre = 0u, w_valid = 0u;
  g_start.px = g_start.py = 0.0; g_start.sr = g_start.sc = 0; g_start.pad0 = g_start.pad1 = 0;
  c_prev.px = c_prev.py = 0.0; c_prev.sr = c_prev.sc = c_prev.shr = c_prev.shc = 0;
  g_pre = g_start;
  c_pre = c_prev;
  if (threadIdx.x < 64u) {
    // (the chain as the walker block of the previous launch left it: used if the scouts confirmed what it assumed)
    w_pre = ms->flags[1];
    w_valid = ms->flags[2];
    g_pre = ms->PE[k];
    c_pre = ms->PC[k];
    if (lane <= k) { pose_x = B.robot_x[lane]; pose_y = B.robot_y[lane]; }
    if (B.prev) {
      const unsigned pk = B.prev_count - 1u;
      g_start = B.prev->E[pk];
      c_prev = B.prev->C[pk];
      f_prev = B.prev->flags[0];
    } else {
      g_start = st->geom[B.scan_no0 & 3u];
    }
  }
  const unsigned n = B.n[k];
  const unsigned i0 = lb * kMBlock + threadIdx.x;  // the thread's points: i0, i0 + 256 (two coalesced sweeps)
  float xs[kMPts], ys[kMPts], zs[kMPts], vs[kMPts];
#pragma unroll
  for (int h = 0; h < kMPts; ++h) {  // the loads go out first
    const unsigned i = i0 + unsigned(h) * 256u;
    xs[h] = ys[h] = zs[h] = vs[h] = 0.f;
    if (i < n) {
      xs[h] = px[i];
      ys[h] = py[i];
      zs[h] = pz[i];
      if (has_int) vs[h] = B.pint[k][i];
    }
  }
#if FDM_MB_PHASES == 2  // (variant: kernel arguments read + loads issued | loads back | chain walked)
  FDM_PHASE(0);
#endif
#pragma unroll
  for (int h = 0; h < kMPts; ++h) {
    const unsigned t = threadIdx.x + unsigned(h) * 256u;
    S.t_key[t] = kEmptyKey;
    S.t_cell[t] = kEmptyCell;
    S.t_zmx[t] = 0u; S.t_imx[t] = 0u; S.t_fst[t] = kNoIdx; S.t_lst[t] = 0u;
  }
#if FDM_MB_PHASES == 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FDM_PHASE(1);
#endif
  // The chain of moves up to this scan, given which of the earlier scans moved the map — walked by the whole first
  // wavefront (lane j holds scan j's pose), because a serial instruction costs ~10 ns on a full chip and a block of
  // the batch's last scan used to spend 6.6 us here.  move() snaps the position to the grid p0 + integer * res, so in
  // exact arithmetic the TOTAL shift after scan j's move is round((pose_j - p0) / res) whatever happened before it:
  // lane j rounds its pose ONCE, with move_candidate_fast's own margins (1e-4 cell off a rounding tie; the accumulated
  // position differs from p0 + V res by ~1e-10 cell at most), and the walk only accumulates the position in scan
  // order, p <- p + double(v) * res and the per-step index wrap — the operations the reference performs.  A pose that
  // sits within the margin of a tie (a robot advancing half a cell per scan) takes the reference's subtract + divide
  // against the position accumulated so far, at its place in the walk.
  auto lane_f64 = [](double v, unsigned j) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), int(j));
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), int(j));
    return __hiloint2double(hi, lo);
  };
  auto chain = [&](unsigned passmask) {  // (every lane of the first wavefront, uniformly)
    if (K.do_move && K.walk && K.dbg != 2 && K.dbg != 3 && uni(w_valid) != 0u &&
        ((uni(w_pre) ^ passmask) & lowbits(k)) == 0u) {  // pre-walked (mwalk_body): nothing to walk
      if (threadIdx.x == 0) {
        S.s_cand = c_pre;
        if (lb == 0u) { ms->E[k] = g_pre; ms->C[k] = c_pre; }
      }
      return;
    }
    DevGeom g = g_start;
    if (B.prev) {  // what the update of the previous batch (the other half of this launch) is about to commit
      const unsigned pk = B.prev_count - 1u;
      if (K.do_move && (!K.gate_on_filter || ((f_prev >> pk) & 1u) != 0u)) {
        g.px = c_prev.px; g.py = c_prev.py; g.sr = c_prev.sr; g.sc = c_prev.sc;
      }
    }
    DevCand c;
    c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
    if (K.do_move) {
      auto nearest = [&](double a, double p0, int& t) -> bool {
        t = 0;
        if (!(fabs(p0) < 1.0e7)) return false;
        if (fabs(a) <= 1e-4) return true;
        const double ue = a + 0.5 * (a > 0 ? 1 : -1);
        t = static_cast<int>(ue);
        const double f = fabs(ue - double(t));
        return f > 1e-4 && f < 1.0 - 1e-4 && fabs(ue) < 1.0e6;
      };
      int tx = 0, ty = 0;
      const bool sx = nearest((pose_x - g.px) * G.inv_res, g.px, tx), sy = nearest((pose_y - g.py) * G.inv_res, g.py, ty);
      const unsigned sure_x = K.dbg == 3 ? 0u : uni(unsigned(__ballot(sx))), sure_y = K.dbg == 3 ? 0u : uni(unsigned(__ballot(sy)));
      int vx = 0, vy = 0;  // total shift so far, in cells
      auto shift_of = [&](unsigned j, int& dx, int& dy) {  // scan j's move against the geometry walked so far
        if ((sure_x >> j) & 1u) {
          dx = __builtin_amdgcn_readlane(tx, int(j)) - vx;
        } else {  // GridMap::move's own arithmetic (the quotient correctly rounded: div_by_res)
          const double t = div_by_res(lane_f64(pose_x, j) - g.px, G.res, G.inv_res);
          dx = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
        }
        if ((sure_y >> j) & 1u) {
          dy = __builtin_amdgcn_readlane(ty, int(j)) - vy;
        } else {
          const double t = div_by_res(lane_f64(pose_y, j) - g.py, G.res, G.inv_res);
          dy = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
        }
      };
      unsigned m = K.dbg == 2 ? 0u : ((K.gate_on_filter ? passmask : 0xFFFFFFFFu) & lowbits(k));  // the moves ahead of scan k
      bool unwrapped = false;
      while (m) {  // (a scan that returned before its move is not in the mask)
        const unsigned j = unsigned(__ffs(int(m))) - 1u;
        m &= m - 1u;
        int dx, dy;
        shift_of(j, dx, dy);
        // The start index is wrapped ONCE behind the walk.  For a shift of fewer cells than the map has,
        // wrapIndexToRange returns the representative in [0, size) of its argument modulo size — wrapping after every
        // move and reducing the sum are then the same number, and the per-move wrap was a third of the walk's serial
        // instructions (0.35 us per move on a full chip: 5 us for the blocks of a batch's sixteenth scan, which is what
        // ended the launch).  A shift of >= size cells (a jump beyond the map) takes the reference's own function at its
        // place in the walk, quirks included (it returns `size` for an argument of -2 size).
        const bool small = unsigned(dx + G.rows - 1) < unsigned(2 * G.rows - 1) && unsigned(dy + G.cols - 1) < unsigned(2 * G.cols - 1);
        if (!small) {
          if (unwrapped) {
            g.sr %= G.rows; if (g.sr < 0) g.sr += G.rows;
            g.sc %= G.cols; if (g.sc < 0) g.sc += G.cols;
            unwrapped = false;
          }
          g.sr -= dx; g.sc -= dy;
          wrap_index(g.sr, G.rows);
          wrap_index(g.sc, G.cols);
        } else {
          g.sr -= dx; g.sc -= dy;
          unwrapped = true;
        }
        g.px = g.px + double(dx) * G.res;
        g.py = g.py + double(dy) * G.res;
        vx += dx; vy += dy;
      }
      if (unwrapped) {  // (the small shifts accumulated since the last wrap)
        g.sr %= G.rows; if (g.sr < 0) g.sr += G.rows;
        g.sc %= G.cols; if (g.sc < 0) g.sc += G.cols;
      }
      int dx, dy;
      shift_of(k, dx, dy);
      c.sr = g.sr - dx; c.sc = g.sc - dy;
      wrap_index(c.sr, G.rows);
      wrap_index(c.sc, G.cols);
      c.px = g.px + double(dx) * G.res;
      c.py = g.py + double(dy) * G.res;
      c.shr = -dx; c.shc = -dy;
    }
    if (threadIdx.x == 0) {
      S.s_cand = c;
      if (lb == 0u) { ms->E[k] = g; ms->C[k] = c; }
    }
  };
  // (which of the scans ahead moved the map is known before the launch starts: the scout blocks of the previous launch —
  // or a small launch of their own ahead of a call's first batch — left it in the state word.  The walk runs in the
  // shadow of the point loads; nothing is published, nothing is polled)
  const bool gated = K.do_move && K.gate_on_filter;
  if (threadIdx.x < 64u) chain((gated && k > 0u) ? uni(ms->flags[0]) : 0xFFFFFFFFu);

#if FDM_MB_PHASES == 2
  FDM_PHASE(2);
#else
  FDM_PHASE(0);  // table initialised, chain walked (the point loads are still in flight; vmcnt returns in order)
#endif
  bool pass[kMPts];
  unsigned npw = 0u;
#pragma unroll
  for (int h = 0; h < kMPts; ++h) {
    const unsigned i = i0 + unsigned(h) * 256u;
    const float sx = xs[h], sy = ys[h], sz = zs[h];
    float w = 1.0f;
    pass[h] = false;
    if (i < n) pass[h] = mcrops(V, xs[h], ys[h], zs[h], w);
    transform4_affine(V.Twb12, xs[h], ys[h], zs[h], w);
    if (RAY && B.cap && i < n) {  // the raycasting stage's input: the preprocessed cloud by point index (as k_bin's capture)
      const size_t at = size_t(k) * B.obs_stride + i, plane = size_t(kMaxBatch) * B.obs_stride;
      B.cap[at] = pass[h] ? xs[h] : __uint_as_float(0x7FC00000u);
      B.cap[plane + at] = ys[h];
      B.cap[2u * plane + at] = zs[h];
    }
    S.s_pt[threadIdx.x + unsigned(h) * 256u] = make_float4(sx, sy, sz, zs[h]);
    npw += unsigned(__popcll(__ballot(pass[h])));
  }
  if (lane == 0u) S.s_pass[wave] = npw;
#if FDM_MB_PHASES != 2
  FDM_PHASE(1);  // points arrived, crops + both transforms done
#endif
  __syncthreads();  // table, s_pt, s_pass, s_cand
  const unsigned np = S.s_pass[0] + S.s_pass[1] + S.s_pass[2] + S.s_pass[3];
  const DevCand cand = S.s_cand;

  unsigned long long* const S_key = B.key + size_t(k) * ncell;
  uint32_t* const S_aux = reinterpret_cast<uint32_t*>(B.aux + size_t(k) * ncell);
  uint32_t* const S_zs = reinterpret_cast<uint32_t*>(B.zs + size_t(k) * ncell);
  float2* const S_obs = B.obs + size_t(k) * B.obs_stride;
  uint32_t* const S_cobs = B.cobs + size_t(k) * B.obs_stride;

  // one cell's reduction goes to the scan's scratch; the merging thread also leaves what the update needs of the
  // block-local winner / last point at the POINT's index
  auto merge = [&](uint32_t c, unsigned long long key, uint32_t zmx, uint32_t imx, uint32_t fst, uint32_t lst) {
    if (K.dbg == 1) return;
    atomicMin(&S_key[c], key);
    uint32_t* a = S_aux + size_t(c) * 4;
    if (zmx) atomicMax(a + 0, zmx);
    if (has_int) {
      if (imx) atomicMax(a + 1, imx);
      atomicMin(a + 2, fst);
    }
    if (has_col) {
      atomicMax(a + 3, lst);
      S_cobs[lst] = B.prgb[k][lst];
    }
    const uint32_t idx = uint32_t(key);
    if (idx != kNoIdx) {
      const float4 p = S.s_pt[idx - lb * kMBlock];
      float var = 0.0f;  // CellObservation default (elevation_mapping.hpp:26-34)
      if (K.has_var) var = B.pvar[k][idx];
      else if (K.integrate_mode) var = sigma_z2(V, p.x, p.y, p.z);
      S_obs[idx] = make_float2(p.w, var);
    }
  };

  // getIndex on a fixed-point estimate, the reference's fp64 arithmetic only for the lanes within 2^-shift cell of
  // a cell edge (axis_fast / axis_exact, as k_tbin); every point then folds straight into the block's LDS cell
  // table — LDS atomics are cheap, the cross-lane run merge of k_bin costs more instructions than it saves here
  unsigned niw = 0u;
  bool any_glob = false;
  const bool any_start = cand.sr != 0 || cand.sc != 0;
  const double off_r = (G.half_x + cand.px) * G.inv_res_k, off_c = (G.half_y + cand.py) * G.inv_res_k;
#pragma unroll
  for (int h = 0; h < kMPts; ++h) {
    const unsigned i = i0 + unsigned(h) * 256u;
    const float x = xs[h], y = ys[h], z = zs[h], vint = vs[h];
    bool sure_r, sure_c;
    int kr = axis_fast(double(x), off_r, G.inv_res_k, G.idx_shift, G.rows, sure_r);
    int kc = axis_fast(double(y), off_c, G.inv_res_k, G.idx_shift, G.cols, sure_c);
    bool in = pass[h];
    if (__ballot(pass[h] && !(sure_r && sure_c))) {  // wave-uniform; well under a percent of the wavefronts
      if (pass[h] && !sure_r) in = axis_exact(double(x), cand.px, G.half_x, G.len_x, G.res, kr);
      if (in && !sure_c) in = axis_exact(double(y), cand.py, G.half_y, G.len_y, G.res, kc);
    }
    const bool okr = axis_wrap(kr, cand.sr, any_start, G.rows);  // (both axes are evaluated, as getIndex does)
    const bool okc = axis_wrap(kc, cand.sc, any_start, G.cols);
    const bool in_map = in && okr && okc;
    const int lr = kr - G.o_r0, lc = kc - G.o_c0;
    const bool inside = in_map && unsigned(lr) < unsigned(G.o_rows) && unsigned(lc) < unsigned(G.o_cols);
    const int cell = (kc - G.s_c0) * G.s_rows + (kr - G.s_r0);
    any_glob = any_glob || in_map;
    if (inside) {
      if (z == 0.0f) atomicMin(S_zs + size_t(cell) * 2, (i << 1) | (__float_as_uint(z) >> 31));
      if (has_int && vint == 0.0f) atomicMin(S_zs + size_t(cell) * 2 + 1, (i << 1) | (__float_as_uint(vint) >> 31));
    }
    // what this point contributes to its cell
    unsigned long long key = kEmptyKey;
    uint32_t zmx = 0u, imx = 0u, fst = kNoIdx, lst = 0u;
    if (inside) {
      key = make_key(z, i);
      zmx = make_zmax(z);
      if (has_int) {
        const bool vnan = isnan(vint);
        imx = vnan ? 0u : ord(vint);
        fst = (i << 1) | (vnan ? 1u : 0u);
      }
      lst = i;
    }
    // Neighbouring lanes are neighbouring points: on an image-ordered cloud (RGB-D rows) they fall into the same cell
    // in runs of ten or twenty, and every LDS atomic of such a wavefront serialises on a handful of table slots
    // (SQ_LDS_ADDR_CONFLICT, profiles/r04/pmc_lds_mbatch_halves.txt).  A wavefront whose runs are long merges them in
    // registers first (segmented scan over runs of equal cell, as k_bin does) and only the run TAILS touch the table;
    // a firing-order LiDAR scan (neighbouring lanes = different beams: every lane its own run) skips the merge —
    // decided per wavefront from two ballots.  min / max are idempotent: either way the table ends up the same.
    bool commit = inside;
    {
      const int prev_cell = __shfl_up(inside ? cell : -1, 1);
      const bool head = inside && (lane == 0u || prev_cell != cell);
      const unsigned n_in = unsigned(__popcll(__ballot(inside))), n_runs = unsigned(__popcll(__ballot(head)));
      if (n_runs * 4u <= n_in) {  // (wave-uniform) average run >= 4 points
        const int mcell = inside ? cell : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int ocell = __shfl_up(mcell, d);
          const unsigned long long okey = __shfl_up(key, d);
          const uint32_t ozmx = __shfl_up(zmx, d);
          const uint32_t oimx = has_int ? __shfl_up(imx, d) : 0u;
          const uint32_t ofst = has_int ? __shfl_up(fst, d) : kNoIdx;
          if (int(lane) >= d && ocell == mcell && inside) {
            key = okey < key ? okey : key;
            zmx = ozmx > zmx ? ozmx : zmx;
            imx = oimx > imx ? oimx : imx;
            fst = ofst < fst ? ofst : fst;
          }
        }
        // (lst: the tail of a run is its highest point index already)
        const int next_cell = __shfl_down(mcell, 1);
        commit = inside && (lane == 63u || next_cell != cell);
      }
    }
    if (commit) {
#ifdef FDM_X_MHASH  // (measurement build, profiles/r06/batch_max.txt: a multiplicative hash instead of cell mod 512)
      uint32_t hh = (uint32_t(cell) * 2654435761u) >> 23;
#else
      uint32_t hh = uint32_t(cell) & (kMBlock - 1u);
#endif
      while (true) {
        const uint32_t prev = atomicCAS(&S.t_cell[hh], kEmptyCell, uint32_t(cell));
        if (prev == kEmptyCell || prev == uint32_t(cell)) break;
        hh = (hh + 1) & (kMBlock - 1u);
      }
      atomicMin(&S.t_key[hh], key);
      atomicMax(&S.t_zmx[hh], zmx);  // (max with 0: no-op)
      if (has_int) {
        atomicMax(&S.t_imx[hh], imx);
        atomicMin(&S.t_fst[hh], fst);
      }
      if (has_col) atomicMax(&S.t_lst[hh], lst);
    }
    niw += unsigned(__popcll(__ballot(inside)));
  }
  // (the ballot BEFORE the lane test: inside `if (lane == 0)` it would see lane 0 alone — a small scan whose first
  // point lies outside the map then never reported "observed", and the obstacle layer kept the previous scan's values)
  const bool wave_glob = __ballot(any_glob) != 0ull;
  if (lane == 0u) {
    S.s_in[wave] = niw;
    if (wave_glob) ms->inside[k] = 1u;
  }
  // ---- flush: the occupied slots are compacted first, so that the sigma_z^2 evaluation and the memory-side
  // atomics of a block's ~100 cells keep two wavefronts busy instead of four, twice ----
#if FDM_MB_PHASES != 2
  FDM_PHASE(2);  // index + LDS fold done
#endif
  __syncthreads();
  uint32_t tc[kMPts];
  unsigned before[kMPts];
#pragma unroll
  for (int h = 0; h < kMPts; ++h) {
    tc[h] = S.t_cell[threadIdx.x + unsigned(h) * 256u];
    const unsigned long long mo = __ballot(tc[h] != kEmptyCell);
    before[h] = unsigned(__popcll(mo & ((1ull << lane) - 1ull)));
    if (lane == 0u) S.s_occ[h * 4 + int(wave)] = unsigned(__popcll(mo));
  }
  __syncthreads();
  unsigned n_occ = 0u;
#pragma unroll
  for (int h = 0; h < kMPts; ++h) {
    unsigned base = 0u;
#pragma unroll
    for (int q = 0; q < 4 * kMPts; ++q) {
      const unsigned c = S.s_occ[q];
      if (q < h * 4 + int(wave)) base += c;
      if (h == 0) n_occ += c;
    }
    if (tc[h] != kEmptyCell) S.s_list[base + before[h]] = uint16_t(threadIdx.x + unsigned(h) * 256u);
  }
  __syncthreads();
  for (unsigned j = threadIdx.x; j < n_occ; j += 256u) {
    const unsigned t = S.s_list[j];
    merge(S.t_cell[t], S.t_key[t], S.t_zmx[t], S.t_imx[t], S.t_fst[t], S.t_lst[t]);
  }
  if (threadIdx.x == 0) {
    const unsigned ni = S.s_in[0] + S.s_in[1] + S.s_in[2] + S.s_in[3];
    B.bin_part[B.first_block[k] + lb] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
}

// ---------------------------------------------------------------------------------------------
// update half: kUpdCells cells of the map (memory order), every scan of the batch.  Four threads per cell — thread
// 4 c + q looks after cell c in scans 4 q .. 4 q + 3 — so that a block holds at most 1024 events (two rounds of
// the exchange) and the densest corner of the map (every cell touched by every scan) is a chain as short as any
// other; the cell's own thread (q == 0) applies the events.
template <typename POLICY, int CH, bool RAY>
__device__ __forceinline__ void mupdate_body(const MUpd& U, const GeomConst& G, DevState* __restrict__ st,
                                             const typename POLICY::Layers& L,
                                             float* const* __restrict__ all_layers, int n_layers,
                                             const unsigned ncell, const unsigned bid,
                                             MUpdLds<(CH & 2) != 0, RAY>& S) {
  constexpr bool has_int = (CH & 1) != 0, has_col = (CH & 2) != 0;
  const float nanv = __uint_as_float(0x7FC00000u);
  const unsigned lt = threadIdx.x, lane = lt & 63u, wave = lt >> 6;
  const unsigned count = U.count;
  const unsigned tpc = upd_threads_per_cell(count), cpb = 256u / tpc;   // threads per cell, cells per block (block-uniform)
  const unsigned cq = lt & (tpc - 1u), cl = tpc == 8u ? lt >> 3 : lt >> 2;
  const unsigned o = bid * cpb + cl;
  const bool valid = o < ncell;
  const bool owner = valid && cq == 0u;  // the thread that applies the cell's events
  MState* const ms = U.ms;

  // ---- round trip 1: the thread's four keys, the per-scan geometry (lane j holds scan j's) ----
  unsigned long long kk[kScansPerThread];
#pragma unroll
  for (int j = 0; j < kScansPerThread; ++j) {
    const unsigned k = cq * unsigned(kScansPerThread) + unsigned(j);
    kk[j] = kEmptyKey;
    if (k < count && valid) kk[j] = U.key[size_t(k) * ncell + o];
  }
  int v_sr = 0, v_sc = 0, v_shr = 0, v_shc = 0;
  unsigned v_in = 0u;
  bool v_run = false;
  if (lane < count) {
    v_sr = ms->E[lane].sr; v_sc = ms->E[lane].sc;
    v_shr = ms->C[lane].shr; v_shc = ms->C[lane].shc;
    v_in = ms->inside[lane];
    if (RAY && U.ray.stamp) v_run = U.ray.rs->any[lane] == U.ray.stamp && U.ray.rs->origin_in[lane] != 0u;
  }
  // raycasting: the thread's four (evidence count, min ray height) pairs join round trip 1 (the images of a scan whose
  // stage did not run are clean: no events)
  uint32_t rc_[kScansPerThread], rh_[kScansPerThread];
  if (RAY) {
#pragma unroll
    for (int j = 0; j < kScansPerThread; ++j) {
      const unsigned k = cq * unsigned(kScansPerThread) + unsigned(j);
      rc_[j] = 0u; rh_[j] = kMRayEmpty;
      if (U.ray.stamp && k < count && valid) {
        rc_[j] = U.ray.rc_cnt[size_t(k) * ncell + o];
        rh_[j] = U.ray.rc_min[size_t(k) * ncell + o];
      }
    }
  }
  // ... and so does the cell's stored state: with raycasting nearly every cell has an event, and asking for the record
  // only once the images are back would be one more dependent round trip
  typename POLICY::State rec_early;
  POLICY::set_nan(rec_early);
  float sint_early = nanv, lo_early = nanv;
  if (RAY && owner) {
    POLICY::load(L, o, rec_early);
    if (has_int) sint_early = L.intensity[size_t(o) * L.istride];
    if (U.ray.stamp) lo_early = U.ray.logodds[o];
  }
  const unsigned runmask = RAY ? uni(unsigned(__ballot(v_run))) : 0u;  // scans whose raycasting stage runs (raycasting.cpp:207-220)
  const unsigned passbits = uni(ms->flags[0]);
  const bool v_applied = lane < count && U.do_move && (!U.gate_on_filter || ((passbits >> lane) & 1u) != 0u);
  const unsigned umask = uni(unsigned(__ballot(lane < count && v_in != 0u)));                     // scans that observed a cell
  unsigned stripmask = uni(unsigned(__ballot(v_applied && (v_shr != 0 || v_shc != 0))));         // scans whose move vacated cells

  if (bid == 0 && lt == 0) {  // commit the geometry ring behind the batch (make_ctx does this per scan)
    const unsigned last = count - 1u;
    const unsigned slot_next = (U.scan_no0 + count) & 3u;
    DevGeom g = ms->E[last];
    const DevCand cl2 = ms->C[last];
    const bool applied_last = U.do_move && (!U.gate_on_filter || ((passbits >> last) & 1u) != 0u);
    if (applied_last) { g.px = cl2.px; g.py = cl2.py; g.sr = cl2.sr; g.sc = cl2.sc; }
    const unsigned ob_prev = st->obst[U.scan_no0 & 3u].scan;
    st->geom[slot_next] = g;
    st->cand[(U.scan_no0 + last) & 3u] = cl2;   // (the synchronous statistics report the last scan's shift)
    // the last scan of the batch that observed a cell (umask: one bit per scan)
    st->obst[slot_next].scan = umask ? U.scan_no0 + (31u - unsigned(__clz(int(umask)))) : ob_prev;
#pragma unroll
    for (int q = 0; q < 4; ++q) { st->flags[q].any_pass = 0u; st->flags[q].any_inside = 0u; st->flags[q].ray_any = 0u; }
    if (umask) {
      const unsigned first_upd = U.scan_no0 + unsigned(__ffs(int(umask))) - 1u;
      if (has_int && st->vis_int == 0u) st->vis_int = 3u * first_upd + 2u;
      if (has_col && st->vis_col == 0u) st->vis_col = 3u * first_upd + 2u;
    }
    if (RAY && runmask && st->vis_ray == 0u)  // the three layers become visible with the first frame that runs (raycasting.cpp:223-226)
      st->vis_ray = 3u * (U.scan_no0 + unsigned(__ffs(int(runmask))) - 1u) + 3u;
  }
  if (bid == 0 && lt >= 64u && lt < 64u + unsigned(kMaxBatch)) {  // re-arm the state of the batch after next
    U.rearm->inside[lt - 64u] = 0u;
    if (lt == 64u) { U.rearm->flags[0] = 0u; U.rearm->flags[1] = 0u; U.rearm->flags[2] = 0u; }
  }

  unsigned nib = 0u;
#pragma unroll
  for (int j = 0; j < kScansPerThread; ++j) nib |= (kk[j] != kEmptyKey) ? (1u << j) : 0u;
  FDM_PHASE(0);  // round trip 1 (keys, geometry) back
  unsigned tmask = nib << (unsigned(kScansPerThread) * cq);  // the cell's scans, all four threads of the cell
  tmask |= unsigned(__shfl_xor(int(tmask), 1));
  tmask |= unsigned(__shfl_xor(int(tmask), 2));
  if (tpc == 8u) tmask |= unsigned(__shfl_xor(int(tmask), 4));
  unsigned rmask = 0u;  // the cell's ray events: scans whose stage left evidence or a ray height in it
  if (RAY) {
    unsigned rb = 0u;
#pragma unroll
    for (int j = 0; j < kScansPerThread; ++j) {
      const unsigned k = cq * unsigned(kScansPerThread) + unsigned(j);
      if (rc_[j] != 0u || rh_[j] != kMRayEmpty) {
        rb |= 1u << j;
        if (rc_[j] != 0u) U.ray.rc_cnt[size_t(k) * ncell + o] = 0u;  // the images are clean again for the next batch
        if (rh_[j] != kMRayEmpty) U.ray.rc_min[size_t(k) * ncell + o] = kMRayEmpty;
      }
      S.rcnt[k * kUpdCells + cl] = rc_[j];
      S.rmin[k * kUpdCells + cl] = rh_[j];
    }
    rmask = rb << (unsigned(kScansPerThread) * cq);
    rmask |= unsigned(__shfl_xor(int(rmask), 1));
    rmask |= unsigned(__shfl_xor(int(rmask), 2));
    if (tpc == 8u) rmask |= unsigned(__shfl_xor(int(rmask), 4));
  }
  // which scans' moves vacated THIS cell
  unsigned smask = 0u;
  if (stripmask) {
    const int r = int(o % unsigned(G.s_rows)) + G.s_r0;
    const int col = int(o / unsigned(G.s_rows)) + G.s_c0;
    while (stripmask) {
      const unsigned k = unsigned(__ffs(int(stripmask))) - 1u;
      stripmask &= stripmask - 1u;
      const int sr = __builtin_amdgcn_readlane(v_sr, int(k)), sc = __builtin_amdgcn_readlane(v_sc, int(k));
      const int shr = __builtin_amdgcn_readlane(v_shr, int(k)), shc = __builtin_amdgcn_readlane(v_shc, int(k));
      if (in_cleared_strip(r, sr, shr, G.rows) || in_cleared_strip(col, sc, shc, G.cols)) smask |= 1u << k;
    }
    if (!valid) smask = 0u;
  }

  // ---- the block's (cell, scan) events, numbered cell-major / scan-minor (= thread order) ----
  const unsigned mine = unsigned(__popc(nib));
  unsigned inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned v = __shfl_up(inc, d);
    if (int(lane) >= d) inc += v;
  }
  if (lane == 63u) S.s_w[wave] = inc;
  __syncthreads();
  unsigned base = inc - mine;
  for (unsigned w2 = 0; w2 < wave; ++w2) base += S.s_w[w2];
  const unsigned total = uni(S.s_w[0] + S.s_w[1] + S.s_w[2] + S.s_w[3]);

  typename POLICY::State stt;
  POLICY::set_nan(stt);
  float sint = nanv;
  unsigned lastp1 = 0u, m = owner ? tmask : 0u, e_next = base;  // (owner: the events still to be applied, the next one's number)
  unsigned rm = (RAY && owner) ? rmask : 0u;                        // ... and its ray events
  unsigned lastu1 = 0u;                                             // (RAY: behind the cell's last OBSERVATION; lastp1 counts ray events too)
  bool evt = false, st_dirty = false, strip_any = false, obst_dirty = false;
  float obst = nanv;
  uint32_t colv = 0x7FC00000u;
  // raycasting state of the cell: visibility log-odds, the elevation the last event left, what the raycasting layer ends
  // up with (the LAST running frame's minimum: map.clear(raycasting) + that frame's rays)
  float lo = nanv, elev = nanv, ray_val = nanv;
  bool lo_dirty = false, ghost_one = false;
  const int k_last_run = runmask ? 31 - __clz(int(runmask)) : -1;
  // NaN in every layer of the cell: a strip GridMap::move vacated, or ElevationMap::clearAt (elevation_map.hpp:131-135)
  auto wipe = [&]() {
    POLICY::set_nan(stt);
    sint = nanv;
    colv = 0x7FC00000u;
    strip_any = true;
    st_dirty = false;
    if (RAY) { obst = nanv; obst_dirty = true; lo = nanv; lo_dirty = false; ghost_one = false; elev = nanv; ray_val = nanv; }
  };
  // resolveGhostCells (raycasting.cpp:175-202) for this cell and scan k: the scan's evidence, then its ghost decision
  auto resolve = [&](const unsigned k) {
    const uint32_t cnt = S.rcnt[k * kUpdCells + cl], hmin = S.rmin[k * kUpdCells + cl];
    if (cnt) {  // observed evidence, once per ray-scan point in the cell (raycasting.cpp:156-163)
      if (isnan(lo)) lo = 0.0f;
      for (uint32_t q = 0; q < cnt; ++q) {
        const float a = lo + U.ray.l_obs;
        const float nx = (U.ray.l_max < a) ? U.ray.l_max : a;  // std::min(a, l_max)
        if (nx == lo) break;  // fixed point reached: the remaining folds change nothing
        lo = nx;
      }
      lo_dirty = true;
    }
    float ray = nanv;
    if (hmin != kMRayEmpty) {
      ray = unord(hmin);
      if (!isnan(elev) && elev > ray + U.ray.conflict_thr) {
        if (isnan(lo)) lo = 0.0f;
        lo -= U.ray.l_ghost;
        lo_dirty = true;
        if (lo < U.ray.clear_thr) {  // clearAt: NaN in every layer, then the marker (no min ray height for the frame)
          wipe();
          ghost_one = true;
          ray = nanv;
        }
      }
    }
    if (int(k) == k_last_run) ray_val = ray;
  };
  if (RAY && owner) {  // (rays cross most cells of the map: the cell's state was fetched with round trip 1, see rec_early)
    if (tmask | rmask) {
      stt = rec_early;
      if (has_int) sint = sint_early;
      elev = POLICY::elevation(stt);
      if (rmask) lo = lo_early;
    }
  }
#pragma unroll 1
  for (unsigned r0 = 0; r0 < total; r0 += unsigned(kEvCap)) {  // one round for all but the densest corners
    const unsigned r1 = min(r0 + unsigned(kEvCap), total);
    {  // every thread lists its (at most four) events of this round
      unsigned e = base;
#pragma unroll
      for (int j = 0; j < kScansPerThread; ++j) {
        if ((nib >> j) & 1u) {
          if (e >= r0 && e < r1) {
            MEvent ev;
            ev.idx = uint32_t(kk[j]);
            ev.cell = uint16_t(cl);
            ev.k = uint16_t(cq * unsigned(kScansPerThread) + unsigned(j));
            S.ev[e - r0] = ev;
          }
          ++e;
        }
      }
    }
    __syncthreads();
    // round trip 2: every event of the round, spread evenly over the block (kEvCap / 256 per thread), and the
    // cells' stored state
    constexpr int kPer = kEvCap / 256;
    MEvent ev_[kPer];
    float2 ob_[kPer];
    uint4 ax_[kPer];
    uint2 zs_[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const unsigned j = lt + unsigned(q) * 256u;
      ob_[q] = make_float2(kFltMax, 0.0f);  // (no finite z: FLT_MAX, variance 0 — elevation_mapping.hpp:26-34)
      ax_[q] = make_uint4(0u, 0u, kNoIdx, 0u);
      zs_[q] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      ev_[q].idx = kNoIdx; ev_[q].cell = 0; ev_[q].k = 0;
      if (r0 + j < r1) {
        ev_[q] = S.ev[j];
        const size_t oc = size_t(ev_[q].k) * ncell + (bid * cpb + ev_[q].cell);
        if (ev_[q].idx != kNoIdx) ob_[q] = U.obs[size_t(ev_[q].k) * U.obs_stride + ev_[q].idx];
        ax_[q] = U.aux[oc];
        zs_[q] = U.zs[oc];
      }
    }
    if (!RAY && r0 == 0u && owner && tmask) {  // (the record joins the same round trip)
      POLICY::load(L, o, stt);
      if (has_int) sint = L.intensity[size_t(o) * L.istride];
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      const unsigned j = lt + unsigned(q) * 256u;
      if (r0 + j >= r1) continue;
      if (has_col) S.rgb[j] = U.cobs[size_t(ev_[q].k) * U.obs_stride + ax_[q].w];  // (one more dependent gather)
      const uint32_t zm = ax_[q].x, imx = ax_[q].y, fst = ax_[q].z;
      MObs ob;
      ob.min_z = ob_[q].x;
      ob.var = ob_[q].y;
      // (a zero maximum takes the sign of the first zero-valued point, see Scratch::zs)
      ob.max_z = zm ? ((zm == 0x80000000u && (zs_[q].x & 1u)) ? -0.0f : unord(zm)) : -kFltMax;
      ob.iobs = 0.0f;
      if (has_int) ob.iobs = (fst & 1u) ? nanv : ((imx == 0x80000000u && (zs_[q].y & 1u)) ? -0.0f : unord(imx));
      S.ob[j] = ob;
      FDM_PHASE(1);  // round trip 2 (observations, record) back
      const size_t oc = size_t(ev_[q].k) * ncell + (bid * cpb + ev_[q].cell);
      U.key[oc] = kEmptyKey;  // the scratch is clean again for the batch after next
      U.aux[oc] = make_uint4(0u, 0u, kNoIdx, 0u);
      if ((zs_[q].x & zs_[q].y) != 0xFFFFFFFFu) U.zs[oc] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    __syncthreads();
    // the cells' own threads apply their events of this round, in scan order
    if constexpr (!RAY) {
      while (m && e_next < r1) {
        const unsigned k = unsigned(__ffs(int(m))) - 1u;
        m &= m - 1u;
        const MObs ob = S.ob[e_next - r0];
        const uint32_t rgb = has_col ? S.rgb[e_next - r0] : 0u;
        ++e_next;
        const unsigned upto = lowbits(k + 1u), from = lowbits(lastp1);  // scans lastp1 .. k
        if (smask & upto & ~from) {  // vacated since the last event: NaN in every layer (GridMap::move)
          POLICY::set_nan(stt);
          sint = nanv;
          colv = 0x7FC00000u;
          strip_any = true;
        }
        POLICY::step(L, stt, ob.min_z, ob.var, ob.max_z);
        obst = (ob.max_z > ob.min_z) ? ob.max_z : nanv;
        obst_dirty = true;
        if (has_int && (isnan(sint) || ob.iobs > sint)) sint = ob.iobs;
        if (has_col) colv = rgb & 0x00FFFFFFu;
        evt = true;
        st_dirty = true;
        lastp1 = k + 1u;
      }
    } else {
      while (m | rm) {
        const unsigned k = unsigned(__ffs(int(m | rm))) - 1u, bit = 1u << k;
        if ((m & bit) && e_next >= r1) break;  // (its observation arrives with the next round)
        const unsigned upto = lowbits(k + 1u), from = lowbits(lastp1);  // scans lastp1 .. k
        if (smask & upto & ~from) wipe();  // vacated since the last event: NaN in every layer (GridMap::move)
        if (m & bit) {
          m &= ~bit;
          const MObs ob = S.ob[e_next - r0];
          const uint32_t rgb = has_col ? S.rgb[e_next - r0] : 0u;
          ++e_next;
          POLICY::step(L, stt, ob.min_z, ob.var, ob.max_z);
          obst = (ob.max_z > ob.min_z) ? ob.max_z : nanv;
          obst_dirty = true;
          if (has_int && (isnan(sint) || ob.iobs > sint)) sint = ob.iobs;
          if (has_col) colv = rgb & 0x00FFFFFFu;
          evt = true;
          st_dirty = true;
          if (RAY) { elev = POLICY::elevation(stt); lastu1 = k + 1u; }
        }
        if (RAY && (rm & bit)) {  // scan k's raycasting stage, behind its observation
          rm &= ~bit;
          resolve(k);
        }
        lastp1 = k + 1u;
      }
    }
    __syncthreads();  // (the next round reuses the lists)
  }
  if (RAY) {  // ray events behind the block's last observation (or of a block without any)
    while (rm) {
      const unsigned k = unsigned(__ffs(int(rm))) - 1u;
      rm &= rm - 1u;
      const unsigned upto = lowbits(k + 1u), from = lowbits(lastp1);
      if (smask & upto & ~from) wipe();
      resolve(k);
      lastp1 = k + 1u;
    }
  }
  FDM_PHASE(2);  // events applied
  // the scans after the cell's last event
  if (owner) {
    const unsigned all = lowbits(count), from = lowbits(lastp1);
    const unsigned tail = all & ~from;
    if (smask & tail) {
      strip_any = true; st_dirty = false; sint = nanv; colv = 0x7FC00000u; obst = nanv; obst_dirty = true;
      if (RAY) { lo = nanv; lo_dirty = false; ghost_one = false; ray_val = nanv; }
    }
    // map_.clear(obstacle) by every scan that observed a cell (elevation_mapping.cpp:144-146), behind the cell's last observation
    const unsigned tail_u = RAY ? (all & ~lowbits(lastu1)) : tail;
    if (umask & tail_u) { obst = nanv; obst_dirty = true; }
    if (strip_any) {
      for (int l0 = 0; l0 < n_layers; l0 += 8) {
        float* p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q] = all_layers[min(l0 + q, n_layers - 1)];
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (l0 + q < n_layers) p[q][o] = nanv;
      }
      POLICY::clear_cell(L, o);
    }
    if (st_dirty) {
      POLICY::finish(stt);
      POLICY::store(L, o, stt);
    }
    if (obst_dirty) L.obstacle[o] = obst;
    if (has_int && (evt || strip_any)) L.intensity[size_t(o) * L.istride] = sint;
    if (has_col && (evt || strip_any)) reinterpret_cast<uint32_t*>(L.color)[o] = colv;
    if (RAY && U.ray.stamp) {
      if (lo_dirty) U.ray.logodds[o] = lo;
      if (ghost_one) U.ray.ghost[o] = 1.0f;
      if (runmask) U.ray.ray_min[o] = ray_val;  // every cell, every running frame: the layer is cleared first (raycasting.cpp:228)
    }
  }
  // touched-cell count of the batch's last scan (what the synchronous statistics report), per update block
  const unsigned long long mt = __ballot(owner && ((tmask >> (count - 1u)) & 1u) != 0u);
  if (lane == 0u) S.s_t[wave] = unsigned(__popcll(mt));
  __syncthreads();
  if (lt == 0 && bid * cpb < ncell) U.upd_part[bid] = S.s_t[0] + S.s_t[1] + S.s_t[2] + S.s_t[3];
}

// [ update of batch b-1 | bin of batch b | crop pass of batch b+1 ] — any of the three may be empty.
// Grid: blockIdx.x = block inside a scan (as wide as the widest scan of the two halves); blockIdx.y = first
// `upd_rows` rows of update blocks (block = row * width + x), then one row per scan of the bin half, then one row per
// scan of the crop half.  Surplus blocks leave at once.  Rows are dispatched in order: the update's chains first.
// (the P2 estimator's 17-float state does not fit the 80 registers of 6 waves per SIMD without spilling: 5 there)
template <typename POLICY> struct MBatchWaves { static constexpr int value = FDM_MB_WAVES; };
template <> struct MBatchWaves<P2RecPolicy> { static constexpr int value = FDM_MB_WAVES - 1; };
template <> struct MBatchWaves<P2Policy> { static constexpr int value = FDM_MB_WAVES - 1; };

template <typename POLICY, int CH, bool RAY>
__global__ __launch_bounds__(256, MBatchWaves<POLICY>::value) void k_mbatch(const MUpd U, const MBin B, const MCrop Cn, const MCommon K,
                                                const GeomConst G, DevState* __restrict__ st,
                                                const typename POLICY::Layers L,
                                                float* const* __restrict__ all_layers, int n_layers,
                                                unsigned ncell, unsigned upd_blocks, unsigned upd_rows) {
  __shared__ __align__(16) unsigned char lds[kMLdsBytes<(CH & 2) != 0, RAY>];
  const unsigned long long t0 = K.timeline ? wall_clock64() : 0ull;
#if FDM_MB_PHASES
  if (threadIdx.x == 0) { g_phase[0] = g_phase[1] = g_phase[2] = unsigned(t0); }
#endif
  const unsigned row = blockIdx.y, x = blockIdx.x;
  if (row < upd_rows) {
    const unsigned ub = row * gridDim.x + x;
    if (ub < upd_blocks)
      mupdate_body<POLICY, CH, RAY>(U, G, st, L, all_layers, n_layers, ncell, ub,
                                    *reinterpret_cast<MUpdLds<(CH & 2) != 0, RAY>*>(lds));
  } else if (row < upd_rows + B.count) {
    const unsigned k = row - upd_rows;
    if (x < (B.n[k] + kMBlock - 1u) / kMBlock) mbin_body<CH, RAY>(B, K, G, st, ncell, k, x, *reinterpret_cast<MBinLds*>(lds));
  } else {  // scout rows: block c of Cn.count * kMScout
    const unsigned c = (row - upd_rows - B.count) * gridDim.x + x;
    if (c < Cn.count * kMScout) mcrop_body(Cn, K, c / kMScout, c % kMScout);
    else if (c == Cn.count * kMScout && Cn.count) mwalk_body(B, Cn, K, G, st);
  }
  if (K.timeline && threadIdx.x == 0) {  // (thread 0's view of the block; scripts/timeline_batch.py)
    const unsigned b = blockIdx.y * gridDim.x + x;
    K.timeline[2u * b] = t0;
#if FDM_MB_PHASES
    K.timeline[2u * b + 1u] = phase_word(t0);
#else
    K.timeline[2u * b + 1u] = wall_clock64();
#endif
  }
}

}  // namespace fdm
