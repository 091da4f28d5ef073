// fdm_multi.hpp — a BATCH of small scans in two launches instead of one launch per scan.
//
// Why: a VLP-16 scan (28.8 K points, 113 bin blocks, 88 update blocks) is launch- and
// latency-bound: one fused launch per scan costs ~6 us of which ~3 us is the queue's own per-kernel floor.
// fdm_engine_integrate_device_batch sees its scans up front, so up to kMaxBatch of them are binned by ONE
// launch into per-scan scratch sets, and ONE update launch lets every cell thread apply the batch's
// observations in scan order.  Per cell the reference only fixes the order of the scans
// (elevation_mapping.cpp:94-125: one independent update per observed cell and scan; GridMap::move strips and
// the obstacle-layer clear happen between scans), and that order is kept exactly — the map after the batch is
// bit-identical to integrating the scans one by one.  As in the single-scan path the update of batch b is held
// back and shares its launch with the bin of batch b+1 (k_mupdate_mbin).
//
//   k_mput          the per-scan parameters (transforms, pose, array pointers) travel as kernel arguments of a
//                   one-block kernel that stores them into a device table: no copy command on the stream.
//   mbin_body       k_bin's body (one point per thread, wave-merge + per-block LDS cell table, memory-side atomics
//                   on the block's unique cells) for scan k of the batch.  Two things are new:
//                   * the geometry chain.  Scan k is binned against the map geometry AFTER the LOCAL-mode moves of
//                     scans 0..k, and whether scan j moved the map depends on whether any of its points survived
//                     the crops (fastdem.cpp:138).  Every wavefront-0 of a block publishes "my block is past
//                     the crops, it had / had no surviving point" with one atomic on done[k] and then waits for
//                     the blocks of scans < k (a block only ever waits for blocks with a LOWER block index, which
//                     the dispatcher started earlier and which never wait before publishing: no deadlock; the
//                     spin is bounded and raises MState::err instead of hanging).  Thread 0 then walks the chain
//                     of k moves (move_candidate_fast: no fp64 divide on the common path).
//                   * no second look at the scan.  The thread that merges a block-local minimum into the scratch
//                     also evaluates that point's sigma_z^2 (the block keeps its points' sensor-frame coordinates
//                     in LDS) and stores {map-frame z, sigma_z^2} at the POINT's index in an observation array;
//                     likewise the colour of a block-local last point.  The update finds the winner's entry through
//                     the index in the reduced key — one 8-byte gather, no transforms, no per-scan parameters —
//                     and the caller's arrays are dead as soon as the bin launch has run.
//   mupdate_body    one thread per cell, one block per 256-cell tile (memory order).  Round trip 1: the cell's
//                   keys of all scans, its estimator record, the per-scan geometry.  Then a per-lane event loop
//                   over the scans that touched the cell: {obs, aux, zs} loads, strips vacated since the previous
//                   event, estimator step in registers.  One record store at the end.
//
// Algorithmic bytes are per scan what they were: 12 B/point (+4 intensity, +4 colour), 72 / 124 B per
// touched cell, 4 B per map cell per scan for the obstacle clear.
#pragma once

#include "fdm_kernels.hpp"

namespace fdm {

constexpr int kMaxBatch = 16;          // scans per launch: one k_mput argument block (16 x 240 B < 4 KB)
constexpr unsigned kSpinMax = 1u << 22;  // polls before a waiting block gives up (seconds; never reached in practice)

struct MScan {  // what is particular to one scan of a batch (240 B)
  float Tbs[16], Twb[16], R[9];  // as ScanParams
  unsigned n;
  double robot_x, robot_y;
  const float *x, *y, *z, *intensity;
  const uint32_t* rgb;
  const float* var;
  unsigned scan_no, pad;
};
struct MScanBlock { MScan s[kMaxBatch]; };

// Device-resident bookkeeping of one batch (double-buffered by batch parity).
struct MState {
  DevGeom E[kMaxBatch];        // geometry before scan k (written by the scan's first bin block)
  DevCand C[kMaxBatch];        // geometry after its move + the index shift
  unsigned done[kMaxBatch];    // blocks of scan k past the crops (low 16 bits) | blocks with a surviving point << 16
  unsigned inside[kMaxBatch];  // some point of scan k landed in the map (elevation_mapping.cpp:118)
  unsigned err;                // a waiting block ran out of polls
  unsigned pad[3];
};

struct MBatch {  // kernel argument
  unsigned count;                        // scans in the batch
  unsigned scan_no0;                     // number of its first scan
  unsigned first_block[kMaxBatch + 1];   // bin blocks before scan k
  const MScan* scans;                    // device table (bin half only)
  MState* ms;
  const MState* prev;                    // bin half: the previous batch's state while its update is still held back
  unsigned prev_count;
  unsigned obs_stride;                   // points per scan slot of obs / cobs
  // per-scan scratch, slot k at + k * ncell (key, aux, zs as in Scratch)
  unsigned long long* key;
  uint4* aux;
  uint2* zs;
  float2* obs;                           // [count][obs_stride] {map-frame z, sigma_z^2} of block-local minima, by point index
  uint32_t* cobs;                        // [count][obs_stride] colour of block-local last points, by point index
  unsigned long long* bin_part;          // [bin blocks]
  uint32_t* upd_part;                    // [tiles] of the batch's LAST scan
  // what all scans of the batch share
  float min_sq, max_sq, z_min, z_max;
  float sp[4];
  int sensor_type, integrate_mode, do_move, gate_on_filter, has_var, bin_table;
  double robot_x[kMaxBatch], robot_y[kMaxBatch];  // T_world_base translation of every scan (the chain of moves)
};

// The members preprocess_point / sigma_z2 read, by name (pointers only: an array member copied in device code is
// spilled, see DevObst).
struct MView {
  const float* Tbs;
  const float* Twb;
  const float* R;
  const float* sp;
  float min_sq, max_sq, z_min, z_max;
  int sensor_type, integrate_mode;
};

// Parameter upload + re-arm of the batch's counters (the set was last read by the update two launches ago).
__global__ __launch_bounds__(256) void k_mput(const MScanBlock src, MScan* __restrict__ dst, unsigned count,
                                              MState* __restrict__ ms) {
  const uint32_t* s = reinterpret_cast<const uint32_t*>(&src);
  uint32_t* d = reinterpret_cast<uint32_t*>(dst);
  const unsigned words = count * unsigned(sizeof(MScan) / 4);
  for (unsigned i = threadIdx.x; i < words; i += 256u) d[i] = s[i];
  if (threadIdx.x < unsigned(kMaxBatch)) {
    ms->done[threadIdx.x] = 0u;
    ms->inside[threadIdx.x] = 0u;
  }
}

// ---------------------------------------------------------------------------------------------
// bin half: block `bid` of the batch's bin grid.  CH: bit 0 intensity, bit 1 colour (compile-time, as k_bin's).
template <int CH>
__device__ __forceinline__ void mbin_body(const MBatch& B, const GeomConst& G, DevState* __restrict__ st,
                                          const unsigned ncell, const unsigned bid) {
  constexpr bool has_int = (CH & 1) != 0, has_col = (CH & 2) != 0;
  __shared__ DevCand s_cand;
  __shared__ unsigned s_pass[4], s_in[4];
  __shared__ unsigned long long t_key[256];
  __shared__ uint32_t t_cell[256], t_zmx[256], t_imx[256], t_fst[256], t_lst[256];
  __shared__ float4 s_pt[256];  // sensor-frame x, y, z | map-frame z of the block's points

  // which scan of the batch (block-uniform): lane j asks "does scan j start at or before this block?"
  const unsigned k = uni(unsigned(__popcll(__ballot((threadIdx.x & 63u) < B.count &&
                                                    bid >= B.first_block[threadIdx.x & 63u])))) - 1u;
  const MScan* __restrict__ M = B.scans + k;
  const unsigned lb = bid - B.first_block[k];
  MState* const ms = B.ms;
  MView V;
  V.Tbs = M->Tbs; V.Twb = M->Twb; V.R = M->R; V.sp = B.sp;
  V.min_sq = B.min_sq; V.max_sq = B.max_sq; V.z_min = B.z_min; V.z_max = B.z_max;
  V.sensor_type = B.sensor_type; V.integrate_mode = B.integrate_mode;

  {
    t_key[threadIdx.x] = kEmptyKey;
    t_cell[threadIdx.x] = kEmptyCell;
    t_zmx[threadIdx.x] = 0u; t_imx[threadIdx.x] = 0u; t_fst[threadIdx.x] = kNoIdx; t_lst[threadIdx.x] = 0u;
  }
  const unsigned n = M->n;
  const unsigned i = lb * 256u + threadIdx.x;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float x = 0.f, y = 0.f, z = 0.f, vint = 0.f;
  if (i < n) {
    x = M->x[i];
    y = M->y[i];
    z = M->z[i];
    if (has_int) vint = M->intensity[i];
  }
  // the crops need no geometry: run them first, publish, then wait for the earlier scans
  const float sx = x, sy = y, sz = z;
  bool pass = false;
  if (i < n) pass = preprocess_point(V, x, y, z);
  s_pt[threadIdx.x] = make_float4(sx, sy, sz, z);
  {
    const unsigned long long mp = __ballot(pass);
    if (lane == 0u) s_pass[wave] = unsigned(__popcll(mp));
  }
  __syncthreads();  // table, s_pt, s_pass
  const unsigned np = s_pass[0] + s_pass[1] + s_pass[2] + s_pass[3];
  if (threadIdx.x < 64u) {
    if (lane == 0u) atomicAdd(&ms->done[k], 1u | (np ? 0x10000u : 0u));
    unsigned passmask = 0xFFFFFFFFu;
    if (B.do_move && B.gate_on_filter && k > 0u) {
      unsigned v = 0u, spins = 0u;
      const unsigned want = lane < k ? B.first_block[lane + 1u] - B.first_block[lane] : 0u;
      while (true) {
        bool ok = true;
        if (lane < k) {
          v = __hip_atomic_load(&ms->done[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (v & 0xFFFFu) == want;
        }
        if (__ballot(!ok) == 0ull) break;
        if (++spins >= kSpinMax) {
          if (lane == 0u) ms->err = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(2);  // (~128 clocks: keeps the pollers off the memory pipeline)
      }
      passmask = unsigned(__ballot(lane < k && (v >> 16) != 0u));
    }
    if (lane == 0u) {
      DevGeom g;
      if (B.prev) {  // what the held-back update of the previous batch is about to commit
        const unsigned pk = B.prev_count - 1u;
        g = B.prev->E[pk];
        if (B.do_move && (!B.gate_on_filter || (B.prev->done[pk] >> 16) != 0u)) {
          const DevCand pc = B.prev->C[pk];
          g.px = pc.px; g.py = pc.py; g.sr = pc.sr; g.sc = pc.sc;
        }
      } else {
        g = st->geom[B.scan_no0 & 3u];
      }
      if (B.do_move) {
        for (unsigned j = 0; j < k; ++j) {
          if (B.gate_on_filter && !((passmask >> j) & 1u)) continue;  // scan j returned before its move
          const DevCand cj = move_candidate_fast(g, G, B.robot_x[j], B.robot_y[j]);
          g.px = cj.px; g.py = cj.py; g.sr = cj.sr; g.sc = cj.sc;
        }
      }
      DevCand c;
      if (B.do_move) {
        c = move_candidate_fast(g, G, B.robot_x[k], B.robot_y[k]);
      } else {
        c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
      }
      s_cand = c;
      if (lb == 0u) { ms->E[k] = g; ms->C[k] = c; }
    }
  }
  __syncthreads();
  const DevCand cand = s_cand;

  int cell = -1;
  if (pass) cell = owned_cell(x, y, cand, G);
  const bool inside = cell >= 0;
  const bool glob = pass && cell != -1;

  unsigned long long* const S_key = B.key + size_t(k) * ncell;
  uint32_t* const S_aux = reinterpret_cast<uint32_t*>(B.aux + size_t(k) * ncell);
  uint32_t* const S_zs = reinterpret_cast<uint32_t*>(B.zs + size_t(k) * ncell);
  float2* const S_obs = B.obs + size_t(k) * B.obs_stride;
  uint32_t* const S_cobs = B.cobs + size_t(k) * B.obs_stride;

  // one cell's reduction goes to the scan's scratch; the merging thread also leaves what the update needs of the
  // block-local winner / last point at the POINT's index
  auto merge = [&](uint32_t c, unsigned long long key, uint32_t zmx, uint32_t imx, uint32_t fst, uint32_t lst) {
    atomicMin(&S_key[c], key);
    uint32_t* a = S_aux + size_t(c) * 4;
    if (zmx) atomicMax(a + 0, zmx);
    if (has_int) {
      if (imx) atomicMax(a + 1, imx);
      atomicMin(a + 2, fst);
    }
    if (has_col) {
      atomicMax(a + 3, lst);
      S_cobs[lst] = M->rgb[lst];
    }
    const uint32_t idx = uint32_t(key);
    if (idx != kNoIdx) {
      const float4 p = s_pt[idx - lb * 256u];
      float var = 0.0f;  // CellObservation default (elevation_mapping.hpp:26-34)
      if (B.has_var) var = M->var[idx];
      else if (B.integrate_mode) var = sigma_z2(V, p.x, p.y, p.z);
      S_obs[idx] = make_float2(p.w, var);
    }
  };

  unsigned long long key = kEmptyKey;
  uint32_t zmx = 0, imx = 0, fst = kNoIdx, lst = 0;
  if (inside) {
    if (z == 0.0f) atomicMin(S_zs + size_t(cell) * 2, (i << 1) | (__float_as_uint(z) >> 31));
    if (has_int && vint == 0.0f) atomicMin(S_zs + size_t(cell) * 2 + 1, (i << 1) | (__float_as_uint(vint) >> 31));
    key = make_key(z, i);
    zmx = make_zmax(z);
    if (has_int) {
      const bool vnan = isnan(vint);
      imx = vnan ? 0u : ord(vint);
      fst = (i << 1) | (vnan ? 1u : 0u);
    }
    lst = i;
  }
  // segmented scan over runs of equal cell in neighbouring lanes (see bin_body)
  bool commit = inside;
  {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ocell = __shfl_up(cell, d);
      const unsigned long long okey = __shfl_up(key, d);
      const uint32_t ozmx = __shfl_up(zmx, d);
      const uint32_t oimx = __shfl_up(imx, d);
      const uint32_t ofst = __shfl_up(fst, d);
      if (int(lane) >= d && ocell == cell && inside) {
        key = okey < key ? okey : key;
        zmx = ozmx > zmx ? ozmx : zmx;
        imx = oimx > imx ? oimx : imx;
        fst = ofst < fst ? ofst : fst;
      }
    }
    const int ncellv = __shfl_down(cell, 1);
    commit = inside && (lane == 63u || ncellv != cell);
  }
  const bool use_table = B.bin_table && 2 * __popcll(__ballot(commit)) > __popcll(__ballot(inside));
  if (use_table) {
    if (commit) {
      uint32_t h = uint32_t(cell) & 255u;
      while (true) {
        const uint32_t seen = t_cell[h];
        if (seen == uint32_t(cell)) break;
        if (seen == kEmptyCell) {
          const uint32_t prev = atomicCAS(&t_cell[h], kEmptyCell, uint32_t(cell));
          if (prev == kEmptyCell || prev == uint32_t(cell)) break;
        }
        h = (h + 1) & 255u;
      }
      atomicMin(&t_key[h], key);
      if (zmx) atomicMax(&t_zmx[h], zmx);
      if (has_int) {
        if (imx) atomicMax(&t_imx[h], imx);
        atomicMin(&t_fst[h], fst);
      }
      if (has_col) atomicMax(&t_lst[h], lst);
    }
  } else if (commit) {
    merge(uint32_t(cell), key, zmx, imx, fst, lst);
  }

  const unsigned long long mi = __ballot(inside), mg = __ballot(glob);
  if (lane == 0u) {
    s_in[wave] = unsigned(__popcll(mi));
    if (mg) ms->inside[k] = 1u;
  }
  __syncthreads();
  if (B.bin_table) {  // one slot per thread
    const uint32_t tc = t_cell[threadIdx.x];
    if (tc != kEmptyCell)
      merge(tc, t_key[threadIdx.x], t_zmx[threadIdx.x], t_imx[threadIdx.x], t_fst[threadIdx.x], t_lst[threadIdx.x]);
  }
  if (threadIdx.x == 0) {
    const unsigned ni = s_in[0] + s_in[1] + s_in[2] + s_in[3];
    B.bin_part[bid] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
}

// ---------------------------------------------------------------------------------------------
// update half: 256 cells of the map, every scan of the batch.
template <typename POLICY, int CH>
__device__ __forceinline__ void mupdate_body(const MBatch& B, const GeomConst& G, DevState* __restrict__ st,
                                             const typename POLICY::Layers& L,
                                             float* const* __restrict__ all_layers, int n_layers,
                                             const unsigned ncell, const unsigned bid) {
  constexpr bool has_int = (CH & 1) != 0, has_col = (CH & 2) != 0;
  const float nanv = __uint_as_float(0x7FC00000u);
  __shared__ unsigned s_t[4];
  const unsigned lt = threadIdx.x, lane = lt & 63u;
  const unsigned o = bid * 256u + lt;
  const bool valid = o < ncell;
  const unsigned count = B.count;
  MState* const ms = B.ms;

  // ---- round trip 1: keys of every scan, the stored state, the per-scan geometry (lane j holds scan j's) ----
  unsigned long long kk[kMaxBatch];
#pragma unroll
  for (int k = 0; k < kMaxBatch; ++k) {
    kk[k] = kEmptyKey;
    if (unsigned(k) < count && valid) kk[k] = B.key[size_t(k) * ncell + o];
  }
  typename POLICY::State stt;
  POLICY::set_nan(stt);
  float sint = nanv;
  if (valid) {
    POLICY::load(L, o, stt);
    if (has_int) sint = L.intensity[o];
  }
  int v_sr = 0, v_sc = 0, v_shr = 0, v_shc = 0;
  unsigned v_done = 0u, v_in = 0u;
  if (lane < count) {
    v_sr = ms->E[lane].sr; v_sc = ms->E[lane].sc;
    v_shr = ms->C[lane].shr; v_shc = ms->C[lane].shc;
    v_done = ms->done[lane];
    v_in = ms->inside[lane];
  }
  const bool v_applied = lane < count && B.do_move && (!B.gate_on_filter || (v_done >> 16) != 0u);
  const unsigned umask = uni(unsigned(__ballot(lane < count && v_in != 0u)));                     // scans that observed a cell
  unsigned stripmask = uni(unsigned(__ballot(v_applied && (v_shr != 0 || v_shc != 0))));         // scans whose move vacated cells

  if (bid == 0 && lt == 0) {  // commit the geometry ring behind the batch (make_ctx does this per scan)
    const unsigned last = count - 1u;
    const unsigned slot_next = (B.scan_no0 + count) & 3u;
    DevGeom g = ms->E[last];
    const DevCand cl = ms->C[last];
    const bool applied_last = B.do_move && (!B.gate_on_filter || (ms->done[last] >> 16) != 0u);
    if (applied_last) { g.px = cl.px; g.py = cl.py; g.sr = cl.sr; g.sc = cl.sc; }
    const unsigned ob_prev = st->obst[B.scan_no0 & 3u].scan;
    st->geom[slot_next] = g;
    st->cand[(B.scan_no0 + last) & 3u] = cl;   // (the synchronous statistics report the last scan's shift)
    // the last scan of the batch that observed a cell (umask: one bit per scan)
    st->obst[slot_next].scan = umask ? B.scan_no0 + (31u - unsigned(__clz(int(umask)))) : ob_prev;
#pragma unroll
    for (int s = 0; s < 4; ++s) { st->flags[s].any_pass = 0u; st->flags[s].any_inside = 0u; st->flags[s].ray_any = 0u; }
    if (umask) {
      const unsigned first_upd = B.scan_no0 + unsigned(__ffs(int(umask))) - 1u;
      if (has_int && st->vis_int == 0u) st->vis_int = 3u * first_upd + 2u;
      if (has_col && st->vis_col == 0u) st->vis_col = 3u * first_upd + 2u;
    }
    if (ms->err) st->fault = 1u;
  }

  unsigned tmask = 0u;
  uint32_t idx[kMaxBatch];
#pragma unroll
  for (int k = 0; k < kMaxBatch; ++k) {
    tmask |= (kk[k] != kEmptyKey) ? (1u << k) : 0u;
    idx[k] = uint32_t(kk[k]);
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

  // ---- the cell's events, in scan order ----
  unsigned m = tmask, lastp1 = 0u;
  bool evt = false, cleared = false, strip_any = false, obst_dirty = false;
  float obst = nanv;
  uint32_t colv = 0x7FC00000u;
  while (__ballot(m != 0u)) {
    const bool act = m != 0u;
    const unsigned k = act ? unsigned(__ffs(int(m))) - 1u : 0u;
    m &= m - 1u;
    uint32_t id = idx[0];
#pragma unroll
    for (int j = 1; j < kMaxBatch; ++j) id = (k == unsigned(j)) ? idx[j] : id;
    float2 ob = make_float2(kFltMax, 0.0f);
    uint4 ax = make_uint4(0u, 0u, kNoIdx, 0u);
    uint2 zsw = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    if (act) {
      if (id != kNoIdx) ob = B.obs[size_t(k) * B.obs_stride + id];
      ax = B.aux[size_t(k) * ncell + o];
      zsw = B.zs[size_t(k) * ncell + o];
    }
    uint32_t rgb = 0u;
    if (has_col && act) rgb = B.cobs[size_t(k) * B.obs_stride + ax.w];
    if (act) {
      const unsigned upto = (2u << k) - 1u, from = (1u << lastp1) - 1u;  // scans lastp1 .. k
      if (smask & upto & ~from) {  // vacated since the last event: NaN in every layer (GridMap::move)
        POLICY::set_nan(stt);
        sint = nanv;
        colv = 0x7FC00000u;
        strip_any = true;
      }
      const float min_z = ob.x, min_z_var = ob.y;  // (no finite z: FLT_MAX, variance 0 — elevation_mapping.hpp:26-34)
      const uint32_t zm = ax.x, imx = ax.y, fst = ax.z;
      const float max_z = zm ? ((zm == 0x80000000u && (zsw.x & 1u)) ? -0.0f : unord(zm)) : -kFltMax;
      POLICY::step(L, stt, min_z, min_z_var, max_z);
      obst = (max_z > min_z) ? max_z : nanv;
      obst_dirty = true;
      if (has_int) {
        const float obs = (fst & 1u) ? nanv : ((imx == 0x80000000u && (zsw.y & 1u)) ? -0.0f : unord(imx));
        if (isnan(sint) || obs > sint) sint = obs;
      }
      if (has_col) colv = rgb & 0x00FFFFFFu;
      B.key[size_t(k) * ncell + o] = kEmptyKey;  // the scratch is clean again for the batch after next
      B.aux[size_t(k) * ncell + o] = make_uint4(0u, 0u, kNoIdx, 0u);
      if ((zsw.x & zsw.y) != 0xFFFFFFFFu) B.zs[size_t(k) * ncell + o] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      evt = true;
      cleared = false;
      lastp1 = k + 1u;
    }
  }
  // the scans after the cell's last event
  if (valid) {
    const unsigned all = count >= 32u ? 0xFFFFFFFFu : (1u << count) - 1u, from = (1u << lastp1) - 1u;
    const unsigned tail = all & ~from;
    if (smask & tail) { strip_any = true; cleared = true; sint = nanv; colv = 0x7FC00000u; obst = nanv; obst_dirty = true; }
    if (umask & tail) { obst = nanv; obst_dirty = true; }  // map_.clear(obstacle), elevation_mapping.cpp:144-146
    if (smask) strip_any = true;
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
    if (evt && !cleared) POLICY::store(L, o, stt);
    if (obst_dirty) L.obstacle[o] = obst;
    if (has_int && (evt || strip_any)) L.intensity[o] = sint;
    if (has_col && (evt || strip_any)) reinterpret_cast<uint32_t*>(L.color)[o] = colv;
  }
  // per-tile touched-cell count of the batch's last scan (what the synchronous statistics report)
  const unsigned long long mt = __ballot(valid && ((tmask >> (count - 1u)) & 1u) != 0u);
  if (lane == 0u) s_t[lt >> 6] = unsigned(__popcll(mt));
  __syncthreads();
  if (lt == 0 && bid * 256u < ncell) B.upd_part[bid] = s_t[0] + s_t[1] + s_t[2] + s_t[3];
}

template <typename POLICY, int CH>
__global__ __launch_bounds__(256) void k_mupdate(const MBatch Bu, const GeomConst G, DevState* __restrict__ st,
                                                 const typename POLICY::Layers L,
                                                 float* const* __restrict__ all_layers, int n_layers, unsigned ncell) {
  mupdate_body<POLICY, CH>(Bu, G, st, L, all_layers, n_layers, ncell, blockIdx.x);
}

template <int CH>
__global__ __launch_bounds__(256) void k_mbin(const MBatch Bb, const GeomConst G, DevState* __restrict__ st,
                                              unsigned ncell) {
  mbin_body<CH>(Bb, G, st, ncell, blockIdx.x);
}

// update of batch b + bin of batch b+1 in one launch (scratch sets, observation arrays and MState double-buffered
// by batch parity)
template <typename POLICY, int CH>
__global__ __launch_bounds__(256) void k_mupdate_mbin(const MBatch Bu, const GeomConst G, DevState* __restrict__ st,
                                                      const typename POLICY::Layers L,
                                                      float* const* __restrict__ all_layers, int n_layers,
                                                      unsigned ncell, unsigned upd_blocks, const MBatch Bb) {
  if (blockIdx.x < upd_blocks) mupdate_body<POLICY, CH>(Bu, G, st, L, all_layers, n_layers, ncell, blockIdx.x);
  else mbin_body<CH>(Bb, G, st, ncell, blockIdx.x - upd_blocks);
}

}  // namespace fdm
