// fdm_kernels.hpp — the two kernels of one scan (gfx950 / CDNA4, wave64).
//
//   k_bin4 / k_bin : per input point.  Fused preprocessScan (fastdem.cpp:164-190: T_base_sensor,
//       cropRange, cropZ, T_world_base) + LOCAL-mode move arithmetic + nanogrid getIndex +
//       per-cell reduction (elevation_mapping.cpp:41-92).  SoA channels are read with 16 B/lane
//       coalesced loads; same-cell points are merged in registers, then in a per-block LDS table
//       ("LDS-staged cell tile"), and only the block's UNIQUE cells go to the device-resident
//       scratch with one atomic set each.  Every flushed cell stamps its 1024-cell map tile with
//       the scan number.
//   k_update : one block per 1024-cell map tile, cells visited in MEMORY ORDER (column-major, so a
//       wave's accesses to each layer coalesce).  A tile that was not stamped by this scan, not
//       stamped by the last updating scan (obstacle clear) and not crossed by a GridMap::move strip
//       costs one scalar load.  Touched cells get the Kalman / P2 update
//       (elevation_mapping.cpp:94-108), min/max, obstacle, intensity, colour
//       (elevation_mapping.cpp:127-175); the scratch is reset; block 0 commits the geometry.
//
// Measured facts this design rests on (MI355X, scripts/ubench/atomics.hip, profiles/):
//   * global atomics execute at the memory side (TCC_EA0_ATOMIC == TCC_ATOMIC): ~26 Gop/s for
//     scattered addresses, ~1 ns per op on one address, 7-12x faster when a wave's addresses are
//     consecutive; returning atomics are slower still.  Atomic COUNT, not bytes, bounds the bin
//     kernel => merge on chip first, never ask for the old value.
//   * same-address counters serialise => statistics are per-block partials summed on demand.
//   * scattered 32 B accesses run at 20-80 G/s => the update walks cells in memory order instead
//     of following a touched-cell list (tried: 1.8x slower).
//
// Roofline: HBM-bound work (no contraction => MFMA is irrelevant).  Algorithmic bytes
// (SURVEY.md §8d): 12 B per input point (+4 intensity, +4 colour); per touched cell 72 B (Kalman)
// / 124 B (P2); 4 B per map cell per scan for the reference's whole-layer obstacle clear (which
// this engine narrows to the tiles that can hold non-NaN obstacle cells).
#pragma once

#include "fdm_device.hpp"

namespace fdm {


constexpr int kMaxLayers = 64;  // layers per map
constexpr int kTileShift = 8;   // k_update tile = 256 storage-linear cells

// Device-resident per-cell scratch of one scan + per-tile stamps.
struct Scratch {
  unsigned long long* bin_part;  // [bin blocks] lo32 = n_after_filter, hi32 = n_in_map
  uint32_t* upd_part;            // [tiles] cells touched in the tile
  uint32_t* tile_stamp;          // [tiles] number of the last scan that touched a cell of the tile
  int dense;                     // 1: every tile is visited (small/medium maps); 0: stamp-gated
  unsigned long long* key;       // (ord(z) << 32 | point index), min-reduced; kEmptyKey = untouched
  // aux[cell] = {zmax, imax, first, last}: one 16 B group per cell, so the update kernel reads and
  // resets the whole group with one access.
  //   zmax  ord(max z), 0 = none
  //   imax  ord(max non-NaN intensity), 0 = none
  //   first (lowest point index in the cell << 1) | that point's intensity is NaN
  //         — the reference keeps the FIRST point's intensity unconditionally, so a NaN there
  //         sticks (elevation_mapping.cpp:73-79); kNoIdx = none
  //   last  highest point index in the cell (colour = last point wins)
  uint4* aux;
  // zs[cell] = {(lowest index of a point with z == +-0) << 1 | that z is -0, the same for the intensity}; all-ones =
  // none.  -0 and +0 tie in the reference's comparisons and the FIRST one seen stays (elevation_mapping.cpp:65-79), so
  // when a cell's max z / max intensity is a zero its sign is the first zero-valued point's.  Touched only by
  // zero-valued points (bin) and read with the aux group (update): free for real scans, exact for synthetic ones.
  uint2* zs;
  // optional captures for the scan callbacks (null unless fdm_engine_capture enabled them)
  float* cap_x;                  // [n] map-frame coordinates of every input point
  float* cap_y;
  float* cap_z;
  float* cap_var;                // [n] sigma_z^2 of every input point (nullable on its own)
  float* cap_cov;                // [9][cap_stride] full R Sigma R^T of every input point (nullable on its own)
  size_t cap_stride;
  int cap_drop_nan;              // 1: a point the crops dropped is stored with x = NaN (ray stage input)
  float* ras_z;                  // [cells] min z observed by this scan (NaN = not observed)
  // optional write-through (fdm_engine_integrate_async on PINNED host arrays): the bin kernel reads
  // the scan over PCIe exactly once and leaves a copy in HBM for the update kernel's gather
  float* wt_x;                   // [n] null = off
  float* wt_y;
  float* wt_z;
  float* wt_var;                 // [n] copy of wt_src_var (channels the bin kernel does not use itself)
  uint32_t* wt_rgb;
  const float* wt_src_var;
  const uint32_t* wt_src_rgb;
};

struct KalmanLayers {
  float *elevation, *elevation_min, *elevation_max, *variance, *n_points, *kalman_p, *sample_mean,
      *sample_m2, *upper, *lower, *obstacle, *intensity, *color;
  float min_var, max_var, q;
  static constexpr int istride = 1;
};
struct P2Layers {
  float *elevation, *elevation_min, *elevation_max, *variance, *n_points, *upper, *lower, *obstacle,
      *intensity, *color;
  float* q[5];
  float* n[5];
  P2Params p;
  static constexpr int istride = 1;
};

// Record layout ("cell records"): the estimator state of ONE cell packed into one 64 B (Kalman) or
// 128 B (P2) line, so a touched cell costs one line read + one line write instead of one
// scattered sector per layer (measured: k_update moved 138 MB for 17 MB of algorithmic bytes
// with the per-layer layout).  The named layers still exist for the API: a record field is a
// strided view (pointer = rec + field, stride = record size); obstacle / intensity / colour and
// user layers stay one-array-per-layer.
constexpr int kKalmanRec = 16;  // floats per record
constexpr int kP2Rec = 32;
// field index inside the record, in the order of kKalmanFields / kP2Fields (fdm_engine.hip)
enum KalmanField { KF_ELEV = 0, KF_MIN, KF_MAX, KF_VAR, KF_N, KF_P, KF_MEAN, KF_M2, KF_UP, KF_LO, KF_COUNT };
enum P2Field { PF_ELEV = 0, PF_MIN, PF_MAX, PF_VAR, PF_N, PF_Q0, PF_N0 = PF_Q0 + 5, PF_UP = PF_N0 + 5, PF_LO, PF_COUNT };

// (intensity: element o of the layer is intensity[o * istride] — with cell records the running maximum of the
// intensity is a record field too (slot kKalmanIntSlot / kP2IntSlot: a touched cell's read-modify-write of it rides in
// the record's line instead of being a scattered access of its own); one array per layer otherwise, istride 1)
struct KalmanRecLayers {
  float* rec;  // [cells][kKalmanRec]
  float *obstacle, *intensity, *color;
  float min_var, max_var, q;
  int istride;
};
struct P2RecLayers {
  float* rec;  // [cells][kP2Rec]
  float *obstacle, *intensity, *color;
  P2Params p;
  int istride;
};
constexpr int kKalmanIntSlot = 10, kP2IntSlot = 17;  // (behind the last estimator field; cleared with the record)

// Bring one input point into the map frame; returns whether it survived the crops.
template <class PT>
__device__ __forceinline__ bool preprocess_point(const PT& P, float& x, float& y, float& z) {
  if (!P.integrate_mode) return true;
  float w = 1.0f;
  transform4(P.Tbs, x, y, z, w);
  const float d2 = sum3(x * x, y * y, z * z);
  bool pass = (d2 >= P.min_sq) && (d2 <= P.max_sq);
  pass = pass && (z >= P.z_min) && (z <= P.z_max);
  transform4(P.Twb, x, y, z, w);
  return pass;
}

// point -> storage-linear cell of this engine's owned window; -1 outside the (global) map,
// -2 inside the map but owned by another tile.  "Some point landed in the map" is a GLOBAL fact
// (it gates the obstacle clear, elevation_mapping.cpp:118-121) that every tile derives by itself
// because every tile sees the whole scan.
__device__ __forceinline__ int owned_cell(float x, float y, const DevCand& cand, const GeomConst& G) {
  int r, c;
  if (!cell_of(x, y, cand, G, r, c)) return -1;
  const int lr = r - G.o_r0, lc = c - G.o_c0;
  if (lr < 0 || lc < 0 || lr >= G.o_rows || lc >= G.o_cols) return -2;
  return (c - G.s_c0) * G.s_rows + (r - G.s_r0);
}

__device__ __forceinline__ unsigned long long make_key(float z, unsigned i) {
  const float zc = (z == 0.0f) ? 0.0f : z;  // -0 and +0 tie, the first index wins
  // strict "z < min_z" starting from FLT_MAX: NaN / +inf / FLT_MAX never become the minimum
  return (z < kFltMax) ? ((unsigned long long)ord(zc) << 32) | i
                       : ((unsigned long long)ord(kFltMax) << 32) | kNoIdx;
}
__device__ __forceinline__ uint32_t make_zmax(float z) {
  const float zc = (z == 0.0f) ? 0.0f : z;
  return (z > -kFltMax) ? ord(zc) : 0u;  // strict "z > max_z" starting from lowest()
}

// One cell's reduction goes to the scratch (non-returning atomics) and stamps the cell's tile.
template <bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ void scratch_merge(const Scratch& S, unsigned scan_no, uint32_t cell,
                                              unsigned long long key, uint32_t zmx, uint32_t imx,
                                              uint32_t fst, uint32_t lst) {
  atomicMin(&S.key[cell], key);
  uint32_t* a = reinterpret_cast<uint32_t*>(S.aux) + size_t(cell) * 4;
  if (zmx) atomicMax(a + 0, zmx);
  if (HAS_INT) {
    if (imx) atomicMax(a + 1, imx);
    atomicMin(a + 2, fst);
  }
  if (HAS_COL) atomicMax(a + 3, lst);
  if (!S.dense) S.tile_stamp[cell >> kTileShift] = scan_no;  // benign race: all writers store the same value
}

// a zero-valued z / intensity of point i in `cell`: remember the first one and its sign (see Scratch::zs)
template <bool HAS_INT>
__device__ __forceinline__ void note_zeros(const Scratch& S, uint32_t cell, unsigned i, float z, float v) {
  if (z == 0.0f)
    atomicMin(reinterpret_cast<uint32_t*>(S.zs) + size_t(cell) * 2, (i << 1) | (__float_as_uint(z) >> 31));
  if (HAS_INT && v == 0.0f)
    atomicMin(reinterpret_cast<uint32_t*>(S.zs) + size_t(cell) * 2 + 1, (i << 1) | (__float_as_uint(v) >> 31));
}

// In two steps, so that a kernel can put work between the state read and the walk (k_tbin: the transforms and crops of
// its points, which need no geometry — 255 threads used to idle at the barrier while thread 0 waited for the state):
//   candidate_begin   thread 0 issues the loads of the geometry ring entry it chains from
//   candidate_finish  thread 0 walks the move, publishes; barrier; every thread returns the candidate
struct CandState {
  DevGeom g;
  DevCand pc;
  unsigned any_pass;
};
__device__ __forceinline__ void candidate_begin(const ScanParams& P, const DevState* __restrict__ st, CandState& cs) {
  if (threadIdx.x == 0) {
    if (P.chain_prev) {  // what k_update of the previous scan commits to geom[P.slot] (make_ctx)
      const int ps = (P.slot + 3) & 3;
      cs.g = st->geom[ps];
      cs.any_pass = st->flags[ps].any_pass;
      cs.pc = st->cand[ps];
    } else {
      cs.g = st->geom[P.slot];
    }
  }
}
__device__ __forceinline__ DevCand candidate_finish(const ScanParams& P, const GeomConst& G, DevState* __restrict__ st,
                                                    const CandState& cs, DevCand* s_cand, unsigned bid) {
  if (threadIdx.x == 0) {
    DevGeom g = cs.g;
    if (P.chain_prev) {
      const bool applied = P.prev_do_move && (!P.prev_gate || cs.any_pass != 0u);
      if (applied) { g.px = cs.pc.px; g.py = cs.pc.py; g.sr = cs.pc.sr; g.sc = cs.pc.sc; }
    }
    DevCand c;
    if (P.do_move) {
      // (every block of a launch walks this on ONE thread: the variant without the two fp64 divides on the common
      // path, bit-identical by construction — fdm_device.hpp)
      c = move_candidate_fast(g, G, P.robot_x, P.robot_y);
    } else {
      c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
    }
    *s_cand = c;
    if (bid == 0) {
      st->cand[P.slot] = c;
      if (P.force_inside) st->flags[P.slot].any_inside = 1u;
    }
  }
  __syncthreads();
  return *s_cand;
}
__device__ __forceinline__ DevCand block_candidate(const ScanParams& P, const GeomConst& G,
                                                   DevState* __restrict__ st, DevCand* s_cand,
                                                   unsigned bid) {
  CandState cs;
  candidate_begin(P, st, cs);
  return candidate_finish(P, G, st, cs, s_cand, bid);
}

// ---------------------------------------------------------------------------------------------
// k_bin4 — production bin kernel for large scans: 1024 consecutive points per 256-thread block,
// four CONSECUTIVE points per thread (dwordx4 loads).
constexpr uint32_t kEmptyCell = 0xFFFFFFFFu;

template <bool HAS_INT, bool HAS_COL, int THREADS, int LEAN = 0>
__device__ __forceinline__ void bin4_body(const ScanParams& P, const GeomConst& G,
                                          DevState* __restrict__ st, const float* __restrict__ px,
                                          const float* __restrict__ py, const float* __restrict__ pz,
                                          const float* __restrict__ pint, const Scratch& S,
                                          int32_t* __restrict__ cell_ids, const unsigned bid) {
  // LEAN (the fused launches of a plain scan): captures, write-through, cell ids, the non-finite filter and
  // the measurement switches are compiled out — dormant, they still cost 3-7 % of the launch.
  // LEAN == 2 keeps the write-through (a held-back update gathers from the engine's copy, so the caller's
  // arrays are free as soon as this kernel has run).
  float* const cap_x = LEAN ? nullptr : S.cap_x;
  float* const cap_var = LEAN ? nullptr : S.cap_var;
  float* const wt_x = LEAN == 1 ? nullptr : S.wt_x;
  int32_t* const ids = LEAN ? nullptr : cell_ids;
  const int dbg_na = LEAN ? 0 : P.dbg_no_atomics;
  const bool drop_nf = LEAN ? false : P.drop_nonfinite != 0;
  constexpr int kHashSlots = THREADS * 4;  // == points per block: room for every point in its own cell
  __shared__ unsigned long long h_key[kHashSlots];
  __shared__ uint32_t h_cell[kHashSlots];
  __shared__ uint32_t h_zmax[kHashSlots];
  __shared__ uint32_t h_imax[HAS_INT ? kHashSlots : 1];
  __shared__ uint32_t h_first[HAS_INT ? kHashSlots : 1];
  __shared__ uint32_t h_last[HAS_COL ? kHashSlots : 1];
  __shared__ DevCand s_cand;
  __shared__ unsigned s_cnt[THREADS / 64];

  // the point loads go out first: they are in flight while the table is initialised and
  // thread 0 works out the post-move geometry
  const unsigned i0 = (bid * unsigned(THREADS) + threadIdx.x) * 4u;
  float xs[4], ys[4], zs[4], vs[4];
  if (i0 + 3 < P.n) {
    const float4 a = *reinterpret_cast<const float4*>(px + i0);
    const float4 b = *reinterpret_cast<const float4*>(py + i0);
    const float4 c = *reinterpret_cast<const float4*>(pz + i0);
    xs[0] = a.x; xs[1] = a.y; xs[2] = a.z; xs[3] = a.w;
    ys[0] = b.x; ys[1] = b.y; ys[2] = b.z; ys[3] = b.w;
    zs[0] = c.x; zs[1] = c.y; zs[2] = c.z; zs[3] = c.w;
    if (HAS_INT) {
      const float4 d = *reinterpret_cast<const float4*>(pint + i0);
      vs[0] = d.x; vs[1] = d.y; vs[2] = d.z; vs[3] = d.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = i0 + j < P.n;
      xs[j] = ok ? px[i0 + j] : 0.f;
      ys[j] = ok ? py[i0 + j] : 0.f;
      zs[j] = ok ? pz[i0 + j] : 0.f;
      if (HAS_INT) vs[j] = ok ? pint[i0 + j] : 0.f;
    }
  }

  if (wt_x) {  // leave the raw scan in HBM for the update kernel (see Scratch::wt_x)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j >= P.n) break;
      wt_x[i0 + j] = xs[j]; S.wt_y[i0 + j] = ys[j]; S.wt_z[i0 + j] = zs[j];
      if (S.wt_var) S.wt_var[i0 + j] = S.wt_src_var[i0 + j];
      if (S.wt_rgb) S.wt_rgb[i0 + j] = S.wt_src_rgb[i0 + j];
    }
  }
  for (int k = threadIdx.x; k < kHashSlots; k += THREADS) {
    h_key[k] = kEmptyKey;
    h_cell[k] = kEmptyCell;
    h_zmax[k] = 0u;
    if (HAS_INT) { h_imax[k] = 0u; h_first[k] = kNoIdx; }
    if (HAS_COL) h_last[k] = 0u;
  }
  const DevCand cand = block_candidate(P, G, st, &s_cand, bid);  // contains the __syncthreads

  // phase 1: all four points through the arithmetic (independent chains -> ILP)
  int cells[4];
  unsigned n_pass = 0, n_in = 0;
  bool any_glob = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (dbg_na >= 3) {  // measurement only: loads without the arithmetic
      cells[j] = (xs[j] + ys[j] + zs[j] == 12345.f) ? 0 : -1;
      continue;
    }
    const bool live = i0 + j < P.n;
    float cvar = 0.f;
    if (cap_var && live && P.integrate_mode) cvar = sigma_z2(P, xs[j], ys[j], zs[j]);
    if (!LEAN && cap_var && S.cap_cov && live && P.integrate_mode) {
      float c9[9];
      cov_full(P, xs[j], ys[j], zs[j], c9);
#pragma unroll
      for (int k = 0; k < 9; ++k) S.cap_cov[size_t(k) * S.cap_stride + i0 + j] = c9[k];
    }
    const bool exists = live && (!drop_nf || (isfinite(xs[j]) && isfinite(ys[j]) && isfinite(zs[j])));
    const bool pass = preprocess_point(P, xs[j], ys[j], zs[j]) && exists;
    if (cap_x && live) {
      cap_x[i0 + j] = (S.cap_drop_nan && !pass) ? __uint_as_float(0x7FC00000u) : xs[j];
      S.cap_y[i0 + j] = ys[j]; S.cap_z[i0 + j] = zs[j];
      if (cap_var) cap_var[i0 + j] = cvar;
    }
    cells[j] = pass ? owned_cell(xs[j], ys[j], cand, G) : -1;
    n_pass += pass ? 1u : 0u;
    n_in += cells[j] >= 0 ? 1u : 0u;
    any_glob = any_glob || (pass && cells[j] != -1);
    if (ids && live) ids[i0 + j] = cells[j] >= 0 ? cells[j] : (!pass ? -1 : (cells[j] == -2 ? -3 : -2));
  }

  // phase 2: merge runs of equal cell in registers, fold each run into the block's LDS table
  int run_cell = -1;
  unsigned long long run_key = kEmptyKey;
  uint32_t run_zmx = 0, run_imx = 0, run_fst = kNoIdx, run_lst = 0;
  auto fold_run = [&]() {
    if (run_cell < 0 || dbg_na >= 2) return;
    uint32_t h = uint32_t(run_cell) & (kHashSlots - 1);
    while (true) {
      const uint32_t seen = h_cell[h];
      if (seen == uint32_t(run_cell)) break;
      if (seen == kEmptyCell) {
        const uint32_t prev = atomicCAS(&h_cell[h], kEmptyCell, uint32_t(run_cell));
        if (prev == kEmptyCell || prev == uint32_t(run_cell)) break;
      }
      h = (h + 1) & (kHashSlots - 1);
    }
    atomicMin(&h_key[h], run_key);
    if (run_zmx) atomicMax(&h_zmax[h], run_zmx);
    if (HAS_INT) {
      if (run_imx) atomicMax(&h_imax[h], run_imx);
      atomicMin(&h_first[h], run_fst);
    }
    if (HAS_COL) atomicMax(&h_last[h], run_lst);
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (cells[j] < 0) continue;
    const unsigned i = i0 + j;
    if (zs[j] == 0.0f || (HAS_INT && vs[j] == 0.0f)) note_zeros<HAS_INT>(S, uint32_t(cells[j]), i, zs[j], HAS_INT ? vs[j] : 1.0f);
    const unsigned long long key = make_key(zs[j], i);
    const uint32_t zmx = make_zmax(zs[j]);
    uint32_t imx = 0;
    bool vnan = false;
    if (HAS_INT) { vnan = isnan(vs[j]); imx = vnan ? 0u : ord(vs[j]); }
    if (cells[j] != run_cell) {
      fold_run();
      run_cell = cells[j];
      run_key = key;
      run_zmx = zmx;
      run_imx = imx;
      run_fst = (i << 1) | (vnan ? 1u : 0u);
    } else {
      run_key = key < run_key ? key : run_key;
      run_zmx = zmx > run_zmx ? zmx : run_zmx;
      run_imx = imx > run_imx ? imx : run_imx;
    }
    run_lst = i;
  }
  fold_run();

  unsigned v = n_pass | (n_in << 16);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = v;
  if (__ballot(any_glob) && (threadIdx.x & 63) == 0) st->flags[P.slot].any_inside = 1u;
  __syncthreads();  // every run of the block is in the table

  // phase 3: one global atomic set per unique cell of the block
  if (!dbg_na) {
#pragma unroll
    for (int q = 0; q < kHashSlots / THREADS; ++q) {
      const int k = threadIdx.x + q * THREADS;
      const uint32_t cell = h_cell[k];
      if (cell == kEmptyCell) continue;
      scratch_merge<HAS_INT, HAS_COL>(S, P.scan_no, cell, h_key[k], h_zmax[k],
                                      HAS_INT ? h_imax[k] : 0u, HAS_INT ? h_first[k] : 0u,
                                      HAS_COL ? h_last[k] : 0u);
    }
  }
  if (threadIdx.x == 0) {
    unsigned np = 0, ni = 0;
    for (int w = 0; w < THREADS / 64; ++w) { np += s_cnt[w] & 0xFFFFu; ni += s_cnt[w] >> 16; }
    if (np) st->flags[P.slot].any_pass = 1u;
    S.bin_part[bid] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
}

template <bool HAS_INT, bool HAS_COL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_bin4(const ScanParams P, const GeomConst G,
                                              DevState* __restrict__ st,
                                              const float* __restrict__ px,
                                              const float* __restrict__ py,
                                              const float* __restrict__ pz,
                                              const float* __restrict__ pint, const Scratch S,
                                              int32_t* __restrict__ cell_ids) {
  bin4_body<HAS_INT, HAS_COL, THREADS>(P, G, st, px, py, pz, pint, S, cell_ids, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_bin — one point per thread (small scans: latency matters more than atomic count, and
// unaligned channel pointers).  Same-cell runs of neighbouring lanes are merged inside the
// wavefront with a segmented scan; run tails go to the scratch.
// CH: the scan's optional channels as a compile-time constant (bit 0 intensity, bit 1 colour) or -1 = read
// them from ScanParams.  A VLP-16 scan is launch/latency-bound: with the channel tests folded away the fused
// launch of configs[1] takes 5.95 instead of 6.35 us.
template <bool WAVE_MERGE, int CH = -1, int LEAN = 0>
__device__ __forceinline__ void bin_body(const ScanParams& P, const GeomConst& G,
                                         DevState* __restrict__ st, const float* __restrict__ px,
                                         const float* __restrict__ py, const float* __restrict__ pz,
                                         const float* __restrict__ pint, const Scratch& S,
                                         int32_t* __restrict__ cell_ids, const unsigned bid) {
  // LEAN (the fused launches of a plain scan): captures, write-through, cell ids, the non-finite filter and
  // the measurement switches are compiled out — dormant, they still cost 3-7 % of the launch.
  // LEAN == 2 keeps the write-through (a held-back update gathers from the engine's copy, so the caller's
  // arrays are free as soon as this kernel has run).
  float* const cap_x = LEAN ? nullptr : S.cap_x;
  float* const cap_var = LEAN ? nullptr : S.cap_var;
  float* const wt_x = LEAN == 1 ? nullptr : S.wt_x;
  int32_t* const ids = LEAN ? nullptr : cell_ids;
  const int dbg_na = LEAN ? 0 : P.dbg_no_atomics;
  const bool drop_nf = LEAN ? false : P.drop_nonfinite != 0;
  const bool has_int = CH < 0 ? P.has_intensity != 0 : (CH & 1) != 0;
  const bool has_col = CH < 0 ? P.has_color != 0 : (CH & 2) != 0;
  __shared__ DevCand s_cand;
  __shared__ unsigned s_pass[4], s_in[4];
  // per-block table of the cells this block's 256 points fall into (P.bin_table): the run tails of
  // the wave merge fold into it with LDS atomics and only its occupied slots go to memory — a firing-
  // order scan puts the 16 samples a block holds of each beam into 2-3 cells, but 16 lanes apart
  __shared__ unsigned long long t_key[256];
  __shared__ uint32_t t_cell[256], t_zmx[256], t_imx[256], t_fst[256], t_lst[256];
  if (P.bin_table) {
    t_key[threadIdx.x] = kEmptyKey;
    t_cell[threadIdx.x] = kEmptyCell;
    t_zmx[threadIdx.x] = 0u; t_imx[threadIdx.x] = 0u; t_fst[threadIdx.x] = kNoIdx; t_lst[threadIdx.x] = 0u;
  }  // (made visible by the barrier inside block_candidate)
  // the point (and intensity) loads go out first: they are in flight while thread 0 reads the
  // geometry and works out the post-move candidate
  const unsigned i = bid * 256u + threadIdx.x;
  float x = 0.f, y = 0.f, z = 0.0f, vint = 0.f;
  if (i < P.n) {
    x = px[i];
    y = py[i];
    z = pz[i];
    if (has_int) vint = pint[i];
    if (wt_x) {  // leave the raw scan in HBM for the update kernel (see Scratch::wt_x)
      wt_x[i] = x; S.wt_y[i] = y; S.wt_z[i] = z;
      if (S.wt_var) S.wt_var[i] = S.wt_src_var[i];
      if (S.wt_rgb) S.wt_rgb[i] = S.wt_src_rgb[i];
    }
  }
  const DevCand cand = block_candidate(P, G, st, &s_cand, bid);

  bool pass = false;
  int cell = -1;
  if (i < P.n) {
    float cvar = 0.f;
    if (cap_var && P.integrate_mode) cvar = sigma_z2(P, x, y, z);
    if (!LEAN && cap_var && S.cap_cov && P.integrate_mode) {
      float c9[9];
      cov_full(P, x, y, z, c9);
#pragma unroll
      for (int k = 0; k < 9; ++k) S.cap_cov[size_t(k) * S.cap_stride + i] = c9[k];
    }
    const bool exists = !drop_nf || (isfinite(x) && isfinite(y) && isfinite(z));
    pass = preprocess_point(P, x, y, z) && exists;
    if (cap_x) {
      cap_x[i] = (S.cap_drop_nan && !pass) ? __uint_as_float(0x7FC00000u) : x;
      S.cap_y[i] = y; S.cap_z[i] = z;
      if (cap_var) cap_var[i] = cvar;
    }
    if (pass) cell = owned_cell(x, y, cand, G);
    if (ids) ids[i] = cell >= 0 ? cell : (!pass ? -1 : (cell == -2 ? -3 : -2));
  }
  const bool inside = cell >= 0;
  const bool glob = pass && cell != -1;

  unsigned long long key = kEmptyKey;
  uint32_t zmx = 0, imx = 0, fst = kNoIdx, lst = 0;
  if (inside) {
    if (z == 0.0f || (has_int && vint == 0.0f)) {
      if (has_int) note_zeros<true>(S, uint32_t(cell), i, z, vint);
      else note_zeros<false>(S, uint32_t(cell), i, z, 1.0f);
    }
    key = make_key(z, i);
    zmx = make_zmax(z);
    if (has_int) {
      const bool vnan = isnan(vint);
      imx = vnan ? 0u : ord(vint);
      fst = (i << 1) | (vnan ? 1u : 0u);
    }
    lst = i;
  }

  bool commit = inside;
  if (WAVE_MERGE) {
    // Segmented inclusive scan over runs of equal `cell` in neighbouring lanes; the LAST lane of
    // each run holds the run's reduction and is the only one to touch memory.  (Merging two
    // non-adjacent lanes of the same cell is harmless: min/max are idempotent.)
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ocell = __shfl_up(cell, d);
      const unsigned long long okey = __shfl_up(key, d);
      const uint32_t ozmx = __shfl_up(zmx, d);
      const uint32_t oimx = __shfl_up(imx, d);
      const uint32_t ofst = __shfl_up(fst, d);
      if (lane >= d && ocell == cell && inside) {
        key = okey < key ? okey : key;
        zmx = ozmx > zmx ? ozmx : zmx;
        imx = oimx > imx ? oimx : imx;
        fst = ofst < fst ? ofst : fst;
      }
    }
    const int ncell = __shfl_down(cell, 1);
    commit = inside && (lane == 63 || ncell != cell);
  }
  // a wave whose adjacent-lane merge already found long runs (ring-major scans: few tails) goes to
  // memory directly; one whose lanes are mostly their own tail (firing order) folds into the table
  const bool use_table = P.bin_table && 2 * __popcll(__ballot(commit)) > __popcll(__ballot(inside));
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
  } else if (commit && !dbg_na) {
    if (has_int && has_col)
      scratch_merge<true, true>(S, P.scan_no, cell, key, zmx, imx, fst, lst);
    else if (has_int)
      scratch_merge<true, false>(S, P.scan_no, cell, key, zmx, imx, fst, lst);
    else if (has_col)
      scratch_merge<false, true>(S, P.scan_no, cell, key, zmx, imx, fst, lst);
    else
      scratch_merge<false, false>(S, P.scan_no, cell, key, zmx, imx, fst, lst);
  }

  const unsigned long long mp = __ballot(pass), mi = __ballot(inside), mg = __ballot(glob);
  if ((threadIdx.x & 63) == 0) {
    s_pass[threadIdx.x >> 6] = __popcll(mp);
    s_in[threadIdx.x >> 6] = __popcll(mi);
    if (mg) st->flags[P.slot].any_inside = 1u;
  }
  __syncthreads();
  if (P.bin_table && !dbg_na) {  // one slot per thread
    const uint32_t tc = t_cell[threadIdx.x];
    if (tc != kEmptyCell) {
      const unsigned long long tk = t_key[threadIdx.x];
      const uint32_t a = t_zmx[threadIdx.x], b2 = t_imx[threadIdx.x], c2 = t_fst[threadIdx.x], d2 = t_lst[threadIdx.x];
      if (has_int && has_col) scratch_merge<true, true>(S, P.scan_no, tc, tk, a, b2, c2, d2);
      else if (has_int) scratch_merge<true, false>(S, P.scan_no, tc, tk, a, b2, c2, d2);
      else if (has_col) scratch_merge<false, true>(S, P.scan_no, tc, tk, a, b2, c2, d2);
      else scratch_merge<false, false>(S, P.scan_no, tc, tk, a, b2, c2, d2);
    }
  }
  if (threadIdx.x == 0) {
    const unsigned np = s_pass[0] + s_pass[1] + s_pass[2] + s_pass[3];
    const unsigned ni = s_in[0] + s_in[1] + s_in[2] + s_in[3];
    if (np) st->flags[P.slot].any_pass = 1u;
    S.bin_part[bid] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
}

template <bool WAVE_MERGE, int CH = -1>
__global__ __launch_bounds__(256) void k_bin(const ScanParams P, const GeomConst G,
                                             DevState* __restrict__ st,
                                             const float* __restrict__ px,
                                             const float* __restrict__ py,
                                             const float* __restrict__ pz,
                                             const float* __restrict__ pint, const Scratch S,
                                             int32_t* __restrict__ cell_ids) {
  bin_body<WAVE_MERGE, CH>(P, G, st, px, py, pz, pint, S, cell_ids, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_update: one thread per map cell, one block per 256-cell tile; consecutive threads are
// consecutive cells of a column, so every layer access of a wave coalesces.  The kernel is a chain
// of dependent memory round trips (context -> key -> winning point + stored state -> stores),
// so it keeps the chain SHORT (loads that do not depend on each other are issued together) and
// the machine FULL (about 30 VGPRs: 8 waves/SIMD):
//   round 1: scan context (scalar loads) + the cell's key, issued together
//            (dense mode reads the key unconditionally: an untouched cell simply holds kEmptyKey;
//             stamp mode - very large maps - first checks the tile stamp and skips idle tiles)
//   round 2: winning point's x/y/z, zmax/intensity/colour scratch, stored estimator state
//   round 3: arithmetic + stores.

struct UpdateCtx {
  bool applied, do_update;
  bool cur;        // tile may hold touched cells
  bool obst_tile;  // obstacle cells of this tile must be cleared (map_.clear(obstacle))
  bool strips;     // a move happened: cells may lie in a vacated strip
  DevGeom E;
  DevCand C;
};

__device__ __forceinline__ void make_ctx(const ScanParams& P, DevState* __restrict__ st,
                                         const Scratch& S, UpdateCtx& u, unsigned tile, unsigned lt,
                                         bool tile_ok) {
  const int slot = P.slot;
  const bool any_pass = st->flags[slot].any_pass != 0u;
  u.do_update = st->flags[slot].any_inside != 0u;
  u.applied = P.do_move && (!P.gate_on_filter || any_pass);
  const unsigned ob_scan = st->obst[slot].scan;
  u.E = st->geom[slot];
  u.C = st->cand[slot];
  if (tile == 0 && lt == 0) {  // commit geometry + ring bookkeeping
    const int nxt = (slot + 1) & 3, nn = (slot + 2) & 3;
    DevGeom g = u.E;
    if (u.applied) { g.px = u.C.px; g.py = u.C.py; g.sr = u.C.sr; g.sc = u.C.sc; }
    st->geom[nxt] = g;
    st->obst[nxt].scan = u.do_update ? P.scan_no : ob_scan;
    st->flags[nn].any_pass = 0u;
    st->flags[nn].any_inside = 0u;
    st->flags[nn].ray_any = 0u;
    if (u.do_update) {
      if (P.has_intensity && st->vis_int == 0u) st->vis_int = 3u * P.scan_no + 2u;
      if (P.has_color && st->vis_col == 0u) st->vis_col = 3u * P.scan_no + 2u;
    }
  }
  if (S.dense) {
    u.cur = u.do_update;
    u.obst_tile = u.do_update;
  } else {
    const unsigned stamp = tile_ok ? S.tile_stamp[tile] : 0xFFFFFFFEu;
    // (stamp == scan_no + 1: the NEXT scan's bin kernel, which shares a launch with this held-back
    // update, re-stamped the tile — it may or may not hold cells of this scan or of the last updating
    // one, so it is swept; untouched cells have an empty key and cost nothing but the sweep)
    u.cur = u.do_update && (stamp == P.scan_no || stamp == P.scan_no + 1u);
    u.obst_tile = u.do_update && (u.cur || stamp == ob_scan);
  }
  u.strips = u.applied && (u.C.shr != 0 || u.C.shc != 0);
}

// Estimator policies: how the state of a cell is loaded / NaN-ed / updated and stored.
// update(): estimator step (elevation_mapping.cpp:94-108) + updateMinMax (:127-142).
struct KalmanPolicy {  // one array per layer
  using Layers = KalmanLayers;
  struct State { KalmanState s; float smin, smax; };
  static __device__ __forceinline__ void load(const Layers& L, unsigned o, State& t) {
    t.s.x = L.elevation[o];
    t.s.P = L.kalman_p[o];
    t.s.count = L.n_points[o];
    t.s.mean = L.sample_mean[o];
    t.s.var = 0.0f;  // write-only: Kalman::update always overwrites the sample variance
    t.s.m2 = L.sample_m2[o];
    t.smin = L.elevation_min[o];
    t.smax = L.elevation_max[o];
  }
  static __device__ __forceinline__ void set_nan(State& t) {
    const float nanv = __uint_as_float(0x7FC00000u);
    t.s.x = t.s.P = t.s.count = t.s.mean = t.s.var = t.s.m2 = t.smin = t.smax = nanv;
  }
  static __device__ __forceinline__ void clear_cell(const Layers&, unsigned) {}
  // (move_basic: a vacated strip clears the three basic layers only)
  static __device__ __forceinline__ void set_nan_basic(State& t) { t.s.x = t.smin = t.smax = __uint_as_float(0x7FC00000u); }
  static __device__ __forceinline__ void clear_cell_basic(const Layers& L, unsigned o) {
    L.elevation[o] = L.elevation_min[o] = L.elevation_max[o] = __uint_as_float(0x7FC00000u);
  }
  static __device__ __forceinline__ void update(const Layers& L, unsigned o, State& t, float min_z,
                                                float var, float max_z) {
    kalman_step(t.s, min_z, var, L.min_var, L.max_var, L.q);
    L.elevation[o] = t.s.x;
    L.kalman_p[o] = t.s.P;
    L.n_points[o] = t.s.count;
    L.sample_mean[o] = t.s.mean;
    L.variance[o] = t.s.var;
    L.sample_m2[o] = t.s.m2;
    L.upper[o] = t.s.upper;
    L.lower[o] = t.s.lower;
    if (isnan(t.smin) || min_z < t.smin) L.elevation_min[o] = min_z;
    if (isnan(t.smax) || max_z > t.smax) L.elevation_max[o] = max_z;
  }
};

struct KalmanRecPolicy {  // cell records
  using Layers = KalmanRecLayers;
  struct State { KalmanState s; float smin, smax; float4 r2, r3; };
  static __device__ __forceinline__ void load(const Layers& L, unsigned o, State& t) {
    const float4* r = reinterpret_cast<const float4*>(L.rec + size_t(o) * kKalmanRec);
    const float4 a = r[0], b = r[1], c = r[2];
    t.s.x = a.x; t.smin = a.y; t.smax = a.z; t.s.var = a.w;
    t.s.count = b.x; t.s.P = b.y; t.s.mean = b.z; t.s.m2 = b.w;
    t.s.upper = c.x; t.s.lower = c.y;
  }
  static __device__ __forceinline__ void set_nan(State& t) {
    const float nanv = __uint_as_float(0x7FC00000u);
    t.s.x = t.s.P = t.s.count = t.s.mean = t.s.var = t.s.m2 = t.smin = t.smax = nanv;
  }
  static __device__ __forceinline__ void clear_cell(const Layers& L, unsigned o) {
    const float nanv = __uint_as_float(0x7FC00000u);
    float4* r = reinterpret_cast<float4*>(L.rec + size_t(o) * kKalmanRec);
    const float4 n4 = make_float4(nanv, nanv, nanv, nanv);
    r[0] = n4; r[1] = n4; r[2] = n4;
  }
  static __device__ __forceinline__ void set_nan_basic(State& t) { t.s.x = t.smin = t.smax = __uint_as_float(0x7FC00000u); }
  static __device__ __forceinline__ void clear_cell_basic(const Layers& L, unsigned o) {
    float* r = L.rec + size_t(o) * kKalmanRec;  // fields 0, 1, 2 = elevation, elevation_min, elevation_max
    r[0] = r[1] = r[2] = __uint_as_float(0x7FC00000u);
  }
  static __device__ __forceinline__ void update(const Layers& L, unsigned o, State& t, float min_z,
                                                float var, float max_z) {
    kalman_step(t.s, min_z, var, L.min_var, L.max_var, L.q);
    const float nmin = (isnan(t.smin) || min_z < t.smin) ? min_z : t.smin;
    const float nmax = (isnan(t.smax) || max_z > t.smax) ? max_z : t.smax;
    float4* r = reinterpret_cast<float4*>(L.rec + size_t(o) * kKalmanRec);
    r[0] = make_float4(t.s.x, nmin, nmax, t.s.var);
    r[1] = make_float4(t.s.count, t.s.P, t.s.mean, t.s.m2);
    reinterpret_cast<float2*>(r + 2)[0] = make_float2(t.s.upper, t.s.lower);
  }
  // the same update split in two for a cell that takes SEVERAL observations in one kernel (fdm_multi.hpp): step()
  // keeps everything in registers, store() writes the record once at the end
  // (the sample variance and the bounds once, in finish(), after the cell's last observation)
  static __device__ __forceinline__ void step(const Layers& L, State& t, float min_z, float var, float max_z) {
    kalman_core(t.s, min_z, var, L.min_var, L.max_var, L.q);
    t.smin = (isnan(t.smin) || min_z < t.smin) ? min_z : t.smin;
    t.smax = (isnan(t.smax) || max_z > t.smax) ? max_z : t.smax;
  }
  static __device__ __forceinline__ void finish(State& t) { kalman_finish(t.s); }
  static __device__ __forceinline__ float elevation(const State& t) { return t.s.x; }  // what store() leaves in the elevation field
  static __device__ __forceinline__ void store(const Layers& L, unsigned o, const State& t) {
    float4* r = reinterpret_cast<float4*>(L.rec + size_t(o) * kKalmanRec);
    r[0] = make_float4(t.s.x, t.smin, t.smax, t.s.var);
    r[1] = make_float4(t.s.count, t.s.P, t.s.mean, t.s.m2);
    reinterpret_cast<float2*>(r + 2)[0] = make_float2(t.s.upper, t.s.lower);
  }
};

struct P2Policy {  // one array per layer
  using Layers = P2Layers;
  struct State { P2State s; float smin, smax; };
  static __device__ __forceinline__ void load(const Layers& L, unsigned o, State& t) {
    t.s.count = L.n_points[o];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      t.s.q[k] = L.q[k][o];
      t.s.n[k] = L.n[k][o];
    }
    t.smin = L.elevation_min[o];
    t.smax = L.elevation_max[o];
  }
  static __device__ __forceinline__ void set_nan(State& t) {
    const float nanv = __uint_as_float(0x7FC00000u);
    t.s.count = t.smin = t.smax = nanv;
#pragma unroll
    for (int k = 0; k < 5; ++k) t.s.q[k] = t.s.n[k] = nanv;
  }
  static __device__ __forceinline__ void clear_cell(const Layers&, unsigned) {}
  static __device__ __forceinline__ void set_nan_basic(State& t) { t.smin = t.smax = __uint_as_float(0x7FC00000u); }
  static __device__ __forceinline__ void clear_cell_basic(const Layers& L, unsigned o) {
    L.elevation[o] = L.elevation_min[o] = L.elevation_max[o] = __uint_as_float(0x7FC00000u);
  }
  static __device__ __forceinline__ void update(const Layers& L, unsigned o, State& t, float min_z,
                                                float /*var*/, float max_z) {
    p2_step(t.s, min_z, L.p);
    L.n_points[o] = t.s.count;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      L.q[k][o] = t.s.q[k];
      L.n[k][o] = t.s.n[k];
    }
    L.elevation[o] = t.s.elevation;
    L.variance[o] = t.s.variance;
    L.upper[o] = t.s.upper;
    L.lower[o] = t.s.lower;
    if (isnan(t.smin) || min_z < t.smin) L.elevation_min[o] = min_z;
    if (isnan(t.smax) || max_z > t.smax) L.elevation_max[o] = max_z;
  }
};

struct P2RecPolicy {  // cell records
  using Layers = P2RecLayers;
  struct State { P2State s; float smin, smax; };
  static __device__ __forceinline__ void load(const Layers& L, unsigned o, State& t) {
    const float4* r = reinterpret_cast<const float4*>(L.rec + size_t(o) * kP2Rec);
    const float4 a = r[0], b = r[1], c = r[2], d = r[3];
    t.s.elevation = a.x;  // (p2_step rewrites it; the batch update's ray events read it for cells without an observation)
    t.smin = a.y; t.smax = a.z;
    t.s.count = b.x;
    t.s.q[0] = b.y; t.s.q[1] = b.z; t.s.q[2] = b.w; t.s.q[3] = c.x; t.s.q[4] = c.y;
    t.s.n[0] = c.z; t.s.n[1] = c.w; t.s.n[2] = d.x; t.s.n[3] = d.y; t.s.n[4] = d.z;
  }
  static __device__ __forceinline__ void set_nan(State& t) {
    const float nanv = __uint_as_float(0x7FC00000u);
    t.s.count = t.smin = t.smax = nanv;
#pragma unroll
    for (int k = 0; k < 5; ++k) t.s.q[k] = t.s.n[k] = nanv;
  }
  static __device__ __forceinline__ void clear_cell(const Layers& L, unsigned o) {
    const float nanv = __uint_as_float(0x7FC00000u);
    float4* r = reinterpret_cast<float4*>(L.rec + size_t(o) * kP2Rec);
    const float4 n4 = make_float4(nanv, nanv, nanv, nanv);
    r[0] = n4; r[1] = n4; r[2] = n4; r[3] = n4; r[4] = n4;
  }
  static __device__ __forceinline__ void set_nan_basic(State& t) { t.s.elevation = t.smin = t.smax = __uint_as_float(0x7FC00000u); }
  static __device__ __forceinline__ void clear_cell_basic(const Layers& L, unsigned o) {
    float* r = L.rec + size_t(o) * kP2Rec;  // fields 0, 1, 2 = elevation, elevation_min, elevation_max
    r[0] = r[1] = r[2] = __uint_as_float(0x7FC00000u);
  }
  static __device__ __forceinline__ void update(const Layers& L, unsigned o, State& t, float min_z,
                                                float /*var*/, float max_z) {
    p2_step(t.s, min_z, L.p);
    const float nmin = (isnan(t.smin) || min_z < t.smin) ? min_z : t.smin;
    const float nmax = (isnan(t.smax) || max_z > t.smax) ? max_z : t.smax;
    float4* r = reinterpret_cast<float4*>(L.rec + size_t(o) * kP2Rec);
    r[0] = make_float4(t.s.elevation, nmin, nmax, t.s.variance);
    r[1] = make_float4(t.s.count, t.s.q[0], t.s.q[1], t.s.q[2]);
    r[2] = make_float4(t.s.q[3], t.s.q[4], t.s.n[0], t.s.n[1]);
    r[3] = make_float4(t.s.n[2], t.s.n[3], t.s.n[4], t.s.upper);
    r[4].x = t.s.lower;
  }
  static __device__ __forceinline__ void step(const Layers& L, State& t, float min_z, float /*var*/, float max_z) {
    p2_step(t.s, min_z, L.p);
    t.smin = (isnan(t.smin) || min_z < t.smin) ? min_z : t.smin;
    t.smax = (isnan(t.smax) || max_z > t.smax) ? max_z : t.smax;
  }
  static __device__ __forceinline__ void finish(State&) {}
  static __device__ __forceinline__ float elevation(const State& t) { return t.s.elevation; }
  static __device__ __forceinline__ void store(const Layers& L, unsigned o, const State& t) {
    float4* r = reinterpret_cast<float4*>(L.rec + size_t(o) * kP2Rec);
    r[0] = make_float4(t.s.elevation, t.smin, t.smax, t.s.variance);
    r[1] = make_float4(t.s.count, t.s.q[0], t.s.q[1], t.s.q[2]);
    r[2] = make_float4(t.s.q[3], t.s.q[4], t.s.n[0], t.s.n[1]);
    r[3] = make_float4(t.s.n[2], t.s.n[3], t.s.n[4], t.s.upper);
    r[4].x = t.s.lower;
  }
};

// BLOCK = 256: one 256-cell tile per block; BLOCK = 512: two (so that the body fits into a launch of
// 512-thread blocks next to the large-scan bin kernel).
template <typename POLICY, int BLOCK = 256>
__device__ __forceinline__ void update_body(
    const ScanParams& P, const GeomConst& G, DevState* __restrict__ st,
    const typename POLICY::Layers& L, float* const* __restrict__ all_layers, int n_layers,
    const Scratch& S, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, const uint32_t* __restrict__ prgb, const float* __restrict__ pvar,
    unsigned ncell, const unsigned bid) {
  const float nanv = __uint_as_float(0x7FC00000u);
  __shared__ unsigned s_t[BLOCK / 64];
  const unsigned tile = bid * unsigned(BLOCK / 256) + (threadIdx.x >> 8), lt = threadIdx.x & 255u;
  const unsigned o = tile * 256u + lt;
  const bool valid = o < ncell;

  // ---- round 1: the cell's key (dense mode: independent of the context) + the scan context ----
  unsigned long long key = kEmptyKey;
  if (S.dense && valid) key = S.key[o];
  UpdateCtx u;
  make_ctx(P, st, S, u, tile, lt, tile * 256u < ncell);
  bool touched = false;
  if (valid && (u.cur || u.obst_tile || u.strips) && P.dbg_upd != 1) {
    if (!S.dense && u.cur) key = S.key[o];
    bool in_strip = false;
    // (a move of >= the map's size on an axis is clearAll() in either reading of move())
    const bool basic = P.move_basic && abs(u.C.shr) < G.rows && abs(u.C.shc) < G.cols;
    if (u.strips) {
      const int r = int(o % unsigned(G.s_rows)) + G.s_r0;
      const int col = int(o / unsigned(G.s_rows)) + G.s_c0;
      in_strip = in_cleared_strip(r, u.E.sr, u.C.shr, G.rows) ||
                 in_cleared_strip(col, u.E.sc, u.C.shc, G.cols);
      if (in_strip && basic) {
        POLICY::clear_cell_basic(L, o);  // (option "move_clear_basic": the three basic layers only)
      } else if (in_strip) {
        // NaN in EVERY layer (GridMap::move).  Pointers are fetched 8 at a time so the loop costs
        // ceil(n/8) round trips, not one per layer.
        for (int l0 = 0; l0 < n_layers; l0 += 8) {
          float* p[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) p[k] = all_layers[min(l0 + k, n_layers - 1)];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (l0 + k < n_layers) p[k][o] = nanv;
        }
        POLICY::clear_cell(L, o);  // record layout: the packed estimator state
      }
    }
    touched = u.cur && key != kEmptyKey && P.dbg_upd != 3;  // dbg_upd: measurement-only switches
    if (!touched) {
      // map_.clear(obstacle) (elevation_mapping.cpp:144-146) for the cells it can matter for
      if (u.obst_tile && (!in_strip || basic)) L.obstacle[o] = nanv;
    } else {
      // ---- round 2: every load the update needs, issued before any is used ----
      const uint32_t idx = uint32_t(key);
      float gx = 0.f, gy = 0.f, gz = 0.f, gvar = 0.f;
      if (idx != kNoIdx) {
        gx = px[idx];
        gy = py[idx];
        gz = pz[idx];
        if (P.has_var) gvar = pvar[idx];
      }
      const uint4 ax = S.aux[o];
      const uint2 zsw = S.zs[o];
      const uint32_t zm = ax.x, imx = ax.y, fst = ax.z, lst = ax.w;
      float sint = nanv;
      typename POLICY::State stt;
      if (in_strip && !basic) {
        POLICY::set_nan(stt);
      } else {
        POLICY::load(L, o, stt);
        if (P.has_intensity) sint = L.intensity[size_t(o) * L.istride];
        if (in_strip) POLICY::set_nan_basic(stt);
      }
      uint32_t rgb = 0u;
      // ---- round 3 (colour channel only): the last point's colour
      if (P.has_color) rgb = prgb[lst];

      // ---- one estimator update per touched cell (elevation_mapping.cpp:94-175) ----
      float min_z = kFltMax, min_z_var = 0.0f;  // CellObservation defaults (elevation_mapping.hpp:26-34)
      if (idx != kNoIdx) {
        float x = gx, y = gy, z = gz;
        if (P.has_var) min_z_var = gvar;
        else if (P.integrate_mode) min_z_var = sigma_z2(P, x, y, z);
        preprocess_point(P, x, y, z);
        min_z = z;
      }
      // (a zero maximum takes the sign of the first zero-valued point, see Scratch::zs)
      const float max_z = zm ? ((zm == 0x80000000u && (zsw.x & 1u)) ? -0.0f : unord(zm)) : -kFltMax;
      if (S.ras_z) S.ras_z[o] = min_z;
      POLICY::update(L, o, stt, min_z, min_z_var, max_z);
      L.obstacle[o] = (max_z > min_z) ? max_z : nanv;
      if (P.has_intensity) {
        const float obs = (fst & 1u) ? nanv  // first point NaN -> NaN (see Scratch)
                                     : ((imx == 0x80000000u && (zsw.y & 1u)) ? -0.0f : unord(imx));
        if (isnan(sint) || obs > sint) L.intensity[size_t(o) * L.istride] = obs;
      }
      if (P.has_color) reinterpret_cast<uint32_t*>(L.color)[o] = rgb & 0x00FFFFFFu;
      S.key[o] = kEmptyKey;  // scratch is clean again for the next scan
      S.aux[o] = make_uint4(0u, 0u, kNoIdx, 0u);
      if ((zsw.x & zsw.y) != 0xFFFFFFFFu) S.zs[o] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    }
  }
  // per-tile touched-cell count (plain store; summed by the host on demand)
  const unsigned long long m = __ballot(touched);
  if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = unsigned(__popcll(m));
  __syncthreads();
  if (lt == 0 && tile * 256u < ncell) {
    const unsigned* w = s_t + (threadIdx.x >> 8) * 4;
    S.upd_part[tile] = w[0] + w[1] + w[2] + w[3];
  }
}

template <typename POLICY>
__global__ __launch_bounds__(256) void k_update(
    const ScanParams P, const GeomConst G, DevState* __restrict__ st,
    const typename POLICY::Layers L, float* const* __restrict__ all_layers, int n_layers,
    const Scratch S, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, const float* __restrict__ /*pint: folded into aux by k_bin*/,
    const uint32_t* __restrict__ prgb, const float* __restrict__ pvar, unsigned ncell) {
  update_body<POLICY>(P, G, st, L, all_layers, n_layers, S, px, py, pz, prgb, pvar, ncell, blockIdx.x);
}

// Stamp-gated maps (> 4 M cells): almost every 256-cell tile is idle in a scan, and a block per tile
// would spend its life loading the scan context only to find that out (64 M cells = 250 K blocks:
// 165 us of nothing).  Here a block looks at kStampTiles consecutive tiles, decides "idle" from the
// tile stamp alone and runs the update body only for the live ones (and always for tile 0, which
// commits the geometry ring).
constexpr unsigned kStampTiles = 32;  // <= 64: one lane of the first wave per tile (16: 62 us, 32: 60 us, 64: 73 us at C5)
// BLOCK threads look at kStampTiles SLOTS of BLOCK/256 consecutive tiles each.
template <typename POLICY, int BLOCK>
__device__ __forceinline__ void update_stamped_body(
    const ScanParams& P, const GeomConst& G, DevState* __restrict__ st,
    const typename POLICY::Layers& L, float* const* __restrict__ all_layers, int n_layers,
    const Scratch& S, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, const uint32_t* __restrict__ prgb, const float* __restrict__ pvar,
    unsigned ncell, const unsigned bid) {
  constexpr unsigned kPer = unsigned(BLOCK) / 256u;
  const int slot = P.slot;
  const bool do_update = st->flags[slot].any_inside != 0u;
  const bool applied = P.do_move && (!P.gate_on_filter || st->flags[slot].any_pass != 0u);
  const int shr = st->cand[slot].shr, shc = st->cand[slot].shc;
  const bool strips = applied && (shr != 0 || shc != 0);  // a move: every tile may hold vacated cells
  const unsigned ob_scan = st->obst[slot].scan;
  const unsigned n_tiles = (ncell + 255u) >> 8;
  // all stamps in ONE round trip: lane q of the first wave looks at slot q
  __shared__ unsigned long long s_live;
  if (threadIdx.x < 64u) {
    bool live = false;
    if (threadIdx.x < kStampTiles) {
      const unsigned first = (bid * kStampTiles + threadIdx.x) * kPer;
#pragma unroll
      for (unsigned k = 0; k < kPer; ++k) {
        const unsigned tile = first + k;
        if (tile < n_tiles) {
          const unsigned stamp = S.tile_stamp[tile];
          live = live || tile == 0u || strips ||
                 (do_update && (stamp == P.scan_no || stamp == P.scan_no + 1u || stamp == ob_scan));  // see make_ctx
        }
      }
      if (!live) {
#pragma unroll
        for (unsigned k = 0; k < kPer; ++k)
          if (first + k < n_tiles) S.upd_part[first + k] = 0u;
      }
    }
    const unsigned long long m = __ballot(live);
    if (threadIdx.x == 0) s_live = m;
  }
  __syncthreads();
  unsigned long long live_mask = s_live;
  while (live_mask) {  // block-uniform
    const unsigned q = unsigned(__ffsll((long long)live_mask)) - 1u;
    live_mask &= live_mask - 1ull;
    update_body<POLICY, BLOCK>(P, G, st, L, all_layers, n_layers, S, px, py, pz, prgb, pvar, ncell,
                               bid * kStampTiles + q);
    __syncthreads();  // the body's shared counters are reused by the next live slot
  }
}

template <typename POLICY>
__global__ __launch_bounds__(256) void k_update_stamped(
    const ScanParams P, const GeomConst G, DevState* __restrict__ st,
    const typename POLICY::Layers L, float* const* __restrict__ all_layers, int n_layers,
    const Scratch S, const float* __restrict__ px, const float* __restrict__ py,
    const float* __restrict__ pz, const float* __restrict__ /*pint*/,
    const uint32_t* __restrict__ prgb, const float* __restrict__ pvar, unsigned ncell) {
  update_stamped_body<POLICY, 256>(P, G, st, L, all_layers, n_layers, S, px, py, pz, prgb, pvar, ncell, blockIdx.x);
}

// Scan statistics for the synchronous entry points (fdm_scan_stats): the per-block / per-tile
// partial counts are summed on the device and the result is stored straight into a host-mapped
// struct, so the host pays one stream sync and no copies (three blocking D2H copies before: ~50 us
// of a 136 us synchronous integrate at C2).
struct StatsAcc {  // device
  unsigned long long n_pass, n_in, n_touched, n_finite;
  unsigned done;
};
struct StatsOut {  // pinned host memory, written by the last block
  unsigned long long n_pass, n_in, n_touched, n_finite;
  int shr, shc;
  unsigned fault, pad;     // DevState::fault
  unsigned long long seq;  // written last (system scope): the host polls it instead of sleeping in a stream wait
  unsigned dense_paid;     // DevState::dense_paid as of the last k_obstacle_dense_paid (the host reads it without a sync)
  unsigned pad2;
};
inline __global__ __launch_bounds__(256) void k_collect_stats(const unsigned long long* __restrict__ bin_part,
                                                       unsigned n_bin, const uint32_t* __restrict__ upd_part,
                                                       unsigned n_tiles, const uint32_t* __restrict__ ingest_part,
                                                       unsigned n_ingest, const DevState* __restrict__ st, int slot,
                                                       StatsAcc* __restrict__ acc, StatsOut* __restrict__ out,
                                                       unsigned long long seq) {
  unsigned long long np = 0, ni = 0, nt = 0, nf = 0;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_ingest; i += gridDim.x * 256u) nf += ingest_part[i];
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_bin; i += gridDim.x * 256u) {
    const unsigned long long v = bin_part[i];
    np += uint32_t(v);
    ni += uint32_t(v >> 32);
  }
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n_tiles; i += gridDim.x * 256u) nt += upd_part[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    np += __shfl_down(np, d);
    ni += __shfl_down(ni, d);
    nt += __shfl_down(nt, d);
    nf += __shfl_down(nf, d);
  }
  __shared__ unsigned long long s_sum[4][4];
  __shared__ unsigned s_last;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_sum[wave][0] = np; s_sum[wave][1] = ni; s_sum[wave][2] = nt; s_sum[wave][3] = nf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc->n_pass, s_sum[0][0] + s_sum[1][0] + s_sum[2][0] + s_sum[3][0]);
    atomicAdd(&acc->n_in, s_sum[0][1] + s_sum[1][1] + s_sum[2][1] + s_sum[3][1]);
    atomicAdd(&acc->n_touched, s_sum[0][2] + s_sum[1][2] + s_sum[2][2] + s_sum[3][2]);
    if (n_ingest) atomicAdd(&acc->n_finite, s_sum[0][3] + s_sum[1][3] + s_sum[2][3] + s_sum[3][3]);
    __threadfence();
    s_last = atomicAdd(&acc->done, 1u) == gridDim.x - 1u ? 1u : 0u;
    if (s_last) {  // every block's sums are in: publish to the host and re-arm the accumulators
      __threadfence();
      out->n_pass = atomicExch(&acc->n_pass, 0ull);
      out->n_in = atomicExch(&acc->n_in, 0ull);
      out->n_touched = atomicExch(&acc->n_touched, 0ull);
      out->n_finite = atomicExch(&acc->n_finite, 0ull);
      out->shr = st->cand[slot].shr;
      out->shc = st->cand[slot].shc;
      out->fault = st->fault;
      acc->done = 0u;
      __threadfence_system();
      __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// One launch for two scans: blocks [0, upd_blocks) finish scan t (its update), the rest start scan
// t+1 (its bin).  The two halves share nothing — the scratch is double-buffered by scan parity and
// the chained bin derives its base geometry from slot t (ScanParams::chain_prev) — so a stream of
// small scans costs one launch and max(bin, update) per scan instead of two launches and their sum.
struct ScanInputs {
  const float *x, *y, *z, *intensity;
  const uint32_t* rgb;
  const float* var;
};
// STAMPED (stamp-gated maps) is a template parameter, not a branch on Su.dense: carrying both update bodies
// in one kernel cost the dense configs[2] launch 4 % (14.9 -> 15.4 us).
template <typename POLICY, bool HAS_INT, bool HAS_COL, int THREADS, bool STAMPED = false, int LEAN = 0>
__global__ __launch_bounds__(THREADS) void k_update_bin4(
    const ScanParams Pu, const GeomConst G, DevState* __restrict__ st, const typename POLICY::Layers L,
    float* const* __restrict__ all_layers, int n_layers, const Scratch Su, const ScanInputs Iu, unsigned ncell,
    unsigned upd_blocks, const ScanParams Pb, const Scratch Sb, const ScanInputs Ib,
    int32_t* __restrict__ cell_ids) {
  // the two kinds of block are interleaved in proportion over the grid, so that the bandwidth-bound
  // update half and the atomic/latency-bound bin half are resident together for the whole launch
  // (update blocks first, then bin blocks, only overlapped where one kind ran out)
  const unsigned long long total = gridDim.x;
  const unsigned u0 = unsigned((blockIdx.x * (unsigned long long)upd_blocks) / total);
  const unsigned u1 = unsigned(((blockIdx.x + 1ull) * (unsigned long long)upd_blocks) / total);
  if (u1 > u0) {
    if (!STAMPED)
      update_body<POLICY, THREADS>(Pu, G, st, L, all_layers, n_layers, Su, Iu.x, Iu.y, Iu.z, Iu.rgb, Iu.var, ncell, u0);
    else  // stamp-gated maps: an update block is kStampTiles slots
      update_stamped_body<POLICY, THREADS>(Pu, G, st, L, all_layers, n_layers, Su, Iu.x, Iu.y, Iu.z, Iu.rgb, Iu.var,
                                           ncell, u0);
  } else {
    bin4_body<HAS_INT, HAS_COL, THREADS, LEAN>(Pb, G, st, Ib.x, Ib.y, Ib.z, Ib.intensity, Sb, cell_ids,
                                         blockIdx.x - u0);
  }
}

template <typename POLICY, bool WAVE_MERGE, bool STAMPED = false, int CH = -1, int LEAN = 0>
__global__ __launch_bounds__(256) void k_update_bin(
    const ScanParams Pu, const GeomConst G, DevState* __restrict__ st, const typename POLICY::Layers L,
    float* const* __restrict__ all_layers, int n_layers, const Scratch Su, const ScanInputs Iu, unsigned ncell,
    unsigned upd_blocks, const ScanParams Pb, const Scratch Sb, const ScanInputs Ib,
    int32_t* __restrict__ cell_ids) {
  if (blockIdx.x < upd_blocks) {
    if (!STAMPED)
      update_body<POLICY>(Pu, G, st, L, all_layers, n_layers, Su, Iu.x, Iu.y, Iu.z, Iu.rgb, Iu.var, ncell,
                          blockIdx.x);
    else
      update_stamped_body<POLICY, 256>(Pu, G, st, L, all_layers, n_layers, Su, Iu.x, Iu.y, Iu.z, Iu.rgb, Iu.var,
                                       ncell, blockIdx.x);
  } else {
    bin_body<WAVE_MERGE, CH, LEAN>(Pb, G, st, Ib.x, Ib.y, Ib.z, Ib.intensity, Sb, cell_ids, blockIdx.x - upd_blocks);
  }
}

// The host wrote the obstacle layer (upload / add): the touched-cell lists no longer bound the
// non-NaN cells, so this scan falls back to the reference's whole-layer clear — still only if
// the scan observed a cell.
inline __global__ void k_obstacle_dense_clear(const ScanParams P, DevState* __restrict__ st,
                                       float* __restrict__ obstacle, size_t n) {
  if (st->flags[P.slot].any_inside == 0u || st->dense_paid == st->dense_owed) return;  // (nothing observed | nothing owed)
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (; i < n; i += stride) obstacle[i] = nanv;
}

// ... and behind it (one thread): the scan that observed a cell has paid the debt
// (the paid number is also left in the pinned statistics block: enqueue-only callers learn that the debt is gone
// without a stream wait — until then every scan is "not plain": no fused launch, no batch launch)
inline __global__ void k_obstacle_dense_paid(const ScanParams P, DevState* __restrict__ st, StatsOut* __restrict__ out) {
  if (st->flags[P.slot].any_inside != 0u) {
    st->dense_paid = st->dense_owed;
    __hip_atomic_store(&out->dense_paid, st->dense_owed, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
inline __global__ void k_obstacle_dense_owe(DevState* __restrict__ st, unsigned seq) { st->dense_owed = seq; }

// ---- small utility kernels (a layer is a strided view: stride 1 or the record size) ----
inline __global__ void k_fill(float* __restrict__ p, float v, size_t n, int es) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i * size_t(es)] = v;
}
// dst[i*ds] = src[i*ss]  (gather a record field into a contiguous array and back)
inline __global__ void k_copy_strided(float* __restrict__ dst, int ds, const float* __restrict__ src, int ss,
                               size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) dst[i * size_t(ds)] = src[i * size_t(ss)];
}
// two fields of the same records in one pass (a record's line is fetched once): uncertainty fusion's private copies
inline __global__ void k_copy_strided2(float* __restrict__ dst0, float* __restrict__ dst1, const float* __restrict__ src0,
                                       const float* __restrict__ src1, int ss, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) {
    const float a = src0[i * size_t(ss)], b = src1[i * size_t(ss)];
    dst0[i] = a;
    dst1[i] = b;
  }
}
inline __global__ void k_fill_aux(uint4* __restrict__ p, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = make_uint4(0u, 0u, kNoIdx, 0u);
}
inline __global__ void k_fill_u64(unsigned long long* __restrict__ p, unsigned long long v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
inline __global__ void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
// rectangle <-> contiguous buffer (halo exchange); thread = (row within rect), blockIdx.y = col
inline __global__ void k_region_copy(float* __restrict__ layer, int es, float* __restrict__ buf, int s_rows,
                              int r0, int c0, int nr, int nc, int to_buf) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (r >= nr || c >= nc) return;
  float* a = layer + (size_t(c0 + c) * s_rows + (r0 + r)) * size_t(es);
  float* b = buf + size_t(c) * nr + r;
  if (to_buf) *b = *a; else *a = *b;
}

// The same for SEVERAL rectangles and layers in one launch (a halo exchange packs <= 8 strips x every visible layer:
// one launch instead of one per strip and layer).  blockIdx.y = rectangle * n_layers + layer; the buffer holds the
// rectangles one after the other, each layer-major, each rectangle column-major — the layout of k_region_copy.
constexpr int kRegionRects = 8, kRegionLayers = 24;
struct RegionArgs {
  int r0[kRegionRects], c0[kRegionRects], nr[kRegionRects], nc[kRegionRects];
  unsigned long long off[kRegionRects];  // first float of the rectangle's block in the buffer
  float* layer[kRegionLayers];
  int es[kRegionLayers];
  int n_rects, n_layers, s_rows, to_buf;
};
inline __global__ void k_regions_copy(const RegionArgs A, float* __restrict__ buf) {
  const int q = blockIdx.y / A.n_layers, l = blockIdx.y % A.n_layers;
  const int nr = A.nr[q], nc = A.nc[q];
  const size_t cells = size_t(nr) * size_t(nc);
  float* const lay = A.layer[l];
  const size_t es = size_t(A.es[l]);
  float* const b0 = buf + A.off[q] + size_t(l) * cells;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < cells; i += size_t(gridDim.x) * blockDim.x) {
    const int c = int(i / size_t(nr)), r = int(i % size_t(nr));
    float* a = lay + (size_t(A.c0[q] + c) * A.s_rows + size_t(A.r0[q] + r)) * es;
    if (A.to_buf) b0[i] = *a; else *a = b0[i];
  }
}

}  // namespace fdm
