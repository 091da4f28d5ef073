// fdm_kernels.hpp — the two kernels of one scan (gfx950 / CDNA4, wave64).
//
//   k_bin4 / k_bin : per input point.  Fused preprocessScan (fastdem.cpp:164-190: T_base_sensor,
//       cropRange, cropZ, T_world_base) + LOCAL-mode move arithmetic + nanogrid getIndex +
//       per-cell reduction (elevation_mapping.cpp:41-92).  SoA channels are read with 16 B/lane
//       coalesced loads; same-cell points are merged in registers, then in a per-block LDS table
//       ("LDS-staged cell tile"), and only the block's UNIQUE cells go to the device-resident
//       scratch with one atomic set each.  A returning atomicMin tells the first toucher of a cell,
//       which appends it to the scan's touched-cell list.
//   k_update : per TOUCHED cell (grid-stride over the list) — never a dense pass over the map.
//       Per-cell Kalman / P2 update (elevation_mapping.cpp:94-108), min/max, obstacle, intensity,
//       colour (elevation_mapping.cpp:127-175); obstacle clear of the cells the previous updating
//       scan touched; GridMap::move strip clear; scratch reset; geometry commit.
//
// Measured facts this design rests on (MI355X, scripts/ubench/atomics.hip, profiles/):
//   * global atomics execute at the memory side (TCC_EA0_ATOMIC == TCC_ATOMIC): ~26 Gop/s for
//     scattered addresses, ~1 ns per op on one address, 7-12x faster when a wave's addresses are
//     consecutive.  Atomic COUNT, not bytes, bounds the bin kernel => merge on chip first.
//   * same-address counters serialise => statistics are per-block partials / one add per block.
//
// Roofline: HBM-bound work (no contraction => MFMA is irrelevant).  Algorithmic bytes
// (SURVEY.md §8d): 12 B per input point (+4 intensity, +4 colour); per touched cell 72 B (Kalman)
// / 124 B (P2); 4 B per map cell per scan for the reference's whole-layer obstacle clear (which
// this engine replaces by clearing only the cells that can be non-NaN).
#pragma once

#include "fdm_device.hpp"

namespace fdm {

// Device-resident per-cell scratch, double-buffered by scan parity so that k_update(t) reads
// buffer t&1 immutably while it resets the entries scan t-1 left in buffer (t-1)&1.
struct Scratch {
  unsigned long long* bin_part;  // [bin blocks] lo32 = n_after_filter, hi32 = n_in_map
  unsigned long long* key[2];    // (ord(z) << 32 | point index), min-reduced; kEmptyKey = untouched
  uint32_t* zmax[2];             // ord(max z), 0 = none
  uint32_t* imax[2];             // ord(max non-NaN intensity), 0 = none
  uint32_t* first[2];            // lowest point index in the cell (intensity NaN-first rule)
  uint32_t* last[2];             // highest point index in the cell (colour = last point wins)
  uint32_t* list[2];             // touched-cell lists (one being written, one = obstacle-dirty set)
};

struct KalmanLayers {
  float *elevation, *elevation_min, *elevation_max, *variance, *n_points, *kalman_p, *sample_mean,
      *sample_m2, *upper, *lower, *obstacle, *intensity, *color;
  float min_var, max_var, q;
};
struct P2Layers {
  float *elevation, *elevation_min, *elevation_max, *variance, *n_points, *upper, *lower, *obstacle,
      *intensity, *color;
  float* q[5];
  float* n[5];
  P2Params p;
};

// Bring one input point into the map frame; returns whether it survived the crops.
__device__ __forceinline__ bool preprocess_point(const ScanParams& P, float& x, float& y, float& z) {
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

// One cell's reduction goes to the scratch; returns true for the first toucher of the cell.
template <bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ bool scratch_merge(const Scratch& S, int b, uint32_t cell,
                                              unsigned long long key, uint32_t zmx, uint32_t imx,
                                              uint32_t fst, uint32_t lst) {
  const unsigned long long old = atomicMin(&S.key[b][cell], key);
  if (zmx) atomicMax(&S.zmax[b][cell], zmx);
  if (HAS_INT) {
    if (imx) atomicMax(&S.imax[b][cell], imx);
    atomicMin(&S.first[b][cell], fst);
  }
  if (HAS_COL) atomicMax(&S.last[b][cell], lst);
  return old == kEmptyKey;
}

__device__ __forceinline__ DevCand block_candidate(const ScanParams& P, const GeomConst& G,
                                                   DevState* __restrict__ st, DevCand* s_cand) {
  if (threadIdx.x == 0) {
    const DevGeom g = st->geom[P.slot];
    DevCand c;
    if (P.do_move) {
      c = move_candidate(g, G, P.robot_x, P.robot_y);
    } else {
      c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
    }
    *s_cand = c;
    if (blockIdx.x == 0) st->cand[P.slot] = c;
  }
  __syncthreads();
  return *s_cand;
}

// ---------------------------------------------------------------------------------------------
// k_bin4 — production bin kernel for large scans: 1024 consecutive points per 256-thread block,
// four CONSECUTIVE points per thread (dwordx4 loads).
constexpr int kHashSlots = 1024;  // == points per block: enough for every point in its own cell
constexpr uint32_t kEmptyCell = 0xFFFFFFFFu;

template <bool HAS_INT, bool HAS_COL>
__global__ __launch_bounds__(256) void k_bin4(const ScanParams P, const GeomConst G,
                                              DevState* __restrict__ st,
                                              const float* __restrict__ px,
                                              const float* __restrict__ py,
                                              const float* __restrict__ pz,
                                              const float* __restrict__ pint, const Scratch S,
                                              int32_t* __restrict__ cell_ids) {
  __shared__ unsigned long long h_key[kHashSlots];
  __shared__ uint32_t h_cell[kHashSlots];
  __shared__ uint32_t h_zmax[kHashSlots];
  __shared__ uint32_t h_imax[HAS_INT ? kHashSlots : 1];
  __shared__ uint32_t h_first[HAS_INT ? kHashSlots : 1];
  __shared__ uint32_t h_last[HAS_COL ? kHashSlots : 1];
  __shared__ DevCand s_cand;
  __shared__ unsigned s_cnt[4];
  __shared__ unsigned s_nfirst, s_base;

  for (int k = threadIdx.x; k < kHashSlots; k += 256) {
    h_key[k] = kEmptyKey;
    h_cell[k] = kEmptyCell;
    h_zmax[k] = 0u;
    if (HAS_INT) { h_imax[k] = 0u; h_first[k] = kNoIdx; }
    if (HAS_COL) h_last[k] = 0u;
  }
  if (threadIdx.x == 0) s_nfirst = 0u;
  const DevCand cand = block_candidate(P, G, st, &s_cand);  // contains the __syncthreads

  const unsigned i0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
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

  // phase 1: all four points through the arithmetic (independent chains -> ILP)
  int cells[4];
  unsigned n_pass = 0, n_in = 0;
  bool any_glob = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool live = i0 + j < P.n;
    const bool pass = preprocess_point(P, xs[j], ys[j], zs[j]) && live;
    cells[j] = pass ? owned_cell(xs[j], ys[j], cand, G) : -1;
    n_pass += pass ? 1u : 0u;
    n_in += cells[j] >= 0 ? 1u : 0u;
    any_glob = any_glob || (pass && cells[j] != -1);
    if (cell_ids && live) cell_ids[i0 + j] = cells[j] >= 0 ? cells[j] : (!pass ? -1 : (cells[j] == -2 ? -3 : -2));
  }

  // phase 2: merge runs of equal cell in registers, fold each run into the block's LDS table
  int run_cell = -1;
  unsigned long long run_key = kEmptyKey;
  uint32_t run_zmx = 0, run_imx = 0, run_fst = kNoIdx, run_lst = 0;
  auto fold_run = [&]() {
    if (run_cell < 0) return;
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
    const unsigned long long key = make_key(zs[j], i);
    const uint32_t zmx = make_zmax(zs[j]);
    uint32_t imx = 0;
    if (HAS_INT) imx = isnan(vs[j]) ? 0u : ord(vs[j]);
    if (cells[j] != run_cell) {
      fold_run();
      run_cell = cells[j];
      run_key = key;
      run_zmx = zmx;
      run_imx = imx;
      run_fst = i;
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

  // phase 3: one global atomic set per unique cell; collect the first touchers
  const int b = P.slot & 1;
  uint32_t mine[kHashSlots / 256];
  unsigned n_mine = 0;
#pragma unroll
  for (int q = 0; q < kHashSlots / 256; ++q) {
    const int k = threadIdx.x + q * 256;
    const uint32_t cell = h_cell[k];
    mine[q] = kEmptyCell;
    if (cell == kEmptyCell || P.dbg_no_atomics) continue;
    const bool first = scratch_merge<HAS_INT, HAS_COL>(S, b, cell, h_key[k], h_zmax[k],
                                                      HAS_INT ? h_imax[k] : 0u,
                                                      HAS_INT ? h_first[k] : 0u,
                                                      HAS_COL ? h_last[k] : 0u);
    if (first) { mine[q] = cell; ++n_mine; }
  }
  unsigned rank = 0;
  if (n_mine) rank = atomicAdd(&s_nfirst, n_mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    const unsigned np = tot & 0xFFFFu, ni = tot >> 16;
    if (np) st->flags[P.slot].any_pass = 1u;
    S.bin_part[blockIdx.x] = (unsigned long long)np | ((unsigned long long)ni << 32);
    s_base = s_nfirst ? atomicAdd(&st->n_list[P.slot], s_nfirst) : 0u;
  }
  __syncthreads();
  if (n_mine) {
    uint32_t* list = S.list[1 - st->obst[P.slot].buf];
    unsigned o = s_base + rank;
#pragma unroll
    for (int q = 0; q < kHashSlots / 256; ++q)
      if (mine[q] != kEmptyCell) list[o++] = mine[q];
  }
}

// ---------------------------------------------------------------------------------------------
// k_bin — one point per thread (small scans: latency matters more than atomic count, and
// unaligned channel pointers).  Same-cell runs of neighbouring lanes are merged inside the
// wavefront with a segmented scan; run tails go to the scratch.
template <bool WAVE_MERGE>
__global__ __launch_bounds__(256) void k_bin(const ScanParams P, const GeomConst G,
                                             DevState* __restrict__ st,
                                             const float* __restrict__ px,
                                             const float* __restrict__ py,
                                             const float* __restrict__ pz,
                                             const float* __restrict__ pint, const Scratch S,
                                             int32_t* __restrict__ cell_ids) {
  __shared__ DevCand s_cand;
  __shared__ unsigned s_pass[4], s_in[4];
  const DevCand cand = block_candidate(P, G, st, &s_cand);

  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  bool pass = false;
  int cell = -1;
  float z = 0.0f;
  if (i < P.n) {
    float x = px[i], y = py[i];
    z = pz[i];
    pass = preprocess_point(P, x, y, z);
    if (pass) cell = owned_cell(x, y, cand, G);
    if (cell_ids) cell_ids[i] = cell >= 0 ? cell : (!pass ? -1 : (cell == -2 ? -3 : -2));
  }
  const bool inside = cell >= 0;
  const bool glob = pass && cell != -1;

  unsigned long long key = kEmptyKey;
  uint32_t zmx = 0, imx = 0, fst = kNoIdx, lst = 0;
  if (inside) {
    key = make_key(z, i);
    zmx = make_zmax(z);
    if (P.has_intensity) {
      const float v = pint[i];
      imx = isnan(v) ? 0u : ord(v);
      fst = i;
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
  bool first = false;
  if (commit && !P.dbg_no_atomics) {
    const int b = P.slot & 1;
    if (P.has_intensity && P.has_color)
      first = scratch_merge<true, true>(S, b, cell, key, zmx, imx, fst, lst);
    else if (P.has_intensity)
      first = scratch_merge<true, false>(S, b, cell, key, zmx, imx, fst, lst);
    else if (P.has_color)
      first = scratch_merge<false, true>(S, b, cell, key, zmx, imx, fst, lst);
    else
      first = scratch_merge<false, false>(S, b, cell, key, zmx, imx, fst, lst);
  }
  // wave-aggregated append of the first touchers
  const unsigned long long mf = __ballot(first);
  if (mf) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mf) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&st->n_list[P.slot], unsigned(__popcll(mf)));
    base = __shfl(base, leader);
    if (first) {
      uint32_t* list = S.list[1 - st->obst[P.slot].buf];
      list[base + __popcll(mf & ((1ull << lane) - 1ull))] = uint32_t(cell);
    }
  }

  const unsigned long long mp = __ballot(pass), mi = __ballot(inside), mg = __ballot(glob);
  if ((threadIdx.x & 63) == 0) {
    s_pass[threadIdx.x >> 6] = __popcll(mp);
    s_in[threadIdx.x >> 6] = __popcll(mi);
    if (mg) st->flags[P.slot].any_inside = 1u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned np = s_pass[0] + s_pass[1] + s_pass[2] + s_pass[3];
    const unsigned ni = s_in[0] + s_in[1] + s_in[2] + s_in[3];
    if (np) st->flags[P.slot].any_pass = 1u;
    S.bin_part[blockIdx.x] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
}

// ---------------------------------------------------------------------------------------------
// k_update: grid-stride over work items
//   [0, n_cur)                      cells touched by this scan         -> estimator update
//   [n_cur, n_cur + n_obst)         cells of the last updating scan    -> obstacle clear + scratch reset
//   [.., .. + n_strip)              cells vacated by GridMap::move     -> NaN in every layer
struct UpdateCtx {
  bool applied, do_update, reset_prev;
  int cb;               // scratch buffer of this scan
  unsigned n_cur, n_obst, n_strip;
  const uint32_t* cur_list;
  const uint32_t* obst_list;
  DevGeom E;
  DevCand C;
  // strip decomposition
  bool clear_all;
  int row_idx, row_n, col_idx, col_n;
};

__device__ __forceinline__ void strip_range(int start, int sh, int size, int& idx, int& n) {
  n = sh > 0 ? sh : -sh;
  idx = sh > 0 ? start : start + sh;
  if (n) wrap_index(idx, size);
}

__device__ __forceinline__ UpdateCtx make_ctx(const ScanParams& P, const GeomConst& G,
                                              DevState* __restrict__ st, const Scratch& S) {
  UpdateCtx u;
  const int slot = P.slot;
  const bool any_pass = st->flags[slot].any_pass != 0u;
  u.do_update = st->flags[slot].any_inside != 0u;
  u.applied = P.do_move && (!P.gate_on_filter || any_pass);
  u.cb = slot & 1;
  const DevObst ob = st->obst[slot];
  u.n_cur = u.do_update ? st->n_list[slot] : 0u;
  u.n_obst = ob.n;
  u.cur_list = S.list[1 - ob.buf];
  u.obst_list = S.list[ob.buf];
  u.reset_prev = ob.scan + 1u == P.scan_no;  // its scratch entries are still dirty
  u.E = st->geom[slot];
  u.C = st->cand[slot];
  u.clear_all = false;
  u.row_idx = u.row_n = u.col_idx = u.col_n = 0;
  u.n_strip = 0;
  if (u.applied && (u.C.shr != 0 || u.C.shc != 0)) {
    strip_range(u.E.sr, u.C.shr, G.rows, u.row_idx, u.row_n);
    strip_range(u.E.sc, u.C.shc, G.cols, u.col_idx, u.col_n);
    if (u.row_n >= G.rows || u.col_n >= G.cols) {
      u.clear_all = true;
      u.n_strip = unsigned(G.rows) * unsigned(G.cols);
    } else {
      u.n_strip = unsigned(u.row_n) * unsigned(G.cols) + unsigned(u.col_n) * unsigned(G.rows);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int nxt = (slot + 1) & 3, nn = (slot + 2) & 3;
    DevGeom g = u.E;
    if (u.applied) { g.px = u.C.px; g.py = u.C.py; g.sr = u.C.sr; g.sc = u.C.sc; }
    st->geom[nxt] = g;
    DevObst o2 = ob;
    if (u.do_update) { o2.buf = 1 - ob.buf; o2.n = u.n_cur; o2.scan = P.scan_no; }
    st->obst[nxt] = o2;
    st->flags[nn].any_pass = 0u;
    st->flags[nn].any_inside = 0u;
    st->n_list[nn] = 0u;
    if (u.do_update) {
      unsigned f = 0;
      if (P.has_intensity) f |= 1u;
      if (P.has_color) f |= 2u;
      if (f) st->sticky |= f;
    }
  }
  return u;
}

struct CellObs {
  int o;
  bool in_strip;
  float min_z, min_z_var, max_z;
  uint32_t first, last, imax;
};

// decode the scan's CellObservation of a touched cell
__device__ __forceinline__ CellObs load_obs(const ScanParams& P, const GeomConst& G, const UpdateCtx& u,
                                            const Scratch& S, uint32_t cell,
                                            const float* __restrict__ px,
                                            const float* __restrict__ py,
                                            const float* __restrict__ pz,
                                            const float* __restrict__ pvar) {
  CellObs c;
  c.o = int(cell);
  c.in_strip = false;
  if (u.n_strip) {
    const int r = int(cell % unsigned(G.s_rows)) + G.s_r0;
    const int col = int(cell / unsigned(G.s_rows)) + G.s_c0;
    c.in_strip = in_cleared_strip(r, u.E.sr, u.C.shr, G.rows) || in_cleared_strip(col, u.E.sc, u.C.shc, G.cols);
  }
  const unsigned long long key = S.key[u.cb][cell];
  const uint32_t idx = uint32_t(key);
  c.min_z = kFltMax;
  c.min_z_var = 0.0f;
  if (idx != kNoIdx) {
    float x = px[idx], y = py[idx], z = pz[idx];
    if (P.has_var) {
      c.min_z_var = pvar[idx];
    } else if (P.integrate_mode) {
      c.min_z_var = sigma_z2(P, x, y, z);
    }
    preprocess_point(P, x, y, z);
    c.min_z = z;
  }
  const uint32_t zm = S.zmax[u.cb][cell];
  c.max_z = zm ? unord(zm) : -kFltMax;
  c.first = kNoIdx;
  c.last = 0u;
  c.imax = 0u;
  if (P.has_intensity) {
    c.first = S.first[u.cb][cell];
    c.imax = S.imax[u.cb][cell];
  }
  if (P.has_color) c.last = S.last[u.cb][cell];
  return c;
}

// updateMinMax / updateObstacle / updateIntensity / updateColor (elevation_mapping.cpp:127-175)
template <typename LAYERS>
__device__ __forceinline__ void cell_epilogue(const ScanParams& P, const LAYERS& L, const CellObs& cx,
                                              const float* __restrict__ pint,
                                              const uint32_t* __restrict__ prgb) {
  const int o = cx.o;
  const float nanv = __uint_as_float(0x7FC00000u);
  const float smin = cx.in_strip ? nanv : L.elevation_min[o];
  const float smax = cx.in_strip ? nanv : L.elevation_max[o];
  if (isnan(smin) || cx.min_z < smin) L.elevation_min[o] = cx.min_z;
  if (isnan(smax) || cx.max_z > smax) L.elevation_max[o] = cx.max_z;
  L.obstacle[o] = (cx.max_z > cx.min_z) ? cx.max_z : nanv;
  if (P.has_intensity) {
    const float vf = pint[cx.first];
    const float obs = isnan(vf) ? vf : unord(cx.imax);
    const float stored = cx.in_strip ? nanv : L.intensity[o];
    if (isnan(stored) || obs > stored) L.intensity[o] = obs;
  }
  if (P.has_color) reinterpret_cast<uint32_t*>(L.color)[o] = prgb[cx.last] & 0x00FFFFFFu;
}

// obstacle clear of a cell the previous updating scan touched + reset of its scratch entry
template <typename LAYERS>
__device__ __forceinline__ void obst_item(const UpdateCtx& u, const Scratch& S, const LAYERS& L,
                                          uint32_t cell) {
  if (u.reset_prev) {
    const int pb = 1 - u.cb;
    S.key[pb][cell] = kEmptyKey;
    S.zmax[pb][cell] = 0u;
    if (S.imax[pb]) { S.imax[pb][cell] = 0u; S.first[pb][cell] = kNoIdx; }
    if (S.last[pb]) S.last[pb][cell] = 0u;
  }
  // map_.clear(obstacle) runs only when this scan observed a cell (elevation_mapping.cpp:118-121)
  if (u.do_update && S.key[u.cb][cell] == kEmptyKey) L.obstacle[cell] = __uint_as_float(0x7FC00000u);
}

// a cell vacated by GridMap::move: NaN in EVERY layer unless this scan re-observes it
__device__ __forceinline__ void strip_item(const GeomConst& G, const UpdateCtx& u, const Scratch& S,
                                           float* const* __restrict__ all_layers, int n_layers,
                                           unsigned w) {
  int r, c;
  if (u.clear_all) {
    r = int(w % unsigned(G.rows));
    c = int(w / unsigned(G.rows));
  } else if (w < unsigned(u.row_n) * unsigned(G.cols)) {
    c = int(w / unsigned(u.row_n));
    r = u.row_idx + int(w % unsigned(u.row_n));
    if (r >= G.rows) r -= G.rows;
  } else {
    const unsigned v = w - unsigned(u.row_n) * unsigned(G.cols);
    c = u.col_idx + int(v / unsigned(G.rows));
    if (c >= G.cols) c -= G.cols;
    r = int(v % unsigned(G.rows));
  }
  const size_t o = size_t(c) * G.rows + r;  // moves exist only for untiled engines
  if (u.do_update && S.key[u.cb][o] != kEmptyKey) return;  // its cur-list item rewrites it
  const float nanv = __uint_as_float(0x7FC00000u);
  for (int l = 0; l < n_layers; ++l) all_layers[l][o] = nanv;
}

__global__ __launch_bounds__(256) void k_update_kalman(
    const ScanParams P, const GeomConst G, DevState* __restrict__ st, const KalmanLayers L,
    float* const* __restrict__ all_layers, int n_layers, const Scratch S,
    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
    const float* __restrict__ pint, const uint32_t* __restrict__ prgb,
    const float* __restrict__ pvar) {
  const UpdateCtx u = make_ctx(P, G, st, S);
  const unsigned total = u.n_cur + u.n_obst + u.n_strip;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (unsigned w = blockIdx.x * 256u + threadIdx.x; w < total; w += gridDim.x * 256u) {
    if (w < u.n_cur) {
      const CellObs cx = load_obs(P, G, u, S, u.cur_list[w], px, py, pz, pvar);
      const int o = cx.o;
      KalmanState s;
      if (cx.in_strip) {
        for (int l = 0; l < n_layers; ++l) all_layers[l][o] = nanv;
        s.x = s.P = s.count = s.mean = s.var = s.m2 = nanv;
      } else {
        s.x = L.elevation[o];
        s.P = L.kalman_p[o];
        s.count = L.n_points[o];
        s.mean = L.sample_mean[o];
        s.var = L.variance[o];
        s.m2 = L.sample_m2[o];
      }
      kalman_step(s, cx.min_z, cx.min_z_var, L.min_var, L.max_var, L.q);
      L.elevation[o] = s.x;
      L.kalman_p[o] = s.P;
      L.n_points[o] = s.count;
      L.sample_mean[o] = s.mean;
      L.variance[o] = s.var;
      L.sample_m2[o] = s.m2;
      L.upper[o] = s.upper;
      L.lower[o] = s.lower;
      cell_epilogue(P, L, cx, pint, prgb);
    } else if (w < u.n_cur + u.n_obst) {
      obst_item(u, S, L, u.obst_list[w - u.n_cur]);
    } else {
      strip_item(G, u, S, all_layers, n_layers, w - u.n_cur - u.n_obst);
    }
  }
}

__global__ __launch_bounds__(256) void k_update_p2(
    const ScanParams P, const GeomConst G, DevState* __restrict__ st, const P2Layers L,
    float* const* __restrict__ all_layers, int n_layers, const Scratch S,
    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
    const float* __restrict__ pint, const uint32_t* __restrict__ prgb,
    const float* __restrict__ pvar) {
  const UpdateCtx u = make_ctx(P, G, st, S);
  const unsigned total = u.n_cur + u.n_obst + u.n_strip;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (unsigned w = blockIdx.x * 256u + threadIdx.x; w < total; w += gridDim.x * 256u) {
    if (w < u.n_cur) {
      const CellObs cx = load_obs(P, G, u, S, u.cur_list[w], px, py, pz, pvar);
      const int o = cx.o;
      P2State s;
      if (cx.in_strip) {
        for (int l = 0; l < n_layers; ++l) all_layers[l][o] = nanv;
        s.count = nanv;
#pragma unroll
        for (int k = 0; k < 5; ++k) s.q[k] = s.n[k] = nanv;
      } else {
        s.count = L.n_points[o];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          s.q[k] = L.q[k][o];
          s.n[k] = L.n[k][o];
        }
      }
      p2_step(s, cx.min_z, L.p);
      L.n_points[o] = s.count;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        L.q[k][o] = s.q[k];
        L.n[k][o] = s.n[k];
      }
      L.elevation[o] = s.elevation;
      L.variance[o] = s.variance;
      L.upper[o] = s.upper;
      L.lower[o] = s.lower;
      cell_epilogue(P, L, cx, pint, prgb);
    } else if (w < u.n_cur + u.n_obst) {
      obst_item(u, S, L, u.obst_list[w - u.n_cur]);
    } else {
      strip_item(G, u, S, all_layers, n_layers, w - u.n_cur - u.n_obst);
    }
  }
}

// The host wrote the obstacle layer (upload / add): the touched-cell lists no longer bound the
// non-NaN cells, so this scan falls back to the reference's whole-layer clear — still only if
// the scan observed a cell.
__global__ void k_obstacle_dense_clear(const ScanParams P, DevState* __restrict__ st,
                                       float* __restrict__ obstacle, size_t n) {
  if (st->flags[P.slot].any_inside == 0u) return;
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (; i < n; i += stride) obstacle[i] = nanv;
}

// ---- small utility kernels ----
__global__ void k_fill(float* __restrict__ p, float v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ void k_fill_u64(unsigned long long* __restrict__ p, unsigned long long v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
// rectangle <-> contiguous buffer (halo exchange); thread = (row within rect), blockIdx.y = col
__global__ void k_region_copy(float* __restrict__ layer, float* __restrict__ buf, int s_rows, int r0,
                              int c0, int nr, int nc, int to_buf) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (r >= nr || c >= nc) return;
  float* a = layer + size_t(c0 + c) * s_rows + (r0 + r);
  float* b = buf + size_t(c) * nr + r;
  if (to_buf) *b = *a; else *a = *b;
}

}  // namespace fdm
