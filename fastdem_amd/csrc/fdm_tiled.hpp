// fdm_tiled.hpp — the large-scan pipeline: per-tile observation buckets instead of per-cell atomics.
//
// Why (profiles/r01): the first large-scan pipeline reduced every cell through memory-side atomics (k_bin4 -> key / aux
// scratch -> k_update).  On configs[3] that was ~240 K (block, cell) pairs x 4 atomics ~ 1 M fabric operations per scan
// at ~26 Gop/s = the whole 39 us of the bin kernel, plus a dense key sweep and a point gather in the update kernel.
// Nothing in this pipeline reduces through memory:
//
//   bin half (fdm_tbin2.hpp): a block of 1 024 consecutive points runs preprocessScan + getIndex, folds same-cell
//       points in a per-block LDS table, SORTS its unique cells by 16x16-cell map tile and writes them as observation
//       records — plain coalesced stores into the block's own region of a record pool.  Per (block, tile) ONE
//       returning atomic on the tile's counter appends a chunk descriptor {first record, count} to the tile's row.  A
//       record carries everything the update needs (min z + its sigma_z^2, max z, intensity, colour): the caller's
//       arrays are dead as soon as the bin half has run.
//   update half (this file): ONE WAVEFRONT per tile, no workgroup barrier anywhere.  A wavefront reads the tile's
//       chunk list (the first 64 descriptors speculatively, in the same round trip as the count), folds the records
//       into a 256-cell LDS image (LDS atomics), compacts the touched cells and gives each ONE lane: estimator step,
//       min / max, obstacle, intensity, colour, move() strips.  Untouched cells only ever need stores.  The update
//       wavefronts of a launch are PERSISTENT: a launch carries a few hundred update blocks (four independent
//       wavefronts each) that walk their tiles one after the other, so that in the fused launch (update of scan t |
//       bin of scan t + 1) the arithmetic-bound bin blocks own most of the chip from the first microsecond — round
//       2-4's one-256-thread-block-per-32x32-tile design held 94 % of the block slots with three dependent round trips
//       each for the first 10 us of every launch (LABNOTES "Timeline of the fused launch").
//
// Records are two arrays: a HOT 16-byte word per record (ord(min z), ord(max z), ord(max intensity), cell in tile |
// flags) — all the fold needs, one 16-byte load — and a COLD 8-byte word (sigma_z^2 of the min-z point, colour) that
// only the cell's winning record is asked for.
//
// Order rules (elevation_mapping.cpp:62-92) carried through both levels without a point index:
//   * records of one cell are ordered by pool position = (bin block, ...) = scan order of their blocks, and each
//     block contributes at most one record per cell, so "first point wins a tie" is "lowest position wins";
//   * the minimum is ONE 64-bit reduction word  ord(z) << 32 | pos << 1 | (z is -0): the lowest z, among equals the
//     first point — "strict z < min_z, first point wins".  -0 and +0 compare equal in the reference, so every value
//     is reduced with zeros canonicalised; for the maxima the sign of the FIRST zero-valued point of a cell rides along
//     in a separate min-reduced word (rare path).  min_z / max_z / intensity come out bit-identical to the reference.
//
// Algorithmic bytes are unchanged: 12 B/point (+4 intensity, +4 colour), 72 / 124 B per touched cell,
// 4 B per map cell per scan for the obstacle clear.
#pragma once

#include "fdm_kernels.hpp"

namespace fdm {

constexpr int kTS = 16;                 // tile height: 16 consecutive rows of a column = 64 B of every layer
constexpr int kTSShift = 4;
constexpr int kTCShift = 4;             // tile width: 16 columns
constexpr int kTC = 1 << kTCShift;
constexpr unsigned kTileCells = unsigned(kTS * kTC);  // 256: one wavefront holds a tile's image in 4-5 KB of LDS
constexpr int kCitBits = 8;             // cell-in-tile bits of tile << 8 | cell
constexpr uint32_t kCitMask = (1u << kCitBits) - 1u;
constexpr uint32_t kNoWinner = 0xFFFFFFFFu;
constexpr uint32_t kOrdZero = 0x80000000u;  // ord(+0.0f)
#ifndef FDM_UPD_WAVES
#define FDM_UPD_WAVES 6  // min waves per SIMD the large-scan kernels are compiled for (<= 80 VGPRs): six blocks per CU are what the launch runs with (tiled_lds_pad)
#endif

// flags beside the cell-in-tile number (8 bits) of a record
constexpr uint32_t kRecMinNeg = 1u << 8;     // the record's min z is -0
constexpr uint32_t kRecNoWin = 1u << 9;      // no point of the block's cell has a z below FLT_MAX (the min stays FLT_MAX)
// (rare: set only by a block that met a -0.0 or a NaN intensity; "the cell holds a zero" needs no flag — a
// zero only matters when it is the cell's maximum, and then the record's zmax / imax IS ord(0))
constexpr uint32_t kRecNanFirst = 1u << 10;  // the block's first point in the cell has a NaN intensity
constexpr uint32_t kRecZNeg = 1u << 12;      // the block's first zero-valued z in the cell is -0
constexpr uint32_t kRecINeg = 1u << 14;      // the same for the intensity
constexpr uint32_t kRecRare = kRecNanFirst | kRecZNeg | kRecINeg;

// One observation record: what a bin block knows about one cell.
struct __align__(16) RecHot {
  uint32_t zmin;   // ord(min z) (ord(FLT_MAX) with kRecNoWin)
  uint32_t zmax;   // ord(max z), 0 = none
  uint32_t imax;   // ord(max intensity), 0 = none
  uint32_t cell;   // cell inside the tile | kRec* flags
};
struct __align__(8) RecCold {
  float var;       // sigma_z^2 of the min-z point
  uint32_t rgb;    // colour of the block's last point in the cell
};
// The record pool of one scan parity (bin of scan t+1 runs beside the update of scan t).
struct TilePool {
  RecHot* hot;                // [cap]
  RecCold* cold;              // [cap]
  unsigned* cnt;              // [n_tiles << cnt_shift] chunks of the tile (put back to 0 by the update); one counter per
                              // 2^cnt_shift words: neighbouring tiles are hit by the same blocks at the same time, and
                              // memory-side atomics on one line queue up
  unsigned cnt_shift;
  unsigned long long* desc;   // [n_tiles][stride] one word per chunk: first record | count << 32
  unsigned stride;            // >= bin blocks of the scan: a block appends at most one chunk per tile
  uint32_t* rare;             // [update wavefronts][3][256] scratch of the update's rare path (first-occurrence words)
};

struct TileGrid {
  int tiles_r, tiles_c;  // tiles over the stored window (rows, cols)
  unsigned n_tiles;
};

struct TileAux {           // what the update kernel keeps per tile between scans
  uint32_t* stamp;         // [n_tiles] last scan that touched a cell of the tile
  uint32_t* upd_part;      // [n_tiles] cells touched by the last scan (statistics)
  float* ras_z;            // [ncell] optional capture (onScanRasterized), NaN-filled by the host
  unsigned long long* timeline;  // measurement only (nullable): per block of a fused launch {start, end} in 100 MHz ticks
};

// Which tiles an update wavefront walks: wavefront w of W, k-th tile = k * W + w — consecutive tiles go to consecutive
// wavefronts.  The live tiles of a scan are neighbours (on a large GLOBAL map a patch of a few thousand among a quarter
// of a million): strided, they spread over all wavefronts.  (Runs of 8 consecutive tiles per wavefront on such a map —
// the tile counters lie one per 128 B, a run coalesces nothing — put configs[4]'s 5 600 live tiles on 700 of the 3 072
// wavefronts: 176 us per launch against 37.)
struct TileWork {
  unsigned W;          // update wavefronts of the launch
  unsigned T;          // tiles per wavefront
  unsigned prio;       // 1: the update wavefronts raise their issue priority (option "upd_prio")
  unsigned stagger;    // fused launch: start delay of the first-round bin blocks, in units of 512 cycles per resident slot (option "bin_stagger")
  unsigned delay;      // fused launch: the first `delay_blocks` bin blocks wait this many units of 512 cycles before their first load
  unsigned delay_blocks;  //            (option "bin_delay": the update wavefronts' first round trip goes ahead of 12 MB of point loads)
};

// value of a canonicalised ord word; `neg`: the first zero seen was -0 (only looked at for a zero)
__device__ __forceinline__ float signed_value(uint32_t ordv, uint32_t neg) {
  return (ordv == kOrdZero && neg) ? -0.0f : unord(ordv);
}
// ord with -0 folded onto +0 (they tie in every comparison of the reference)
__device__ __forceinline__ uint32_t ord_canon(float v) { return ord(v == 0.0f ? 0.0f : v); }

// point -> (tile << 8 | cell in tile) of this engine's owned window; -1 outside the (global) map,
// -2 inside the map but owned by another engine tile (see owned_cell).  lin = storage-linear id.
__device__ __forceinline__ int owned_tcell(float x, float y, const DevCand& cand, const GeomConst& G,
                                           const TileGrid& TG, int& lin) {
  int r, c;
  lin = -1;
  if (!cell_of(x, y, cand, G, r, c)) return -1;
  const int lr = r - G.o_r0, lc = c - G.o_c0;
  lin = -2;
  if (lr < 0 || lc < 0 || lr >= G.o_rows || lc >= G.o_cols) return -2;
  const int sr = r - G.s_r0, sc = c - G.s_c0;
  lin = sc * G.s_rows + sr;
  const int tile = (sc >> kTCShift) * TG.tiles_r + (sr >> kTSShift);
  return (tile << kCitBits) | ((sc & (kTC - 1)) << kTSShift) | (sr & (kTS - 1));
}

// set bits of `m` below this lane
__device__ __forceinline__ unsigned lane_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
}
// inclusive wave64 prefix sum / prefix maximum, six DPP steps (row_shr 1 2 4 8 inside the rows of 16, then the row
// totals): 24 issue cycles against six ds_bpermute shuffles at 24 each (scripts/ubench/valu_issue2.hip)
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));  // row_shr:1
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));  // row_shr:2
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));  // row_shr:4
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));  // row_shr:8
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ unsigned wave_scan_max(unsigned v) {
  v = max(v, unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false)));
  v = max(v, unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false)));
  v = max(v, unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false)));
  v = max(v, unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false)));
  v = max(v, unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false)));
  v = max(v, unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false)));
  return v;
}
// LDS that only ONE wavefront touches needs no s_barrier: its DS instructions execute in order.  This keeps the
// compiler from moving accesses of different lanes across the point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Kernel arguments are scalar loads the compiler hoists to the kernel's entry and keeps live to their LAST use: with
// the transforms read in phase 1 and again in the flush (sigma_z^2) and the rare walk, a hundred scalar registers were
// live through the whole body and spilled to vector lanes (183 spills, 800 v_readlane in the first build of fdm_tbin2.hpp).
// late(ref, z) with an opaque zero `z` re-reads the argument where it is used: the early copy dies with phase 1.
__device__ __forceinline__ unsigned opaque_zero() {
  unsigned z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}
#ifndef FDM_LATE_MASK
#define FDM_LATE_MASK 1  // which sites re-read: 1 rare walk, 2 flush, 4 index, 8 update fold, 16 update cells, 32 update pass (A/B: LABNOTES round 5)
#endif
template <int SITE, class T>
__device__ __forceinline__ const T& late(const T& ref, unsigned z) {
  if constexpr ((FDM_LATE_MASK & SITE) != 0) return (&ref)[z];
  else return ref;
}

// ---------------------------------------------------------------------------------------------
// The bin half.  Dynamic LDS: 1 024-slot arrays cell u32 | key u64 | zmax u32 [| imax u32] [| last u32], then the
// record list (u16): 18-26 B per point.
__host__ __device__ constexpr unsigned tbin_lds_bytes(bool has_int, bool has_col, unsigned threads) {
  return threads * 4u * (16u + (has_int ? 4u : 0u) + (has_col ? 4u : 0u) + 2u);
}

// Loads + phase 1 of one block: the four points of this thread through T_base_sensor, the crops,
// T_world_base and getIndex.  cells[j] = tile << 10 | cell in tile of an owned cell, -1 outside the map
// (or dropped by the crops), -2 inside the map but owned by another engine tile.  zs[] = map-frame z,
// vs[] = intensity.  SIDE: also the captures, cell ids and statistics (the block's first walk only).
// tbin_prep: the part that needs no map geometry (T_base_sensor, the crops, T_world_base, captures) — it runs while
// thread 0 waits for the state it chains the geometry from.  tbin_points: getIndex + tile of the surviving points.
template <bool HAS_INT, int THREADS, bool LEAN, bool SIDE, class PT>
__device__ __forceinline__ void tbin_prep(const PT& P, const Scratch& S, const unsigned bid,
                                          const float (&xin)[4], const float (&yin)[4], const float (&zin)[4],
                                          float (&xs)[4], float (&ys)[4], float (&zs)[4], bool (&pass)[4],
                                          unsigned& n_pass) {
  float* const cap_x = (LEAN || !SIDE) ? nullptr : S.cap_x;
  float* const cap_var = (LEAN || !SIDE) ? nullptr : S.cap_var;
  const bool drop_nf = LEAN ? false : P.drop_nonfinite != 0;
  const unsigned i0 = bid * unsigned(THREADS * 4) + threadIdx.x * 4u;
  // Branch-lean on purpose (the first version spent as many issue slots on exec-mask bookkeeping as on
  // arithmetic): the transforms and the fixed-point index estimate run for all four points without a branch;
  // the reference's exact index arithmetic is one shared, rarely taken branch for the lanes whose estimate
  // sits within 2^-shift cell of a cell edge.
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xs[j] = xin[j]; ys[j] = yin[j]; zs[j] = zin[j];
    const bool live = i0 + j < P.n;
    float cvar = 0.f;
    if (cap_var && live && P.integrate_mode) cvar = sigma_z2(P, xs[j], ys[j], zs[j]);
    if (!LEAN && SIDE && cap_var && S.cap_cov && live && P.integrate_mode) {
      float c9[9];
      cov_full(P, xs[j], ys[j], zs[j], c9);
#pragma unroll
      for (int k = 0; k < 9; ++k) S.cap_cov[size_t(k) * S.cap_stride + i0 + j] = c9[k];
    }
    const bool exists = live && (!drop_nf || (isfinite(xs[j]) && isfinite(ys[j]) && isfinite(zs[j])));
    pass[j] = preprocess_point(P, xs[j], ys[j], zs[j]) && exists;
    if (cap_var && live) cap_var[i0 + j] = cvar;
    n_pass += pass[j] ? 1u : 0u;
  }
  // the preprocessed cloud (scan callbacks, the raycasting stage's input): the thread's four points per channel in ONE
  // 16-byte store (the channels start on 16-byte boundaries: the engine rounds their pitch) — as four 4-byte stores per
  // channel the capture was a quarter of the launch's write requests
  if (cap_x) {
    float cx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cx[j] = (S.cap_drop_nan && !pass[j]) ? __uint_as_float(0x7FC00000u) : xs[j];
    if (i0 + 3u < P.n) {
      *reinterpret_cast<float4*>(cap_x + i0) = make_float4(cx[0], cx[1], cx[2], cx[3]);
      *reinterpret_cast<float4*>(S.cap_y + i0) = make_float4(ys[0], ys[1], ys[2], ys[3]);
      *reinterpret_cast<float4*>(S.cap_z + i0) = make_float4(zs[0], zs[1], zs[2], zs[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i0 + unsigned(j) < P.n) { cap_x[i0 + j] = cx[j]; S.cap_y[i0 + j] = ys[j]; S.cap_z[i0 + j] = zs[j]; }
    }
  }
}
template <bool HAS_INT, int THREADS, bool LEAN, bool SIDE, class PT>
__device__ __forceinline__ void tbin_points(const PT& P, const GeomConst& G, const TileGrid& TG,
                                            int32_t* __restrict__ cell_ids, const DevCand& cand, const unsigned bid,
                                            const float (&xs)[4], const float (&ys)[4], const bool (&pass)[4],
                                            int (&cells)[4], unsigned& n_in, bool& any_glob) {
  int32_t* const ids = (LEAN || !SIDE) ? nullptr : cell_ids;
  const unsigned i0 = bid * unsigned(THREADS * 4) + threadIdx.x * 4u;
  const bool any_start = cand.sr != 0 || cand.sc != 0;
  int kr[4], kc[4];
  bool sure_r[4], sure_c[4], inside[4];
  bool unsure = false;
  const double off_r = (G.half_x + cand.px) * G.inv_res_k, off_c = (G.half_y + cand.py) * G.inv_res_k;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    kr[j] = axis_fast(double(xs[j]), off_r, G.inv_res_k, G.idx_shift, G.rows, sure_r[j]);
    kc[j] = axis_fast(double(ys[j]), off_c, G.inv_res_k, G.idx_shift, G.cols, sure_c[j]);
    inside[j] = pass[j];
    unsure = unsure || (pass[j] && !(sure_r[j] && sure_c[j]));
  }
  if (__ballot(unsure)) {  // wave-uniform; well under a percent of the wavefronts
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (pass[j] && !sure_r[j]) inside[j] = axis_exact(double(xs[j]), cand.px, G.half_x, G.len_x, G.res, kr[j]);
      if (inside[j] && !sure_c[j]) inside[j] = axis_exact(double(ys[j]), cand.py, G.half_y, G.len_y, G.res, kc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int r = kr[j], c = kc[j];
    const bool okr = axis_wrap(r, cand.sr, any_start, G.rows);  // (both axes are evaluated, as getIndex does)
    const bool okc = axis_wrap(c, cand.sc, any_start, G.cols);
    const bool in_map = inside[j] && okr && okc;
    const int lr = r - G.o_r0, lc = c - G.o_c0;
    const bool owned = in_map && unsigned(lr) < unsigned(G.o_rows) && unsigned(lc) < unsigned(G.o_cols);
    const int sr = r - G.s_r0, sc = c - G.s_c0;
    const int tile = (sc >> kTCShift) * TG.tiles_r + (sr >> kTSShift);
    const int tcell = (tile << kCitBits) | ((sc & (kTC - 1)) << kTSShift) | (sr & (kTS - 1));
    cells[j] = owned ? tcell : (in_map ? -2 : -1);
    n_in += owned ? 1u : 0u;
    any_glob = any_glob || in_map;
    if (ids && i0 + j < P.n)
      ids[i0 + j] = owned ? sc * G.s_rows + sr : (!pass[j] ? -1 : (in_map ? -3 : -2));
  }
}

// Where a bin block gets the post-move geometry from and where its scan-wide flags go.  One scan per launch: the
// DevState ring (candidate_begin / candidate_finish).
struct TbinRing {
  const ScanParams& P;
  DevState* st;
  CandState cs;
  __device__ __forceinline__ TbinRing(const ScanParams& p, DevState* s) : P(p), st(s) {}
  __device__ __forceinline__ void begin() { candidate_begin(P, st, cs); }
  // (n_pass: this thread's points that survived the crops — the batch hook publishes "some point passed" here)
  __device__ __forceinline__ DevCand finish(const GeomConst& G, DevCand* s_cand, unsigned bid, unsigned /*n_pass*/) {
    return candidate_finish(P, G, st, cs, s_cand, bid);
  }
  __device__ __forceinline__ void note_inside() { st->flags[P.slot].any_inside = 1u; }
  __device__ __forceinline__ void note_pass() { st->flags[P.slot].any_pass = 1u; }
};


}  // namespace fdm
#include "fdm_tbin2.hpp"  // tbin2_body: the bin half
namespace fdm {

template <bool HAS_INT, bool HAS_COL, int THREADS, bool LEAN>
__global__ __launch_bounds__(THREADS, FDM_UPD_WAVES) void k_tbin(const ScanParams P, const GeomConst G, const TileGrid TG,
                                                                 DevState* __restrict__ st, const ScanInputs I,
                                                                 const Scratch S, const TilePool Q,
                                                                 int32_t* __restrict__ cell_ids) {
  static_assert(THREADS == 256, "bin blocks are 256 threads");
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  TbinRing H(P, st);
  tbin2_body<HAS_INT, HAS_COL, LEAN>(P, G, TG, H, I, S, S.bin_part, Q, cell_ids, dyn_lds, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// The update half.
struct TileCtx {  // (integers only: the geometry's doubles are the committer's business, not a register of every wavefront)
  bool applied, do_update, strips;
  unsigned ob_scan;
  int e_sr, e_sc;  // start index BEFORE the scan's move
  int shr, shc;    // the move's index shift
};

__device__ __forceinline__ void make_tile_ctx(const ScanParams& P, DevState* __restrict__ st, TileCtx& u,
                                              bool committer) {
  const int slot = P.slot;
  const bool any_pass = st->flags[slot].any_pass != 0u;
  u.do_update = st->flags[slot].any_inside != 0u;
  u.applied = P.do_move && (!P.gate_on_filter || any_pass);
  u.ob_scan = st->obst[slot].scan;
  u.e_sr = st->geom[slot].sr;
  u.e_sc = st->geom[slot].sc;
  u.shr = st->cand[slot].shr;
  u.shc = st->cand[slot].shc;
  if (committer) {  // commit geometry + ring bookkeeping (see make_ctx)
    const int nxt = (slot + 1) & 3, nn = (slot + 2) & 3;
    DevGeom g = st->geom[slot];
    const DevCand c = st->cand[slot];
    if (u.applied) { g.px = c.px; g.py = c.py; g.sr = c.sr; g.sc = c.sc; }
    st->geom[nxt] = g;
    st->obst[nxt].scan = u.do_update ? P.scan_no : u.ob_scan;
    st->flags[nn].any_pass = 0u;
    st->flags[nn].any_inside = 0u;
    st->flags[nn].ray_any = 0u;
    if (u.do_update) {
      if (P.has_intensity && st->vis_int == 0u) st->vis_int = 3u * P.scan_no + 2u;
      if (P.has_color && st->vis_col == 0u) st->vis_col = 3u * P.scan_no + 2u;
    }
  }
  u.strips = u.applied && (u.shr != 0 || u.shc != 0);
}

// Does the run of buffer indices [b0, b0 + len) on one axis meet the strip GridMap::move vacates there?
// (in_cleared_strip for a whole tile edge: the strip is [index, index + n) modulo size)
__device__ __forceinline__ bool span_hits_strip(int b0, int len, int start, int sh, int size) {
  if (sh == 0) return false;
  const int n = sh > 0 ? sh : -sh;
  if (n >= size) return true;
  int index = sh > 0 ? start : start + sh;
  wrap_index(index, size);
  int d = b0 - index;  // distance of the span's first index behind the strip's first index, modulo size
  if (d < 0) d += size;
  return d < n || d + len > size;  // starts inside the strip, or wraps around into its start
}
__device__ __forceinline__ bool tile_hits_strips(const TileCtx& u, const GeomConst& G, const TileGrid& TG, unsigned tile) {
  const int tr = int(tile % unsigned(TG.tiles_r)), tc = int(tile / unsigned(TG.tiles_r));
  return span_hits_strip(tr * kTS + G.s_r0, kTS, u.e_sr, u.shr, G.rows) ||
         span_hits_strip(tc * kTC + G.s_c0, kTC, u.e_sc, u.shc, G.cols);
}


// LDS of ONE update wavefront: key u64[256] | zmax u32[256] | chunk first-record u32[64] | chunk offset u32[64] |
// chunk-of-record u32[128] | touched list u8[256] | obstacle value f32[256] [| imax u32[256]] [| last u32[256]]
__host__ __device__ constexpr unsigned tile_wave_lds_bytes(bool has_int, bool has_col) {
  return 2048u + 1024u + 256u + 256u + 512u + 256u + 1024u + (has_int ? 1024u : 0u) + (has_col ? 1024u : 0u);
}
// ... of a 256-thread update block (four independent wavefronts)
__host__ __device__ constexpr unsigned tile_lds_bytes(bool has_int, bool has_col) {
  return 4u * tile_wave_lds_bytes(has_int, has_col);
}

// One tile by one wavefront.  Everything here is wave-uniform control flow around wave-private LDS.
template <typename POLICY, bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ void tupdate_tile(
    const ScanParams& P0, const GeomConst& G0, const TileCtx& u,
    const typename POLICY::Layers& L0, float* const* __restrict__ all_layers, int n_layers,
    const TilePool& Q0, const TileAux& A0, unsigned char* lds, const unsigned tile, const unsigned tr, const unsigned tc,
    const unsigned n_chunks, const bool obst_tile, const bool strips, const unsigned lane, const unsigned rare_slot) {
  const float nanv = __uint_as_float(0x7FC00000u);
  // (kernel arguments re-read per tile, where they are used: see late() in fdm_tbin2.hpp)
  const unsigned lz = opaque_zero();
  const TilePool& Q = late<8>(Q0, lz);
  constexpr bool has_int = HAS_INT, has_col = HAS_COL;
  unsigned long long* const s_key = reinterpret_cast<unsigned long long*>(lds);  // min of the records' keys
  uint32_t* const s_zmax = reinterpret_cast<uint32_t*>(s_key + kTileCells);
  uint32_t* const s_dpos = s_zmax + kTileCells;   // [64] first record of chunk c of the current batch
  uint32_t* const s_doff = s_dpos + 64;           // [64] ... and where its records start in the batch's record sequence
  uint32_t* const s_own = s_doff + 64;            // [128] (chunk + 1) at the first record of a chunk of the current window, 0 elsewhere
  uint8_t* const s_tl = reinterpret_cast<uint8_t*>(s_own + 128);  // [256] touched cells, compacted
  float* const s_obst = reinterpret_cast<float*>(s_tl + 256);     // [256] the touched cells' obstacle values (stored densely at the end)
  uint32_t* const s_imax = reinterpret_cast<uint32_t*>(s_obst + kTileCells);
  uint32_t* const s_last = s_imax + (has_int ? kTileCells : 0u);
  // rare-event words of the tile's cells (see the bin half): global scratch of this wavefront, only ever touched by a
  // tile that holds a record flagged kRecRare
  uint32_t* const g_zs = Q.rare + size_t(rare_slot) * (3u * kTileCells);  // (pos << 1 | is -0) of the first record whose zmax is a zero
  uint32_t* const g_izs = g_zs + kTileCells;
  uint32_t* const g_first = g_zs + 2u * kTileCells;                       // (pos << 1 | first intensity is NaN) of the first record
  const unsigned long long* const row = Q.desc + size_t(tile) * Q.stride;

  bool rare_tile = false;  // wave-uniform
  if (n_chunks) {
    // the first 64 descriptors leave at once; the image is initialised in their shadow
    unsigned long long d_first = 0ull;
    if (lane < n_chunks) d_first = row[lane];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned k = lane + unsigned(q) * 64u;
      s_key[k] = kEmptyKey;
      s_zmax[k] = 0u;
      if (has_int) s_imax[k] = 0u;
      if (has_col) s_last[k] = 0u;
    }
    // ---- fold the tile's records into the LDS image, 64 chunks at a time, 128 records per window.  step 0: the
    // values; step 1 (only a tile holding a record flagged kRecRare): the order of first occurrences, into the
    // wavefront's global scratch ----
    bool rare_seen = false;
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
#pragma unroll 1
      for (unsigned c0 = 0; c0 < n_chunks; c0 += 64u) {
        unsigned long long d = 0ull;
        if (c0 == 0u) d = d_first;
        else if (c0 + lane < n_chunks) d = row[c0 + lane];
        const unsigned cnt = unsigned(d >> 32);
        const unsigned inc = wave_scan_incl(cnt);
        const unsigned total = uni(unsigned(__builtin_amdgcn_readlane(int(inc), 63)));
        const unsigned off = inc - cnt;  // exclusive
        wave_sync();  // (the previous batch's readers of s_dpos / s_doff are done)
        s_dpos[lane] = unsigned(d);
        s_doff[lane] = off;
#pragma unroll 1
        for (unsigned r0 = 0; r0 < total; r0 += 128u) {
          // which chunk holds record r of the window: chunk starts are marked, a running maximum spreads them
          wave_sync();
          s_own[lane] = 0u;
          s_own[lane + 64u] = 0u;
          wave_sync();
          if (cnt && off >= r0 && off < r0 + 128u) s_own[off - r0] = lane + 1u;
          const unsigned carry = unsigned(__popcll(__ballot(cnt != 0u && off < r0)));  // chunks that start before the window
          wave_sync();
          unsigned m0 = wave_scan_max(s_own[lane]);
          unsigned m1 = wave_scan_max(s_own[lane + 64u]);
          m0 = max(m0, carry);
          m1 = max(m1, uni(unsigned(__builtin_amdgcn_readlane(int(m0), 63))));
          unsigned pos_[2];
          uint4 h_[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const unsigned r = r0 + unsigned(b) * 64u + lane;
            const unsigned ch = (b ? m1 : m0) - 1u;
            pos_[b] = 0xFFFFFFFFu;
            h_[b] = make_uint4(0u, 0u, 0u, 0u);
            if (r < total) {  // (then ch is a real chunk: the first one starts at record 0)
              pos_[b] = s_dpos[ch] + (r - s_doff[ch]);
              h_[b] = *reinterpret_cast<const uint4*>(Q.hot + pos_[b]);
            }
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (pos_[b] == 0xFFFFFFFFu) continue;
            const unsigned pos = pos_[b];
            const uint32_t cw = h_[b].w, lc = cw & kCitMask;
            if (step == 0) {
              const uint32_t lo = (cw & kRecNoWin) ? kNoWinner : ((pos << 1) | ((cw & kRecMinNeg) ? 1u : 0u));
              atomicMin(&s_key[lc], ((unsigned long long)h_[b].x << 32) | lo);
              atomicMax(&s_zmax[lc], h_[b].y);  // (max with 0: no-op)
              if (has_int) atomicMax(&s_imax[lc], h_[b].z);
              if (has_col) atomicMax(&s_last[lc], pos + 1u);
              rare_seen = rare_seen || (cw & kRecRare) != 0u;
            } else {
              if (h_[b].y == kOrdZero) atomicMin(&g_zs[lc], (pos << 1) | ((cw & kRecZNeg) ? 1u : 0u));
              if (has_int) {
                if (h_[b].z == kOrdZero) atomicMin(&g_izs[lc], (pos << 1) | ((cw & kRecINeg) ? 1u : 0u));
                atomicMin(&g_first[lc], (pos << 1) | ((cw & kRecNanFirst) ? 1u : 0u));
              }
            }
          }
        }
      }
      if (step == 0) {
        rare_tile = __ballot(rare_seen) != 0ull;
        if (!rare_tile) break;
        for (unsigned k = lane; k < 3u * kTileCells; k += 64u) g_zs[k] = 0xFFFFFFFFu;  // (zs | izs | first)
        __threadfence();  // the initialisation is at the memory side before the atomics
      } else {
        __threadfence();  // every atomic has landed before the cells read the words
      }
    }
    wave_sync();
  }

  const unsigned lz2 = opaque_zero();
  const ScanParams& P = late<16>(P0, lz2);
  const GeomConst& G = late<16>(G0, lz2);
  const typename POLICY::Layers& L = late<16>(L0, lz2);
  const TileAux& A = late<16>(A0, lz2);
  // ---- the tile's cells.  First the ones that only need stores, four per lane in memory order: the strips move()
  // vacates (NaN in EVERY layer, touched or not) and the obstacle clear of the untouched cells.  Then the touched ones,
  // compacted into a list so that each is one lane's only cell and all their record / sigma loads are ONE round trip.
  // (tr, tc: the tile's row / column in the tile grid; `strips`: a vacated strip crosses the tile) ----
  const bool fold = n_chunks != 0u && u.do_update;
  // (a move of >= the map's size on an axis is clearAll() in either reading of move())
  const bool basic = P.move_basic && abs(u.shr) < G.rows && abs(u.shc) < G.cols;
  unsigned n_touched = 0;  // wave-uniform
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned lc = lane + unsigned(q) * 64u;
    const int sr = int(tr * kTS + (lc & 15u)), sc = int(tc * kTC + (lc >> 4));
    const bool inside = sr < G.s_rows && sc < G.s_cols;
    const bool t = fold && inside && s_key[lc] != kEmptyKey;
    const unsigned long long m = __ballot(t);
    if (t) s_tl[n_touched + lane_rank(m)] = uint8_t(lc);
    n_touched += unsigned(__popcll(m));
    if (!(obst_tile || strips) || !inside) continue;
    const unsigned o = unsigned(sc) * unsigned(G.s_rows) + unsigned(sr);
    bool in_strip = false;
    if (strips) {
      in_strip = in_cleared_strip(sr + G.s_r0, u.e_sr, u.shr, G.rows) || in_cleared_strip(sc + G.s_c0, u.e_sc, u.shc, G.cols);
      if (in_strip && basic) {
        POLICY::clear_cell_basic(L, o);  // (option "move_clear_basic": the three basic layers only)
      } else if (in_strip) {
        for (int l0 = 0; l0 < n_layers; l0 += 8) {
          float* p[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) p[k] = all_layers[min(l0 + k, n_layers - 1)];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (l0 + k < n_layers) p[k][o] = nanv;
        }
        POLICY::clear_cell(L, o);
      }
    }
  }
  wave_sync();
#pragma unroll 1
  for (unsigned j0 = 0; j0 < n_touched; j0 += 64u) {
    const unsigned j = j0 + lane;
    if (j < n_touched) {
      const unsigned lc = s_tl[j];
      const int sr = int(tr * kTS + (lc & 15u)), sc = int(tc * kTC + (lc >> 4));
      const unsigned o = unsigned(sc) * unsigned(G.s_rows) + unsigned(sr);
      // (a touched cell inside a vacated strip: its layers were NaN-filled above, its state starts from NaN)
      const bool strip = strips && (in_cleared_strip(sr + G.s_r0, u.e_sr, u.shr, G.rows) ||
                                    in_cleared_strip(sc + G.s_c0, u.e_sc, u.shc, G.cols));
      const unsigned long long key = s_key[lc];
      const uint32_t zm = s_zmax[lc];
      uint32_t im = 0u, zsw = 0u, izw = 0u, fst = 0u, rgb = 0u;
      if (has_int) im = s_imax[lc];
      if (rare_tile) {  // (written by memory-side atomics of this wavefront: read past the L1 / L2 copies)
        zsw = __hip_atomic_load(&g_zs[lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has_int) {
          izw = __hip_atomic_load(&g_izs[lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fst = __hip_atomic_load(&g_first[lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const uint32_t wl = uint32_t(key);
      float var = 0.0f;  // (CellObservation default, elevation_mapping.hpp:26-34)
      if (wl != kNoWinner) var = Q.cold[wl >> 1].var;
      if (has_col) rgb = Q.cold[s_last[lc] - 1u].rgb;
      typename POLICY::State stt;
      float sint = nanv;
      if (strip && !basic) {
        POLICY::set_nan(stt);
      } else {
        POLICY::load(L, o, stt);
        if (has_int && !(P.dbg_upd & 2)) sint = L.intensity[size_t(o) * L.istride];
        if (strip) POLICY::set_nan_basic(stt);
      }
      const float min_z = wl != kNoWinner ? signed_value(uint32_t(key >> 32), wl & 1u) : kFltMax;
      const float max_z = zm ? signed_value(zm, zsw & 1u) : -kFltMax;
      if (A.ras_z) A.ras_z[o] = min_z;
      POLICY::update(L, o, stt, min_z, var, max_z);
      s_obst[lc] = (max_z > min_z) ? max_z : nanv;  // (stored with the tile's other cells below)
      if (has_int && !(P.dbg_upd & 2)) {
        const float obs = (fst & 1u) ? nanv  // first point NaN -> stays NaN (elevation_mapping.cpp:73-79)
                                     : signed_value(im, izw & 1u);
        if (isnan(sint) || obs > sint) L.intensity[size_t(o) * L.istride] = obs;
      }
      if (has_col) reinterpret_cast<uint32_t*>(L.color)[o] = rgb & 0x00FFFFFFu;
    }
  }
  // The obstacle layer of the whole tile, four 64-byte column segments per store: map_.clear(obstacle)
  // (elevation_mapping.cpp:144-146) for the untouched cells and the touched cells' values in ONE dense pass — as 163 K
  // scattered 4-byte stores the touched cells were a third of the launch's write requests.
  if (obst_tile && !(P.dbg_upd & 1)) {  // (dbg_upd: measurement only — results are wrong)
    wave_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned lc = lane + unsigned(q) * 64u;
      const int sr = int(tr * kTS + (lc & 15u)), sc = int(tc * kTC + (lc >> 4));
      if (sr >= G.s_rows || sc >= G.s_cols) continue;
      const bool t = fold && s_key[lc] != kEmptyKey;
      L.obstacle[unsigned(sc) * unsigned(G.s_rows) + unsigned(sr)] = t ? s_obst[lc] : nanv;
    }
  }
  // bookkeeping: the chunk list is consumed, the tile remembers who touched it last
  if (lane == 0u) {
    A.upd_part[tile] = n_touched;
    if (n_chunks) {
      Q.cnt[size_t(tile) << Q.cnt_shift] = 0u;
      if (u.do_update) A.stamp[tile] = P.scan_no;
    }
  }
  wave_sync();  // (the next tile re-initialises the image)
}

// One update wavefront: its tiles one after the other.  Per pass the counters and stamps of up to 64 tiles come back
// in one round trip (lane = tile) and only the live ones are visited.
template <typename POLICY, bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ void tupdate_wave(
    const ScanParams& P, const GeomConst& G, const TileGrid& TG, DevState* __restrict__ st,
    const typename POLICY::Layers& L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool& Q, const TileAux& A, const TileWork& K, unsigned char* lds, const unsigned w) {
  const unsigned lane = threadIdx.x & 63u;
  // An update wavefront is a chain of dependent round trips with a few hundred instructions between them; beside
  // six arithmetic-bound bin wavefronts per SIMD every one of those instructions waits its turn.  Priority 3: its
  // instructions issue first (it is idle most of the time, the bin wavefronts lose next to nothing).
  if (K.prio) __builtin_amdgcn_s_setprio(3);
  TileCtx u;
  make_tile_ctx(P, st, u, w == 0u && lane == 0u);
  if (w >= K.W) return;  // (a surplus wavefront of the grid owns nothing)
#pragma unroll 1
  for (unsigned p0 = 0; p0 < K.T; p0 += 64u) {
    // (kernel arguments re-read per pass, where they are used: see late())
    const unsigned lz = opaque_zero();
    const TilePool& Qp = late<32>(Q, lz);
    const TileAux& Ap = late<32>(A, lz);
    const TileGrid& TGp = late<32>(TG, lz);
    const GeomConst& Gp = late<32>(G, lz);
    const unsigned k = p0 + lane;
    const unsigned tile = k * K.W + w;
    const bool valid = k < K.T && tile < TGp.n_tiles;
    unsigned nch = 0u, stamp = 0xFFFFFFFFu;
    if (valid) { nch = Qp.cnt[size_t(tile) << Qp.cnt_shift]; stamp = Ap.stamp[tile]; }
    // (the tile's row / column in the tile grid: one division per lane and pass instead of one per tile)
    const unsigned tr = tile % unsigned(TGp.tiles_r), tc = tile / unsigned(TGp.tiles_r);
    const bool hits = valid && u.strips &&
                      (span_hits_strip(int(tr) * kTS + Gp.s_r0, kTS, u.e_sr, u.shr, Gp.rows) ||
                       span_hits_strip(int(tc) * kTC + Gp.s_c0, kTC, u.e_sc, u.shc, Gp.cols));
    const bool ob = valid && u.do_update && (nch != 0u || stamp == u.ob_scan);
    const bool live = valid && (nch != 0u || ob || hits);
    if (valid && !live) Ap.upd_part[tile] = 0u;
    unsigned long long m = __ballot(live);
    const unsigned long long mo = __ballot(ob), mh = __ballot(hits);
    while (m) {  // wave-uniform walk over the live tiles of the pass
      const unsigned q = unsigned(__ffsll((long long)m)) - 1u;
      m &= m - 1ull;
      const unsigned tq = uni(unsigned(__builtin_amdgcn_readlane(int(tile), int(q))));
      const unsigned nq = uni(unsigned(__builtin_amdgcn_readlane(int(nch), int(q))));
      const unsigned trq = uni(unsigned(__builtin_amdgcn_readlane(int(tr), int(q))));
      const unsigned tcq = uni(unsigned(__builtin_amdgcn_readlane(int(tc), int(q))));
      tupdate_tile<POLICY, HAS_INT, HAS_COL>(P, G, u, L, all_layers, n_layers, Q, A, lds, tq, trq, tcq, nq,
                                             ((mo >> q) & 1ull) != 0ull, ((mh >> q) & 1ull) != 0ull, lane, w);
    }
  }
}

template <typename POLICY, bool HAS_INT, bool HAS_COL>
__global__ __launch_bounds__(256, FDM_UPD_WAVES) void k_tupdate(
    const ScanParams P, const GeomConst G, const TileGrid TG, DevState* __restrict__ st,
    const typename POLICY::Layers L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool Q, const TileAux A, const TileWork K) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  const unsigned wave = uni(threadIdx.x >> 6);  // (wave-uniform: everything derived from it lives in scalar registers)
  tupdate_wave<POLICY, HAS_INT, HAS_COL>(P, G, TG, st, L, all_layers, n_layers, Q, A, K,
                                         dyn_lds + wave * tile_wave_lds_bytes(HAS_INT, HAS_COL), blockIdx.x * 4u + wave);
}

// update of scan t + bin of scan t+1 in one launch (the pools are double-buffered by scan parity).  The update blocks
// come first in the grid (their wavefronts are chains of dependent round trips: they start at once and the bin blocks
// fill the rest of the chip beside them).
template <typename POLICY, bool HAS_INT, bool HAS_COL, int THREADS, bool LEAN>
__global__ __launch_bounds__(THREADS, FDM_UPD_WAVES) void k_tupdate_tbin(
    const ScanParams Pu, const GeomConst G, const TileGrid TG, DevState* __restrict__ st,
    const typename POLICY::Layers L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool Qu, const TileAux A, const TileWork K, unsigned upd_blocks, const ScanParams Pb,
    const ScanInputs Ib, const Scratch Sb, const TilePool Qb, int32_t* __restrict__ cell_ids) {
  static_assert(THREADS == 256, "256-thread blocks");
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  const unsigned long long t0 = A.timeline ? wall_clock64() : 0ull;
#if FDM_MB_PHASES
  if (threadIdx.x == 0) { g_phase[0] = g_phase[1] = g_phase[2] = unsigned(t0); }
#endif
  if (blockIdx.x < upd_blocks) {
    const unsigned wave = uni(threadIdx.x >> 6);  // (wave-uniform: everything derived from it lives in scalar registers)
    tupdate_wave<POLICY, HAS_INT, HAS_COL>(Pu, G, TG, st, L, all_layers, n_layers, Qu, A, K,
                                           dyn_lds + wave * tile_wave_lds_bytes(HAS_INT, HAS_COL), blockIdx.x * 4u + wave);
  } else {
    const unsigned bb = blockIdx.x - upd_blocks;
    // The first ~1 300 bin blocks start in the same microsecond and ask for most of the scan at once: for 6-7 us
    // the chip waits for memory, then every SIMD's seven wavefronts compete for issue slots in the same phase.  A
    // stagger (block k of a CU's first round waits k x `stagger` x 512 cycles) lets the early blocks' arithmetic run
    // under the late blocks' loads.
    if (K.stagger) {
      const unsigned slot = (bb >> 8) & 7u;
      for (unsigned i = 0; i < slot * K.stagger && bb < 2048u; ++i) __builtin_amdgcn_s_sleep(8);
    }
    if (K.delay && bb < K.delay_blocks)
      for (unsigned i = 0; i < K.delay; ++i) __builtin_amdgcn_s_sleep(8);
    TbinRing H(Pb, st);
    tbin2_body<HAS_INT, HAS_COL, LEAN>(Pb, G, TG, H, Ib, Sb, Sb.bin_part, Qb, cell_ids, dyn_lds, bb);
  }
  if (A.timeline && threadIdx.x == 0) {  // (thread 0's view of the block; bench A/B tool, see scripts/timeline.py)
    A.timeline[2u * blockIdx.x] = t0;
#if FDM_MB_PHASES
    A.timeline[2u * blockIdx.x + 1u] = phase_word(t0);
#else
    A.timeline[2u * blockIdx.x + 1u] = wall_clock64();
#endif
  }
}

}  // namespace fdm
