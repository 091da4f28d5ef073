// fdm_tiled.hpp — the large-scan pipeline: per-tile observation buckets instead of per-cell atomics.
//
// Why (profiles/r01): the first large-scan pipeline reduced every cell through
// memory-side atomics (k_bin4 -> key/aux scratch -> k_update).  On configs[3] that was ~240 K
// (block, cell) pairs x 4 atomics ~ 1 M fabric operations per scan at ~26 Gop/s = the whole 39 us of
// the bin kernel, plus a dense 8 B/cell key sweep and a 3-sector point gather in the update kernel
// (160 MB of traffic for 52 MB of algorithmic bytes).  Nothing in this pipeline reduces through memory:
//
//   k_tbin   : a block of 4*THREADS consecutive points runs preprocessScan + getIndex, merges
//              same-cell points in registers and in a per-block LDS table (as before), then SORTS its
//              unique cells by 32x32-cell map tile in LDS and writes them as observation records —
//              plain coalesced stores into the block's own region of a record pool.  Per (block, tile)
//              ONE returning atomic appends a chunk descriptor {first record, count} to the tile's list
//              (configs[3]: ~16 per block, 16 K per scan, instead of ~1 M).  A record carries
//              everything the update needs (min z + its sigma_z^2, max z, intensity, colour), so the
//              update kernel never looks at the scan again: the caller's arrays are dead as soon as
//              this kernel has run.
//   k_tupdate: one block per 32x32 tile.  It reads the tile's chunk list, folds the records into an
//              LDS image of the tile (LDS atomics), and then walks the tile's cells in memory order:
//              estimator step, min/max, obstacle, intensity, colour, move() strips.  Only touched
//              cells read and write their 64 B / 128 B record; there is no per-cell scratch in memory.
//
// Order rules (elevation_mapping.cpp:62-92) carried through both levels without a point index:
//   * records of one cell are ordered by pool position = (bin block, ...) = scan order of their
//     blocks, and each block contributes at most one record per cell, so "first point wins a tie"
//     is "lowest position wins";
//   * the minimum is ONE 64-bit reduction word  ord(z) << 32 | order << 1 | (z is -0): the lowest z,
//     among equals the first point — "strict z < min_z, first point wins" — and the winner's order
//     leads to its sigma_z^2.  -0 and +0 compare equal in the reference, so every value is reduced
//     with zeros canonicalised; for the maxima (32-bit words) the sign of the FIRST zero-valued
//     point of a cell rides along in a separate min-reduced word.  min_z / max_z / intensity come
//     out bit-identical to the reference's first-seen zero.
//
// Algorithmic bytes are unchanged: 12 B/point (+4 intensity, +4 colour), 72 / 124 B
// per touched cell, 4 B per map cell per scan for the obstacle clear.
#pragma once

#include "fdm_kernels.hpp"

namespace fdm {

#ifndef FDM_TILE_COL_SHIFT
#define FDM_TILE_COL_SHIFT 5
#endif
constexpr int kTS = 32;                 // tile height: 32 consecutive rows of a column = 128 B of every layer
constexpr int kTSShift = 5;
constexpr int kTCShift = FDM_TILE_COL_SHIFT;   // tile width: 32 columns (16: the update alone 21.9 -> 18.7 us, but twice the tile
                                               // blocks ahead of the bin blocks: fused launch 34.2 -> 37.5 us)
constexpr int kTC = 1 << kTCShift;
constexpr unsigned kTileCells = unsigned(kTS * kTC);
constexpr int kCellsPerThread = int(kTileCells) / 256;
constexpr uint32_t kNoWinner = 0xFFFFFFFFu;
constexpr uint32_t kOrdZero = 0x80000000u;  // ord(+0.0f)
#ifndef FDM_UPD_WAVES
#define FDM_UPD_WAVES 6  // min waves per SIMD the tile kernels are compiled for (<= 72 VGPRs): LDS lets 7-8 blocks per CU in
#endif
#ifndef FDM_REC_BATCH
#define FDM_REC_BATCH 2
#endif
constexpr int kRecBatch = FDM_REC_BATCH;    // k_tupdate: records per thread whose loads are in flight together
constexpr int kCellBatch = 1;   // ... and touched cells

// flags beside the cell-in-tile number (10 bits) of a record
// (rare: set only by a block that met a -0.0 or a NaN intensity; "the cell holds a zero" needs no flag — a
// zero only matters when it is the cell's maximum, and then the record's zmax / imax IS ord(0))
constexpr uint32_t kRecNanFirst = 1u << 10;  // the block's first point in the cell has a NaN intensity
constexpr uint32_t kRecZNeg = 1u << 12;      // the block's first zero-valued z in the cell is -0
constexpr uint32_t kRecINeg = 1u << 14;      // the same for the intensity
constexpr uint32_t kRecRare = kRecNanFirst | kRecZNeg | kRecINeg;

// One observation record: what a bin block knows about one cell (32 B, two 16 B words; a chunk's records are
// contiguous, so a tile's update reads them as whole 128 B lines — as five separate arrays the same records
// cost 18 MB of fetches for 6.7 MB of payload at configs[3]).
struct __align__(16) TileRec {
  unsigned long long key;  // ord(min z) << 32 | pos << 1 | (that z is -0)   (low word kNoWinner: no finite z)
  uint32_t zmax;           // ord(max z), 0 = none
  uint32_t imax;           // ord(max intensity), 0 = none
  uint32_t cell;           // cell inside the tile | kRec* flags
  float var;               // sigma_z^2 of the min-z point
  uint32_t rgb;            // colour of the block's last point in the cell
  uint32_t pad;
};
// The record pool of one scan parity (bin of scan t+1 runs beside the update of scan t).
struct TilePool {
  TileRec* rec;               // [cap]
  unsigned long long* desc;   // [n_tiles][stride]  row of a tile: word 0 = number of chunks (put back to 0 by the update
                              // kernel), then one word per chunk: first record | count << 32
  unsigned stride;            // > bin blocks of the scan: a block appends at most one chunk per tile
  uint32_t* rare;             // [update groups][3][1024] scratch of k_tupdate's rare path (first-occurrence words)
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

// value of a canonicalised ord word; `neg`: the first zero seen was -0 (only looked at for a zero)
__device__ __forceinline__ float signed_value(uint32_t ordv, uint32_t neg) {
  return (ordv == kOrdZero && neg) ? -0.0f : unord(ordv);
}
// ord with -0 folded onto +0 (they tie in every comparison of the reference)
__device__ __forceinline__ uint32_t ord_canon(float v) { return ord(v == 0.0f ? 0.0f : v); }

// point -> (tile << 10 | cell in tile) of this engine's owned window; -1 outside the (global) map,
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
  return (tile << 10) | ((sc & (kTC - 1)) << kTSShift) | (sr & (kTS - 1));
}

// ---------------------------------------------------------------------------------------------
// k_tbin.  Dynamic LDS: 4*THREADS-slot arrays cell u32 | key u64 | zmax u32 [| imax u32] [| last u32], then
// the compaction list (u16): 18-26 B per point.  The RARE-EVENT words of a cell — the sign of its first
// zero-valued z / intensity, whether its first point's intensity is NaN — are not in the table: a block
// that meets no -0.0 and no NaN (every block of a real scan) never needs them, and keeping 12 B per slot for
// them cost three resident blocks per CU (configs[3]: 27 -> 22 us).  A block that does meet one re-walks
// its points after the main fold, into table arrays that are dead by then (rare path, exact).
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
    if (cap_x && live) {
      cap_x[i0 + j] = (S.cap_drop_nan && !pass[j]) ? __uint_as_float(0x7FC00000u) : xs[j];
      S.cap_y[i0 + j] = ys[j]; S.cap_z[i0 + j] = zs[j];
      if (cap_var) cap_var[i0 + j] = cvar;
    }
    n_pass += pass[j] ? 1u : 0u;
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
    const int tcell = (tile << 10) | ((sc & (kTC - 1)) << kTSShift) | (sr & (kTS - 1));
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

template <bool HAS_INT, bool HAS_COL, int THREADS, bool LEAN, class PT, class HOOK>
__device__ __forceinline__ void tbin_body(const PT& P, const GeomConst& G, const TileGrid& TG,
                                          HOOK& H, const ScanInputs& I,
                                          const Scratch& S, unsigned long long* __restrict__ bin_part, const TilePool& Q,
                                          int32_t* __restrict__ cell_ids, unsigned char* lds,
                                          const unsigned bid) {
  const int dbg = LEAN ? 0 : P.dbg_no_atomics;  // measurement only: leave the kernel after a phase (results are wrong)
  constexpr int kSlots = THREADS * 4;  // == points per block: room for every point in its own cell
  constexpr int kSlotBits = THREADS == 512 ? 11 : (THREADS == 256 ? 10 : 9);
  static_assert((1 << kSlotBits) == kSlots, "block size");
  constexpr int kWaves = THREADS / 64;
  uint32_t* const h_cell = reinterpret_cast<uint32_t*>(lds);
  // ord(min z) << 32 | order in block << 1 | z is -0, min-reduced: the lowest z, among equals the first point
  unsigned long long* const h_key = reinterpret_cast<unsigned long long*>(h_cell + kSlots);
  uint32_t* const h_zmax = h_cell + 3 * kSlots;  // ord(max z), 0 = none
  uint32_t* const h_imax = h_zmax + kSlots;      // (HAS_INT) ord(max intensity), 0 = none
  uint32_t* const h_last = h_zmax + (HAS_INT ? 2 : 1) * kSlots;  // (HAS_COL) order + 1 of the last point
  uint16_t* const s_list = reinterpret_cast<uint16_t*>(h_zmax + (1 + (HAS_INT ? 1 : 0) + (HAS_COL ? 1 : 0)) * kSlots);
  __shared__ DevCand s_cand;
  __shared__ unsigned s_cnt[kWaves];
  __shared__ unsigned s_wsum[kWaves];
  __shared__ unsigned s_rare;

  const float* __restrict__ px = I.x;
  const float* __restrict__ py = I.y;
  const float* __restrict__ pz = I.z;
  const float* __restrict__ pint = I.intensity;

  // the point loads go out first: they are in flight while the table is initialised and
  // thread 0 works out the post-move geometry
  const unsigned b0 = bid * unsigned(kSlots);
  const unsigned l0 = threadIdx.x * 4u;
  const unsigned i0 = b0 + l0;
  float xs[4], ys[4], zin[4], vs[4];
  auto load_points = [&]() {
    if (i0 + 3 < P.n) {
      const float4 a = *reinterpret_cast<const float4*>(px + i0);
      const float4 b = *reinterpret_cast<const float4*>(py + i0);
      const float4 c = *reinterpret_cast<const float4*>(pz + i0);
      xs[0] = a.x; xs[1] = a.y; xs[2] = a.z; xs[3] = a.w;
      ys[0] = b.x; ys[1] = b.y; ys[2] = b.z; ys[3] = b.w;
      zin[0] = c.x; zin[1] = c.y; zin[2] = c.z; zin[3] = c.w;
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
        zin[j] = ok ? pz[i0 + j] : 0.f;
        if (HAS_INT) vs[j] = ok ? pint[i0 + j] : 0.f;
      }
    }
  };
  load_points();
  {  // table initialisation, 16 bytes per LDS store (4 slots per thread and array)
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), zero = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(h_cell)[threadIdx.x] = ones;                  // kEmptyCell
    reinterpret_cast<uint4*>(h_key)[2 * threadIdx.x] = ones;               // kEmptyKey (two keys per store)
    reinterpret_cast<uint4*>(h_key)[2 * threadIdx.x + 1] = ones;
    reinterpret_cast<uint4*>(h_zmax)[threadIdx.x] = zero;
    if (HAS_INT) reinterpret_cast<uint4*>(h_imax)[threadIdx.x] = zero;
    if (HAS_COL) reinterpret_cast<uint4*>(h_last)[threadIdx.x] = zero;
  }
  if (threadIdx.x == 0) s_rare = 0u;
  H.begin();  // (thread 0's state loads leave; the walk follows the transforms below)

  // phase 1: all four points through the arithmetic — first what needs no geometry (both transforms, the crops), in
  // the shadow of thread 0's state read, then the geometry candidate (barrier), then getIndex
  int cells[4];
  float xm[4], ym[4], zs[4];
  bool pass[4];
  unsigned n_pass = 0, n_in = 0;
  bool any_glob = false;
  tbin_prep<HAS_INT, THREADS, LEAN, true>(P, S, bid, xs, ys, zin, xm, ym, zs, pass, n_pass);
  const DevCand cand = H.finish(G, &s_cand, bid, n_pass);  // contains the __syncthreads
  FDM_PHASE(0);  // table initialised, points transformed and cropped, candidate known
  tbin_points<HAS_INT, THREADS, LEAN, true>(P, G, TG, cell_ids, cand, bid, xm, ym, pass, cells, n_in, any_glob);
  FDM_PHASE(1);  // index done
  if (dbg == 2) {
    bin_part[bid] = (cells[0] + cells[1] + cells[2] + cells[3] == 0x7FFFFFF1) ? 1ull : 0x100000001ull;
    return;
  }

  unsigned v = n_pass | (n_in << 16);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = v;
  if (__ballot(any_glob) && (threadIdx.x & 63) == 0) H.note_inside();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // probe of the block's cell table: claim-or-find in ONE LDS operation per step.  (Multiplicative hash: the low
  // bits of tile << 10 | cell repeat from tile to tile along a wedge, and linear probing through such clusters
  // cost 49 of the first version's 65 us.)
  auto find_slot = [&](uint32_t cell) -> uint32_t {
    uint32_t h = (cell * 2654435761u) >> (32 - kSlotBits);
    while (true) {
      const uint32_t prev = atomicCAS(&h_cell[h], kEmptyCell, cell);
      if (prev == kEmptyCell || prev == cell) return h;
      h = (h + 1) & (kSlots - 1);
    }
  };

  // phase 2: merge runs of equal cell in registers, fold each run into the table
  bool rare = false;  // a -0.0 or a NaN intensity among this thread's points
  {
    int run_cell = -1;
    unsigned long long run_key = kEmptyKey;
    uint32_t run_zmx = 0u, run_imx = 0u, run_lst = 0u;
    auto fold_run = [&]() {
      if (run_cell < 0) return;
      const uint32_t h = find_slot(uint32_t(run_cell));
      // (no-op operands instead of branches: max with 0)
      atomicMin(&h_key[h], run_key);
      atomicMax(&h_zmax[h], run_zmx);
      if (HAS_INT) atomicMax(&h_imax[h], run_imx);
      if (HAS_COL) atomicMax(&h_last[h], run_lst);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (cells[j] < 0) continue;
      const uint32_t li = l0 + j;  // order inside the block
      const float z = zs[j];
      const uint32_t zneg = __float_as_uint(z) == 0x80000000u ? 1u : 0u;
      const uint32_t oz = ord_canon(z);
      // strict "z < min_z" from FLT_MAX / "z > max_z" from lowest(): NaN, FLT_MAX and beyond never win
      const unsigned long long key = (z < kFltMax) ? ((unsigned long long)oz << 32) | (li << 1) | zneg
                                                   : ((unsigned long long)ord(kFltMax) << 32) | kNoWinner;
      const uint32_t zmx = (z > -kFltMax) ? oz : 0u;
      uint32_t imx = 0u;
      rare = rare || zneg != 0u;
      if (HAS_INT) {
        const float vv = vs[j];
        const uint32_t vb = __float_as_uint(vv);
        const bool vnan = (vb & 0x7FFFFFFFu) > 0x7F800000u;
        imx = vnan ? 0u : ord_canon(vv);
        rare = rare || vnan || vb == 0x80000000u;
      }
      if (cells[j] != run_cell) {
        fold_run();
        run_cell = cells[j];
        run_key = key;
        run_zmx = zmx;
        run_imx = imx;
      } else {
        run_key = key < run_key ? key : run_key;
        run_zmx = zmx > run_zmx ? zmx : run_zmx;
        run_imx = imx > run_imx ? imx : run_imx;
      }
      run_lst = li + 1u;
    }
    fold_run();
    // a -0.0 or a NaN intensity among the block's points: the order of first occurrences matters
    if (__ballot(rare) && lane == 0) s_rare = 1u;
  }
  __syncthreads();  // every run of the block is in the table
  if (threadIdx.x == 0) {
    unsigned np = 0, ni = 0;
    for (int w = 0; w < kWaves; ++w) { np += s_cnt[w] & 0xFFFFu; ni += s_cnt[w] >> 16; }
    if (np) H.note_pass();
    bin_part[bid] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
  const bool rare_block = s_rare != 0u;  // block-uniform

  // ---- flush: the block's unique cells become observation records, grouped by map tile ----
  // (a) compact the occupied slots: record j of the block is slot s_list[j]
  unsigned n_rec;
  {
    const uint4 c4 = *reinterpret_cast<const uint4*>(h_cell + l0);  // the thread's 4 consecutive slots
    const uint32_t cc[4] = {c4.x, c4.y, c4.z, c4.w};
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) mine += cc[q] != kEmptyCell ? 1u : 0u;
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const unsigned ws = s_wsum[w];
      base += w < wave ? ws : 0u;
      total += ws;
    }
    n_rec = total;
    unsigned p = base + inc - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (cc[q] != kEmptyCell) s_list[p++] = uint16_t(l0 + q);
  }
  __syncthreads();
  FDM_PHASE(2);  // LDS fold + compaction done
  // (b) record j = threadIdx.x + q * THREADS leaves the table for registers; its sigma_z^2 is evaluated
  // from the winning point (re-read from L2: the block loaded it a few microseconds ago)
  constexpr int kRounds = 4;
  uint32_t c_[kRounds], kz_[kRounds], kw_[kRounds], zm_[kRounds], im_[kRounds], fl_[kRounds], col_[kRounds];
  float var_[kRounds];
#pragma unroll
  for (int q = 0; q < kRounds; ++q) {
    c_[q] = kEmptyCell; kz_[q] = 0u; kw_[q] = kNoWinner; zm_[q] = 0u; im_[q] = 0u; var_[q] = 0.0f; fl_[q] = 0u; col_[q] = 0u;
    if (unsigned(q * THREADS) >= n_rec) continue;  // block-uniform
    const unsigned j = threadIdx.x + unsigned(q * THREADS);
    if (j >= n_rec) continue;
    const unsigned slot = s_list[j];
    c_[q] = h_cell[slot];  // tile << 10 | cell in tile
    const unsigned long long k64 = h_key[slot];
    kz_[q] = uint32_t(k64 >> 32);
    kw_[q] = uint32_t(k64);
    zm_[q] = h_zmax[slot];
    if (HAS_INT) im_[q] = h_imax[slot];
    if (HAS_COL) col_[q] = I.rgb[b0 + h_last[slot] - 1u];
    const uint32_t wl = kw_[q];           // winner: order << 1 | sign, or kNoWinner
    if (wl != kNoWinner) {                // (else: CellObservation default 0, elevation_mapping.hpp:26-34)
      const unsigned gi = b0 + (wl >> 1);
      if (P.has_var) var_[q] = I.var[gi];
      else if (P.integrate_mode) var_[q] = sigma_z2(P, px[gi], py[gi], pz[gi]);
    }
  }
  __syncthreads();
  if (rare_block) {
    // Rare path: some point of the block is a -0.0 or has a NaN intensity, so the order of first occurrences
    // matters.  The points are walked once more (their cells, z and intensity stayed in registers) into
    // three table arrays that are dead now: per slot the (order << 1 | is -0) of the first zero-valued z, of
    // the first zero-valued intensity, and (order << 1 | is NaN) of the first point.
    uint32_t* const r_zs = h_cell + kSlots;      // (the key's memory)
    uint32_t* const r_izs = h_cell + 2 * kSlots;
    uint32_t* const r_first = h_zmax;
    for (int k = threadIdx.x; k < kSlots; k += THREADS) {
      r_zs[k] = 0xFFFFFFFFu; r_izs[k] = 0xFFFFFFFFu; r_first[k] = 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (cells[j] < 0) continue;
      const uint32_t li = l0 + j;
      const uint32_t h = find_slot(uint32_t(cells[j]));  // (present: the main fold put it there)
      if (zs[j] == 0.0f) atomicMin(&r_zs[h], (li << 1) | (__float_as_uint(zs[j]) == 0x80000000u ? 1u : 0u));
      if (HAS_INT) {
        const float vv = vs[j];
        if (vv == 0.0f) atomicMin(&r_izs[h], (li << 1) | (__float_as_uint(vv) == 0x80000000u ? 1u : 0u));
        atomicMin(&r_first[h], (li << 1) | (isnan(vv) ? 1u : 0u));
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRounds; ++q) {
      if (c_[q] == kEmptyCell) continue;
      const unsigned slot = s_list[threadIdx.x + unsigned(q * THREADS)];
      uint32_t fl = 0u;
      const uint32_t zsw = r_zs[slot];
      if (zsw != 0xFFFFFFFFu && (zsw & 1u)) fl |= kRecZNeg;
      if (HAS_INT) {
        const uint32_t izw = r_izs[slot];
        if (izw != 0xFFFFFFFFu && (izw & 1u)) fl |= kRecINeg;
        if (r_first[slot] & 1u) fl |= kRecNanFirst;
      }
      fl_[q] = fl;
    }
    __syncthreads();
  }
  // (c) the table's memory becomes the block's TILE table: tile id -> how many of the block's cells
  uint32_t* const t_tile = h_cell + kSlots;      // [kSlots]
  uint32_t* const t_cnt = h_cell + 2 * kSlots;   // [kSlots]
  uint32_t* const t_off = h_zmax;                // [kSlots]
  for (int k = threadIdx.x; k < kSlots; k += THREADS) {
    t_tile[k] = kEmptyCell;
    t_cnt[k] = 0u;
  }
  __syncthreads();
  uint32_t th_[kRounds], rk_[kRounds];
#pragma unroll
  for (int q = 0; q < kRounds; ++q) {
    th_[q] = 0u; rk_[q] = 0u;
    if (c_[q] == kEmptyCell) continue;
    const uint32_t tile = c_[q] >> 10;
    uint32_t h = (tile * 2654435761u) >> (32 - kSlotBits);
    while (true) {  // (at most n_rec <= kSlots distinct tiles: always terminates)
      const uint32_t prev = atomicCAS(&t_tile[h], kEmptyCell, tile);
      if (prev == kEmptyCell || prev == tile) break;
      h = (h + 1) & (kSlots - 1);
    }
    th_[q] = h;
    rk_[q] = atomicAdd(&t_cnt[h], 1u);
  }
  __syncthreads();
  // (d) exclusive scan of the tile counts; thread t owns entries t, t + THREADS, ... (neighbouring
  // tiles of a wedge go to different threads), appends one chunk per occupied entry to the tile's row
  {
    uint32_t nn[4], tt[4];
    unsigned mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      nn[k] = t_cnt[threadIdx.x + k * THREADS];
      tt[k] = t_tile[threadIdx.x + k * THREADS];
      mine += nn[k];
    }
    unsigned slot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // the returning atomics leave together, ahead of the scan
      slot[k] = 0u;
      if (nn[k])
        slot[k] = atomicAdd(reinterpret_cast<unsigned*>(Q.desc + size_t(tt[k]) * Q.stride), 1u);
    }
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    __syncthreads();  // ((a)'s readers of s_wsum are long done)
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned p = inc - mine;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) p += w < wave ? s_wsum[w] : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (nn[k]) {
        t_off[threadIdx.x + k * THREADS] = p;
        Q.desc[size_t(tt[k]) * Q.stride + 1u + slot[k]] = (unsigned long long)(b0 + p) | ((unsigned long long)nn[k] << 32);
        p += nn[k];
      }
    }
  }
  __syncthreads();
  // (e) the records, grouped by tile, into the block's own region of the pool
#pragma unroll
  for (int q = 0; q < kRounds; ++q) {
    if (c_[q] == kEmptyCell) continue;
    const uint32_t pos = b0 + t_off[th_[q]] + rk_[q];
    const uint32_t wl = kw_[q];
    TileRec r;
    r.key = ((unsigned long long)kz_[q] << 32) | (wl != kNoWinner ? (pos << 1) | (wl & 1u) : kNoWinner);
    r.zmax = zm_[q];
    r.imax = HAS_INT ? im_[q] : 0u;
    r.cell = (c_[q] & 1023u) | fl_[q];
    r.var = var_[q];
    r.rgb = HAS_COL ? col_[q] : 0u;
    r.pad = 0u;
    uint4* const dst = reinterpret_cast<uint4*>(Q.rec + pos);
    const uint4* const src = reinterpret_cast<const uint4*>(&r);
    dst[0] = src[0];
    dst[1] = src[1];
  }
}

}  // namespace fdm
#include "fdm_tbin2.hpp"  // tbin2_body: the second edition of the bin half (VER = 2 below)
namespace fdm {

template <bool HAS_INT, bool HAS_COL, int THREADS, bool LEAN, int VER = 2>
__global__ __launch_bounds__(THREADS) void k_tbin(const ScanParams P, const GeomConst G, const TileGrid TG,
                                                  DevState* __restrict__ st, const ScanInputs I,
                                                  const Scratch S, const TilePool Q,
                                                  int32_t* __restrict__ cell_ids) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  TbinRing H(P, st);
  if constexpr (VER == 2)
    tbin2_body<HAS_INT, HAS_COL, LEAN>(P, G, TG, H, I, S, S.bin_part, Q, cell_ids, dyn_lds, blockIdx.x);
  else
    tbin_body<HAS_INT, HAS_COL, THREADS, LEAN>(P, G, TG, H, I, S, S.bin_part, Q, cell_ids, dyn_lds, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_tupdate — 256 threads per tile.  LDS image of the tile (dynamic shared memory, per 256-thread
// group), 1024-entry arrays: key u64 | zmax | descriptors u64[256] | offsets u32[260]
// [| imax | izs | first] [| last].
__host__ __device__ constexpr unsigned tile_lds_bytes(bool has_int, bool has_col) {
  return kTileCells * (12u + (has_int ? 4u : 0u) + (has_col ? 4u : 0u)) + 260u * 4u + 256u * 8u;
}


struct TileCtx {
  bool applied, do_update, strips;
  unsigned ob_scan;
  DevGeom E;
  DevCand C;
};

__device__ __forceinline__ void make_tile_ctx(const ScanParams& P, DevState* __restrict__ st, TileCtx& u,
                                              bool committer) {
  const int slot = P.slot;
  const bool any_pass = st->flags[slot].any_pass != 0u;
  u.do_update = st->flags[slot].any_inside != 0u;
  u.applied = P.do_move && (!P.gate_on_filter || any_pass);
  u.ob_scan = st->obst[slot].scan;
  u.E = st->geom[slot];
  u.C = st->cand[slot];
  if (committer) {  // commit geometry + ring bookkeeping (see make_ctx)
    const int nxt = (slot + 1) & 3, nn = (slot + 2) & 3;
    DevGeom g = u.E;
    if (u.applied) { g.px = u.C.px; g.py = u.C.py; g.sr = u.C.sr; g.sc = u.C.sc; }
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
  u.strips = u.applied && (u.C.shr != 0 || u.C.shc != 0);
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
  return span_hits_strip(tr * kTS + G.s_r0, kTS, u.E.sr, u.C.shr, G.rows) ||
         span_hits_strip(tc * kTC + G.s_c0, kTC, u.E.sc, u.C.shc, G.cols);
}

// What differs between the scans a tile group works through (one per launch, or the scans of a batch in order).
struct TileJob {
  unsigned scan_no;
  int dbg_upd;       // measurement only (ScanParams::dbg_upd)
  bool write_obst;   // touched cells write the obstacle layer (a batch: only its LAST updating scan — every updating scan
                     // clears the whole layer first, elevation_mapping.cpp:144-146, so earlier values never survive)
  bool set_stamp;    // ... and that scan stamps the tiles it touched
};

// One tile by one 256-thread group (`lt` = thread inside the group).  Block-uniform control flow
// around the barriers: `n_chunks_max` is the largest chunk count among the block's groups.
// `d0` = word `lt` of the tile's descriptor row, already loaded by the caller (word 0 is the count).
template <typename POLICY, int BLOCK, bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ void tupdate_tile(
    const TileJob& J, const GeomConst& G, const TileGrid& TG, const TileCtx& u,
    const typename POLICY::Layers& L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool& Q, const TileAux& A, unsigned char* lds, const unsigned tile, const bool tile_ok,
    const unsigned n_chunks, const unsigned n_chunks_max, const unsigned long long d0, const bool obst_tile,
    const unsigned lt, const unsigned rare_slot, unsigned* s_rare /* one word per block */) {
  const float nanv = __uint_as_float(0x7FC00000u);
  constexpr bool has_int = HAS_INT, has_col = HAS_COL;  // (compile-time: the LDS layout and a dozen uniform values fold away)
  unsigned long long* const s_key = reinterpret_cast<unsigned long long*>(lds);  // min of the records' keys
  uint32_t* const s_zmax = reinterpret_cast<uint32_t*>(s_key + kTileCells);
  unsigned long long* const s_desc = reinterpret_cast<unsigned long long*>(s_zmax + kTileCells);  // [256]
  uint32_t* const s_off = reinterpret_cast<uint32_t*>(s_desc + 256);                      // [260]
  uint32_t* const s_imax = s_off + 260;                            // (intensity scans)
  uint32_t* const s_last = s_off + 260 + (has_int ? kTileCells : 0u);     // (colour scans)
  // rare-event words of the tile's cells (see k_tbin): global scratch of this group, only ever touched by a tile
  // that holds a record flagged kRecRare
  uint32_t* const g_zs = Q.rare + size_t(rare_slot) * (3u * kTileCells);  // (pos << 1 | is -0) of the first record whose zmax is a zero
  uint32_t* const g_izs = g_zs + kTileCells;
  uint32_t* const g_first = g_zs + 2u * kTileCells;                       // (pos << 1 | first intensity is NaN) of the first record
  const unsigned long long* const row = Q.desc + size_t(tile_ok ? tile : 0u) * Q.stride;

  if (J.dbg_upd == 1) {
    if (d0 == 0x7FFFFFF1ull) A.upd_part[0] = 1u;
    if (lt == 0 && tile_ok && n_chunks) Q.desc[size_t(tile) * Q.stride] = 0ull;
    return;
  }
  bool rare_tile = false;  // block-uniform
  if (n_chunks_max) {
#pragma unroll
    for (int q = 0; q < kCellsPerThread; ++q) {
      const unsigned k = lt + q * 256u;
      s_key[k] = kEmptyKey;
      s_zmax[k] = 0u;
      if (has_int) s_imax[k] = 0u;
      if (has_col) s_last[k] = 0u;
    }
    // ---- fold the tile's records into the LDS image, 256 row words (255 chunks) at a time.  step 0: the
    // values; step 1 (only a tile holding a record flagged kRecRare): the order of first occurrences, into the
    // group's global scratch ----
    bool rare_seen = false;
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
#pragma unroll 1
      for (unsigned c0 = 0; c0 <= n_chunks_max; c0 += 256u) {
        unsigned long long d = 0ull;
        if (c0 == 0u) d = lt ? d0 : 0ull;                      // (word 0 is the count)
        else if (c0 + lt <= n_chunks) d = row[c0 + lt];
        if (c0 + lt > n_chunks) d = 0ull;                      // beyond the list: whatever an earlier scan left
        const unsigned cnt = unsigned(d >> 32);
        // inclusive scan of the chunk sizes over the group's four wavefronts
        unsigned inc = cnt;
        const unsigned lane = lt & 63u;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
          const unsigned o = __shfl_up(inc, dd);
          if (lane >= unsigned(dd)) inc += o;
        }
        __syncthreads();  // (the previous batch's readers of s_off / s_desc are done)
        if (lane == 63u) s_off[256u + (lt >> 6)] = inc;
        __syncthreads();
        unsigned base = 0;
        for (unsigned w = 0; w < (lt >> 6); ++w) base += s_off[256u + w];
        s_off[lt] = base + inc - cnt;  // exclusive
        s_desc[lt] = d;
        __syncthreads();
        const unsigned total = uni(s_off[256u] + s_off[257u] + s_off[258u] + s_off[259u]);
        // kRecBatch records per thread and pass: all their loads are in flight before the first atomic
#pragma unroll 1
        for (unsigned r0 = lt; r0 < total; r0 += 256u * kRecBatch) {
          unsigned pos_[kRecBatch];
          uint32_t cw_[kRecBatch], zm_[kRecBatch], im_[kRecBatch];
          unsigned long long k_[kRecBatch];
#pragma unroll
          for (int b = 0; b < kRecBatch; ++b) {
            const unsigned r = r0 + unsigned(b) * 256u;
            pos_[b] = 0xFFFFFFFFu; cw_[b] = 0u; zm_[b] = 0u; im_[b] = 0u; k_[b] = 0ull;
            if (r >= total) continue;
            // which chunk holds record r: the last one whose offset is <= r (real chunks are never empty;
            // the batch's unused lanes sit at offset == total, its leading count word at offset 0 with
            // size 0 — the search takes the LAST lane with offset <= r, never that one)
            unsigned lo = 0, hi = 255u;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const unsigned mid = (lo + hi + 1u) >> 1;
              const bool le = s_off[mid] <= r;
              lo = le ? mid : lo;
              hi = le ? hi : mid - 1u;
            }
            const unsigned pos = unsigned(s_desc[lo]) + (r - s_off[lo]);
            pos_[b] = pos;
            const uint4 w0 = reinterpret_cast<const uint4*>(Q.rec + pos)[0];  // key | zmax | imax
            cw_[b] = Q.rec[pos].cell;
            k_[b] = (unsigned long long)w0.x | ((unsigned long long)w0.y << 32);
            zm_[b] = w0.z;
            im_[b] = w0.w;
          }
#pragma unroll
          for (int b = 0; b < kRecBatch; ++b) {
            if (pos_[b] == 0xFFFFFFFFu) continue;
            const unsigned pos = pos_[b];
            const uint32_t cw = cw_[b], lc = cw & 1023u;
            if (step == 0) {
              atomicMin(&s_key[lc], k_[b]);
              atomicMax(&s_zmax[lc], zm_[b]);  // (max with 0: no-op)
              if (has_int) atomicMax(&s_imax[lc], im_[b]);
              if (has_col) atomicMax(&s_last[lc], pos + 1u);
              rare_seen = rare_seen || (cw & kRecRare) != 0u;
            } else {
              if (zm_[b] == kOrdZero) atomicMin(&g_zs[lc], (pos << 1) | ((cw & kRecZNeg) ? 1u : 0u));
              if (has_int) {
                if (im_[b] == kOrdZero) atomicMin(&g_izs[lc], (pos << 1) | ((cw & kRecINeg) ? 1u : 0u));
                atomicMin(&g_first[lc], (pos << 1) | ((cw & kRecNanFirst) ? 1u : 0u));
              }
            }
          }
        }
      }
      if (step == 0) {
        if (__ballot(rare_seen) && (lt & 63u) == 0u) *s_rare = 1u;
        __syncthreads();
        rare_tile = uni(*s_rare) != 0u;
        if (!rare_tile) break;
        for (unsigned k = lt; k < 3u * kTileCells; k += 256u) g_zs[k] = 0xFFFFFFFFu;  // (zs | izs | first)
      }
      __syncthreads();  // step 0: the image is complete / the scratch is initialised; step 1: every atomic has landed
    }
  }

  FDM_PHASE(1);  // records folded into the tile image
  if (J.dbg_upd == 2) {
    if (lt == 0 && tile_ok && n_chunks) Q.desc[size_t(tile) * Q.stride] = 0ull;
    return;  // measurement only (block-uniform)
  }
  // ---- the tile's cells.  Untouched ones (most) only ever need stores — the obstacle clear, the strips
  // move() vacates — and are walked in memory order, four per thread (a wavefront = two columns x 32
  // rows).  The touched ones are compacted into a list first, so that each is one thread's only cell and
  // all their record / sigma loads are ONE round trip instead of four dependent ones. ----
  const unsigned tr = tile % unsigned(TG.tiles_r), tc = tile / unsigned(TG.tiles_r);
  uint16_t* const s_tlist = reinterpret_cast<uint16_t*>(s_desc);  // [kTileCells] (the descriptors are consumed)
  // (a move vacates a few rows / columns: only the tiles they cross look at their cells for it)
  const bool strips = u.strips && tile_hits_strips(u, G, TG, tile);
  const bool work = tile_ok && (n_chunks || obst_tile || strips);
  unsigned n_touched = 0;  // of the whole tile (group-uniform)
  if (n_chunks_max) {      // block-uniform: barriers inside
    unsigned long long tm[kCellsPerThread];
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < kCellsPerThread; ++q) {
      const unsigned lc = lt + unsigned(q) * 256u;
      const bool t = work && n_chunks && u.do_update && s_key[lc] != kEmptyKey &&
                     int(tr * kTS + (lc & 31u)) < G.s_rows && int(tc * kTC + (lc >> 5)) < G.s_cols;
      tm[q] = __ballot(t);
      mine += unsigned(__popcll(tm[q]));  // (wave total)
    }
    if ((lt & 63u) == 0u) s_off[256u + (lt >> 6)] = mine;
    __syncthreads();
    unsigned base = 0;
    for (unsigned w = 0; w < (lt >> 6); ++w) base += s_off[256u + w];
    n_touched = uni(s_off[256u] + s_off[257u] + s_off[258u] + s_off[259u]);
    const unsigned long long below = (1ull << (lt & 63u)) - 1ull;
#pragma unroll
    for (int q = 0; q < kCellsPerThread; ++q) {
      if ((tm[q] >> (lt & 63u)) & 1ull) s_tlist[base + unsigned(__popcll(tm[q] & below))] = uint16_t(lt + unsigned(q) * 256u);
      base += unsigned(__popcll(tm[q]));
    }
    __syncthreads();
  }
  if (work) {
    // touched cells: one per thread (kCellBatch per pass when the tile holds more than 256)
#pragma unroll 1
    for (unsigned j0 = lt; j0 < n_touched; j0 += 256u * kCellBatch) {
      unsigned o_[kCellBatch];
      bool on_[kCellBatch], strip_[kCellBatch];
      unsigned long long key_[kCellBatch];
      uint32_t zm_[kCellBatch], zsw_[kCellBatch], im_[kCellBatch], izw_[kCellBatch], fst_[kCellBatch], rgb_[kCellBatch];
      float var_[kCellBatch], sint_[kCellBatch];
      typename POLICY::State stt_[kCellBatch];
#pragma unroll
      for (int b = 0; b < kCellBatch; ++b) {
        const unsigned j = j0 + unsigned(b) * 256u;
        on_[b] = j < n_touched;
        o_[b] = 0u; strip_[b] = false; key_[b] = kEmptyKey; zm_[b] = 0u; zsw_[b] = 0u; im_[b] = 0u; izw_[b] = 0u;
        fst_[b] = 0u; rgb_[b] = 0u; var_[b] = 0.0f; sint_[b] = nanv;  // (CellObservation defaults: var 0)
        if (!on_[b]) continue;
        const unsigned lc = s_tlist[j];
        const int sr = int(tr * kTS + (lc & 31u)), sc = int(tc * kTC + (lc >> 5));
        o_[b] = unsigned(sc) * unsigned(G.s_rows) + unsigned(sr);
        strip_[b] = strips && (in_cleared_strip(sr + G.s_r0, u.E.sr, u.C.shr, G.rows) ||
                                 in_cleared_strip(sc + G.s_c0, u.E.sc, u.C.shc, G.cols));
        key_[b] = s_key[lc];
        zm_[b] = s_zmax[lc];
        if (has_int) im_[b] = s_imax[lc];
        if (rare_tile) {  // (written by memory-side atomics of this block: read past the L1 / L2 copies)
          zsw_[b] = __hip_atomic_load(&g_zs[lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (has_int) {
            izw_[b] = __hip_atomic_load(&g_izs[lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fst_[b] = __hip_atomic_load(&g_first[lc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        const uint32_t wl = uint32_t(key_[b]);
        if (wl != kNoWinner) var_[b] = Q.rec[wl >> 1].var;
        if (has_col) rgb_[b] = Q.rec[s_last[lc] - 1u].rgb;
        if (strip_[b]) {
          POLICY::set_nan(stt_[b]);
        } else {
          POLICY::load(L, o_[b], stt_[b]);
          if (has_int) sint_[b] = L.intensity[o_[b]];
        }
      }
#pragma unroll
      for (int b = 0; b < kCellBatch; ++b) {
        if (!on_[b]) continue;
        const unsigned o = o_[b];
        if (strip_[b]) {  // NaN in EVERY layer (GridMap::move); the estimator's record is rewritten below
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
        const uint32_t wl = uint32_t(key_[b]);
        const float min_z = wl != kNoWinner ? signed_value(uint32_t(key_[b] >> 32), wl & 1u) : kFltMax;
        const float max_z = zm_[b] ? signed_value(zm_[b], zsw_[b] & 1u) : -kFltMax;
        if (A.ras_z) A.ras_z[o] = min_z;
        POLICY::update(L, o, stt_[b], min_z, var_[b], max_z);
        if (J.write_obst) L.obstacle[o] = (max_z > min_z) ? max_z : nanv;
        if (has_int) {
          const float obs = (fst_[b] & 1u) ? nanv  // first point NaN -> stays NaN (elevation_mapping.cpp:73-79)
                                           : signed_value(im_[b], izw_[b] & 1u);
          if (isnan(sint_[b]) || obs > sint_[b]) L.intensity[o] = obs;
        }
        if (has_col) reinterpret_cast<uint32_t*>(L.color)[o] = rgb_[b] & 0x00FFFFFFu;
      }
    }
    FDM_PHASE(2);  // touched cells updated
    // untouched cells: stores only
    if (obst_tile || strips) {
#pragma unroll
      for (int q = 0; q < kCellsPerThread; ++q) {
        const unsigned lc = lt + unsigned(q) * 256u;
        const int sr = int(tr * kTS + (lc & 31u)), sc = int(tc * kTC + (lc >> 5));
        if (sr >= G.s_rows || sc >= G.s_cols) continue;
        if (n_chunks && u.do_update && s_key[lc] != kEmptyKey) continue;  // touched: done above
        const unsigned o = unsigned(sc) * unsigned(G.s_rows) + unsigned(sr);
        bool in_strip = false;
        if (strips) {
          in_strip = in_cleared_strip(sr + G.s_r0, u.E.sr, u.C.shr, G.rows) ||
                     in_cleared_strip(sc + G.s_c0, u.E.sc, u.C.shc, G.cols);
          if (in_strip) {
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
        if (obst_tile && !in_strip) L.obstacle[o] = nanv;  // map_.clear(obstacle), elevation_mapping.cpp:144-146
      }
    }
  }
  // bookkeeping: the chunk list is consumed, the tile remembers who touched it last
  if (lt == 0 && tile_ok) {
    A.upd_part[tile] = n_touched;
    if (n_chunks) {
      Q.desc[size_t(tile) * Q.stride] = 0ull;
      if (u.do_update && J.set_stamp) A.stamp[tile] = J.scan_no;
    }
  }
}

// One 256-thread block looks after `span` tiles (span = 1 on maps whose tile count fills the chip by itself —
// the block then reads its tile's whole descriptor row, count included, in ONE round trip — and 32 on very
// large maps where nearly every tile is idle: the first wavefront reads the 32 chunk counts / stamps in one
// round trip and only the live tiles are visited).
template <typename POLICY, int BLOCK, bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ void tupdate_body(
    const ScanParams& P, const GeomConst& G, const TileGrid& TG, DevState* __restrict__ st,
    const typename POLICY::Layers& L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool& Q, const TileAux& A, const unsigned span, unsigned char* dyn_lds, const unsigned bid) {
  static_assert(BLOCK == 256, "one tile group per block");
  __shared__ unsigned long long s_live, s_ob;
  __shared__ unsigned s_nch[64];
  __shared__ unsigned s_rare;
  const unsigned lt = threadIdx.x;
  // slot q of block i is tile i + q * n_groups: the tiles a scan touches are neighbours in the map (and in tile
  // order), so consecutive slots would put all of them into a few blocks that then walk them one after the
  // other (configs[4] on one GPU: 209 us); strided, the live tiles spread over all blocks
  const unsigned n_groups = (TG.n_tiles + span - 1u) / span;
  const unsigned first = bid;

  // round trip 1: the tile's descriptor row (span 1) or the chunk counts of the block's tiles, the
  // stamps, and the scan context
  unsigned long long d0 = 0ull;
  if (span == 1u && first < TG.n_tiles && lt < Q.stride) d0 = Q.desc[size_t(first) * Q.stride + lt];
  TileCtx u;
  make_tile_ctx(P, st, u, bid == 0 && threadIdx.x == 0);
  if (lt < 64u) {
    bool live = false, ob = false;
    unsigned nch = 0;
    if (lt < span) {
      const unsigned tile = first + lt * n_groups;
      if (first < n_groups && tile < TG.n_tiles) {  // (a surplus block of the grid owns nothing)
        nch = span == 1u ? unsigned(d0) : unsigned(Q.desc[size_t(tile) * Q.stride]);
        const unsigned stamp = A.stamp[tile];
        ob = u.do_update && (nch != 0u || stamp == u.ob_scan);
        live = nch != 0u || ob || (u.strips && tile_hits_strips(u, G, TG, tile));
        if (!live) A.upd_part[tile] = 0u;
      }
      s_nch[lt] = nch;
    }
    const unsigned long long m = __ballot(live), mo = __ballot(ob);
    if (lt == 0) { s_live = m; s_ob = mo; s_rare = 0u; }
  }
  __syncthreads();
  FDM_PHASE(0);  // round trip 1 (descriptor row, stamps, context) back
  unsigned live_lo = uni(unsigned(s_live)), live_hi = uni(unsigned(s_live >> 32));
  const unsigned ob_lo = uni(unsigned(s_ob)), ob_hi = uni(unsigned(s_ob >> 32));
  while (live_lo | live_hi) {  // block-uniform walk over the live slots
    const unsigned q = live_lo ? unsigned(__ffs(int(live_lo))) - 1u : 32u + unsigned(__ffs(int(live_hi))) - 1u;
    if (q < 32u) live_lo &= live_lo - 1u; else live_hi &= live_hi - 1u;
    const unsigned tile = first + q * n_groups;
    const unsigned nch = uni(s_nch[q]);
    unsigned long long dq = d0;
    if (span != 1u) {
      dq = 0ull;
      if (lt <= nch && lt < Q.stride) dq = Q.desc[size_t(tile) * Q.stride + lt];
    }
    const bool obst_tile = ((q < 32u ? ob_lo >> q : ob_hi >> (q - 32u)) & 1u) != 0u;
    const TileJob J{P.scan_no, P.dbg_upd, true, true};
    tupdate_tile<POLICY, BLOCK, HAS_INT, HAS_COL>(J, G, TG, u, L, all_layers, n_layers, Q, A, dyn_lds, tile, true, nch,
                                                  nch, dq, obst_tile, lt, first, &s_rare);
    __syncthreads();
    if (lt == 0) s_rare = 0u;  // (read only behind barriers inside tupdate_tile)
  }
}

template <typename POLICY, bool HAS_INT, bool HAS_COL>
__global__ __launch_bounds__(256, FDM_UPD_WAVES) void k_tupdate(
    const ScanParams P, const GeomConst G, const TileGrid TG, DevState* __restrict__ st,
    const typename POLICY::Layers L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool Q, const TileAux A, unsigned span) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  tupdate_body<POLICY, 256, HAS_INT, HAS_COL>(P, G, TG, st, L, all_layers, n_layers, Q, A, span, dyn_lds, blockIdx.x);
}

// update of scan t + bin of scan t+1 in one launch (the pools are double-buffered by scan parity)
template <typename POLICY, bool HAS_INT, bool HAS_COL, int THREADS, bool LEAN, int VER = 2>
__global__ __launch_bounds__(THREADS, FDM_UPD_WAVES) void k_tupdate_tbin(
    const ScanParams Pu, const GeomConst G, const TileGrid TG, DevState* __restrict__ st,
    const typename POLICY::Layers L, float* const* __restrict__ all_layers, int n_layers,
    const TilePool Qu, const TileAux A, unsigned span, unsigned upd_blocks, const ScanParams Pb,
    const ScanInputs Ib, const Scratch Sb, const TilePool Qb, int32_t* __restrict__ cell_ids) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  // Update blocks first: they are short chains of dependent memory round trips that barely use the
  // vector units, so they get going at once and the bin blocks (arithmetic-bound) fill the chip behind them
  // (C4, final kernels: 35.3 us against 39.5 with the two kinds interleaved in proportion, and 37.7-40.1 with a
  // heavy-tiles-first square ahead of an interleaved rest — bin blocks that start early slow the latency-bound
  // tile chains down by more than they gain).
  const unsigned u0 = blockIdx.x < upd_blocks ? blockIdx.x : upd_blocks;
  const unsigned u1 = blockIdx.x < upd_blocks ? blockIdx.x + 1u : upd_blocks;
  const unsigned long long t0 = A.timeline ? wall_clock64() : 0ull;
#if FDM_MB_PHASES
  if (threadIdx.x == 0) { g_phase[0] = g_phase[1] = g_phase[2] = unsigned(t0); }
#endif
  if (u1 > u0)
    tupdate_body<POLICY, THREADS, HAS_INT, HAS_COL>(Pu, G, TG, st, L, all_layers, n_layers, Qu, A, span, dyn_lds, u0);
  else {
    TbinRing H(Pb, st);
    if constexpr (VER == 2)
      tbin2_body<HAS_INT, HAS_COL, LEAN>(Pb, G, TG, H, Ib, Sb, Sb.bin_part, Qb, cell_ids, dyn_lds, blockIdx.x - u0);
    else
      tbin_body<HAS_INT, HAS_COL, THREADS, LEAN>(Pb, G, TG, H, Ib, Sb, Sb.bin_part, Qb, cell_ids, dyn_lds, blockIdx.x - u0);
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
