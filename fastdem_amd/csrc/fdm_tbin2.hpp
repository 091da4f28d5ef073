// fdm_tbin2.hpp — the bin half of the large-scan pipeline (second edition, round 5).
//
// A block of 1 024 consecutive points becomes observation records grouped by map tile in the block's own pool region
// + one chunk descriptor per (block, tile).  The launch is ISSUE-bound (scripts/ubench/valu_issue*.hip: a plain VOP2
// add / mul / logic instruction costs a SIMD 2 cycles; a compare, select, VOP3, conversion, fp64 or scalar instruction
// 4; a ds_bpermute 24), so the body is built around instruction count (the first edition ran 1 125 vector + 484
// scalar instructions per wavefront and 13 barriers):
//   * no register run merge and no branchy claim loop: all four points of a thread probe the table with ONE batch
//     of four ds_cmpst (a lane re-probes in a loop only on a real collision), then three LDS atomics each;
//   * records are numbered when their table slot is CLAIMED (ballot + one LDS add per wavefront) — no compaction scan
//     over the 1 024 slots; a block's records are in claim order, which is as good as any: only the order of BLOCKS
//     carries "the first point wins" (fdm_tiled.hpp);
//   * tiles are numbered the same way when their tile-table entry is claimed, so the exclusive scan over the tile
//     counts is ONE wavefront scanning a list of ~30 entries with DPP adds;
//   * the RARE values (a -0.0, a zero or NaN intensity, a non-finite or FLT_MAX z) are detected before the fold.  A
//     block without one (every block of a real scan) folds plain ord(z) words: no canonical zeros, no sign bits, no
//     validity selects.  A block with one folds the general words and walks its points a second time, after the
//     records have left the table, for the order of first occurrences (tbin2_rare_walk).
// Six barriers (nine in a rare block).  LDS: 22 B per point + the u16 list.
// Included by fdm_tiled.hpp: not a header of its own.

namespace fdm {

// Second walk of a rare block: per table slot the (order << 1 | is -0) of the first zero-valued z, of the first
// zero-valued intensity, and (order << 1 | is NaN) of the first point.  The points are re-read (L2) and taken through
// the arithmetic again — the fast path keeps none of it in registers for this.
template <bool HAS_INT, bool LEAN, class PT>
__device__ __forceinline__ void tbin2_rare_walk(const PT& P0, const GeomConst& G0, const TileGrid& TG0, const ScanInputs& I0,
                                                const DevCand& cand, const unsigned i0, const unsigned l0,
                                                const uint32_t* const h_cell, uint32_t* const r_zs, uint32_t* const r_izs,
                                                uint32_t* const r_first) {
  constexpr int kSlots = 1024, kSlotBits = 10;
  const unsigned lz = opaque_zero();
  const PT& P = late<1>(P0, lz);
  const GeomConst& G = late<1>(G0, lz);
  const TileGrid& TG = late<1>(TG0, lz);
  const ScanInputs& I = late<1>(I0, lz);
  const bool drop_nf = LEAN ? false : P.drop_nonfinite != 0;
#pragma unroll 1
  for (unsigned j = 0; j < 4u; ++j) {
    if (i0 + j >= P.n) break;
    float x = I.x[i0 + j], y = I.y[i0 + j], z = I.z[i0 + j];
    const float v = HAS_INT ? I.intensity[i0 + j] : 0.0f;
    const bool exists = !drop_nf || (isfinite(x) && isfinite(y) && isfinite(z));
    if (!(preprocess_point(P, x, y, z) && exists)) continue;
    int lin;
    const int cell = owned_tcell(x, y, cand, G, TG, lin);
    if (cell < 0) continue;
    uint32_t h = (uint32_t(cell) * 2654435761u) >> (32 - kSlotBits);
    int probes = 0;  // (the cell is present: the fold put it there; the bound only guards the loop)
    while (h_cell[h] != uint32_t(cell) && probes < kSlots) { h = (h + 1) & (kSlots - 1); ++probes; }
    if (probes >= kSlots) continue;
    const uint32_t li = l0 + j;
    if (z == 0.0f) atomicMin(&r_zs[h], (li << 1) | (__float_as_uint(z) == 0x80000000u ? 1u : 0u));
    if (HAS_INT) {
      if (v == 0.0f) atomicMin(&r_izs[h], (li << 1) | (__float_as_uint(v) == 0x80000000u ? 1u : 0u));
      atomicMin(&r_first[h], (li << 1) | (isnan(v) ? 1u : 0u));
    }
  }
}

// The flush of `ROUNDS` x 256 records (ROUNDS = 1: a block with at most 256 distinct cells — every block of a real
// scan; 4: anything).  Table memory is reused as the tile table once the records sit in registers.
template <bool HAS_INT, bool HAS_COL, bool LEAN, int ROUNDS, class PT>
__device__ __forceinline__ void tbin2_flush(const PT& P0, const GeomConst& G, const TileGrid& TG, const ScanInputs& I0,
                                            const TilePool& Q0, const DevCand& cand, const bool rare_block, const unsigned b0,
                                            const unsigned n_rec, uint32_t* const h_cell, unsigned long long* const h_key,
                                            uint32_t* const h_zmax, uint32_t* const h_imax, uint32_t* const h_last,
                                            uint16_t* const s_list, unsigned* const s_ntile) {
  constexpr int kSlots = 1024, kSlotBits = 10;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned lz = opaque_zero();
  const PT& P = late<2>(P0, lz);
  const ScanInputs& I = late<2>(I0, lz);
  const TilePool& Q = late<2>(Q0, lz);
  // (b) record j leaves the table for registers; sigma_z^2 of its winning point (re-read from L2)
  uint32_t c_[ROUNDS], kz_[ROUNDS], zm_[ROUNDS], im_[ROUNDS], col_[ROUNDS], fl_[ROUNDS];
  float var_[ROUNDS];
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    c_[q] = kEmptyCell; kz_[q] = 0u; zm_[q] = 0u; im_[q] = 0u; var_[q] = 0.0f; col_[q] = 0u; fl_[q] = 0u;
    if (ROUNDS > 1 && unsigned(q * 256) >= n_rec) continue;  // block-uniform
    const unsigned j = threadIdx.x + unsigned(q * 256);
    if (j >= n_rec) continue;
    const unsigned slot = s_list[j];
    c_[q] = h_cell[slot];  // tile << 8 | cell in tile
    const unsigned long long k64 = h_key[slot];
    kz_[q] = uint32_t(k64 >> 32);
    const uint32_t wl = uint32_t(k64);  // winner: order << 1 | its z is -0, or kNoWinner (rare blocks only)
    zm_[q] = h_zmax[slot];
    if (HAS_INT) im_[q] = h_imax[slot];
    if (HAS_COL) col_[q] = I.rgb[b0 + h_last[slot] - 1u];
    if (rare_block && wl == kNoWinner) {  // (CellObservation default 0, elevation_mapping.hpp:26-34)
      fl_[q] = kRecNoWin;
    } else {
      if (rare_block && (wl & 1u)) fl_[q] = kRecMinNeg;
      const unsigned gi = b0 + (wl >> 1);
#ifndef FDM_X_NOSIGMA
      if (P.has_var) var_[q] = I.var[gi];
      else if (P.integrate_mode) var_[q] = sigma_z2(P, I.x[gi], I.y[gi], I.z[gi]);
#else
      var_[q] = __uint_as_float(gi);
#endif
    }
  }
  __syncthreads();  // every record has left the table (the cell words stay: the rare walk probes them)
  if (rare_block) {
    // Rare path: some point of the block is a -0.0 or has a NaN intensity, so the order of first occurrences matters.
    uint32_t* const r_zs = reinterpret_cast<uint32_t*>(h_key);  // (the key's memory)
    uint32_t* const r_izs = r_zs + kSlots;
    uint32_t* const r_first = h_zmax;
    for (int k = threadIdx.x; k < kSlots; k += 256) {
      r_zs[k] = 0xFFFFFFFFu; r_izs[k] = 0xFFFFFFFFu; r_first[k] = 0xFFFFFFFFu;
    }
    __syncthreads();
    tbin2_rare_walk<HAS_INT, LEAN>(P0, G, TG, I0, cand, b0 + threadIdx.x * 4u, threadIdx.x * 4u, h_cell, r_zs, r_izs, r_first);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < ROUNDS; ++q) {
      if (c_[q] == kEmptyCell) continue;
      const unsigned slot = s_list[threadIdx.x + unsigned(q * 256)];
      const uint32_t zsw = r_zs[slot];
      if (zsw != 0xFFFFFFFFu && (zsw & 1u)) fl_[q] |= kRecZNeg;
      if (HAS_INT) {
        const uint32_t izw = r_izs[slot];
        if (izw != 0xFFFFFFFFu && (izw & 1u)) fl_[q] |= kRecINeg;
        if (r_first[slot] & 1u) fl_[q] |= kRecNanFirst;
      }
    }
    __syncthreads();
  }
  // (c) the key array's memory becomes the block's TILE table: tile id -> how many of the block's cells, later -> offset
  uint32_t* const t_tile = reinterpret_cast<uint32_t*>(h_key);  // [kSlots]
  uint32_t* const t_cnt = t_tile + kSlots;                      // [kSlots]
  uint16_t* const t_list = s_list;                              // tile-table entries in claim order
  {
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), zero = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(t_tile)[threadIdx.x] = ones;
    reinterpret_cast<uint4*>(t_cnt)[threadIdx.x] = zero;
  }
  __syncthreads();
  uint32_t th_[ROUNDS], rk_[ROUNDS];
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    th_[q] = 0u; rk_[q] = 0u;
    if (ROUNDS > 1 && unsigned(q * 256) >= n_rec) continue;  // block-uniform
    const bool on = c_[q] != kEmptyCell;
    const uint32_t tile = c_[q] >> kCitBits;
    uint32_t h = (tile * 2654435761u) >> (32 - kSlotBits);
    bool claimed = false;
    if (on) {
      while (true) {  // (at most n_rec <= kSlots distinct tiles: always terminates)
        const uint32_t prev = atomicCAS(&t_tile[h], kEmptyCell, tile);
        claimed = prev == kEmptyCell;
        if (claimed || prev == tile) break;
        h = (h + 1) & (kSlots - 1);
      }
      rk_[q] = atomicAdd(&t_cnt[h], 1u);
    }
    th_[q] = h;
    // tiles in claim order
    const unsigned long long m = __ballot(claimed);
    if (m) {
      unsigned base = 0u;
      if (lane == 0u) base = atomicAdd(s_ntile, unsigned(__popcll(m)));
      base = uni(base);
      if (claimed) t_list[base + lane_rank(m)] = uint16_t(h);
    }
  }
  __syncthreads();
  // (d) wavefront 0: exclusive scan of the tile counts in claim order (count -> offset inside the block's region).
  // The first 64 tiles (every block of a real scan has fewer) keep their entry in registers: their chunk descriptors
  // leave AFTER the barrier, the returning add on the tile's counter in flight behind the record stores (a barrier
  // waits for outstanding memory operations).  Tiles beyond 64: add and descriptor at once.
  const unsigned n_tile = uni(*s_ntile);
  unsigned t0_tile = 0u, t0_n = 0u, t0_off = 0u;
  if (wave == 0u) {
    unsigned base = 0u;
#pragma unroll 1
    for (unsigned p0 = 0u; p0 < n_tile; p0 += 64u) {  // wave-uniform
      const unsigned k = p0 + lane;
      unsigned n = 0u, h = 0u, tile = 0u;
      if (k < n_tile) { h = t_list[k]; n = t_cnt[h]; tile = t_tile[h]; }
      const unsigned inc = wave_scan_incl(n);
      const unsigned off = base + inc - n;
      if (k < n_tile) t_cnt[h] = off;
      if (p0 == 0u) {
        t0_tile = tile; t0_n = n; t0_off = off;
      } else if (k < n_tile) {
        const unsigned slot = atomicAdd(Q.cnt + (size_t(tile) << Q.cnt_shift), 1u);
        Q.desc[size_t(tile) * Q.stride + slot] = (unsigned long long)(b0 + off) | ((unsigned long long)n << 32);
      }
      base += uni(unsigned(__builtin_amdgcn_readlane(int(inc), 63)));
    }
  }
  __syncthreads();
  unsigned t0_slot = 0u;
#ifndef FDM_X_NOATOM
  if (wave == 0u && t0_n) t0_slot = atomicAdd(Q.cnt + (size_t(t0_tile) << Q.cnt_shift), 1u);
#endif
  // (e) the records, grouped by tile, into the block's own region of the pool
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    if (c_[q] == kEmptyCell) continue;
    const uint32_t pos = b0 + t_cnt[th_[q]] + rk_[q];
    *reinterpret_cast<uint4*>(Q.hot + pos) = make_uint4(kz_[q], zm_[q], HAS_INT ? im_[q] : 0u, (c_[q] & kCitMask) | fl_[q]);
    *reinterpret_cast<uint2*>(Q.cold + pos) = make_uint2(__float_as_uint(var_[q]), HAS_COL ? col_[q] : 0u);
  }
  if (wave == 0u && t0_n)  // the descriptors, once the returning adds are back
    Q.desc[size_t(t0_tile) * Q.stride + t0_slot] = (unsigned long long)(b0 + t0_off) | ((unsigned long long)t0_n << 32);
}

template <bool HAS_INT, bool HAS_COL, bool LEAN, class PT, class HOOK>
__device__ __forceinline__ void tbin2_body(const PT& P, const GeomConst& G, const TileGrid& TG, HOOK& H,
                                           const ScanInputs& I, const Scratch& S,
                                           unsigned long long* __restrict__ bin_part, const TilePool& Q,
                                           int32_t* __restrict__ cell_ids, unsigned char* lds, const unsigned bid) {
  constexpr int THREADS = 256;
  constexpr int kSlots = THREADS * 4;  // == points per block: room for every point in its own cell
  constexpr int kSlotBits = 10;
  uint32_t* const h_cell = reinterpret_cast<uint32_t*>(lds);
  // ord(min z) << 32 | order in block << 1 | z is -0, min-reduced: the lowest z, among equals the first point
  unsigned long long* const h_key = reinterpret_cast<unsigned long long*>(h_cell + kSlots);
  uint32_t* const h_zmax = h_cell + 3 * kSlots;  // ord(max z), 0 = none
  uint32_t* const h_imax = h_zmax + kSlots;      // (HAS_INT) ord(max intensity), 0 = none
  uint32_t* const h_last = h_zmax + (HAS_INT ? 2 : 1) * kSlots;  // (HAS_COL) order + 1 of the last point
  uint16_t* const s_list = reinterpret_cast<uint16_t*>(h_zmax + (1 + (HAS_INT ? 1 : 0) + (HAS_COL ? 1 : 0)) * kSlots);
  __shared__ DevCand s_cand;
  __shared__ unsigned s_cnt[4], s_rare[4], s_any[4];
  __shared__ unsigned s_nrec, s_ntile;

  // the point loads go out first: they are in flight while the table is initialised and
  // thread 0 works out the post-move geometry
  const unsigned b0 = bid * unsigned(kSlots);
  const unsigned l0 = threadIdx.x * 4u;
  const unsigned i0 = b0 + l0;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  float xs[4], ys[4], zin[4], vs[4];
  if (i0 + 3 < P.n) {
    const float4 a = *reinterpret_cast<const float4*>(I.x + i0);
    const float4 b = *reinterpret_cast<const float4*>(I.y + i0);
    const float4 c = *reinterpret_cast<const float4*>(I.z + i0);
    xs[0] = a.x; xs[1] = a.y; xs[2] = a.z; xs[3] = a.w;
    ys[0] = b.x; ys[1] = b.y; ys[2] = b.z; ys[3] = b.w;
    zin[0] = c.x; zin[1] = c.y; zin[2] = c.z; zin[3] = c.w;
    if (HAS_INT) {
      const float4 d = *reinterpret_cast<const float4*>(I.intensity + i0);
      vs[0] = d.x; vs[1] = d.y; vs[2] = d.z; vs[3] = d.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = i0 + j < P.n;
      xs[j] = ok ? I.x[i0 + j] : 0.f;
      ys[j] = ok ? I.y[i0 + j] : 0.f;
      zin[j] = ok ? I.z[i0 + j] : 0.f;
      if (HAS_INT) vs[j] = ok ? I.intensity[i0 + j] : 0.f;
    }
  }
  {  // table initialisation, 16 bytes per LDS store (4 slots per thread and array)
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), zero = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(h_cell)[threadIdx.x] = ones;                  // kEmptyCell
    reinterpret_cast<uint4*>(h_key)[2 * threadIdx.x] = ones;               // kEmptyKey (two keys per store)
    reinterpret_cast<uint4*>(h_key)[2 * threadIdx.x + 1] = ones;
    reinterpret_cast<uint4*>(h_zmax)[threadIdx.x] = zero;
    if (HAS_INT) reinterpret_cast<uint4*>(h_imax)[threadIdx.x] = zero;
    if (HAS_COL) reinterpret_cast<uint4*>(h_last)[threadIdx.x] = zero;
  }
  if (threadIdx.x == 0) { s_nrec = 0u; s_ntile = 0u; }
  H.begin();  // (thread 0's state loads leave; the walk follows the transforms below)

  // phase 1: all four points through the arithmetic — first what needs no geometry (both transforms, the crops), in
  // the shadow of thread 0's state read, then the geometry candidate (barrier), then getIndex
  int cells[4];
  float xm[4], ym[4], zs[4];
  bool pass[4];
  unsigned n_pass = 0, n_in = 0;
  bool any_glob = false;
  tbin_prep<HAS_INT, THREADS, LEAN, true>(P, S, bid, xs, ys, zin, xm, ym, zs, pass, n_pass);
  {  // rare values among the block's points: z = +-0, NaN, +-inf, +-FLT_MAX; the same for the intensity
    bool rare = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = i0 + j < P.n;
      const uint32_t a = __float_as_uint(zs[j]) & 0x7FFFFFFFu;
      rare = rare || (live && a - 1u >= 0x7F7FFFFEu);
      if (HAS_INT) {
        const uint32_t av = __float_as_uint(vs[j]) & 0x7FFFFFFFu;
        rare = rare || (live && av - 1u >= 0x7F7FFFFEu);
      }
    }
    const unsigned long long mr = __ballot(rare);
    if (lane == 0u) s_rare[wave] = mr ? 1u : 0u;
  }
  const DevCand cand = H.finish(G, &s_cand, bid, n_pass);  // contains the __syncthreads
  const bool rare_block = uni(s_rare[0] | s_rare[1] | s_rare[2] | s_rare[3]) != 0u;  // block-uniform
  FDM_PHASE(0);  // table initialised, points transformed and cropped, candidate known
  const int dbg = LEAN ? 0 : P.dbg_no_atomics;  // measurement only: leave the kernel after a stage (results are wrong)
  if (dbg == 1) { bin_part[bid] = (zs[0] + zs[1] + zs[2] + zs[3] == 12345.f) ? 1ull : 0x100000001ull; return; }
  {
    const unsigned lz = opaque_zero();
    tbin_points<HAS_INT, THREADS, LEAN, true>(late<4>(P, lz), late<4>(G, lz), late<4>(TG, lz), cell_ids, cand, bid, xm, ym, pass, cells, n_in,
                                              any_glob);
  }
  FDM_PHASE(1);  // index done
  if (dbg == 2) { bin_part[bid] = (cells[0] + cells[1] + cells[2] + cells[3] == 0x7FFFFFF1) ? 1ull : 0x100000001ull; return; }
  {  // statistics of this wavefront: surviving points / points in the owned window (ballots: no shuffles)
    unsigned np = 0u, ni = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      np += unsigned(__popcll(__ballot(pass[j])));
      ni += unsigned(__popcll(__ballot(cells[j] >= 0)));
    }
    const unsigned long long mg = __ballot(any_glob);
    if (lane == 0u) { s_cnt[wave] = np | (ni << 16); s_any[wave] = mg ? 1u : 0u; }
  }

  // ---- fold: four probes per thread in one batch, then the atomics ----
  uint32_t hs[4], prev[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hs[j] = (uint32_t(cells[j]) * 2654435761u) >> (32 - kSlotBits);
    prev[j] = uint32_t(cells[j]);
    if (cells[j] >= 0) prev[j] = atomicCAS(&h_cell[hs[j]], kEmptyCell, uint32_t(cells[j]));
  }
  unsigned long long mc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool live = cells[j] >= 0;
    bool claimed = live && prev[j] == kEmptyCell;
    if (live && !claimed && prev[j] != uint32_t(cells[j])) {  // a real collision: linear probing (rare)
      uint32_t h = hs[j];
      while (true) {
        h = (h + 1) & (kSlots - 1);
        const uint32_t pv = atomicCAS(&h_cell[h], kEmptyCell, uint32_t(cells[j]));
        claimed = pv == kEmptyCell;
        if (claimed || pv == uint32_t(cells[j])) break;
      }
      hs[j] = h;
    }
    mc[j] = __ballot(claimed);
    if (live) {
      const uint32_t li = l0 + unsigned(j);
      const float z = zs[j];
      uint32_t kh = ord(z), kl = li << 1, zmx = kh, imx = HAS_INT ? ord(vs[j]) : 0u;
      if (rare_block) {  // the general words (block-uniform branch)
        const uint32_t oz = ord_canon(z);
        // strict "z < min_z" from FLT_MAX / "z > max_z" from lowest(): NaN, FLT_MAX and beyond never win
        const bool vmin = z < kFltMax;
        kh = vmin ? oz : ord(kFltMax);
        kl = vmin ? (kl | (__float_as_uint(z) == 0x80000000u ? 1u : 0u)) : kNoWinner;
        zmx = (z > -kFltMax) ? oz : 0u;
        if (HAS_INT) {
          const float vv = vs[j];
          imx = ((__float_as_uint(vv) & 0x7FFFFFFFu) > 0x7F800000u) ? 0u : ord_canon(vv);
        }
      }
      atomicMin(&h_key[hs[j]], ((unsigned long long)kh << 32) | kl);
      atomicMax(&h_zmax[hs[j]], zmx);  // (max with 0: no-op)
      if (HAS_INT) atomicMax(&h_imax[hs[j]], imx);
      if (HAS_COL) atomicMax(&h_last[hs[j]], li + 1u);
    }
  }
  {  // the claimed slots become records, numbered per wavefront with one LDS add
    const unsigned c0 = unsigned(__popcll(mc[0])), c1 = unsigned(__popcll(mc[1])), c2 = unsigned(__popcll(mc[2])),
                   c3 = unsigned(__popcll(mc[3]));
    const unsigned tot = c0 + c1 + c2 + c3;
    if (tot) {  // wave-uniform
      unsigned base = 0u;
      if (lane == 0u) base = atomicAdd(&s_nrec, tot);
      base = uni(base);
      if ((mc[0] >> lane) & 1ull) s_list[base + lane_rank(mc[0])] = uint16_t(hs[0]);
      if ((mc[1] >> lane) & 1ull) s_list[base + c0 + lane_rank(mc[1])] = uint16_t(hs[1]);
      if ((mc[2] >> lane) & 1ull) s_list[base + c0 + c1 + lane_rank(mc[2])] = uint16_t(hs[2]);
      if ((mc[3] >> lane) & 1ull) s_list[base + c0 + c1 + c2 + lane_rank(mc[3])] = uint16_t(hs[3]);
    }
  }
  __syncthreads();  // the table and the record list are complete
  if (threadIdx.x == 0) {
    unsigned np = 0, ni = 0, ag = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { np += s_cnt[w] & 0xFFFFu; ni += s_cnt[w] >> 16; ag |= s_any[w]; }
#ifndef FDM_X_NOFLAGS  // (FDM_X_*: measurement builds only, `make variant`)
    if (np) H.note_pass();
    if (ag) H.note_inside();
#endif
    bin_part[bid] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
  FDM_PHASE(2);  // LDS fold done
  if (dbg == 5) return;
  const unsigned n_rec = uni(s_nrec);
  if (n_rec <= 256u)
    tbin2_flush<HAS_INT, HAS_COL, LEAN, 1>(P, G, TG, I, Q, cand, rare_block, b0, n_rec, h_cell, h_key, h_zmax, h_imax, h_last,
                                           s_list, &s_ntile);
  else
    tbin2_flush<HAS_INT, HAS_COL, LEAN, 4>(P, G, TG, I, Q, cand, rare_block, b0, n_rec, h_cell, h_key, h_zmax, h_imax, h_last,
                                           s_list, &s_ntile);
}

}  // namespace fdm
