// fdm_tbin2.hpp — the bin half of the large-scan pipeline, second edition (round 5).
//
// Same contract as tbin_body (fdm_tiled.hpp): a block of 1024 consecutive points becomes observation records grouped
// by map tile in the block's own pool region + one chunk descriptor per (block, tile).  What changed is the cost:
// the first edition ran 1 125 vector + 484 scalar instructions per wavefront and 13 barriers, and the launch is
// ISSUE-bound (scripts/ubench/valu_issue*.hip: a plain VOP2 add / mul / logic instruction costs a SIMD 2 cycles, a
// compare, select, VOP3, conversion, fp64 or scalar instruction 4, a ds_bpermute 24) — so this edition is built
// around instruction count:
//   * no register run merge and no branchy claim loop: all four points of a thread probe the table with ONE batch
//     of four ds_cmpst (a lane re-probes in a loop only on a real collision), then three LDS atomics each;
//   * records are numbered when their table slot is CLAIMED (ballot + one LDS add per wavefront) — the compaction
//     scan over the 1 024 slots, its shuffles and two barriers are gone; a block's records are in claim order, which
//     is as good as any: only the order of BLOCKS carries "the first point wins" (fdm_tiled.hpp);
//   * tiles are numbered the same way when their tile-table entry is claimed, so the exclusive scan over the tile
//     counts is ONE wavefront scanning a list of ~16 entries with DPP adds instead of 256 threads scanning 1 024
//     table entries through ds_bpermute shuffles and three barriers;
//   * the RARE values (a -0.0, a zero or NaN intensity, a non-finite or FLT_MAX z) are detected before the fold; a
//     block that holds one restarts in the first edition's body, which carries the order-of-first-occurrence
//     bookkeeping.  The fast path therefore folds plain ord(z) words: no canonical zeros, no sign bits, no validity
//     selects.
// Six barriers.  LDS as before (22 B per point + the u16 list).
// Included by fdm_tiled.hpp between tbin_body and the kernels: not a header of its own.

namespace fdm {

// set bits of `m` below this lane
__device__ __forceinline__ unsigned lane_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u));
}
// inclusive wave64 prefix sum, six DPP adds (row_shr 1 2 4 8 inside the rows of 16, then the row totals)
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));  // row_shr:1
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));  // row_shr:2
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));  // row_shr:4
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));  // row_shr:8
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
  return v;
}

// The flush of `ROUNDS` x 256 records (ROUNDS = 1: a block with at most 256 distinct cells — every block of a real
// scan; 4: anything).  Table memory is reused as the tile table once the records sit in registers.
template <bool HAS_INT, bool HAS_COL, int ROUNDS, class PT>
__device__ __forceinline__ void tbin2_flush(const PT& P, const ScanInputs& I, const TilePool& Q, const unsigned b0,
                                            const unsigned n_rec, uint32_t* const h_cell, unsigned long long* const h_key,
                                            uint32_t* const h_zmax, uint32_t* const h_imax, uint32_t* const h_last,
                                            uint16_t* const s_list, unsigned* const s_ntile) {
  constexpr int kSlots = 1024, kSlotBits = 10;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // (b) record j leaves the table for registers; sigma_z^2 of its winning point (re-read from L2)
  uint32_t c_[ROUNDS], kz_[ROUNDS], kw_[ROUNDS], zm_[ROUNDS], im_[ROUNDS], col_[ROUNDS];
  float var_[ROUNDS];
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    c_[q] = kEmptyCell; kz_[q] = 0u; kw_[q] = kNoWinner; zm_[q] = 0u; im_[q] = 0u; var_[q] = 0.0f; col_[q] = 0u;
    if (ROUNDS > 1 && unsigned(q * 256) >= n_rec) continue;  // block-uniform
    const unsigned j = threadIdx.x + unsigned(q * 256);
    if (j >= n_rec) continue;
    const unsigned slot = s_list[j];
    c_[q] = h_cell[slot];  // tile << 10 | cell in tile
    const unsigned long long k64 = h_key[slot];
    kz_[q] = uint32_t(k64 >> 32);
    kw_[q] = uint32_t(k64);
    zm_[q] = h_zmax[slot];
    if (HAS_INT) im_[q] = h_imax[slot];
    if (HAS_COL) col_[q] = I.rgb[b0 + h_last[slot] - 1u];
    const unsigned gi = b0 + (kw_[q] >> 1);  // (the fast path has no record without a winner)
    if (P.has_var) var_[q] = I.var[gi];
    else if (P.integrate_mode) var_[q] = sigma_z2(P, I.x[gi], I.y[gi], I.z[gi]);
  }
  __syncthreads();  // every record has left the table
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
    const uint32_t tile = c_[q] >> 10;
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
  // leave AFTER the barrier, the returning add on the tile's row in flight behind the record stores (a barrier waits
  // for outstanding memory operations).  Tiles beyond 64: add and descriptor at once.
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
        const unsigned slot = atomicAdd(reinterpret_cast<unsigned*>(Q.desc + size_t(tile) * Q.stride), 1u);
        Q.desc[size_t(tile) * Q.stride + 1u + slot] = (unsigned long long)(b0 + off) | ((unsigned long long)n << 32);
      }
      base += uni(unsigned(__builtin_amdgcn_readlane(int(inc), 63)));
    }
  }
  __syncthreads();
  unsigned t0_slot = 0u;
  if (wave == 0u && t0_n)
    t0_slot = atomicAdd(reinterpret_cast<unsigned*>(Q.desc + size_t(t0_tile) * Q.stride), 1u);
  // (e) the records, grouped by tile, into the block's own region of the pool
#pragma unroll
  for (int q = 0; q < ROUNDS; ++q) {
    if (c_[q] == kEmptyCell) continue;
    const uint32_t pos = b0 + t_cnt[th_[q]] + rk_[q];
    TileRec r;
    r.key = ((unsigned long long)kz_[q] << 32) | ((pos << 1) | (kw_[q] & 1u));
    r.zmax = zm_[q];
    r.imax = HAS_INT ? im_[q] : 0u;
    r.cell = c_[q] & 1023u;
    r.var = var_[q];
    r.rgb = HAS_COL ? col_[q] : 0u;
    r.pad = 0u;
    uint4* const dst = reinterpret_cast<uint4*>(Q.rec + pos);
    const uint4* const src = reinterpret_cast<const uint4*>(&r);
    dst[0] = src[0];
    dst[1] = src[1];
  }
  if (wave == 0u && t0_n)  // the descriptors, once the returning adds are back
    Q.desc[size_t(t0_tile) * Q.stride + 1u + t0_slot] = (unsigned long long)(b0 + t0_off) | ((unsigned long long)t0_n << 32);
}

template <bool HAS_INT, bool HAS_COL, bool LEAN, class PT, class HOOK>
__device__ __forceinline__ void tbin2_body(const PT& P, const GeomConst& G, const TileGrid& TG, HOOK& H,
                                           const ScanInputs& I, const Scratch& S,
                                           unsigned long long* __restrict__ bin_part, const TilePool& Q,
                                           int32_t* __restrict__ cell_ids, unsigned char* lds, const unsigned bid) {
  constexpr int THREADS = 256;
  constexpr int kSlots = THREADS * 4;
  constexpr int kSlotBits = 10;
  uint32_t* const h_cell = reinterpret_cast<uint32_t*>(lds);
  unsigned long long* const h_key = reinterpret_cast<unsigned long long*>(h_cell + kSlots);
  uint32_t* const h_zmax = h_cell + 3 * kSlots;
  uint32_t* const h_imax = h_zmax + kSlots;
  uint32_t* const h_last = h_zmax + (HAS_INT ? 2 : 1) * kSlots;
  uint16_t* const s_list = reinterpret_cast<uint16_t*>(h_zmax + (1 + (HAS_INT ? 1 : 0) + (HAS_COL ? 1 : 0)) * kSlots);
  __shared__ DevCand s_cand2;
  __shared__ unsigned s_cnt2[4], s_rare2[4], s_any2[4];
  __shared__ unsigned s_nrec, s_ntile;

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
  {  // table initialisation, 16 bytes per LDS store
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu), zero = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(h_cell)[threadIdx.x] = ones;
    reinterpret_cast<uint4*>(h_key)[2 * threadIdx.x] = ones;
    reinterpret_cast<uint4*>(h_key)[2 * threadIdx.x + 1] = ones;
    reinterpret_cast<uint4*>(h_zmax)[threadIdx.x] = zero;
    if (HAS_INT) reinterpret_cast<uint4*>(h_imax)[threadIdx.x] = zero;
    if (HAS_COL) reinterpret_cast<uint4*>(h_last)[threadIdx.x] = zero;
  }
  if (threadIdx.x == 0) { s_nrec = 0u; s_ntile = 0u; }
  H.begin();

  int cells[4];
  float xm[4], ym[4], zs[4];
  bool pass[4];
  unsigned n_pass = 0, n_in = 0;
  bool any_glob = false;
  tbin_prep<HAS_INT, THREADS, LEAN, true>(P, S, bid, xs, ys, zin, xm, ym, zs, pass, n_pass);
  // rare values among the block's points (see the header): z = +-0, NaN, +-inf, +-FLT_MAX; the same for the intensity
  {
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
    if (lane == 0u) s_rare2[wave] = mr ? 1u : 0u;
  }
  const DevCand cand = H.finish(G, &s_cand2, bid, n_pass);  // contains the __syncthreads
  if (uni(s_rare2[0] | s_rare2[1] | s_rare2[2] | s_rare2[3])) {  // block-uniform: the first edition knows what to do
#ifndef FDM_TBIN2_NO_RESTART  // (measurement builds: the fast path's own resource usage)
    tbin_body<HAS_INT, HAS_COL, THREADS, LEAN>(P, G, TG, H, I, S, bin_part, Q, cell_ids, lds, bid);
#endif
    return;
  }
  FDM_PHASE(0);
  tbin_points<HAS_INT, THREADS, LEAN, true>(P, G, TG, cell_ids, cand, bid, xm, ym, pass, cells, n_in, any_glob);
  FDM_PHASE(1);
  {  // statistics of this wavefront: surviving points / points in the owned window (ballots: no shuffles)
    unsigned np = 0u, ni = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      np += unsigned(__popcll(__ballot(pass[j])));
      ni += unsigned(__popcll(__ballot(cells[j] >= 0)));
    }
    const unsigned long long mg = __ballot(any_glob);
    if (lane == 0u) { s_cnt2[wave] = np | (ni << 16); s_any2[wave] = mg ? 1u : 0u; }
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
      const uint32_t oz = ord(zs[j]);
      atomicMin(&h_key[hs[j]], ((unsigned long long)oz << 32) | (li << 1));
      atomicMax(&h_zmax[hs[j]], oz);
      if (HAS_INT) atomicMax(&h_imax[hs[j]], ord(vs[j]));
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
    for (int w = 0; w < 4; ++w) { np += s_cnt2[w] & 0xFFFFu; ni += s_cnt2[w] >> 16; ag |= s_any2[w]; }
    if (np) H.note_pass();
    if (ag) H.note_inside();
    bin_part[bid] = (unsigned long long)np | ((unsigned long long)ni << 32);
  }
  FDM_PHASE(2);
  const unsigned n_rec = uni(s_nrec);
  if (n_rec <= 256u)
    tbin2_flush<HAS_INT, HAS_COL, 1>(P, I, Q, b0, n_rec, h_cell, h_key, h_zmax, h_imax, h_last, s_list, &s_ntile);
  else
    tbin2_flush<HAS_INT, HAS_COL, 4>(P, I, Q, b0, n_rec, h_cell, h_key, h_zmax, h_imax, h_last, s_list, &s_ntile);
}

}  // namespace fdm
