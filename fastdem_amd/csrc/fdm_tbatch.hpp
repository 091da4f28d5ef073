// fdm_tbatch.hpp — a BATCH of large scans in ONE launch on the per-tile record pools (fdm_tiled.hpp).
//
// Why: one fused launch per 2 M-point scan (k_tupdate_tbin) pays, per LAUNCH, a ramp in which
// 1 444 latency-bound tile groups hold the wave slots while the vector units idle, and a tail in which the last of
// its 2 048 bin blocks run on a draining chip (two quantised rounds on 1 792 slots): 33.7 us per scan for 19 us of
// vector issue.  fdm_engine_integrate_device_batch sees its scans up front, so K consecutive large scans leave as
//
//   k_tbatch = [ update of batch b-1 : a FEW tile groups that pull tiles from a queue | bin of batch b : K x blocks |
//                scouts of batch b+1 : "does scan k have a point that survives the crops?" ]
//
//   bin half     tbin_body as it is (fdm_tiled.hpp) — scan k of the batch writes its records into pool k of the
//                batch's parity.  The geometry scan k is binned against depends on whether the scans before it moved
//                the map, i.e. on whether any of their points survived the crops (fastdem.cpp:138) — device-side
//                data, decided ONE LAUNCH AHEAD by a few scout blocks per scan (tscout_body: first survivor, early
//                exit): when a bin block starts, the flags of the scans ahead of it are final, and thread 0 walks the
//                k moves (GridMap::move arithmetic, move_candidate_fast, in scan order).  No in-launch waiting.
//   update half  `n_groups` (<= 512) groups instead of one per tile, so that the bin blocks own most of the chip
//                from t = 0.  A group pulls tiles off a queue (one atomic per tile, heavy and idle tiles mix), reads
//                the tile's K chunk counts in one round trip and then works through the scans IN ORDER — per cell
//                the reference fixes nothing else (elevation_mapping.cpp:94-125; move() strips and the obstacle
//                clear happen between scans) — with tupdate_tile, the single-scan path's code: fold the scan's
//                records into the LDS image, touched cells, strips.  The next scan's descriptor row is in flight
//                while the current one is folded.  Only the batch's LAST updating scan writes the obstacle layer
//                (every updating scan clears the whole layer first, elevation_mapping.cpp:144-146: what an earlier
//                scan of the batch wrote could never be seen), so a tile is cleared at most once per batch.
//                The LAST group to leave commits the geometry ring (every group has read the ring by then).
//
// The map after a batch is bit-identical to integrating its scans one by one (tests/test_tbatch_gpu.py).
// Algorithmic bytes per scan are what they were.
#pragma once

#include "fdm_multi.hpp"
#include "fdm_tiled.hpp"

namespace fdm {

constexpr int kTBMax = 8;        // scans per launch at most (a power of two: lane = tile slot * kTBMax + scan)
constexpr int kTBShift = 3;
static_assert((1 << kTBShift) == kTBMax && kTBMax <= kMaxBatch, "tile batch size");
constexpr unsigned kTBSpanMax = 64u / unsigned(kTBMax);  // tiles whose counts one wavefront reads in a round trip

struct TBCommon {  // what all scans of a batch share
  float Tbs[16];   // T_base_sensor (one sensor per batch)
  float sp[4];
  float min_sq, max_sq, z_min, z_max;
  int sensor_type, do_move, gate_on_filter, has_var;
  int dbg, pad;
  unsigned long long* timeline;  // measurement only (nullable): {start, end} of every block in 100 MHz ticks
};
struct TBBin {     // bin half: batch b
  unsigned count, scan_no0;
  unsigned first_block[kTBMax + 1];  // bin blocks before scan k
  unsigned n[kTBMax];
  MState* ms;
  const MState* prev;                // the previous batch's state while its update shares this launch (else null)
  unsigned prev_count, stride;       // stride: words per descriptor row
  double robot_x[kTBMax], robot_y[kTBMax];
  const float* px[kTBMax];
  const float* py[kTBMax];
  const float* pz[kTBMax];
  const float* pint[kTBMax];
  const uint32_t* prgb[kTBMax];
  const float* pvar[kTBMax];
  float Twb[kTBMax][16];             // column-major, as ScanParams::Twb
  float R[kTBMax][12];               // 9 used
  TileRec* rec0;                     // pool of scan k: rec0 + k * rec_stride, desc0 + k * desc_stride
  unsigned long long* desc0;
  size_t rec_stride, desc_stride;
  uint32_t* rare;
  unsigned long long* bin_part;      // [bin blocks of the batch]
};
struct TBUpd {     // update half: batch b-1
  unsigned count, scan_no0;
  int do_move, gate_on_filter;
  unsigned span, n_pops, n_groups, stride;
  MState* ms;
  MState* rearm;                     // the state of the batch after next: zeroed by the committing group
  TileRec* rec0;
  unsigned long long* desc0;
  size_t rec_stride, desc_stride;
  uint32_t* rare;
};

// The members tbin_body's arithmetic reads, by name (pointers: an array member copied in device code is spilled).
struct TBScanView {
  const float* Tbs;
  const float* Twb;
  const float* R;
  const float* sp;
  float min_sq, max_sq, z_min, z_max;
  unsigned n;
  int integrate_mode, sensor_type, has_var, drop_nonfinite, dbg_no_atomics;
};

// tbin_body's hook for a scan inside a batch (see TbinRing for the one-scan-per-launch version).
//
// Which of the scans ahead of scan k moved the map is known BEFORE the launch starts: MState::done[j * kLineWords + 1]
// != 0 <=> scan j has a point that survives the crops (fastdem.cpp:138-145), written by the SCOUT blocks of the previous
// launch (or of a small launch of their own ahead of the first batch of a chain) — tscout_body below.  A bin block
// reads the flags ahead of it at its very start, in the shadow of its point loads, and thread 0 walks the k moves
// (GridMap::move arithmetic, move_candidate_fast: the operations the single-scan path performs, in the same order).
// Nothing is published, nothing is polled: no in-launch protocol at all.
// (What was measured on the way, profiles/r04/phases_*.json: an in-launch protocol — every block or wavefront adds /
// stores its "some point survived" to its scan's word, later scans' blocks poll it — makes the 1 536 blocks of a
// launch's FIRST round hit one address at once; a returning atomicAdd per block: bin blocks 17 us instead of 11; one
// fire-and-forget add per wavefront: those blocks waited 30-70 us for 6 000 same-address atomics to drain at the
// memory side; one agent-scope store per wavefront: 90-210 us.)
struct TbinChain {
  const TBBin& B;
  const TBCommon& K;
  DevState* st;
  const unsigned k;
  DevGeom g0;
  DevCand cprev;
  unsigned dprev;  // the previous batch's last scan: its flag
  unsigned fj;     // lane j < k of the first wavefront: scan j's flag
  __device__ __forceinline__ TbinChain(const TBBin& b, const TBCommon& kk, DevState* s, unsigned scan) : B(b), K(kk), st(s), k(scan) {}
  __device__ __forceinline__ bool gated() const { return K.do_move != 0 && K.gate_on_filter != 0; }
  __device__ __forceinline__ void begin() {
    fj = 0u; dprev = 0u;
    if (threadIdx.x < 64u) {
      if (gated() && threadIdx.x < k) fj = B.ms->done[threadIdx.x * kLineWords + 1u];
      if (threadIdx.x == 0) {  // the geometry the chain starts from (in flight while the points are transformed)
        cprev.px = cprev.py = 0.0; cprev.sr = cprev.sc = cprev.shr = cprev.shc = 0;
        if (B.prev) {
          const unsigned pk = B.prev_count - 1u;
          g0 = B.prev->E[pk];
          cprev = B.prev->C[pk];
          dprev = B.prev->done[pk * kLineWords + 1u];
        } else {
          g0 = st->geom[B.scan_no0 & 3u];
        }
      }
    }
  }
  __device__ __forceinline__ DevCand finish(const GeomConst& G, DevCand* s_cand, unsigned lb, unsigned /*n_pass*/) {
    if (threadIdx.x < 64u) {
      const unsigned passmask = gated() ? uni(unsigned(__ballot((threadIdx.x & 63u) < k && fj != 0u))) : 0xFFFFu;
      if (threadIdx.x == 0) {
        DevGeom g = g0;
        if (B.prev) {  // what the update of the previous batch (the other half of this launch) is about to commit
          if (K.do_move && (!K.gate_on_filter || dprev != 0u)) {
            g.px = cprev.px; g.py = cprev.py; g.sr = cprev.sr; g.sc = cprev.sc;
          }
        }
        DevCand c;
        c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
        if (K.do_move) {
          for (unsigned j = 0; j < k; ++j) {  // the moves ahead of scan k (a scan that returned before its move: none)
            if (!((passmask >> j) & 1u)) continue;
            const DevCand m = move_candidate_fast(g, G, B.robot_x[j], B.robot_y[j]);
            g.px = m.px; g.py = m.py; g.sr = m.sr; g.sc = m.sc;
          }
          c = move_candidate_fast(g, G, B.robot_x[k], B.robot_y[k]);
        }
        *s_cand = c;
        if (lb == 0u) { B.ms->E[k] = g; B.ms->C[k] = c; }
      }
    }
    __syncthreads();
    return *s_cand;
  }
  __device__ __forceinline__ void note_inside() { B.ms->inside[k] = 1u; }
  __device__ __forceinline__ void note_pass() {}
};

// ---------------------------------------------------------------------------------------------
// scout blocks: does scan k of the NEXT batch hold a point that survives the crops?  kScoutBlocks blocks per scan walk
// the scan with a grid stride, four points per thread and step, and leave at the first step in which any thread of the
// block sees a survivor — for a real scan that is the first step (one round trip); only a scan without survivors (a
// covered sensor) is read to its end.  The crops are preprocess_point's own float operations (T_base_sensor,
// cropRange, cropZ: crop_impl.hpp:79-96,167-178), so the flag is exactly what the bin blocks would find.
constexpr unsigned kScoutBlocks = 32u;
struct TBScout {   // scans of the batch after this one (count == 0: none)
  unsigned count, pad;
  unsigned n[kTBMax];
  MState* ms;
  const float* px[kTBMax];
  const float* py[kTBMax];
  const float* pz[kTBMax];
};
__device__ __forceinline__ void tscout_body(const TBScout& C, const TBCommon& K, const unsigned k, const unsigned sb) {
  MView V;
  V.Tbs = K.Tbs; V.Twb12 = nullptr; V.R = nullptr; V.sp = K.sp;
  V.min_sq = K.min_sq; V.max_sq = K.max_sq; V.z_min = K.z_min; V.z_max = K.z_max;
  V.sensor_type = K.sensor_type; V.integrate_mode = 1;
  const float* __restrict__ const px = C.px[k];
  const float* __restrict__ const py = C.py[k];
  const float* __restrict__ const pz = C.pz[k];
  const unsigned n = C.n[k];
#pragma unroll 1
  for (unsigned i0 = sb * 1024u; i0 < n; i0 += kScoutBlocks * 1024u) {  // (block-uniform loop)
    bool pass = false;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const unsigned i = i0 + unsigned(h) * 256u + threadIdx.x;
      if (i < n) {
        float x = px[i], y = py[i], z = pz[i], w;
        pass = mcrops(V, x, y, z, w) || pass;
      }
    }
    if (__syncthreads_or(pass ? 1 : 0)) {
      if (threadIdx.x == 0) C.ms->done[k * kLineWords + 1u] = 1u;  // (<= kScoutBlocks plain stores of the same value)
      return;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// update half: group `gid` of U.n_groups.
template <typename POLICY, bool HAS_INT, bool HAS_COL>
__device__ __forceinline__ void tbupdate_body(const TBUpd& U, const GeomConst& G, const TileGrid& TG,
                                              DevState* __restrict__ st, const typename POLICY::Layers& L,
                                              float* const* __restrict__ all_layers, int n_layers, const TileAux& A,
                                              unsigned char* dyn_lds, const unsigned gid) {
  __shared__ unsigned s_nch[64];
  __shared__ unsigned s_stamp[kTBSpanMax];
  __shared__ unsigned long long s_live;
  __shared__ unsigned s_pop, s_rare, s_masks[3];
  const unsigned lt = threadIdx.x, lane = lt & 63u;
  const unsigned count = U.count;
  MState* const ms = U.ms;

  // per-scan context: scan j's geometry in LDS (uniform reads: scalar registers, no vector register held across a tile)
  __shared__ int s_ctx[kTBMax][4];  // start row / column before the scan's move, the move's index shift
  unsigned v_in = 0u;
  bool v_shift = false;
  if (lt < count) {
    const int sr = ms->E[lt].sr, sc = ms->E[lt].sc, shr = ms->C[lt].shr, shc = ms->C[lt].shc;
    s_ctx[lt][0] = sr; s_ctx[lt][1] = sc; s_ctx[lt][2] = shr; s_ctx[lt][3] = shc;
    v_in = ms->inside[lt];
    v_shift = shr != 0 || shc != 0;
  }
  const unsigned ob_prev = uni(st->obst[U.scan_no0 & 3u].scan);  // (the ring is committed by the LAST group to leave)
  if (lt < 64u) {
    // (a scan has a surviving point iff some block said so in its `done` word: final, the bin ran one launch ago)
    const bool v_pass = lane < count && ms->done[lane * kLineWords + 1u] != 0u;
    const bool v_applied = lane < count && U.do_move && (!U.gate_on_filter || v_pass);
    const unsigned long long mu = __ballot(lane < count && v_in != 0u), msx = __ballot(v_applied && v_shift), mp = __ballot(v_pass);
    if (lt == 0) { s_masks[0] = unsigned(mu); s_masks[1] = unsigned(msx); s_masks[2] = unsigned(mp); s_rare = 0u; }
  }
  __syncthreads();
  const unsigned umask = uni(s_masks[0]);      // scans that observed a cell
  const unsigned stripmask = uni(s_masks[1]);  // scans whose move vacated cells
  const unsigned passbits = uni(s_masks[2]);   // scans with a surviving point
  const int s_star = umask ? 31 - __clz(int(umask)) : -1;  // the batch's last updating scan: the one whose obstacle values stay

  unsigned pop = gid;  // (the first pop needs no queue)
#pragma unroll 1
  while (pop < U.n_pops) {  // block-uniform
    // round trip 1: chunk counts of the pop's tiles in every scan of the batch + the tiles' stamps.  Slot q of pop i is
    // tile i + q * n_pops (the tiles a scan touches are neighbours: strided, a pop holds few live ones).  One tile per
    // pop: the descriptor row of the batch's FIRST scan rides in the same round trip (word 0 is its count).
    unsigned long long d_spec = 0ull;
    if (U.span == 1u && lt < U.stride) d_spec = U.desc0[size_t(pop) * U.stride + lt];
    if (lt < 64u) {
      const unsigned q = lt >> kTBShift, k = lt & unsigned(kTBMax - 1);
      const unsigned tile = pop + q * U.n_pops;
      const bool tile_ok = q < U.span && tile < TG.n_tiles;
      unsigned nch = 0u, stamp = 0u;
      if (tile_ok && k < count) nch = unsigned(U.desc0[size_t(k) * U.desc_stride + size_t(tile) * U.stride]);
      if (tile_ok && k == 0u) stamp = A.stamp[tile];
      stamp = unsigned(__shfl(int(stamp), int(lane & ~unsigned(kTBMax - 1))));
      bool live = nch != 0u;
      TileCtx t;
      t.E.sr = s_ctx[k][0]; t.E.sc = s_ctx[k][1];
      t.C.shr = s_ctx[k][2]; t.C.shc = s_ctx[k][3];
      if (tile_ok && k < count) {
        if ((stripmask >> k) & 1u) live = live || tile_hits_strips(t, G, TG, tile);  // scan k's move vacates cells of this tile
        if (int(k) == s_star && stamp == ob_prev) live = true;  // obstacle cells of the last updating scan before the batch
      }
      s_nch[lt] = nch;
      if (k == 0u) s_stamp[q] = stamp;
      const unsigned long long m = __ballot(live);
      if (tile_ok && k == 0u && ((m >> (q << kTBShift)) & unsigned((1 << kTBMax) - 1)) == 0ull) A.upd_part[tile] = 0u;  // idle tile
      if (lt == 0) s_live = m;
    }
    __syncthreads();
    unsigned live_lo = uni(unsigned(s_live)), live_hi = uni(unsigned(s_live >> 32));
#pragma unroll 1
    for (unsigned q = 0; q < U.span; ++q) {  // block-uniform walk over the pop's tiles
      const unsigned sh = q << kTBShift;
      unsigned kbits = ((sh < 32u ? live_lo >> sh : live_hi >> (sh - 32u)) & unsigned((1 << kTBMax) - 1));
      if (!kbits) continue;
      const unsigned tile = pop + q * U.n_pops;
      const unsigned stamp = uni(s_stamp[q]);
      const bool last_seen = ((kbits >> (count - 1u)) & 1u) != 0u;
      const size_t row0 = size_t(tile) * U.stride;
      unsigned long long d = 0ull;
      {  // the first scan's descriptor row (the next one's is fetched while this one is worked on)
        const unsigned k0 = unsigned(__ffs(int(kbits))) - 1u;
        const unsigned n0 = uni(s_nch[sh + k0]);
        if (U.span == 1u && k0 == 0u) d = lt <= n0 ? d_spec : 0ull;
        else if (lt <= n0 && lt < U.stride) d = U.desc0[size_t(k0) * U.desc_stride + row0 + lt];
      }
#pragma unroll 1
      while (kbits) {
        const unsigned k = unsigned(__ffs(int(kbits))) - 1u;
        kbits &= kbits - 1u;
        const unsigned nch = uni(s_nch[sh + k]);
        unsigned long long dn = 0ull;
        if (kbits) {
          const unsigned k2 = unsigned(__ffs(int(kbits))) - 1u;
          const unsigned n2 = uni(s_nch[sh + k2]);
          if (lt <= n2 && lt < U.stride) dn = U.desc0[size_t(k2) * U.desc_stride + row0 + lt];
        }
        TileCtx u;
        u.E.px = u.E.py = 0.0; u.E.pad0 = u.E.pad1 = 0;
        u.C.px = u.C.py = 0.0; u.C.sr = u.C.sc = 0;
        u.E.sr = int(uni(unsigned(s_ctx[k][0]))); u.E.sc = int(uni(unsigned(s_ctx[k][1])));
        u.C.shr = int(uni(unsigned(s_ctx[k][2]))); u.C.shc = int(uni(unsigned(s_ctx[k][3])));
        u.applied = ((stripmask >> k) & 1u) != 0u;  // (only asked together with a shift)
        u.do_update = ((umask >> k) & 1u) != 0u;
        u.strips = ((stripmask >> k) & 1u) != 0u;
        u.ob_scan = ob_prev;
        const bool star = int(k) == s_star;
        const TileJob J{U.scan_no0 + k, 0, star, star};
        const bool obst_tile = star && (nch != 0u || stamp == ob_prev);
        TilePool Q;
        Q.rec = U.rec0 + size_t(k) * U.rec_stride;
        Q.desc = U.desc0 + size_t(k) * U.desc_stride;
        Q.stride = U.stride;
        Q.rare = U.rare;
        tupdate_tile<POLICY, 256, HAS_INT, HAS_COL>(J, G, TG, u, L, all_layers, n_layers, Q, A, dyn_lds, tile, true, nch,
                                                    nch, d, obst_tile, lt, gid, &s_rare);
        __syncthreads();
        if (lt == 0) s_rare = 0u;  // (read only behind barriers inside tupdate_tile)
        d = dn;
      }
      // (the synchronous statistics report the cells touched by the batch's LAST scan)
      if (lt == 0 && !last_seen) A.upd_part[tile] = 0u;
    }
    if (lt == 0) s_pop = U.n_groups + atomicAdd(&ms->tq, 1u);
    __syncthreads();
    pop = uni(s_pop);
  }

  // the last group to leave commits the geometry ring behind the batch (make_tile_ctx does this per scan) and re-arms
  // the state of the batch after next
  if (lt == 0) {
    __threadfence();
    if (atomicAdd(&ms->gdone, 1u) == U.n_groups - 1u) {
      const unsigned last = count - 1u;
      const unsigned slot_next = (U.scan_no0 + count) & 3u;
      DevGeom g = ms->E[last];
      const DevCand cl = ms->C[last];
      const bool applied_last = U.do_move && (!U.gate_on_filter || ((passbits >> last) & 1u) != 0u);
      if (applied_last) { g.px = cl.px; g.py = cl.py; g.sr = cl.sr; g.sc = cl.sc; }
      st->geom[slot_next] = g;
      st->cand[(U.scan_no0 + last) & 3u] = cl;  // (the synchronous statistics report the last scan's shift)
      st->obst[slot_next].scan = umask ? U.scan_no0 + unsigned(s_star) : ob_prev;
#pragma unroll
      for (int q = 0; q < 4; ++q) { st->flags[q].any_pass = 0u; st->flags[q].any_inside = 0u; st->flags[q].ray_any = 0u; }
      if (umask) {
        const unsigned first_upd = U.scan_no0 + unsigned(__ffs(int(umask))) - 1u;
        if (HAS_INT && st->vis_int == 0u) st->vis_int = 3u * first_upd + 2u;
        if (HAS_COL && st->vis_col == 0u) st->vis_col = 3u * first_upd + 2u;
      }
      for (int q = 0; q < kMaxBatch; ++q) {
        U.rearm->done[q * kLineWords] = 0u; U.rearm->done[q * kLineWords + 1] = 0u; U.rearm->inside[q] = 0u;
      }
      U.rearm->flags[0] = 0u;
      U.rearm->flags[1] = 0u;  // (k_mbatch's pre-walked chain word)
      U.rearm->err = 0u;
      U.rearm->tq = 0u;
      U.rearm->gdone = 0u;
    }
  }
}

// [ update of batch b-1 | bin of batch b ] — either may be empty.  Update groups come first in the grid (short chains
// that start at once), then the bin blocks scan by scan.
template <typename POLICY, bool HAS_INT, bool HAS_COL>
__global__ __launch_bounds__(256, FDM_UPD_WAVES) void k_tbatch(const TBUpd U, const TBBin B, const TBScout C, const TBCommon K,
                                                              const GeomConst G, const TileGrid TG,
                                                              DevState* __restrict__ st,
                                                              const typename POLICY::Layers L,
                                                              float* const* __restrict__ all_layers, int n_layers,
                                                              const TileAux A, unsigned upd_groups) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  const unsigned long long t0 = K.timeline ? wall_clock64() : 0ull;
#if FDM_MB_PHASES
  if (threadIdx.x == 0) { g_phase[0] = g_phase[1] = g_phase[2] = unsigned(t0); }
#endif
#ifndef FDM_TB_ONLY
#define FDM_TB_ONLY 0  // measurement builds: 1 = the bin half only, 2 = the update half only (register budgets of the halves)
#endif
  if (FDM_TB_ONLY != 1 && blockIdx.x < upd_groups) {
    tbupdate_body<POLICY, HAS_INT, HAS_COL>(U, G, TG, st, L, all_layers, n_layers, A, dyn_lds, blockIdx.x);
  } else if (blockIdx.x >= upd_groups + (B.count ? B.first_block[B.count] : 0u)) {
    const unsigned sb = blockIdx.x - upd_groups - (B.count ? B.first_block[B.count] : 0u);
    tscout_body(C, K, sb / kScoutBlocks, sb % kScoutBlocks);
  } else if (FDM_TB_ONLY != 2) {
    const unsigned b = blockIdx.x - upd_groups;
    unsigned k = 0u;  // the block's scan: a short prefix table, uniform compares
#pragma unroll
    for (int j = 1; j < kTBMax; ++j) k += (unsigned(j) < B.count && b >= B.first_block[j]) ? 1u : 0u;
    const unsigned lb = b - B.first_block[k];
    TBScanView P;
    P.Tbs = K.Tbs; P.Twb = B.Twb[k]; P.R = B.R[k]; P.sp = K.sp;
    P.min_sq = K.min_sq; P.max_sq = K.max_sq; P.z_min = K.z_min; P.z_max = K.z_max;
    P.n = B.n[k];
    P.integrate_mode = 1; P.sensor_type = K.sensor_type; P.has_var = K.has_var;
    P.drop_nonfinite = 0; P.dbg_no_atomics = 0;
    const ScanInputs I{B.px[k], B.py[k], B.pz[k], B.pint[k], B.prgb[k], B.pvar[k]};
    TilePool Q;
    Q.rec = B.rec0 + size_t(k) * B.rec_stride;
    Q.desc = B.desc0 + size_t(k) * B.desc_stride;
    Q.stride = B.stride;
    Q.rare = B.rare;
    const Scratch S{};  // (LEAN bin body: no captures, no write-through)
    TbinChain H(B, K, st, k);
    tbin_body<HAS_INT, HAS_COL, 256, true>(P, G, TG, H, I, S, B.bin_part + B.first_block[k], Q, nullptr, dyn_lds, lb);
  }
  if (K.timeline && threadIdx.x == 0) {  // (thread 0's view of the block; scripts/timeline.py)
    K.timeline[2u * blockIdx.x] = t0;
#if FDM_MB_PHASES
    K.timeline[2u * blockIdx.x + 1u] = phase_word(t0);
#else
    K.timeline[2u * blockIdx.x + 1u] = wall_clock64();
#endif
  }
}

}  // namespace fdm
