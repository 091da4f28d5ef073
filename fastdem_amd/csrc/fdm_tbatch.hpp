// fdm_tbatch.hpp — a BATCH of large scans in ONE launch on the per-tile record pools (fdm_tiled.hpp).
//
// Why: one fused launch per 2 M-point scan (k_tupdate_tbin) pays, per LAUNCH, a ramp in which
// 1 444 latency-bound tile groups hold the wave slots while the vector units idle, and a tail in which the last of
// its 2 048 bin blocks run on a draining chip (two quantised rounds on 1 792 slots): 33.7 us per scan for 19 us of
// vector issue.  fdm_engine_integrate_device_batch sees its scans up front, so K consecutive large scans leave as
//
//   k_tbatch = [ update of batch b-1 : a FEW tile groups that pull tiles from a queue | bin of batch b : K x blocks ]
//
//   bin half     tbin_body as it is (fdm_tiled.hpp) — scan k of the batch writes its records into pool k of the
//                batch's parity.  The geometry scan k is binned against depends on whether the scans before it moved
//                the map, i.e. on whether any of their points survived the crops (fastdem.cpp:138) — device-side
//                data.  In-launch protocol (as fdm_multi.hpp's): after its crops a block adds itself to its scan's
//                counter; the FIRST block of a scan that holds a surviving point raises the scan's pass bit at once,
//                the block that completes a scan raises its done bit; a block of scan k > 0 polls ONE word until
//                every earlier scan is decided, then walks the k moves ahead of its own (GridMap::move arithmetic,
//                move_candidate_fast: the operations the single-scan path performs, in the same order).  A block only
//                waits for blocks with a lower index, which publish before they wait and are dispatched first: no
//                deadlock; the spin is bounded and raises MState::err.
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
// The in-launch protocol lives in ONE word per scan, MState::done[k * kLineWords] (its own 128-byte line): every block
// of scan k adds 1 | (it holds a surviving point) << 16 once, after its crops — a fire-and-forget atomic, nothing
// waits for it.  Scan j is DECIDED for a later scan's block as soon as the word's upper half is non-zero (some block
// saw a surviving point: the scan moves the map) or its lower half has reached the scan's block count (no block did).
// A block of scan k reads the k words ahead of it at its very start (lane j of the first wavefront: word j), in the
// shadow of its point loads; in the steady state of a launch they are long decided and the block never waits.
// (First version: a returning atomicAdd + a flags word polled after the crops — two dependent memory round trips on
// every block's critical path: bin blocks lived 17 us instead of 11.4, profiles/r04/timeline_c4_tbatch_v1.json.)
__device__ __forceinline__ bool tb_decided(unsigned d, unsigned nblocks) { return (d >> 16) != 0u || (d & 0xFFFFu) == nblocks; }

struct TbinChain {
  const TBBin& B;
  const TBCommon& K;
  DevState* st;
  const unsigned k, nblocks;
  unsigned* s_w;  // [4] LDS: per wavefront "some point survived the crops"
  DevGeom g0;
  DevCand cprev;
  unsigned dprev;  // the previous batch's last scan: its `done` word (final)
  unsigned dj;     // lane j < k of the first wavefront: scan j's `done` word
  unsigned nbj;    // ... and its block count
  __device__ __forceinline__ TbinChain(const TBBin& b, const TBCommon& kk, DevState* s, unsigned scan, unsigned nb, unsigned* w)
      : B(b), K(kk), st(s), k(scan), nblocks(nb), s_w(w) {}
  __device__ __forceinline__ bool gated() const { return K.do_move != 0 && K.gate_on_filter != 0; }
  __device__ __forceinline__ void begin() {
    dj = 0u; nbj = 0u; dprev = 0u;
    if (threadIdx.x < 64u) {
      if (gated() && threadIdx.x < k) {
        dj = __hip_atomic_load(&B.ms->done[threadIdx.x * kLineWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nbj = B.first_block[threadIdx.x + 1u] - B.first_block[threadIdx.x];
      }
      if (threadIdx.x == 0) {  // the geometry the chain starts from (in flight while the points are transformed)
        cprev.px = cprev.py = 0.0; cprev.sr = cprev.sc = cprev.shr = cprev.shc = 0;
        if (B.prev) {
          const unsigned pk = B.prev_count - 1u;
          g0 = B.prev->E[pk];
          cprev = B.prev->C[pk];
          dprev = B.prev->done[pk * kLineWords];
        } else {
          g0 = st->geom[B.scan_no0 & 3u];
        }
      }
    }
  }
  __device__ __forceinline__ DevCand finish(const GeomConst& G, DevCand* s_cand, unsigned lb, unsigned n_pass) {
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (gated()) {
      const bool wp = __ballot(n_pass != 0u) != 0ull;
      if (lane == 0u) s_w[wave] = wp ? 1u : 0u;
      __syncthreads();
    }
    if (threadIdx.x < 64u) {  // the first wavefront: lane j looks after scan j's word, thread 0 publishes and walks
      MState* const ms = B.ms;
      unsigned passmask = 0xFFFFu;
      if (gated()) {
        if (threadIdx.x == 0) {
          const unsigned np = s_w[0] | s_w[1] | s_w[2] | s_w[3];
          (void)atomicAdd(&ms->done[k * kLineWords], 1u | (np ? 0x10000u : 0u));  // (result unused: no round trip)
        }
        unsigned spins = 0u;
        if (K.dbg == 4 && k > 0u && lane == 0u) ms->err = 1u;  // (dbg 4: tests provoke the fault)
        while (true) {  // (wave-uniform) every earlier scan either has a surviving point or is through its crops
          const bool open = lane < k && !tb_decided(dj, nbj);
          if (!__ballot(open)) break;
          if (++spins >= kSpinMax) {
            if (lane == 0u) ms->err = 1u;
            break;
          }
          __builtin_amdgcn_s_sleep(8);
          if (open) dj = __hip_atomic_load(&ms->done[lane * kLineWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        passmask = uni(unsigned(__ballot(lane < k && (dj >> 16) != 0u)));
      }
      if (threadIdx.x == 0) {
        DevGeom g = g0;
        if (B.prev) {  // what the update of the previous batch (the other half of this launch) is about to commit
          if (K.do_move && (!K.gate_on_filter || (dprev >> 16) != 0u)) {
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
        if (lb == 0u) { ms->E[k] = g; ms->C[k] = c; }
      }
    }
    __syncthreads();
    return *s_cand;
  }
  __device__ __forceinline__ void note_inside() { B.ms->inside[k] = 1u; }
  __device__ __forceinline__ void note_pass() {}
};

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
    const bool v_pass = lane < count && (ms->done[lane * kLineWords] >> 16) != 0u;
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
    // tile i + q * n_pops (the tiles a scan touches are neighbours: strided, a pop holds few live ones)
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
        if (lt <= n0 && lt < U.stride) d = U.desc0[size_t(k0) * U.desc_stride + row0 + lt];
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
      if (ms->err) st->fault = 1u;
      for (int q = 0; q < kMaxBatch; ++q) { U.rearm->done[q * kLineWords] = 0u; U.rearm->inside[q] = 0u; }
      U.rearm->flags[0] = 0u;
      U.rearm->err = 0u;
      U.rearm->tq = 0u;
      U.rearm->gdone = 0u;
    }
  }
}

// [ update of batch b-1 | bin of batch b ] — either may be empty.  Update groups come first in the grid (short chains
// that start at once), then the bin blocks scan by scan.
template <typename POLICY, bool HAS_INT, bool HAS_COL>
__global__ __launch_bounds__(256, FDM_UPD_WAVES) void k_tbatch(const TBUpd U, const TBBin B, const TBCommon K,
                                                              const GeomConst G, const TileGrid TG,
                                                              DevState* __restrict__ st,
                                                              const typename POLICY::Layers L,
                                                              float* const* __restrict__ all_layers, int n_layers,
                                                              const TileAux A, unsigned upd_groups) {
  extern __shared__ __align__(16) unsigned char dyn_lds[];
  __shared__ unsigned s_w[4];
  const unsigned long long t0 = K.timeline ? wall_clock64() : 0ull;
#if FDM_MB_PHASES
  if (threadIdx.x == 0) { g_phase[0] = g_phase[1] = g_phase[2] = unsigned(t0); }
#endif
#ifndef FDM_TB_ONLY
#define FDM_TB_ONLY 0  // measurement builds: 1 = the bin half only, 2 = the update half only (register budgets of the halves)
#endif
  if (FDM_TB_ONLY != 1 && blockIdx.x < upd_groups) {
    tbupdate_body<POLICY, HAS_INT, HAS_COL>(U, G, TG, st, L, all_layers, n_layers, A, dyn_lds, blockIdx.x);
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
    TbinChain H(B, K, st, k, B.first_block[k + 1u] - B.first_block[k], s_w);
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
