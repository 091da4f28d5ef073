// fdm_rsort.hpp — stable LSD radix sort of (key, point index) pairs for the voxel filter of the raycasting stage
// (fdm_raycast.hpp: scans too large for the sort-free path, 63-bit keys of clouds without a range bound).  gfx950 only.
//
// Reference being served: lib/nanoPCL/include/nanopcl/filters/impl/voxel_grid_impl.hpp:56-63 (sort of the
// (key, index) array) — the engine needs the STABLE order (ties in point order, see fdm_raycast.hpp).
//
// Eight bits per pass, three launches per pass, tiles of 4096 pairs (1024 for sorts of up to 600 K pairs):
//   k_rs_hist     per tile: digit histogram (LDS atomics), written bin-major  hist[bin][tile]
//   k_rs_scan     one block per bin: exclusive prefix over the tiles in place, the bin's total
//   k_rs_scatter  per tile: bases = scan of the 256 totals (every block, in LDS) + hist[bin][tile]; the tile's pairs stay
//                 in registers; wavefront w owns 1024 consecutive pairs, walked in 16 rounds of 64 consecutive pairs
//                 (memory order = (wavefront, round, lane), so the loads coalesce and the order is the stable one);
//                 a pair's place = base of its digit for this wavefront + pairs of the same digit in earlier rounds
//                 (a wavefront-private running counter in LDS) + lanes below it with the same digit in this round
//                 (match over the 8 digit bits: 8 ballots).
// No look-back, no temporary-storage protocol, no memsets: the histogram is overwritten by every pass.
// (Digits of 10 / 11 bits — three passes instead of four over a 30-bit key — measured the same at 2.1 M pairs, 0.783 vs
// 0.777 ms for the stage: ten ballots and a 4 x 1024-entry counter table per round cost what the fourth pass costs.)
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace fdm {

constexpr unsigned kRsTile = 4096u;       // pairs per block ...
constexpr unsigned kRsTileSmall = 1024u;  // ... and for sorts of up to kRsSmallMax pairs (a 272 K-point scan is 67 tiles of 4 096:
constexpr unsigned kRsSmallMax = 600000u; // a quarter of the chip; 266 of 1 024)
__host__ __device__ constexpr unsigned rs_tile(unsigned n) { return n <= kRsSmallMax ? kRsTileSmall : kRsTile; }

template <typename KEY, unsigned TILE>
__global__ __launch_bounds__(256) void k_rs_hist(unsigned n, const KEY* __restrict__ keys, unsigned shift,
                                                 unsigned ntiles, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned base = blockIdx.x * TILE;
#pragma unroll 4
  for (int r = 0; r < int(TILE / 256u); ++r) {
    const unsigned i = base + unsigned(r) * 256u + threadIdx.x;
    if (i < n) atomicAdd(&h[unsigned(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[size_t(threadIdx.x) * ntiles + blockIdx.x] = h[threadIdx.x];
}

// block d: hist[d][0 .. ntiles) -> exclusive prefix in place, total[d]
inline __global__ __launch_bounds__(256) void k_rs_scan(unsigned ntiles, uint32_t* __restrict__ hist, uint32_t* __restrict__ total) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_carry;
  uint32_t* const row = hist + size_t(blockIdx.x) * ntiles;
  const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  if (t == 0) s_carry = 0u;
  __syncthreads();
  for (unsigned c0 = 0; c0 < ntiles; c0 += 1024u) {  // four consecutive entries per thread and step
    uint32_t v[4];
    uint32_t mine = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned q = c0 + t * 4u + unsigned(j);
      v[j] = q < ntiles ? row[q] : 0u;
      mine += v[j];
    }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if (int(lane) >= d) inc += o;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t run = s_carry + inc - mine;
    for (unsigned w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned q = c0 + t * 4u + unsigned(j);
      if (q < ntiles) row[q] = run;
      run += v[j];
    }
    __syncthreads();
    if (t == 255u) s_carry = run;
    __syncthreads();
  }
  if (t == 0) total[blockIdx.x] = s_carry;
}

// HAS_IDX false: the first pass, a pair's index is its position.  ROUNDS: 64-pair rounds per wavefront (tile = 256 x ROUNDS)
template <typename KEY, bool HAS_IDX, int ROUNDS>
__global__ __launch_bounds__(256) void k_rs_scatter(unsigned n, const KEY* __restrict__ keys_in,
                                                    const uint32_t* __restrict__ idx_in, KEY* __restrict__ keys_out,
                                                    uint32_t* __restrict__ idx_out, unsigned shift, unsigned ntiles,
                                                    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ total) {
  __shared__ uint32_t s_cnt[4][256];  // per wavefront and digit: pairs so far, then the running place
  __shared__ uint32_t s_wave[4];
  const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  // the pairs of this thread: round r of its wavefront, lane `lane` (issued first)
  constexpr int kRsRounds = ROUNDS;
  const unsigned w_base = blockIdx.x * (256u * unsigned(ROUNDS)) + wave * (64u * unsigned(ROUNDS));
  KEY k_[kRsRounds];
  uint32_t i_[kRsRounds];
#pragma unroll
  for (int r = 0; r < kRsRounds; ++r) {
    const unsigned i = w_base + unsigned(r) * 64u + lane;
    k_[r] = KEY(0);
    i_[r] = 0u;
    if (i < n) { k_[r] = keys_in[i]; i_[r] = HAS_IDX ? idx_in[i] : i; }
  }
  // first place of every digit in this tile: exclusive scan of the 256 totals + the tile's prefix
  uint32_t tile_base;
  {
    const uint32_t mine = total[t];
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d);
      if (int(lane) >= d) inc += o;
    }
    if (lane == 63u) s_wave[wave] = inc;
    s_cnt[0][t] = 0u; s_cnt[1][t] = 0u; s_cnt[2][t] = 0u; s_cnt[3][t] = 0u;
    __syncthreads();
    uint32_t run = inc - mine;
    for (unsigned w = 0; w < wave; ++w) run += s_wave[w];
    tile_base = run + hist[size_t(t) * ntiles + blockIdx.x];
  }
  // pairs per wavefront and digit
#pragma unroll
  for (int r = 0; r < kRsRounds; ++r) {
    const unsigned i = w_base + unsigned(r) * 64u + lane;
    if (i < n) atomicAdd(&s_cnt[wave][unsigned(k_[r] >> shift) & 255u], 1u);
  }
  __syncthreads();
  {  // digit t: the four wavefronts' first places
    uint32_t run = tile_base;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = s_cnt[w][t];
      s_cnt[w][t] = run;
      run += c;
    }
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < kRsRounds; ++r) {
    const unsigned i = w_base + unsigned(r) * 64u + lane;
    const bool live = i < n;
    const unsigned d = unsigned(k_[r] >> shift) & 255u;
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long ones = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? ones : ~ones;
    }
    if (live) {
      const uint32_t first = s_cnt[wave][d];  // (read by every peer before the leader moves it on)
      const unsigned rank = unsigned(__popcll(peers & below));
      if (rank == 0u) s_cnt[wave][d] = first + unsigned(__popcll(peers));
      keys_out[first + rank] = k_[r];
      idx_out[first + rank] = i_[r];
    }
  }
}

}  // namespace fdm
