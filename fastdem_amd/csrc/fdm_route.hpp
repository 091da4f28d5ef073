// fdm_route.hpp — scan routing for the spatially tiled global map.
//
// One logical scan is the concatenation, in rank order, of per-rank SLICES (one 2 M-point scan cut in N pieces, or
// N sensors' clouds merged into one scan — the same thing to integrate()).  A rank runs its slice through the first
// half of the path — T_base_sensor, cropRange, cropZ, T_world_base, getIndex against the GLOBAL geometry: the very
// float / double operations the owner's bin kernel will repeat — only to learn which rank OWNS the cell each point
// falls into, and partitions the slice's raw points by owner, order preserved.  The owners then receive, from every
// rank in rank order, exactly the points of the logical scan that land in their window, in scan order: "first point
// wins a tie" and "last colour wins" hold as on the single map, and every owned cell comes out bit-identical.
// Points the crops drop or that fall outside the map go nowhere (they only count in the statistics).
//
//   k_route_count    owner of every point (one byte), per-block / per-owner counts, n_after_filter / n_in_map
//   k_route_scan     exclusive scans of the block counts per owner, the owners' totals (what the host reads back
//                    to size the all-to-all) and their base offsets in the send buffer
//   k_route_scatter  stable partition: rank inside the block from wave ballots, raw {x, y, z, intensity} as one
//                    16-byte store per point
// Algorithmic bytes: 2 x 12 B/point read (+4 intensity), 1 B/point owner write + read, 16 B per routed point.
#pragma once

#include "fdm_kernels.hpp"

namespace fdm {

constexpr int kMaxRanks = 16;
constexpr unsigned kNoOwner = 0xFFu;

struct RoutePlan {  // the tiling plan as the kernels need it: owned rect of rank i * pc + j is
  int world, pr, pc, pad;                  // rows [row_edge[i], row_edge[i + 1]) x cols [col_edge[j], col_edge[j + 1])
  int row_edge[kMaxRanks + 1], col_edge[kMaxRanks + 1];
};

__device__ __forceinline__ unsigned route_owner(const RoutePlan& R, int r, int c) {
  int i = 0, j = 0;
  for (int k = 1; k < R.pr; ++k) i += r >= R.row_edge[k] ? 1 : 0;
  for (int k = 1; k < R.pc; ++k) j += c >= R.col_edge[k] ? 1 : 0;
  return unsigned(i * R.pc + j);
}

// cnt layout: [blocks][world + 2] — per owner, then surviving points, then points inside the map
inline __global__ __launch_bounds__(256) void k_route_count(const ScanParams P, const GeomConst G, const RoutePlan R,
                                                     const DevState* __restrict__ st, const float* __restrict__ px,
                                                     const float* __restrict__ py, const float* __restrict__ pz,
                                                     uint8_t* __restrict__ owner_out, uint32_t* __restrict__ cnt) {
  __shared__ unsigned s_cnt[4][kMaxRanks + 2];
  const unsigned i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const DevGeom g = st->geom[P.slot];
  DevCand cand;
  cand.px = g.px; cand.py = g.py; cand.sr = g.sr; cand.sc = g.sc; cand.shr = 0; cand.shc = 0;
  unsigned owner = kNoOwner;
  bool pass = false, inside = false;
  if (i < P.n) {
    float x = px[i], y = py[i], z = pz[i];
    pass = preprocess_point(P, x, y, z);
    int r, c;
    if (pass && cell_of(x, y, cand, G, r, c)) {
      inside = true;
      owner = route_owner(R, r, c);
    }
    owner_out[i] = uint8_t(owner);
  }
  for (int d = 0; d < R.world; ++d) {
    const unsigned long long m = __ballot(owner == unsigned(d));
    if (lane == 0u) s_cnt[wave][d] = unsigned(__popcll(m));
  }
  {
    const unsigned long long mp = __ballot(pass), mi = __ballot(inside);
    if (lane == 0u) { s_cnt[wave][R.world] = unsigned(__popcll(mp)); s_cnt[wave][R.world + 1] = unsigned(__popcll(mi)); }
  }
  __syncthreads();
  if (threadIdx.x < unsigned(R.world + 2))
    cnt[size_t(blockIdx.x) * unsigned(R.world + 2) + threadIdx.x] =
        s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
}

// one 1024-thread block per column of cnt (world + 2 columns): exclusive scan over the scan's blocks, eight
// consecutive entries per thread and pass so that their loads are in flight together (one wavefront walking the
// column entry by entry spent 100 us of dependent load latency on a 2 M-point slice)
inline __global__ __launch_bounds__(1024) void k_route_scan(uint32_t* __restrict__ cnt, unsigned blocks, int world,
                                                      uint32_t* __restrict__ totals /* [world + 2] */) {
  __shared__ unsigned s_w[16];
  __shared__ unsigned s_run;
  const unsigned col = blockIdx.x, cols = unsigned(world + 2);
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_run = 0u;
  __syncthreads();
  for (unsigned b0 = 0; b0 < blocks; b0 += 8192u) {
    unsigned v[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned b = b0 + threadIdx.x * 8u + unsigned(j);
      v[j] = b < blocks ? cnt[size_t(b) * cols + col] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) mine += v[j];
    unsigned inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(inc, d);
      if (int(lane) >= d) inc += o;
    }
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    unsigned before = s_run + inc - mine;
    for (unsigned w = 0; w < wave; ++w) before += s_w[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned b = b0 + threadIdx.x * 8u + unsigned(j);
      if (b < blocks) cnt[size_t(b) * cols + col] = before;  // exclusive, in place
      before += v[j];
    }
    __syncthreads();
    if (threadIdx.x == 1023u) s_run = before;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[col] = s_run;
}
// the owners' shares in the send buffer start where the shares of the lower ranks end
// (soa: every share is padded to a multiple of four points, so that each of its four channels starts 16-byte aligned)
inline __global__ void k_route_base(const uint32_t* __restrict__ totals, int world, uint32_t* __restrict__ base, int soa) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned acc = 0;
    for (int d = 0; d < world; ++d) { base[d] = acc; acc += soa ? ((totals[d] + 3u) & ~3u) : totals[d]; }
  }
}

// SOA: the share of owner d is four channel blocks x | y | z | intensity of P_d = pad4(count_d) floats each, starting
// at float 4 * base[d] — what the bin kernels read in place (fdm_engine_integrate_soa4_device): no de-interleave pass
// on the receiving side, and the rank's own share is never copied.  AoS ({x, y, z, intensity} records): the layout a
// scan cut into SLICES needs (its owners integrate all sources as ONE scan: the records of consecutive sources must
// be contiguous).
template <bool SOA>
__global__ __launch_bounds__(256) void k_route_scatter(unsigned n, int world, const uint8_t* __restrict__ owner_in,
                                                       const uint32_t* __restrict__ off /* scanned cnt */,
                                                       const uint32_t* __restrict__ base,
                                                       const uint32_t* __restrict__ totals, const float* __restrict__ px,
                                                       const float* __restrict__ py, const float* __restrict__ pz,
                                                       const float* __restrict__ pint, float4* __restrict__ send) {
  __shared__ unsigned s_cnt[4][kMaxRanks];
  const unsigned i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const unsigned owner = i < n ? unsigned(owner_in[i]) : kNoOwner;
  unsigned rank_in_wave = 0;
  for (int d = 0; d < world; ++d) {
    const unsigned long long m = __ballot(owner == unsigned(d));
    if (owner == unsigned(d)) rank_in_wave = unsigned(__popcll(m & ((1ull << lane) - 1ull)));
    if (lane == 0u) s_cnt[wave][d] = unsigned(__popcll(m));
  }
  __syncthreads();
  if (owner == kNoOwner) return;
  unsigned before = 0;
  for (unsigned w = 0; w < wave; ++w) before += s_cnt[w][owner];
  const unsigned j = off[size_t(blockIdx.x) * unsigned(world + 2) + owner] + before + rank_in_wave;  // place inside the share
  if (SOA) {
    const unsigned P = (totals[owner] + 3u) & ~3u;
    float* const blk = reinterpret_cast<float*>(send) + 4u * size_t(base[owner]);
    blk[j] = px[i];
    blk[size_t(P) + j] = py[i];
    blk[2u * size_t(P) + j] = pz[i];
    blk[3u * size_t(P) + j] = pint ? pint[i] : 0.0f;
  } else {
    send[base[owner] + j] = make_float4(px[i], py[i], pz[i], pint ? pint[i] : 0.0f);
  }
}

// received points {x, y, z, intensity} -> the SoA channels the bin kernels read
inline __global__ void k_points4_to_soa(const float4* __restrict__ pts, size_t n, float* __restrict__ x, float* __restrict__ y,
                                 float* __restrict__ z, float* __restrict__ a) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) {
    const float4 p = pts[i];
    x[i] = p.x; y[i] = p.y; z[i] = p.z;
    if (a) a[i] = p.w;
  }
}

}  // namespace fdm
