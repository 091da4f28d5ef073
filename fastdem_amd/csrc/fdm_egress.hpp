// fdm_egress.hpp — map -> PointCloud2 byte stream on the device.  gfx950 only.
//
// Reference being reproduced: fastdem/include/fastdem/bridge/ros/impl.hpp:28-166 (toPointCloud2Impl):
// one point per cell with a FINITE elevation, visited column by column through the submap
// [sub_start, sub_start + sub_size) of the circular buffer; per point x, y (cell centre, computed
// in double from the unwrapped index, cast to float), z = elevation, then every non-internal
// layer in getLayers() order, then the packed colour.  The reference walks the map twice on the
// CPU (count, fill); here: count per block -> exclusive scan of the block counts -> fill, so the
// output order is exactly the reference's and the host receives ONE contiguous D2H copy instead of
// a download per layer.
#pragma once

#include "fdm_device.hpp"

namespace fdm {

constexpr int kPackMaxFields = 64;

struct PackParams {
  int sub_r0, sub_c0, sub_rows, sub_cols;  // buffer indices; sub_rows < 0: the whole map from the start index
  int slot;                                // geometry ring slot
  int n_float;                             // float layers after x, y, z
  int has_color;
};
struct PackLayers {
  const float* elev;
  int elev_stride;
  const float* color;
  const float* ptr[kPackMaxFields];
  int stride[kPackMaxFields];
};

struct PackCell {
  bool valid;
  size_t o;     // storage-linear cell
  float x, y, z;
};

__device__ __forceinline__ PackCell pack_cell(const PackParams& Q, const GeomConst& G, const DevGeom& g,
                                              const PackLayers& L, unsigned long long t,
                                              unsigned long long total) {
  PackCell pc;
  pc.valid = false;
  pc.o = 0;
  pc.x = pc.y = pc.z = 0.f;
  if (t >= total) return pc;
  const bool full = Q.sub_rows < 0;
  const int sub_rows = full ? G.rows : Q.sub_rows;
  const int r0 = full ? g.sr : Q.sub_r0, c0 = full ? g.sc : Q.sub_c0;
  const int j = int(t / unsigned(sub_rows)), i = int(t - (unsigned long long)j * unsigned(sub_rows));
  int r = r0 + i, c = c0 + j;  // (sub_start + i) % size
  r -= r >= G.rows ? G.rows : 0;
  c -= c >= G.cols ? G.cols : 0;
  const int lr = r - G.s_r0, lc = c - G.s_c0;  // tiled engines hold a window of the buffer
  if (lr < 0 || lc < 0 || lr >= G.s_rows || lc >= G.s_cols) return pc;
  pc.o = size_t(lc) * G.s_rows + lr;
  pc.z = L.elev[pc.o * L.elev_stride];
  pc.valid = isfinite(pc.z);
  int ur = r - g.sr, uc = c - g.sc;  // (r - start + size) % size
  ur += ur < 0 ? G.rows : 0;
  uc += uc < 0 ? G.cols : 0;
  const double origin_x = g.px + G.len_x / 2.0 - G.res / 2.0;
  const double origin_y = g.py + G.len_y / 2.0 - G.res / 2.0;
  pc.x = static_cast<float>(origin_x - double(ur) * G.res);
  pc.y = static_cast<float>(origin_y - double(uc) * G.res);
  return pc;
}

__device__ __forceinline__ unsigned long long pack_total(const PackParams& Q, const GeomConst& G) {
  return Q.sub_rows < 0 ? (unsigned long long)G.rows * G.cols
                        : (unsigned long long)Q.sub_rows * Q.sub_cols;
}

// valid cells per block of 256 consecutive visits
inline __global__ __launch_bounds__(256) void k_pack_count(const PackParams Q, const GeomConst G,
                                                    const DevState* __restrict__ st, const PackLayers L,
                                                    uint32_t* __restrict__ counts) {
  __shared__ unsigned s_w[4];
  const DevGeom g = st->geom[Q.slot];
  const unsigned long long t = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
  const PackCell pc = pack_cell(Q, G, g, L, t, pack_total(Q, G));
  const unsigned long long m = __ballot(pc.valid);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = unsigned(__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// in-place exclusive scan of the block counts (one block, carries across 1024-entry chunks);
// counts[n] receives the total
inline __global__ __launch_bounds__(1024) void k_pack_scan(uint32_t* __restrict__ counts, unsigned n) {
  __shared__ unsigned s_wave[16];
  __shared__ unsigned s_carry;
  if (threadIdx.x == 0) s_carry = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (unsigned base = 0; base < n; base += 1024u) {
    const unsigned k = base + threadIdx.x;
    const unsigned v = k < n ? counts[k] : 0u;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    unsigned woff = 0;
    for (int q = 0; q < w; ++q) woff += s_wave[q];
    const unsigned carry = s_carry;
    if (k < n) counts[k] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[n] = s_carry;
}

inline __global__ __launch_bounds__(256) void k_pack_write(const PackParams Q, const GeomConst G,
                                                    const DevState* __restrict__ st, const PackLayers L,
                                                    const uint32_t* __restrict__ offsets,
                                                    float* __restrict__ out) {
  __shared__ unsigned s_w[4];
  const DevGeom g = st->geom[Q.slot];
  const unsigned long long t = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
  const PackCell pc = pack_cell(Q, G, g, L, t, pack_total(Q, G));
  const unsigned long long m = __ballot(pc.valid);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) s_w[w] = unsigned(__popcll(m));
  __syncthreads();
  // records are staged in LDS (rank-major) and leave as one contiguous, fully coalesced run
  extern __shared__ float s_rec[];
  const int nf = 3 + Q.n_float + (Q.has_color ? 1 : 0);
  if (pc.valid) {
    unsigned rank = unsigned(__popcll(m & ((1ull << lane) - 1ull)));
    for (int q = 0; q < w; ++q) rank += s_w[q];
    float* p = s_rec + size_t(rank) * nf;
    p[0] = pc.x;
    p[1] = pc.y;
    p[2] = pc.z;
    for (int k = 0; k < Q.n_float; ++k) p[3 + k] = L.ptr[k][pc.o * size_t(L.stride[k])];
    if (Q.has_color) p[3 + Q.n_float] = L.color[pc.o];
  }
  __syncthreads();
  const unsigned n_out = (s_w[0] + s_w[1] + s_w[2] + s_w[3]) * unsigned(nf);
  float* dst = out + size_t(offsets[blockIdx.x]) * size_t(nf);
  for (unsigned k = threadIdx.x; k < n_out; k += 256u) dst[k] = s_rec[k];
}

}  // namespace fdm
