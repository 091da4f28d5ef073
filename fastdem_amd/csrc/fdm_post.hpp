// fdm_post.hpp — stencil post-processing on the device.  gfx950 only.
//
// Reference being reproduced (file:line under /root/reference/fastdem):
//   src/inpainting.cpp:21-67                                  applyInpainting
//   include/fastdem/postprocess/spatial_smoothing.hpp:38-67   applySpatialSmoothing
//   src/uncertainty_fusion.cpp:28-186                         weighted-ECDF bound fusion
//   src/feature_extraction.cpp:28-118                         local PCA features
//   lib/nanoPCL/include/nanopcl/geometry/impl/pca.hpp:66-88   computePCA -> Eigen computeDirect (3x3)
// One thread per cell, threads of a wave are consecutive LOGICAL rows of one column, so the centre
// loads and (away from the buffer seam) every neighbour load of a wave coalesce; the 3x3 .. 13x13
// neighbourhoods overlap almost entirely between neighbouring lanes and are served by L1/L2.
// Neighbourhood semantics (logical coordinates, clipped at the border, centre included, entry order
// dr-major / dc-minor) are the ones stated in DESIGN.md §7 f2: nanoGrid's cells()/region()/neighbors()
// are not on disk, so these are this repo's definition.
#pragma once

#include "fdm_device.hpp"

namespace fdm {

constexpr int kMaxRegion = 256;  // entries of a disc / box neighbourhood the kernels accept
constexpr int kS3R_ = 32, kS3C_ = 8;  // the stencil kernels' block tile: 32 rows (the contiguous axis) x 8 columns

struct RegionEntry { int dr, dc; float dist_sq; float w; };  // w: per-entry weight the host precomputed (fusion)

struct PostGeom {
  int rows, cols, sr, sc;
};
// The neighbourhood space of a kernel = the cells this engine STORES: the whole circular buffer
// (logical coordinates through the start index), or — spatial tiles of a GLOBAL map, start index 0 —
// the stored window incl. its halo ring.  At a window edge that is not a map edge the neighbourhood
// is clipped, which only ever affects halo cells as long as the halo is at least as wide as the
// stencil reaches (checked by the host); halos are refreshed from their owners afterwards.
__device__ __forceinline__ PostGeom post_geom(const DevState* __restrict__ st, int slot, const GeomConst& G) {
  PostGeom p;
  p.rows = G.s_rows; p.cols = G.s_cols;
  const bool whole = G.s_rows == G.rows && G.s_cols == G.cols;
  p.sr = whole ? st->geom[slot].sr : 0;
  p.sc = whole ? st->geom[slot].sc : 0;
  return p;
}
__device__ __forceinline__ size_t post_index(const PostGeom& p, int lr, int lc) {
  int r = lr + p.sr, c = lc + p.sc;
  r -= r >= p.rows ? p.rows : 0;
  c -= c >= p.cols ? p.cols : 0;
  return size_t(c) * p.rows + r;
}
__device__ __forceinline__ bool post_inside(const PostGeom& p, int lr, int lc) {
  return unsigned(lr) < unsigned(p.rows) && unsigned(lc) < unsigned(p.cols);
}

// ---- inpainting: one pass (inpainting.cpp:41-64); the caller ping-pongs two buffers ----
inline __global__ __launch_bounds__(256) void k_inpaint_pass(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                      const float* __restrict__ in, int in_stride,
                                                      float* __restrict__ out, int out_stride,
                                                      int min_valid, unsigned ncell) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= ncell) return;
  const PostGeom p = post_geom(st, slot, G);
  const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
  const size_t ci = post_index(p, lr, lc);
  float v = in[ci * in_stride];
  if (isnan(v)) {
    float sum = 0.0f;
    int count = 0;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
      for (int dc = -1; dc <= 1; ++dc) {
        if (dr == 0 && dc == 0) continue;
        if (!post_inside(p, lr + dr, lc + dc)) continue;
        const float n = in[post_index(p, lr + dr, lc + dc) * in_stride];
        if (isfinite(n)) {
          sum += n;
          ++count;
        }
      }
    if (count >= min_valid) v = sum / float(count);
  }
  out[ci * out_stride] = v;
}

// ---- spatial median smoothing (spatial_smoothing.hpp:52-66); `in` is a private copy ----
inline __global__ __launch_bounds__(256) void k_median(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                const float* __restrict__ in, float* __restrict__ out,
                                                int out_stride, int kernel, int min_valid, unsigned ncell) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= ncell) return;
  const PostGeom p = post_geom(st, slot, G);
  const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
  const size_t ci = post_index(p, lr, lc);
  if (!isfinite(in[ci])) return;
  float win[kMaxRegion];
  int n = 0;
  const int h = kernel / 2;
  for (int dr = -h; dr <= h; ++dr)
    for (int dc = -h; dc <= h; ++dc) {
      if (!post_inside(p, lr + dr, lc + dc)) continue;
      const float v = in[post_index(p, lr + dr, lc + dc)];
      if (!isfinite(v)) continue;
      int k = n++;  // insertion sort: the window is at most kernel^2 values
      while (k > 0 && win[k - 1] > v) { win[k] = win[k - 1]; --k; }
      win[k] = v;
    }
  if (n < min_valid) return;
  out[ci * out_stride] = win[n / 2];  // nth_element(size/2)
}

// Neighbourhoods beyond kMaxRegion cells (a 17 x 17 median, a 0.3 m disc on a 0.02 m map = 707 cells): the
// per-thread lists live in a global pool, slot-major / thread-minor (`list[k * pitch]`, pitch = threads in flight,
// so that the lanes of a wavefront touch consecutive words), and a thread walks the map with a grid stride.  The same
// insertion-sorted lists as the small-neighbourhood kernels — O(n^2) per cell: a slow path that exists so that no
// call the reference accepts is refused (spatial_smoothing.hpp:38-67, config/postprocess.hpp:35,45).
struct PoolList {
  float* base;
  unsigned pitch;
  __device__ __forceinline__ float& operator[](int k) const { return base[size_t(k) * pitch]; }
};
inline __global__ __launch_bounds__(256) void k_median_big(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                    const float* __restrict__ in, float* __restrict__ out,
                                                    int out_stride, int kernel, int min_valid, unsigned ncell,
                                                    float* __restrict__ pool) {
  const unsigned tid = blockIdx.x * 256u + threadIdx.x, pitch = gridDim.x * 256u;
  const PoolList win{pool + tid, pitch};
  const PostGeom p = post_geom(st, slot, G);
  const int h = kernel / 2;
  for (unsigned t = tid; t < ncell; t += pitch) {
    const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
    const size_t ci = post_index(p, lr, lc);
    if (!isfinite(in[ci])) continue;
    int n = 0;
    for (int dr = -h; dr <= h; ++dr)
      for (int dc = -h; dc <= h; ++dc) {
        if (!post_inside(p, lr + dr, lc + dc)) continue;
        const float v = in[post_index(p, lr + dr, lc + dc)];
        if (!isfinite(v)) continue;
        int k = n++;
        while (k > 0 && win[k - 1] > v) { win[k] = win[k - 1]; --k; }
        win[k] = v;
      }
    if (n < min_valid) continue;
    out[ci * out_stride] = win[n / 2];
  }
}

// Windows beyond kMaxRegion cells, the fast way (round 5): an order statistic does not need a sort.  One block per
// 32 x 8 cells; the tile and its ring of kernel / 2 cells are staged in LDS ONCE as monotone integer keys (ord(v); a
// cell outside the stored window or without a finite value is the all-ones key), and every thread finds element
// n / 2 of its window by BISECTION on the key's 32 bits: per bit one pass over the window counting the keys below the
// candidate.  33 x k^2 LDS reads per cell instead of ~k^4 / 4 global-memory moves (17 x 17 on the 1200 x 1200 map:
// 423 ms with the pooled insertion sort).  Among equal values the order is the keys' (-0 before +0: std::nth_element
// leaves it unspecified, spatial_smoothing.hpp:52-66).
__device__ __forceinline__ uint32_t post_key(float v) {  // monotone; non-finite -> all ones (never below a candidate)
  const uint32_t b = __float_as_uint(v);
  return (b & 0x7FFFFFFFu) >= 0x7F800000u ? 0xFFFFFFFFu : b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float post_unkey(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }
__host__ __device__ constexpr unsigned median_sel_lds_bytes(int kernel) {
  return unsigned(kS3R_ + 2 * (kernel / 2)) * unsigned(kS3C_ + 2 * (kernel / 2)) * 4u;
}
inline __global__ __launch_bounds__(256) void k_median_sel(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                    const float* __restrict__ in, float* __restrict__ out,
                                                    int out_stride, int kernel, int min_valid) {
  extern __shared__ uint32_t s_keys[];
  const PostGeom p = post_geom(st, slot, G);
  const int h = kernel / 2, pitch = kS3R_ + 2 * h, width = kS3C_ + 2 * h;
  const int tiles_r = (p.rows + kS3R_ - 1) / kS3R_;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  const int r0 = tr * kS3R_ - h, c0 = tc * kS3C_ - h;
  for (int k = int(threadIdx.x); k < pitch * width; k += 256) {
    const int cc = k / pitch, rr = k - cc * pitch;
    s_keys[k] = post_inside(p, r0 + rr, c0 + cc) ? post_key(in[post_index(p, r0 + rr, c0 + cc)]) : 0xFFFFFFFFu;
  }
  __syncthreads();
  const int lrl = int(threadIdx.x) & (kS3R_ - 1), lcl = int(threadIdx.x) >> 5;
  const int lr = tr * kS3R_ + lrl, lc = tc * kS3C_ + lcl;
  if (!post_inside(p, lr, lc)) return;
  const uint32_t* const win = s_keys + lcl * pitch + lrl;  // the window's corner (dr = dc = -h)
  if (win[h * pitch + h] == 0xFFFFFFFFu) return;          // the centre holds no finite value
  const int side = 2 * h + 1;
  auto below = [&](uint32_t t) {  // keys of the window below t
    int c = 0;
    for (int dc = 0; dc < side; ++dc) {
      const uint32_t* const col = win + dc * pitch;
      for (int dr = 0; dr < side; ++dr) c += col[dr] < t ? 1 : 0;
    }
    return c;
  };
  const int n = below(0xFFFFFFFFu);  // finite values
  if (n < min_valid) return;
  const int r = n / 2;  // nth_element(size / 2)
  uint32_t key = 0u;
  for (int b = 31; b >= 0; --b) {
    const uint32_t t = key | (1u << b);
    if (below(t) <= r) key = t;  // at most r keys below t: element r is >= t
  }
  out[post_index(p, lr, lc) * size_t(out_stride)] = post_unkey(key);
}

// 3x3 window (the default kernel): the nine values are named registers, missing / non-finite
// neighbours are +inf, a 25-step sorting network (verified on all 512 0-1 inputs) orders them and
// element n/2 of the n finite ones is the median — no window in scratch memory, no dependent loop.
inline __global__ __launch_bounds__(256) void k_median3(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                 const float* __restrict__ in, float* __restrict__ out,
                                                 int out_stride, int min_valid, unsigned ncell) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= ncell) return;
  const PostGeom p = post_geom(st, slot, G);
  const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
  const size_t ci = post_index(p, lr, lc);
  if (!isfinite(in[ci])) return;
  constexpr float kInf = __builtin_huge_valf();
  int n = 0;
#define FDM_W(k, dr, dc)                                                        \
  float w##k = kInf;                                                            \
  if (post_inside(p, lr + (dr), lc + (dc))) {                                   \
    const float v = in[post_index(p, lr + (dr), lc + (dc))];                    \
    if (isfinite(v)) { w##k = v; ++n; }                                         \
  }
  FDM_W(0, -1, -1) FDM_W(1, -1, 0) FDM_W(2, -1, 1) FDM_W(3, 0, -1) FDM_W(4, 0, 0) FDM_W(5, 0, 1)
  FDM_W(6, 1, -1) FDM_W(7, 1, 0) FDM_W(8, 1, 1)
#undef FDM_W
  if (n < min_valid) return;
#define FDM_CE(i, j) { const float lo_ = fminf(w##i, w##j); w##j = fmaxf(w##i, w##j); w##i = lo_; }
  FDM_CE(0, 3) FDM_CE(1, 7) FDM_CE(2, 5) FDM_CE(4, 8) FDM_CE(0, 7) FDM_CE(2, 4) FDM_CE(3, 8) FDM_CE(5, 6)
  FDM_CE(0, 2) FDM_CE(1, 3) FDM_CE(4, 5) FDM_CE(7, 8) FDM_CE(1, 4) FDM_CE(3, 6) FDM_CE(5, 7) FDM_CE(0, 1)
  FDM_CE(2, 4) FDM_CE(3, 5) FDM_CE(6, 8) FDM_CE(2, 3) FDM_CE(4, 5) FDM_CE(6, 7) FDM_CE(1, 2) FDM_CE(3, 4)
  FDM_CE(5, 6)
#undef FDM_CE
  const int m = n / 2;  // nth_element(size/2): 0..4
  out[ci * out_stride] = m == 0 ? w0 : (m == 1 ? w1 : (m == 2 ? w2 : (m == 3 ? w3 : w4)));
}

// ---- the two 3x3 stencils with the neighbourhood staged in LDS: one block per 32 x 8 cells, a ring of one cell,
// cells outside the stored window as NaN ("outside" and "no data" are one test), neighbours visited in the
// reference's order ----
constexpr int kS3R = 32, kS3C = 8, kS3Pitch = kS3R + 2, kS3Width = kS3C + 2;
__device__ __forceinline__ void stage_tile3(const PostGeom& p, const float* __restrict__ in, int in_stride, int tr,
                                            int tc, float* __restrict__ s_t) {
  const int r0 = tr * kS3R - 1, c0 = tc * kS3C - 1;
  for (int k = int(threadIdx.x); k < kS3Pitch * kS3Width; k += 256) {
    const int cc = k / kS3Pitch, rr = k - cc * kS3Pitch;
    s_t[k] = post_inside(p, r0 + rr, c0 + cc) ? in[post_index(p, r0 + rr, c0 + cc) * size_t(in_stride)]
                                              : __uint_as_float(0x7FC00000u);
  }
  __syncthreads();
}
inline __global__ __launch_bounds__(256) void k_inpaint_pass_tiled(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                            const float* __restrict__ in, int in_stride,
                                                            float* __restrict__ out, int out_stride, int min_valid) {
  __shared__ float s_t[kS3Pitch * kS3Width];
  const PostGeom p = post_geom(st, slot, G);
  const int tiles_r = (p.rows + kS3R - 1) / kS3R;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  stage_tile3(p, in, in_stride, tr, tc, s_t);
  const int lrl = int(threadIdx.x) & (kS3R - 1), lcl = int(threadIdx.x) >> 5;
  const int lr = tr * kS3R + lrl, lc = tc * kS3C + lcl;
  if (!post_inside(p, lr, lc)) return;
  const int base = (lcl + 1) * kS3Pitch + lrl + 1;
  float v = s_t[base];
  if (isnan(v)) {
    float sum = 0.0f;
    int count = 0;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
      for (int dc = -1; dc <= 1; ++dc) {
        if (dr == 0 && dc == 0) continue;
        const float n = s_t[base + dc * kS3Pitch + dr];
        if (isfinite(n)) {
          sum += n;
          ++count;
        }
      }
    if (count >= min_valid) v = sum / float(count);
  }
  out[post_index(p, lr, lc) * size_t(out_stride)] = v;
}
inline __global__ __launch_bounds__(256) void k_median3_tiled(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                       const float* __restrict__ in, float* __restrict__ out,
                                                       int out_stride, int min_valid) {
  __shared__ float s_t[kS3Pitch * kS3Width];
  const PostGeom p = post_geom(st, slot, G);
  const int tiles_r = (p.rows + kS3R - 1) / kS3R;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  stage_tile3(p, in, 1, tr, tc, s_t);
  const int lrl = int(threadIdx.x) & (kS3R - 1), lcl = int(threadIdx.x) >> 5;
  const int lr = tr * kS3R + lrl, lc = tc * kS3C + lcl;
  if (!post_inside(p, lr, lc)) return;
  const int base = (lcl + 1) * kS3Pitch + lrl + 1;
  if (!isfinite(s_t[base])) return;
  constexpr float kInf = __builtin_huge_valf();
  int n = 0;
#define FDM_W(k, dr, dc)                                   \
  float w##k = kInf;                                       \
  {                                                        \
    const float v = s_t[base + (dc) * kS3Pitch + (dr)];    \
    if (isfinite(v)) { w##k = v; ++n; }                    \
  }
  FDM_W(0, -1, -1) FDM_W(1, -1, 0) FDM_W(2, -1, 1) FDM_W(3, 0, -1) FDM_W(4, 0, 0) FDM_W(5, 0, 1)
  FDM_W(6, 1, -1) FDM_W(7, 1, 0) FDM_W(8, 1, 1)
#undef FDM_W
  if (n < min_valid) return;
#define FDM_CE(i, j) { const float lo_ = fminf(w##i, w##j); w##j = fmaxf(w##i, w##j); w##i = lo_; }
  FDM_CE(0, 3) FDM_CE(1, 7) FDM_CE(2, 5) FDM_CE(4, 8) FDM_CE(0, 7) FDM_CE(2, 4) FDM_CE(3, 8) FDM_CE(5, 6)
  FDM_CE(0, 2) FDM_CE(1, 3) FDM_CE(4, 5) FDM_CE(7, 8) FDM_CE(1, 4) FDM_CE(3, 6) FDM_CE(5, 7) FDM_CE(0, 1)
  FDM_CE(2, 4) FDM_CE(3, 5) FDM_CE(6, 8) FDM_CE(2, 3) FDM_CE(4, 5) FDM_CE(6, 7) FDM_CE(1, 2) FDM_CE(3, 4)
  FDM_CE(5, 6)
#undef FDM_CE
  const int m = n / 2;  // nth_element(size/2): 0..4
  out[post_index(p, lr, lc) * size_t(out_stride)] = m == 0 ? w0 : (m == 1 ? w1 : (m == 2 ? w2 : (m == 3 ? w3 : w4)));
}

// The stencil kernels work only for cells that hold data, and a map is rings and patches: with thread = cell a
// wavefront runs its ~3 000 instructions for whichever of its 64 cells are there (configs[3] after 12 scans: half).
// compact_cells hands the block's cells that pass `live` to consecutive threads in cell order (ballot + popcount, two
// barriers); the others get -1.  Every thread of the block must call it.
template <int THREADS, class LIVE>
__device__ __forceinline__ int compact_cells(LIVE live) {
  static_assert(THREADS % 64 == 0 && THREADS <= 512, "whole wavefronts");
  __shared__ uint16_t s_cell[THREADS];
  __shared__ int s_live[THREADS / 64];
  const int t = int(threadIdx.x), w = t >> 6;
  const bool mine = live(t);
  const unsigned long long bal = __ballot(mine);
  if ((t & 63) == 0) s_live[w] = __popcll(bal);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int k = 0; k < THREADS / 64; ++k) {
    const int n = s_live[k];
    before += k < w ? n : 0;
    total += n;
  }
  if (mine) s_cell[before + __popcll(bal & ((1ull << (t & 63)) - 1ull))] = uint16_t(t);
  __syncthreads();
  return t < total ? int(s_cell[t]) : -1;
}

// ---- uncertainty fusion (uncertainty_fusion.cpp:135-181); upper/lower are private copies ----
struct FusionParams {
  float inv_2s2, q_lower, q_upper;
  int min_valid, n_entries;
};
// SimpleWeightedECDF::quantile over samples sorted by value (uncertainty_fusion.cpp:63-91): ecdf_quantile_t
// Per-thread sample lists live in LDS when the disc is small enough (slot-major, thread-minor: the
// lanes of a wave hit consecutive banks), in scratch otherwise.  The lists are kept sorted by
// insertion, which is what makes the kernel O(n^2) per cell and is the price of summing the weights
// in the reference's (sorted) order.
constexpr int kFusionThreads = 128;
constexpr int kFusionLdsEntries = 64;  // 4 lists x 64 entries x 128 threads x 4 B = 128 KB of the CU's 160 KB

template <bool USE_LDS>
struct SampleList {
  float* base;  // LDS: list[slot * kFusionThreads]; scratch: list[slot]
  __device__ __forceinline__ float& operator[](int k) const { return USE_LDS ? base[k * kFusionThreads] : base[k]; }
};
template <class LIST>
__device__ __forceinline__ float ecdf_quantile_l(const LIST& val, const LIST& wgt, int n, float p) {
  if (n == 0) return __uint_as_float(0x7FC00000u);
  if (n == 1) return val[0];
  float total = 0.0f;
  for (int k = 0; k < n; ++k) total += wgt[k];
  if (total <= 0.0f) return __uint_as_float(0x7FC00000u);
  const float target = p * total;
  float cumulative = 0.0f;
  for (int k = 0; k < n; ++k) {
    cumulative += wgt[k];
    if (cumulative >= target) return val[k];
  }
  return val[n - 1];
}
template <bool USE_LDS>
__device__ __forceinline__ float ecdf_quantile_t(const SampleList<USE_LDS>& val, const SampleList<USE_LDS>& wgt, int n,
                                                 float p) {
  if (n == 0) return __uint_as_float(0x7FC00000u);
  if (n == 1) return val[0];
  float total = 0.0f;
  for (int k = 0; k < n; ++k) total += wgt[k];
  if (total <= 0.0f) return __uint_as_float(0x7FC00000u);
  const float target = p * total;
  float cumulative = 0.0f;
  for (int k = 0; k < n; ++k) {
    cumulative += wgt[k];
    if (cumulative >= target) return val[k];
  }
  return val[n - 1];
}

template <bool USE_LDS>
__global__ __launch_bounds__(kFusionThreads) void k_fusion(const GeomConst G, const DevState* __restrict__ st,
                                                           int slot, const RegionEntry* __restrict__ reg,
                                                           const FusionParams F, const float* __restrict__ up_in,
                                                           const float* __restrict__ lo_in,
                                                           float* __restrict__ up_out, int up_stride,
                                                           float* __restrict__ lo_out, int lo_stride,
                                                           unsigned ncell) {
  extern __shared__ float s_lists[];
  float scratch[USE_LDS ? 1 : 4 * kMaxRegion];
  const int cap = USE_LDS ? F.n_entries : kMaxRegion;  // LDS is sized for the region actually used
  float* pool = USE_LDS ? s_lists + threadIdx.x : scratch;
  const int pitch = USE_LDS ? cap * kFusionThreads : cap;
  const SampleList<USE_LDS> lv{pool}, lw{pool + pitch}, uv{pool + 2 * pitch}, uw{pool + 3 * pitch};
  const unsigned t = blockIdx.x * unsigned(kFusionThreads) + threadIdx.x;
  if (t >= ncell) return;
  const PostGeom p = post_geom(st, slot, G);
  const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
  const size_t ci = post_index(p, lr, lc);
  if (!isfinite(up_in[ci]) || !isfinite(lo_in[ci])) return;
  int nl = 0, nu = 0, valid = 0;
  for (int e = 0; e < F.n_entries; ++e) {
    const RegionEntry re = reg[e];
    if (!post_inside(p, lr + re.dr, lc + re.dc)) continue;
    const size_t ni = post_index(p, lr + re.dr, lc + re.dc);
    const float nu_v = up_in[ni], nl_v = lo_in[ni];
    if (!isfinite(nu_v) || !isfinite(nl_v)) continue;
    const float w_spatial = re.w;  // std::exp(-dist_sq * inv_2sigma_sq): a property of the entry, from the host
    const float range = nu_v - nl_v;
    const float w_range = 1.0f / (range + 1e-4f);
    const float weight = w_spatial * w_range;
    if (weight > 1e-6f) {  // SimpleWeightedECDF::add; the values are finite here
      int k = nl++;
      while (k > 0 && lv[k - 1] > nl_v) { lv[k] = lv[k - 1]; lw[k] = lw[k - 1]; --k; }
      lv[k] = nl_v; lw[k] = weight;
      k = nu++;
      while (k > 0 && uv[k - 1] > nu_v) { uv[k] = uv[k - 1]; uw[k] = uw[k - 1]; --k; }
      uv[k] = nu_v; uw[k] = weight;
    }
    ++valid;
  }
  if (valid < F.min_valid) return;
  const float lower = ecdf_quantile_t<USE_LDS>(lv, lw, nl, F.q_lower);
  const float upper = ecdf_quantile_t<USE_LDS>(uv, uw, nu, F.q_upper);
  if (isfinite(lower) && isfinite(upper)) {
    up_out[ci * up_stride] = upper;
    lo_out[ci * lo_stride] = lower;
  }
}

// Discs of more than 32 cells: ONE WAVEFRONT per cell, the samples sorted in LDS.  The insertion sorts of k_fusion /
// k_fusion_big keep a list per THREAD (O(n^2) moves through LDS, scratch or a global pool: 134 ms for a 317-cell disc on
// the 1200 x 1200 map); here a wavefront gathers the disc once (lane = entry), sorts 64-bit keys ord(value) << 32 | entry
// — ties in entry order, which is what the reference's insertion into a sorted list leaves — with a bitonic network
// in wave-private LDS, lays the weights out in sorted order and walks SimpleWeightedECDF::quantile's two sequential sums
// (uncertainty_fusion.cpp:63-91: the float sums are in sorted order) with the values handed from lane to lane.
// N: the entry count padded to a power of two (<= kFusionWaveMax); LDS per wavefront = 24 N bytes.
constexpr int kFusionWaveMax = 1024;
constexpr unsigned kFusionWaveThreads = 128u;  // two wavefronts per block
__host__ __device__ constexpr unsigned fusion_wave_lds_bytes(unsigned n_pad) { return 2u * 24u * n_pad; }
// (LDS that only one wavefront touches needs no s_barrier: its DS instructions execute in order; this keeps the compiler
// from moving accesses of different lanes across the point)
__device__ __forceinline__ void fusion_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float fusion_wave_quantile(const unsigned long long* __restrict__ key, const float* __restrict__ ws,
                                                      const float* __restrict__ val, const unsigned n, const float p,
                                                      const unsigned lane) {
  const float nanv = __uint_as_float(0x7FC00000u);
  if (n == 0u) return nanv;
  if (n == 1u) return val[uint32_t(key[0])];
  // total, then the cumulative sum up to the first sample that reaches p * total — every lane walks the same chain
  float total = 0.0f;
  for (unsigned c = 0; c < n; c += 64u) {
    const float mine = c + lane < n ? ws[c + lane] : 0.0f;
    const unsigned m = min(64u, n - c);
    for (unsigned l = 0; l < m; ++l) total += __uint_as_float(unsigned(__builtin_amdgcn_readlane(int(__float_as_uint(mine)), int(l))));
  }
  if (total <= 0.0f) return nanv;
  const float target = p * total;
  float cumulative = 0.0f;
  for (unsigned c = 0; c < n; c += 64u) {
    const float mine = c + lane < n ? ws[c + lane] : 0.0f;
    const unsigned m = min(64u, n - c);
    for (unsigned l = 0; l < m; ++l) {
      cumulative += __uint_as_float(unsigned(__builtin_amdgcn_readlane(int(__float_as_uint(mine)), int(l))));
      if (cumulative >= target) return val[uint32_t(key[c + l])];
    }
  }
  return val[uint32_t(key[n - 1u])];
}
inline __global__ __launch_bounds__(kFusionWaveThreads) void k_fusion_wave(const GeomConst G, const DevState* __restrict__ st,
                                                                    int slot, const RegionEntry* __restrict__ reg,
                                                                    const FusionParams F, const unsigned n_pad,
                                                                    const float* __restrict__ up_in,
                                                                    const float* __restrict__ lo_in,
                                                                    float* __restrict__ up_out, int up_stride,
                                                                    float* __restrict__ lo_out, int lo_stride,
                                                                    unsigned ncell) {
  extern __shared__ unsigned char s_fw[];
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  unsigned long long* const key = reinterpret_cast<unsigned long long*>(s_fw + size_t(wave) * 24u * n_pad);
  float* const s_w = reinterpret_cast<float*>(key + n_pad);  // weight by entry
  float* const s_ws = s_w + n_pad;                            // weight by sorted position
  float* const s_lo = s_ws + n_pad;                           // the entries' lower / upper values
  float* const s_up = s_lo + n_pad;
  const PostGeom p = post_geom(st, slot, G);
  const unsigned waves = gridDim.x * (kFusionWaveThreads / 64u);
  for (unsigned t = blockIdx.x * (kFusionWaveThreads / 64u) + wave; t < ncell; t += waves) {  // (wave-uniform)
    const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
    const size_t ci = post_index(p, lr, lc);
    if (!isfinite(up_in[ci]) || !isfinite(lo_in[ci])) continue;
    fusion_wave_sync();  // (the previous cell's readers are done with the arrays)
    unsigned n = 0u, valid = 0u;
    for (unsigned e = lane; e < n_pad; e += 64u) {
      bool ok = false, take = false;
      float nu_v = 0.0f, nl_v = 0.0f, weight = 0.0f;
      if (e < unsigned(F.n_entries)) {
        const RegionEntry re = reg[e];
        if (post_inside(p, lr + re.dr, lc + re.dc)) {
          const size_t ni = post_index(p, lr + re.dr, lc + re.dc);
          nu_v = up_in[ni]; nl_v = lo_in[ni];
          if (isfinite(nu_v) && isfinite(nl_v)) {
            ok = true;
            weight = re.w * (1.0f / ((nu_v - nl_v) + 1e-4f));
            take = weight > 1e-6f;  // SimpleWeightedECDF::add; the values are finite here
          }
        }
      }
      s_w[e] = take ? weight : 0.0f;
      s_lo[e] = nl_v;
      s_up[e] = nu_v;
      // (sorted behind every sample)
      key[e] = take ? (((unsigned long long)ord(nl_v == 0.0f ? 0.0f : nl_v) << 32) | e) : ~0ull;
      n += unsigned(__popcll(__ballot(take)));
      valid += unsigned(__popcll(__ballot(ok)));
    }
    if (int(valid) < F.min_valid) continue;
    float q[2];
#pragma unroll 1
    for (int list = 0; list < 2; ++list) {
      if (list == 1) {  // the upper list: the same entries keyed by their upper value
        fusion_wave_sync();
        for (unsigned e = lane; e < n_pad; e += 64u)
          key[e] = s_w[e] != 0.0f ? (((unsigned long long)ord(s_up[e] == 0.0f ? 0.0f : s_up[e]) << 32) | e) : ~0ull;
      }
      // bitonic sort, ascending
      for (unsigned k = 2u; k <= n_pad; k <<= 1) {
        for (unsigned j = k >> 1; j > 0u; j >>= 1) {
          fusion_wave_sync();
          for (unsigned i = lane; i < n_pad / 2u; i += 64u) {
            const unsigned a = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), b = a | j;
            const unsigned long long x = key[a], y = key[b];
            const bool asc = (a & k) == 0u;
            if ((x > y) == asc) { key[a] = y; key[b] = x; }
          }
        }
      }
      fusion_wave_sync();
      for (unsigned pos = lane; pos < n; pos += 64u) s_ws[pos] = s_w[uint32_t(key[pos])];
      fusion_wave_sync();
      q[list] = fusion_wave_quantile(key, s_ws, list == 0 ? s_lo : s_up, n, list == 0 ? F.q_lower : F.q_upper, lane);
    }
    if (lane == 0u && isfinite(q[0]) && isfinite(q[1])) {
      up_out[ci * up_stride] = q[1];
      lo_out[ci * lo_stride] = q[0];
    }
  }
}

// discs of more than kFusionWaveMax cells: the four lists in the global pool (see k_median_big)
inline __global__ __launch_bounds__(kFusionThreads) void k_fusion_big(const GeomConst G, const DevState* __restrict__ st,
                                                               int slot, const RegionEntry* __restrict__ reg,
                                                               const FusionParams F, const float* __restrict__ up_in,
                                                               const float* __restrict__ lo_in,
                                                               float* __restrict__ up_out, int up_stride,
                                                               float* __restrict__ lo_out, int lo_stride,
                                                               unsigned ncell, float* __restrict__ pool) {
  const unsigned tid = blockIdx.x * unsigned(kFusionThreads) + threadIdx.x, pitch = gridDim.x * unsigned(kFusionThreads);
  const size_t list = size_t(F.n_entries) * pitch;
  const PoolList lv{pool + tid, pitch}, lw{pool + list + tid, pitch}, uv{pool + 2 * list + tid, pitch},
      uw{pool + 3 * list + tid, pitch};
  const PostGeom p = post_geom(st, slot, G);
  for (unsigned t = tid; t < ncell; t += pitch) {
    const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
    const size_t ci = post_index(p, lr, lc);
    if (!isfinite(up_in[ci]) || !isfinite(lo_in[ci])) continue;
    int nl = 0, nu = 0, valid = 0;
    for (int e = 0; e < F.n_entries; ++e) {
      const RegionEntry re = reg[e];
      if (!post_inside(p, lr + re.dr, lc + re.dc)) continue;
      const size_t ni = post_index(p, lr + re.dr, lc + re.dc);
      const float nu_v = up_in[ni], nl_v = lo_in[ni];
      if (!isfinite(nu_v) || !isfinite(nl_v)) continue;
      const float weight = re.w * (1.0f / ((nu_v - nl_v) + 1e-4f));
      if (weight > 1e-6f) {
        int k = nl++;
        while (k > 0 && lv[k - 1] > nl_v) { lv[k] = lv[k - 1]; lw[k] = lw[k - 1]; --k; }
        lv[k] = nl_v; lw[k] = weight;
        k = nu++;
        while (k > 0 && uv[k - 1] > nu_v) { uv[k] = uv[k - 1]; uw[k] = uw[k - 1]; --k; }
        uv[k] = nu_v; uw[k] = weight;
      }
      ++valid;
    }
    if (valid < F.min_valid) continue;
    const float lower = ecdf_quantile_l(lv, lw, nl, F.q_lower);
    const float upper = ecdf_quantile_l(uv, uw, nu, F.q_upper);
    if (isfinite(lower) && isfinite(upper)) {
      up_out[ci * up_stride] = upper;
      lo_out[ci * lo_stride] = lower;
    }
  }
}

// Regions of up to 32 cells (the default radius: 29): no lists in memory at all.  A sample is one
// 64-bit register, ord(value) << 32 | entry index, so that ties keep the entry order the insertion
// sort above gives them; a fixed 191-step merge-exchange network (fdm_sortnet32.inc) sorts the
// lower and the upper samples in registers, and the weights — shared by both lists — sit in LDS by
// entry index.  The list kernel holds one wave per SIMD (59 KB of LDS per 128 threads) and walks a
// dependent chain of ~1900 LDS operations per cell; this one is straight-line ALU code.
#include "fdm_sortnet32.inc"
#define FDM_E32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
  X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
// One list (lower or upper bounds) of one cell: gather, sort, weighted quantile.  The two lists are
// done one after the other so that only 32 samples are live (2 x 32 x 64 bit would hold the kernel
// at one wave per SIMD); the second pass re-reads the neighbours from L2 and the weights from LDS.
// TILED: up_in / lo_in are the block's LDS tiles (cells outside the stored window hold NaN), `lr` is the thread's
// cell inside them and `lc` the tile pitch — a neighbour is one scalar multiply-add and two LDS reads.
template <bool UPPER, bool TILED = false>
__device__ __forceinline__ float fusion_list32(const PostGeom& p, int lr, int lc, const RegionEntry* __restrict__ reg,
                                               const FusionParams& F, const float* __restrict__ up_in,
                                               const float* __restrict__ lo_in, float (*s_w)[kFusionThreads],
                                               int& valid, uint32_t& taken) {
  constexpr unsigned long long kNone = ~0ull;  // sorts behind every sample
  int n = 0;
#define FDM_DECL(e) unsigned long long a##e = kNone;
  FDM_E32(FDM_DECL)
#undef FDM_DECL
  if (!UPPER) {  // the first list decides which entries count (bit e of `taken`) and leaves their weights in LDS
    valid = 0;
    taken = 0u;
#define FDM_GATHER(e)                                                                         \
  if (e < F.n_entries) {                                                                       \
    const RegionEntry re = reg[e];                                                             \
    if (TILED || post_inside(p, lr + re.dr, lc + re.dc)) {                                     \
      const size_t ni = TILED ? size_t(lr + re.dc * lc + re.dr) : post_index(p, lr + re.dr, lc + re.dc); \
      const float nu_v = up_in[ni], nl_v = lo_in[ni];                                          \
      if (isfinite(nu_v) && isfinite(nl_v)) {                                                  \
        const float weight = re.w * (1.0f / ((nu_v - nl_v) + 1e-4f));                          \
        if (weight > 1e-6f) {                                                                  \
          s_w[e][threadIdx.x] = weight;                                                        \
          a##e = ((unsigned long long)ord(nl_v == 0.0f ? 0.0f : nl_v) << 32) | unsigned(e);    \
          taken |= 1u << e;                                                                    \
          ++n;                                                                                 \
        }                                                                                      \
        ++valid;                                                                               \
      }                                                                                        \
    }                                                                                          \
  }
    FDM_E32(FDM_GATHER)
#undef FDM_GATHER
  } else {  // the second list: the same entries keyed by their upper value — no test, no weight to work out again
    n = __popc(taken);
#define FDM_GATHER_U(e)                                                                       \
  if ((taken >> e) & 1u) {                                                                     \
    const RegionEntry re = reg[e];                                                             \
    const size_t ni = TILED ? size_t(lr + re.dc * lc + re.dr) : post_index(p, lr + re.dr, lc + re.dc); \
    const float nu_v = up_in[ni];                                                              \
    a##e = ((unsigned long long)ord(nu_v == 0.0f ? 0.0f : nu_v) << 32) | unsigned(e);          \
  }
    FDM_E32(FDM_GATHER_U)
#undef FDM_GATHER_U
  }
  if (valid < F.min_valid) return __uint_as_float(0x7FC00000u);
#define FDM_CE(i, j) { const bool sw = a##i > a##j; const unsigned long long lo_ = sw ? a##j : a##i; a##j = sw ? a##i : a##j; a##i = lo_; }
  FDM_NET32(FDM_CE)
#undef FDM_CE
  // SimpleWeightedECDF::quantile over the sorted samples (uncertainty_fusion.cpp:63-91)
  float q = __uint_as_float(0x7FC00000u);
  if (n == 1) {
    q = unord(uint32_t(a0 >> 32));
  } else if (n > 1) {
    float total = 0.0f;
#define FDM_TOTAL(k) if (k < n) total += s_w[a##k & 31u][threadIdx.x];
    FDM_E32(FDM_TOTAL)
#undef FDM_TOTAL
    if (total > 0.0f) {
      const float target = (UPPER ? F.q_upper : F.q_lower) * total;
      float cum = 0.0f;
      bool found = false;
#define FDM_Q(k) if (k < n && !found) { cum += s_w[a##k & 31u][threadIdx.x]; q = unord(uint32_t(a##k >> 32)); found = cum >= target; }
      FDM_E32(FDM_Q)
#undef FDM_Q
    }
  }
  return q;
}

inline __global__ __launch_bounds__(kFusionThreads) void k_fusion_net32(const GeomConst G, const DevState* __restrict__ st,
                                                                 int slot, const RegionEntry* __restrict__ reg,
                                                                 const FusionParams F, const float* __restrict__ up_in,
                                                                 const float* __restrict__ lo_in,
                                                                 float* __restrict__ up_out, int up_stride,
                                                                 float* __restrict__ lo_out, int lo_stride,
                                                                 unsigned ncell) {
  __shared__ float s_w[32][kFusionThreads];
  const unsigned t = blockIdx.x * unsigned(kFusionThreads) + threadIdx.x;
  if (t >= ncell) return;
  const PostGeom p = post_geom(st, slot, G);
  const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
  const size_t ci = post_index(p, lr, lc);
  if (!isfinite(up_in[ci]) || !isfinite(lo_in[ci])) return;
  int valid = 0;
  uint32_t taken = 0u;
  const float lower = fusion_list32<false>(p, lr, lc, reg, F, up_in, lo_in, s_w, valid, taken);
  if (valid < F.min_valid) return;
  const float upper = fusion_list32<true>(p, lr, lc, reg, F, up_in, lo_in, s_w, valid, taken);
  if (isfinite(lower) && isfinite(upper)) {
    up_out[ci * up_stride] = upper;
    lo_out[ci * lo_stride] = lower;
  }
}
// The same with the neighbourhood staged in LDS: one block per 32 x 4 cells, both layers' tiles with a ring of
// `halo` cells (NaN outside the stored window, so "outside" and "no data" are one test).  The two passes (lower
// list, then upper) re-read LDS instead of L2, and a neighbour costs no 64-bit index arithmetic.
constexpr int kFusTileR = 32, kFusTileC = kFusionThreads / kFusTileR, kFusHaloMax = 8;
inline __global__ __launch_bounds__(kFusionThreads) void k_fusion_net32_tiled(const GeomConst G, const DevState* __restrict__ st,
                                                                       int slot, const RegionEntry* __restrict__ reg,
                                                                       const FusionParams F, int halo,
                                                                       const float* __restrict__ up_in,
                                                                       const float* __restrict__ lo_in,
                                                                       float* __restrict__ up_out, int up_stride,
                                                                       float* __restrict__ lo_out, int lo_stride) {
  __shared__ float s_w[32][kFusionThreads];
  __shared__ float s_up[(kFusTileR + 2 * kFusHaloMax) * (kFusTileC + 2 * kFusHaloMax)];
  __shared__ float s_lo[(kFusTileR + 2 * kFusHaloMax) * (kFusTileC + 2 * kFusHaloMax)];
  const PostGeom p = post_geom(st, slot, G);
  const int tiles_r = (p.rows + kFusTileR - 1) / kFusTileR;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  const int pitch = kFusTileR + 2 * halo, width = kFusTileC + 2 * halo;
  const int r0 = tr * kFusTileR - halo, c0 = tc * kFusTileC - halo;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (int k = int(threadIdx.x); k < pitch * width; k += kFusionThreads) {
    const int cc = k / pitch, rr = k - cc * pitch;
    const bool in = post_inside(p, r0 + rr, c0 + cc);
    const size_t gi = in ? post_index(p, r0 + rr, c0 + cc) : 0;
    s_up[k] = in ? up_in[gi] : nanv;
    s_lo[k] = in ? lo_in[gi] : nanv;
  }
  __syncthreads();
  const int lrl = int(threadIdx.x) & (kFusTileR - 1), lcl = int(threadIdx.x) / kFusTileR;
  const int lr = tr * kFusTileR + lrl, lc = tc * kFusTileC + lcl;
  if (!post_inside(p, lr, lc)) return;
  const int base = (lcl + halo) * pitch + lrl + halo;
  if (!isfinite(s_up[base]) || !isfinite(s_lo[base])) return;
  const size_t ci = post_index(p, lr, lc);
  int valid = 0;
  uint32_t taken = 0u;
  const float lower = fusion_list32<false, true>(p, base, pitch, reg, F, s_up, s_lo, s_w, valid, taken);
  if (valid < F.min_valid) return;
  const float upper = fusion_list32<true, true>(p, base, pitch, reg, F, s_up, s_lo, s_w, valid, taken);
  if (isfinite(lower) && isfinite(upper)) {
    up_out[ci * up_stride] = upper;
    lo_out[ci * lo_stride] = lower;
  }
}
// Round 6: the same stage with each sample as a POSITIVE DOUBLE of one fixed exponent (2^33): mantissa bits 50..19 hold
// ord(value), bits 13..9 the entry index, so that v_min_f64 / v_max_f64 order samples by (value, entry) and a
// compare-exchange is TWO instructions (the 64-bit integer version: v_cmp_u64 + four v_cndmask + two wait states).
// Every entry of the region is a sample in both lists — one whose weight does not count (no data, or weight <= 1e-6)
// has the weight 0.0 in its LDS slot: x + 0.0f == x for the sums of positive weights the reference forms, and the
// quantile can only stop at a sample that moved the cumulative sum (the host sends quantiles outside [1e-6, 1] to
// k_fusion_net32_tiled).  So the gather has no branches, the upper list needs no tests at all, and the walk over
// the sorted samples is branch-free: pass one replaces each sample by {ord(value), weight} and sums the total,
// pass two keeps the value of the first sample whose cumulative weight reaches the target.  N = the disc's size (5, 9,
// 13, 21, 25 or 29 cells: there is no disc of 30 .. 32): only the compare-exchanges of Batcher's 32-network between
// slots below N are kept — slots from N on would hold padding that never moves (171 exchanges for 29, 26 for 9).
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
constexpr uint32_t kFusEShift = 9u, kFusEMask = 31u << kFusEShift;  // s_w[e][t]: byte offset e * 512 + t * 4
static_assert(kFusionThreads * 4 == 1 << kFusEShift, "the entry field of a sample is the byte offset of its weight row");
__device__ __forceinline__ double fusion_sample(float v, uint32_t e) {
  const uint32_t i = ord(v + 0.0f);  // -0 -> +0: the reference's `<` calls them equal
  return __hiloint2double(int(__builtin_amdgcn_alignbit(0x840u, i, 13)), int((i << 19) | (e << kFusEShift)));
}
__device__ __forceinline__ double fusion_pad(uint32_t e) {  // behind every value, weight slot e (0.0)
  return __hiloint2double(int(0x4207FFFFu), int(0xFFF80000u | (e << kFusEShift)));
}
template <int N, bool FULL, bool UPPER>
__device__ __forceinline__ float fusion_list_f64(int base, int pitch, const RegionEntry* __restrict__ reg,
                                                 const FusionParams& F, const float2* __restrict__ s_ul,
                                                 float (*s_w)[kFusionThreads], int& valid, bool& any) {
  const uint32_t t4 = threadIdx.x * 4u;
#define FDM_DECL(e) double d##e = 0.0;
  FDM_E32(FDM_DECL)
#undef FDM_DECL
  if (!UPPER) {
    valid = 0;
    any = false;
#define FDM_GATHER(e)                                                                    \
  if (e < N) {                                                                            \
    if (FULL || e < F.n_entries) {                                                        \
      const RegionEntry re = reg[e];                                                      \
      const float2 ul = s_ul[base + re.dc * pitch + re.dr];                               \
      const bool fin = isfinite(ul.x) && isfinite(ul.y);                                  \
      const float weight = re.w * (1.0f / ((ul.x - ul.y) + 1e-4f));                       \
      const bool tk = fin && weight > 1e-6f;                                              \
      s_w[e][threadIdx.x] = tk ? weight : 0.0f;                                           \
      any = any || tk;                                                                    \
      valid += fin ? 1 : 0;                                                               \
      d##e = fusion_sample(ul.y, e);                                                      \
    } else {                                                                              \
      s_w[e][threadIdx.x] = 0.0f;                                                         \
      d##e = fusion_pad(e);                                                               \
    }                                                                                     \
  }
    FDM_E32(FDM_GATHER)
#undef FDM_GATHER
    if (valid < F.min_valid) return __uint_as_float(0x7FC00000u);
  } else {
#define FDM_GATHER_U(e)                                                                  \
  if (e < N) {                                                                            \
    if (FULL || e < F.n_entries) {                                                        \
      const RegionEntry re = reg[e];                                                      \
      d##e = fusion_sample(s_ul[base + re.dc * pitch + re.dr].x, e);                      \
    } else {                                                                              \
      d##e = fusion_pad(e);                                                               \
    }                                                                                     \
  }
    FDM_E32(FDM_GATHER_U)
#undef FDM_GATHER_U
  }
#define FDM_CE(i, j) if (j < N) { const double lo_ = vmin_f64(d##i, d##j); d##j = vmax_f64(d##i, d##j); d##i = lo_; }
  FDM_NET32(FDM_CE)
#undef FDM_CE
  // SimpleWeightedECDF::quantile (uncertainty_fusion.cpp:63-91): total and cumulative sums in sorted order
  float total = 0.0f;
#define FDM_W1(k)                                                                                          \
  uint32_t v##k = 0u;                                                                                      \
  float w##k = 0.0f;                                                                                       \
  if (k < N) {                                                                                             \
    const uint32_t lo_ = uint32_t(__double2loint(d##k)), hi_ = uint32_t(__double2hiint(d##k));            \
    v##k = __builtin_amdgcn_alignbit(hi_, lo_, 19);                                                        \
    w##k = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(&s_w[0][0]) + ((lo_ & kFusEMask) | t4)); \
    total += w##k;                                                                                         \
  }
  FDM_E32(FDM_W1)
#undef FDM_W1
  const float target = (UPPER ? F.q_upper : F.q_lower) * total;
  float cum = 0.0f;
  uint32_t qi = v0;
#define FDM_W2(k) if (k < N) { qi = cum >= target ? qi : v##k; cum += w##k; }
  FDM_E32(FDM_W2)
#undef FDM_W2
  return any ? unord(qi) : __uint_as_float(0x7FC00000u);
}

// one block per 32 x 4 cells; dynamic LDS: the tile of {upper, lower} pairs with its ring of `halo` cells (NaN outside
// the stored window) — fusion_f64_lds_bytes(halo); FULL: the region has exactly N entries
__host__ __device__ constexpr unsigned fusion_f64_lds_bytes(int halo) {
  return unsigned(kFusTileR + 2 * halo) * unsigned(kFusTileC + 2 * halo) * 8u;
}
template <int N, bool FULL>
__global__ __launch_bounds__(kFusionThreads, 4) void k_fusion_f64_tiled(const GeomConst G, const DevState* __restrict__ st,
                                                                     int slot, const RegionEntry* __restrict__ reg,
                                                                     const FusionParams F, int halo,
                                                                     const float* __restrict__ up_in,
                                                                     const float* __restrict__ lo_in,
                                                                     float* __restrict__ up_out, int up_stride,
                                                                     float* __restrict__ lo_out, int lo_stride) {
  __shared__ float s_w[N][kFusionThreads];
  extern __shared__ float2 s_ul[];
  const PostGeom p = post_geom(st, slot, G);
  const int tiles_r = (p.rows + kFusTileR - 1) / kFusTileR;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  const int pitch = kFusTileR + 2 * halo, width = kFusTileC + 2 * halo;
  const int r0 = tr * kFusTileR - halo, c0 = tc * kFusTileC - halo;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (int k = int(threadIdx.x); k < pitch * width; k += kFusionThreads) {
    const int cc = k / pitch, rr = k - cc * pitch;
    const bool in = post_inside(p, r0 + rr, c0 + cc);
    const size_t gi = in ? post_index(p, r0 + rr, c0 + cc) : 0;
    s_ul[k] = make_float2(in ? up_in[gi] : nanv, in ? lo_in[gi] : nanv);
  }
  __syncthreads();
  const int cell = compact_cells<kFusionThreads>([&](int t) {
    const int rr = t & (kFusTileR - 1), cc = t / kFusTileR;
    const float2 ul = s_ul[(cc + halo) * pitch + rr + halo];
    return post_inside(p, tr * kFusTileR + rr, tc * kFusTileC + cc) && isfinite(ul.x) && isfinite(ul.y);
  });
  if (cell < 0) return;
  const int lrl = cell & (kFusTileR - 1), lcl = cell / kFusTileR;
  const int lr = tr * kFusTileR + lrl, lc = tc * kFusTileC + lcl;
  const int base = (lcl + halo) * pitch + lrl + halo;
  int valid = 0;
  bool any = false;
  const float lower = fusion_list_f64<N, FULL, false>(base, pitch, reg, F, s_ul, s_w, valid, any);
  if (valid < F.min_valid) return;
  const float upper = fusion_list_f64<N, FULL, true>(base, pitch, reg, F, s_ul, s_w, valid, any);
  if (isfinite(lower) && isfinite(upper)) {
    const size_t ci = post_index(p, lr, lc);
    up_out[ci * up_stride] = upper;
    lo_out[ci * lo_stride] = lower;
  }
}
#undef FDM_E32

// ---- Eigen::SelfAdjointEigenSolver<Matrix3f>::computeDirect (Eigen 3.4, 3x3 closed form) ----
// transcendental steps are evaluated in double and rounded (closest to the host libm's float results)
__device__ __forceinline__ void eig3_cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float eig3_sqnorm(const float* a) { return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]); }
__device__ __forceinline__ void eig3_kernel(const float* mat, float* res, float* representative) {
  int i0 = 0;
  float best = fabsf(mat[0]);
  if (fabsf(mat[4]) > best) { best = fabsf(mat[4]); i0 = 1; }
  if (fabsf(mat[8]) > best) { best = fabsf(mat[8]); i0 = 2; }
  const int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
  float rep[3], a[3], b[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rep[r] = i0 == 0 ? mat[r] : (i0 == 1 ? mat[3 + r] : mat[6 + r]);
    a[r] = i1 == 0 ? mat[r] : (i1 == 1 ? mat[3 + r] : mat[6 + r]);
    b[r] = i2 == 0 ? mat[r] : (i2 == 1 ? mat[3 + r] : mat[6 + r]);
    representative[r] = rep[r];
  }
  float c0[3], c1[3];
  eig3_cross(rep, a, c0);
  eig3_cross(rep, b, c1);
  const float n0 = eig3_sqnorm(c0), n1 = eig3_sqnorm(c1);
  if (n0 > n1) {
    const float s = sqrtf(n0);
    res[0] = c0[0] / s; res[1] = c0[1] / s; res[2] = c0[2] / s;
  } else {
    const float s = sqrtf(n1);
    res[0] = c1[0] / s; res[1] = c1[1] / s; res[2] = c1[2] / s;
  }
}
// cov column-major; returns eigenvalues ascending in val, the eigenvector of val[0] in v0
__device__ __forceinline__ void eig3_direct(const float* cov, float* val, float* v0) {
  const float shift = (cov[0] + cov[4] + cov[8]) / 3.0f;
  float sm[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) sm[c * 3 + r] = r >= c ? cov[c * 3 + r] : cov[r * 3 + c];
  sm[0] -= shift; sm[4] -= shift; sm[8] -= shift;
  float scale = 0.0f;
#pragma unroll
  for (int k = 0; k < 9; ++k) scale = fmaxf(scale, fabsf(sm[k]));
  if (scale > 0.0f) {
#pragma unroll
    for (int k = 0; k < 9; ++k) sm[k] /= scale;
  }
  // computeRoots
  const float s_inv3 = 1.0f / 3.0f, s_sqrt3 = sqrtf(3.0f);
  const float c0 = sm[0] * sm[4] * sm[8] + 2.0f * sm[1] * sm[2] * sm[5] - sm[0] * sm[5] * sm[5] -
                   sm[4] * sm[2] * sm[2] - sm[8] * sm[1] * sm[1];
  const float c1 = sm[0] * sm[4] - sm[1] * sm[1] + sm[0] * sm[8] - sm[2] * sm[2] + sm[4] * sm[8] - sm[5] * sm[5];
  const float c2 = sm[0] + sm[4] + sm[8];
  const float c2_over_3 = c2 * s_inv3;
  float a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
  a_over_3 = fmaxf(a_over_3, 0.0f);
  const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
  float q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
  q = fmaxf(q, 0.0f);
  const float rho = sqrtf(a_over_3);
  const float theta = static_cast<float>(atan2(static_cast<double>(sqrtf(q)), static_cast<double>(half_b))) * s_inv3;
  double sin_d, cos_d;  // one argument reduction for the pair (the same doubles sin() and cos() return)
  sincos(static_cast<double>(theta), &sin_d, &cos_d);
  const float cos_theta = static_cast<float>(cos_d);
  const float sin_theta = static_cast<float>(sin_d);
  float ev[3];
  ev[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  ev[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  ev[2] = c2_over_3 + 2.0f * rho * cos_theta;
  // only the eigenvector of the SMALLEST eigenvalue is consumed (feature_extraction.cpp:96)
  const float eps = 1.1920929e-07f;
  if ((ev[2] - ev[0]) <= eps) {
    v0[0] = 1.0f; v0[1] = 0.0f; v0[2] = 0.0f;  // eivecs.setIdentity()
  } else {
    float d0 = ev[2] - ev[1];
    const float d1 = ev[1] - ev[0];
    int k = 0, l = 2;
    if (d0 > d1) { k = 2; l = 0; d0 = d1; }
    float tmp[9], vk[3], vl[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) tmp[i] = sm[i];
    const float evk = k == 0 ? ev[0] : ev[2], evl = l == 0 ? ev[0] : ev[2];
    tmp[0] -= evk; tmp[4] -= evk; tmp[8] -= evk;
    eig3_kernel(tmp, vk, vl);
    if (k == 0) {
      v0[0] = vk[0]; v0[1] = vk[1]; v0[2] = vk[2];
    } else {  // column 0 is the l-th: second kernel or the re-orthonormalised representative
      if (d0 <= 2.0f * eps * d1) {
        const float dot = vk[0] * vl[0] + (vk[1] * vl[1] + vk[2] * vl[2]);
        vl[0] -= dot * vl[0]; vl[1] -= dot * vl[1]; vl[2] -= dot * vl[2];
        const float n = sqrtf(eig3_sqnorm(vl));
        vl[0] /= n; vl[1] /= n; vl[2] /= n;
      } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) tmp[i] = sm[i];
        tmp[0] -= evl; tmp[4] -= evl; tmp[8] -= evl;
        float dummy[3];
        eig3_kernel(tmp, vl, dummy);
      }
      v0[0] = vl[0]; v0[1] = vl[1]; v0[2] = vl[2];
    }
  }
  val[0] = ev[0] * scale + shift;
  val[1] = ev[1] * scale + shift;
  val[2] = ev[2] * scale + shift;
}

// ---- feature extraction (feature_extraction.cpp:59-116) ----
struct FeatureParams {
  float resf, lo_pct, hi_pct;
  int min_valid, n_entries;
};
struct FeatureOut {
  float *step, *slope, *roughness, *curvature, *nx, *ny, *nz;
};
// covariance -> PCA -> the seven layers (feature_extraction.cpp:85-116).  sum / sq are the accumulated
// displacement sums in the reference's order; z_lo / z_hi come from the caller's order statistics.
__device__ __forceinline__ bool features_cov(const float* sum, const float* sq, int count, float* cov, float* trace) {
  const float inv_n = 1.0f / float(count);
  const float mean[3] = {sum[0] * inv_n, sum[1] * inv_n, sum[2] * inv_n};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) cov[c * 3 + r] = sq[c * 3 + r] * inv_n - mean[r] * mean[c];
  *trace = cov[0] + cov[4] + cov[8];
  return !(*trace < 1.1920929e-07f);  // computePCA: degenerate covariance
}
__device__ __forceinline__ void features_store(const FeatureOut& O, size_t ci, const float* val, float* normal,
                                               float trace, float z_lo, float z_hi) {
  if (normal[2] < 0.0f) { normal[0] = -normal[0]; normal[1] = -normal[1]; normal[2] = -normal[2]; }
  O.step[ci] = z_hi - z_lo;
  O.slope[ci] = static_cast<float>(acos(static_cast<double>(fabsf(normal[2])))) * 180.0f / 3.14159274101257324f;
  O.roughness[ci] = sqrtf(val[0]);
  O.curvature[ci] = (trace > 0.0f) ? fabsf(val[0] / trace) : 0.0f;
  O.nx[ci] = normal[0];
  O.ny[ci] = normal[1];
  O.nz[ci] = normal[2];
}

// `step` needs two order statistics of the neighbourhood's heights (feature_extraction.cpp:100-103).
// TOPK > 0: they are among the TOPK smallest / TOPK largest values, which are kept in registers by
// unrolled compare-exchange chains (the host checks the percentiles make that true for the region);
// TOPK == 0: any percentile pair, full insertion sort in scratch.
template <int TOPK>
__global__ __launch_bounds__(256) void k_features(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                  const RegionEntry* __restrict__ reg, const FeatureParams F,
                                                  const float* __restrict__ elev, int elev_stride,
                                                  const FeatureOut O, unsigned ncell) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= ncell) return;
  const PostGeom p = post_geom(st, slot, G);
  const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
  const size_t ci = post_index(p, lr, lc);
  const float center_z = elev[ci * elev_stride];
  if (!isfinite(center_z)) return;
  float sum[3] = {0.f, 0.f, 0.f};
  float sq[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float zs[TOPK > 0 ? 1 : kMaxRegion];
  float small[TOPK > 0 ? TOPK : 1], large[TOPK > 0 ? TOPK : 1];
  if (TOPK > 0) {
#pragma unroll
    for (int j = 0; j < TOPK; ++j) { small[j] = 3.402823466e+38f; large[j] = -3.402823466e+38f; }
  }
  int count = 0;
  for (int e = 0; e < F.n_entries; ++e) {
    const RegionEntry re = reg[e];
    if (!post_inside(p, lr + re.dr, lc + re.dc)) continue;
    const float nz = elev[post_index(p, lr + re.dr, lc + re.dc) * elev_stride];
    if (!isfinite(nz)) continue;
    const float d[3] = {float(-re.dr) * F.resf, float(-re.dc) * F.resf, nz - center_z};
#pragma unroll
    for (int k = 0; k < 3; ++k) sum[k] += d[k];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) sq[c * 3 + r] += d[r] * d[c];
    if (TOPK > 0) {
      float a = nz, b = nz;
#pragma unroll
      for (int j = 0; j < TOPK; ++j) {  // small[] ascending, large[] descending
        const float lo_j = small[j], hi_j = large[j];
        small[j] = a < lo_j ? a : lo_j;
        a = a < lo_j ? lo_j : a;
        large[j] = b > hi_j ? b : hi_j;
        b = b > hi_j ? hi_j : b;
      }
      ++count;
    } else {
      int k = count++;  // z_vals kept sorted (std::sort at feature_extraction.cpp:100)
      while (k > 0 && zs[k - 1] > nz) { zs[k] = zs[k - 1]; --k; }
      zs[k] = nz;
    }
  }
  if (count < F.min_valid) return;
  float cov[9], trace;
  if (!features_cov(sum, sq, count, cov, &trace)) return;
  float val[3], normal[3];
  eig3_direct(cov, val, normal);
  if (val[1] < 1e-8f) return;  // kMinEigenvalue
  const int lo = static_cast<int>(F.lo_pct * float(count - 1));
  const int hi = static_cast<int>(F.hi_pct * float(count - 1));
  float z_lo, z_hi;
  if (TOPK > 0) {
    const int from_top = count - 1 - hi;
    z_lo = small[0];
    z_hi = large[0];
#pragma unroll
    for (int j = 1; j < TOPK; ++j) {
      z_lo = lo == j ? small[j] : z_lo;
      z_hi = from_top == j ? large[j] : z_hi;
    }
  } else {
    z_lo = zs[lo];
    z_hi = zs[hi];
  }
  features_store(O, ci, val, normal, trace, z_lo, z_hi);
}

// discs of more than kMaxRegion cells, any percentile pair: the sorted heights in the global pool (see k_median_big)
inline __global__ __launch_bounds__(256) void k_features_big(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                      const RegionEntry* __restrict__ reg, const FeatureParams F,
                                                      const float* __restrict__ elev, int elev_stride,
                                                      const FeatureOut O, unsigned ncell, float* __restrict__ pool) {
  const unsigned tid = blockIdx.x * 256u + threadIdx.x, pitch = gridDim.x * 256u;
  const PoolList zs{pool + tid, pitch};
  const PostGeom p = post_geom(st, slot, G);
  for (unsigned t = tid; t < ncell; t += pitch) {
    const int lc = int(t / unsigned(p.rows)), lr = int(t - unsigned(lc) * unsigned(p.rows));
    const size_t ci = post_index(p, lr, lc);
    const float center_z = elev[ci * elev_stride];
    if (!isfinite(center_z)) continue;
    float sum[3] = {0.f, 0.f, 0.f};
    float sq[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int count = 0;
    for (int e = 0; e < F.n_entries; ++e) {
      const RegionEntry re = reg[e];
      if (!post_inside(p, lr + re.dr, lc + re.dc)) continue;
      const float nz = elev[post_index(p, lr + re.dr, lc + re.dc) * elev_stride];
      if (!isfinite(nz)) continue;
      const float d[3] = {float(-re.dr) * F.resf, float(-re.dc) * F.resf, nz - center_z};
#pragma unroll
      for (int k = 0; k < 3; ++k) sum[k] += d[k];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) sq[c * 3 + r] += d[r] * d[c];
      int k = count++;  // z_vals kept sorted (std::sort at feature_extraction.cpp:100)
      while (k > 0 && zs[k - 1] > nz) { zs[k] = zs[k - 1]; --k; }
      zs[k] = nz;
    }
    if (count < F.min_valid) continue;
    float cov[9], trace;
    if (!features_cov(sum, sq, count, cov, &trace)) continue;
    float val[3], normal[3];
    eig3_direct(cov, val, normal);
    if (val[1] < 1e-8f) continue;  // kMinEigenvalue
    const int lo = static_cast<int>(F.lo_pct * float(count - 1));
    const int hi = static_cast<int>(F.hi_pct * float(count - 1));
    features_store(O, ci, val, normal, trace, zs[lo], zs[hi]);
  }
}

// Any disc, any percentile pair, without a sort (round 5; replaces the pooled insertion sort of k_features_big wherever
// the tile fits the LDS): one block per 32 x 8 cells, the tile and its ring of `halo` cells staged in LDS once — as
// floats for the displacement sums (accumulated in the region's order, as the reference does) and as monotone keys
// for the two order statistics, which are found by bisection on the keys' bits: per bit ONE pass over the region
// counts the keys below both candidates.  The region's LDS offsets are a table in LDS (one broadcast read per entry).
__host__ __device__ constexpr unsigned features_sel_lds_bytes(int halo, int n_entries) {
  return unsigned(kS3R_ + 2 * halo) * unsigned(kS3C_ + 2 * halo) * 8u + unsigned(n_entries) * 4u;
}
inline __global__ __launch_bounds__(256) void k_features_sel(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                      const RegionEntry* __restrict__ reg, const FeatureParams F, int halo,
                                                      const float* __restrict__ elev, const FeatureOut O) {
  extern __shared__ uint32_t s_feat[];
  const PostGeom p = post_geom(st, slot, G);
  const int pitch = kS3R_ + 2 * halo, width = kS3C_ + 2 * halo, cells = pitch * width;
  float* const s_v = reinterpret_cast<float*>(s_feat);
  uint32_t* const s_k = s_feat + cells;
  int* const s_off = reinterpret_cast<int*>(s_feat + 2 * cells);
  const int tiles_r = (p.rows + kS3R_ - 1) / kS3R_;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  const int r0 = tr * kS3R_ - halo, c0 = tc * kS3C_ - halo;
  for (int k = int(threadIdx.x); k < cells; k += 256) {
    const int cc = k / pitch, rr = k - cc * pitch;
    const float v = post_inside(p, r0 + rr, c0 + cc) ? elev[post_index(p, r0 + rr, c0 + cc)] : __uint_as_float(0x7FC00000u);
    s_v[k] = v;
    s_k[k] = post_key(v);
  }
  for (int e = int(threadIdx.x); e < F.n_entries; e += 256) s_off[e] = reg[e].dc * pitch + reg[e].dr;
  __syncthreads();
  const int lrl = int(threadIdx.x) & (kS3R_ - 1), lcl = int(threadIdx.x) >> 5;
  const int lr = tr * kS3R_ + lrl, lc = tc * kS3C_ + lcl;
  if (!post_inside(p, lr, lc)) return;
  const int base = (lcl + halo) * pitch + lrl + halo;
  const float center_z = s_v[base];
  if (!isfinite(center_z)) return;
  float sum[3] = {0.f, 0.f, 0.f};
  float sq[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int count = 0;
  for (int e = 0; e < F.n_entries; ++e) {
    const RegionEntry re = reg[e];
    const float nz = s_v[base + re.dc * pitch + re.dr];
    if (!isfinite(nz)) continue;  // outside the stored window, or no data
    const float d[3] = {float(-re.dr) * F.resf, float(-re.dc) * F.resf, nz - center_z};
#pragma unroll
    for (int k = 0; k < 3; ++k) sum[k] += d[k];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r) sq[c * 3 + r] += d[r] * d[c];
    ++count;
  }
  if (count < F.min_valid) return;
  float cov[9], trace;
  if (!features_cov(sum, sq, count, cov, &trace)) return;
  float val[3], normal[3];
  eig3_direct(cov, val, normal);
  if (val[1] < 1e-8f) return;  // kMinEigenvalue
  const int lo = static_cast<int>(F.lo_pct * float(count - 1));
  const int hi = static_cast<int>(F.hi_pct * float(count - 1));
  uint32_t k_lo = 0u, k_hi = 0u;
  for (int b = 31; b >= 0; --b) {
    const uint32_t t_lo = k_lo | (1u << b), t_hi = k_hi | (1u << b);
    int c_lo = 0, c_hi = 0;
    for (int e = 0; e < F.n_entries; ++e) {
      const uint32_t key = s_k[base + s_off[e]];
      c_lo += key < t_lo ? 1 : 0;
      c_hi += key < t_hi ? 1 : 0;
    }
    if (c_lo <= lo) k_lo = t_lo;
    if (c_hi <= hi) k_hi = t_hi;
  }
  const size_t ci = post_index(p, lr, lc);
  features_store(O, ci, val, normal, trace, post_unkey(k_lo), post_unkey(k_hi));
}

// The same stage for dense layers (stride 1) and a region that reaches at most kFeatHaloMax cells: one block per
// 32 x 8 cells (rows are the contiguous axis of the storage), the tile and its ring staged in LDS once — cells
// outside the stored window as NaN, so "outside" and "no data" are one test.  The region table is pre-digested by the
// host (`FeatEntry`: LDS offset of the neighbour, the two horizontal displacements and their three products — the
// same float expressions the reference evaluates per neighbour, evaluated once): an entry is one scalar load, a
// neighbour one LDS read.  The order statistics are kept by v_min / v_max chains (TOPK smallest, TOPK largest;
// among zeros -0 orders before +0, where std::sort leaves the order of the two unspecified).  Per neighbour
// ~14 + 4*TOPK vector instructions instead of ~130 (index arithmetic, a dependent L1/L2 gather and compare-select
// chains) — the kernel is bound by instruction issue, not by its 46 MB of traffic.
// v_min_f32 / v_max_f32 on values known to be finite: the builtins put a canonicalising v_max x, x in front of
// every operand that was loaded rather than computed (IEEE mode), which doubles the chain.
__device__ __forceinline__ float vmin_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax_f32(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Inserting a into an ascending list s that keeps its K smallest: new s[j] = max(s[j-1], min(a, s[j])) = the MEDIAN of
// {s[j-1], a, s[j]} because s[j-1] <= s[j] — one v_med3_f32 per slot, every slot from the OLD list (no chain of
// dependent min / max pairs: round 6, half the instructions of the two-instruction step).
__device__ __forceinline__ float vmed3_f32(float a, float b, float c) {
  float r;
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
#ifndef FDM_FEAT_UNROLL
#define FDM_FEAT_UNROLL 4  // (8 entries in flight measured the same: 0.1228 against 0.1211 ms)
#endif
constexpr int kFeatTileR = 32, kFeatTileC = 16, kFeatHaloMax = 16, kFeatThreads = kFeatTileR * kFeatTileC;  // (32 x 16 cells: 2.4 cells staged per cell at the default radius, 3.4 with 32 x 8)

// 32 B: one s_load_dwordx8; {d0, d1}, {p00, p01}, {p11, 1.0f} are even-aligned scalar pairs — the second operand of one
// v_pk_add_f32 each (`one` counts the finite neighbours in the lane beside s11: exact far beyond any region's size)
struct FeatEntry { float d0, d1, p00, p01, p11, one; int off, pad; };
typedef float feat_v2f __attribute__((ext_vector_type(2)));

// keeps a table row's scalar loads where they are written (the compiler otherwise sinks them into the branch that
// uses them, one dependent scalar-cache round trip per neighbour)
__device__ __forceinline__ void pin_sgpr(const FeatEntry& f) {
  asm volatile("" ::"s"(f.off), "s"(f.d0), "s"(f.d1), "s"(f.p00), "s"(f.p01), "s"(f.p11), "s"(f.one));
}

template <int KLO, int KHI = KLO, bool MED3 = true>
__global__ __launch_bounds__(kFeatThreads) void k_features_tiled(const GeomConst G, const DevState* __restrict__ st, int slot,
                                                        const FeatEntry* __restrict__ tab, const FeatureParams F,
                                                        int halo, const float* __restrict__ elev, int elev_stride,
                                                        const FeatureOut O) {
  __shared__ float s_z[(kFeatTileR + 2 * kFeatHaloMax) * (kFeatTileC + 2 * kFeatHaloMax)];
  const PostGeom p = post_geom(st, slot, G);
  const int tiles_r = (p.rows + kFeatTileR - 1) / kFeatTileR;
  const int tr = int(blockIdx.x) % tiles_r, tc = int(blockIdx.x) / tiles_r;
  const int pitch = kFeatTileR + 2 * halo, width = kFeatTileC + 2 * halo;
  const int r0 = tr * kFeatTileR - halo, c0 = tc * kFeatTileC - halo;
  const float nanv = __uint_as_float(0x7FC00000u);
  for (int k = int(threadIdx.x); k < pitch * width; k += kFeatThreads) {
    const int cc = k / pitch, rr = k - cc * pitch;
    const int lr = r0 + rr, lc = c0 + cc;
    s_z[k] = post_inside(p, lr, lc) ? elev[post_index(p, lr, lc) * size_t(elev_stride)] : nanv;
  }
  __syncthreads();
  const int cell = compact_cells<kFeatThreads>([&](int t) {  // (round 6) cells with data take consecutive threads: full wavefronts
    const int rr = t & (kFeatTileR - 1), cc = t >> 5;
    return post_inside(p, tr * kFeatTileR + rr, tc * kFeatTileC + cc) && isfinite(s_z[(cc + halo) * pitch + rr + halo]);
  });
  if (cell < 0) return;
  const int lrl = cell & (kFeatTileR - 1), lcl = cell >> 5;
  const int lr = tr * kFeatTileR + lrl, lc = tc * kFeatTileC + lcl;
  const int base = (lcl + halo) * pitch + lrl + halo;
  const float center_z = s_z[base];
  const size_t ci = post_index(p, lr, lc);
  // nine running sums + the count as five register pairs: every pair one v_pk_add_f32 per neighbour
  feat_v2f a01 = {0.f, 0.f};   // sum[0], sum[1]
  feat_v2f b01 = {0.f, 0.f};   // s00, s01
  feat_v2f c11 = {0.f, 0.f};   // s11, count
  feat_v2f m02 = {0.f, 0.f};   // s02, s12
  feat_v2f z22 = {0.f, 0.f};   // sum[2], s22
  float small[KLO], large[KHI];
#pragma unroll
  for (int j = 0; j < KLO; ++j) small[j] = 3.402823466e+38f;
#pragma unroll
  for (int j = 0; j < KHI; ++j) large[j] = -3.402823466e+38f;
  auto visit = [&](const FeatEntry& fe, float nz) {
    if (!isfinite(nz)) return;
    const float d2 = nz - center_z;
    a01 += feat_v2f{fe.d0, fe.d1};
    b01 += feat_v2f{fe.p00, fe.p01};
    c11 += feat_v2f{fe.p11, fe.one};
    m02 += feat_v2f{fe.d0, fe.d1} * feat_v2f{d2, d2};
    z22 += feat_v2f{d2, d2 * d2};
    if (MED3) {  // small[] ascending, large[] descending; slot j from the old slots j - 1 and j
#pragma unroll
      for (int j = KLO - 1; j > 0; --j) small[j] = vmed3_f32(small[j - 1], nz, small[j]);
      small[0] = vmin_f32(nz, small[0]);
#pragma unroll
      for (int j = KHI - 1; j > 0; --j) large[j] = vmed3_f32(large[j - 1], nz, large[j]);
      large[0] = vmax_f32(nz, large[0]);
    } else {
      float a = nz, b = nz;
#pragma unroll
      for (int j = 0; j < KLO; ++j) {
        const float lo_j = small[j];
        small[j] = vmin_f32(a, lo_j);
        a = vmax_f32(a, lo_j);
      }
#pragma unroll
      for (int j = 0; j < KHI; ++j) {
        const float hi_j = large[j];
        large[j] = vmax_f32(b, hi_j);
        b = vmin_f32(b, hi_j);
      }
    }
  };
  int e = 0;
#if FDM_FEAT_UNROLL == 8
  for (; e + 8 <= F.n_entries; e += 8) {  // eight entries' scalar loads and LDS reads in flight together
    const FeatEntry f0 = tab[e], f1 = tab[e + 1], f2 = tab[e + 2], f3 = tab[e + 3];
    const FeatEntry f4 = tab[e + 4], f5 = tab[e + 5], f6 = tab[e + 6], f7 = tab[e + 7];
    pin_sgpr(f0); pin_sgpr(f1); pin_sgpr(f2); pin_sgpr(f3); pin_sgpr(f4); pin_sgpr(f5); pin_sgpr(f6); pin_sgpr(f7);
    const float z0 = s_z[base + f0.off], z1 = s_z[base + f1.off], z2 = s_z[base + f2.off], z3 = s_z[base + f3.off];
    const float z4 = s_z[base + f4.off], z5 = s_z[base + f5.off], z6 = s_z[base + f6.off], z7 = s_z[base + f7.off];
    visit(f0, z0); visit(f1, z1); visit(f2, z2); visit(f3, z3); visit(f4, z4); visit(f5, z5); visit(f6, z6); visit(f7, z7);
  }
#endif
  for (; e + 4 <= F.n_entries; e += 4) {  // four entries' scalar loads and LDS reads in flight together
    const FeatEntry f0 = tab[e], f1 = tab[e + 1], f2 = tab[e + 2], f3 = tab[e + 3];
    pin_sgpr(f0); pin_sgpr(f1); pin_sgpr(f2); pin_sgpr(f3);
    const float z0 = s_z[base + f0.off], z1 = s_z[base + f1.off], z2 = s_z[base + f2.off], z3 = s_z[base + f3.off];
    visit(f0, z0); visit(f1, z1); visit(f2, z2); visit(f3, z3);
  }
  for (; e < F.n_entries; ++e) {
    const FeatEntry fe = tab[e];
    visit(fe, s_z[base + fe.off]);
  }
  const int count = int(c11.y);
  if (count < F.min_valid) return;
  const float sum[3] = {a01.x, a01.y, z22.x};
  const float s00 = b01.x, s01 = b01.y, s11 = c11.x, s02 = m02.x, s12 = m02.y, s22 = z22.y;
  const float sq[9] = {s00, s01, s02, s01, s11, s12, s02, s12, s22};  // d[r] * d[c] is the same float either way round
  float cov[9], trace;
  if (!features_cov(sum, sq, count, cov, &trace)) return;
  float val[3], normal[3];
  eig3_direct(cov, val, normal);
  if (val[1] < 1e-8f) return;  // kMinEigenvalue
  const int lo = static_cast<int>(F.lo_pct * float(count - 1));
  const int from_top = count - 1 - static_cast<int>(F.hi_pct * float(count - 1));
  float z_lo = small[0], z_hi = large[0];
#pragma unroll
  for (int j = 1; j < KLO; ++j) z_lo = lo == j ? small[j] : z_lo;
#pragma unroll
  for (int j = 1; j < KHI; ++j) z_hi = from_top == j ? large[j] : z_hi;
  features_store(O, ci, val, normal, trace, z_lo, z_hi);
}

}  // namespace fdm
