// fdm_raycast.hpp — ghost-obstacle removal stage of FastDEM::integrateImpl on the device
// (SURVEY.md §8 row f1).  gfx950 only.
//
// Reference being reproduced (file:line under /root/reference/fastdem):
//   src/fastdem.cpp:152-159                               sensor origin, voxelGrid(ANY), applyRaycasting
//   lib/nanoPCL/include/nanopcl/core/voxel.hpp:28-43      voxel key [z:21][y:21][x:21]
//   lib/nanoPCL/include/nanopcl/filters/impl/voxel_grid_impl.hpp:30-60,171-189   VoxelMode::ANY
//   src/raycasting.cpp:46-140 traceRay, :142-173 processScan, :175-202 resolveGhostCells, :204-249
//   include/fastdem/elevation_map.hpp:131-135             clearAt
//
// The reference walks the scan sequentially; every step it takes is order-free once restated per
// cell, which is what the kernels below exploit:
//   * observed evidence: logodds = min(logodds + L_obs, L_max) once per ray-scan point in the cell
//     -> the same constant folded k times; k is counted with an integer atomicAdd;
//   * min ray height: min over rays of the height at the cell exit -> atomicMin on the monotone
//     uint encoding of the float (a plain L2 load filters out the rays that cannot lower it);
//   * ghost resolution: one independent decision per traversed cell, after both of the above.
// Pipeline (one stream, no host round trip):
//   k_voxel_keys -> stable radix sort of (key, point index) -> k_voxel_mark   (small scans: k_vs_count -> k_vs_scatter
//   -> k_vs_mark, no sort) -> k_ray_compact
//   (-> k_ray_bin_sum -> k_ray_bin_scan -> k_ray_scatter for large scans) -> k_ray -> k_ray_resolve
// VoxelMode::ANY picks idx[start + (count*7 + start*13) % count] of each voxel's run in the sorted
// array.  The reference sorts with std::sort on the key only (unstable: the order inside a voxel is
// whatever that libstdc++'s introsort leaves); the engine sorts stably, i.e. ties in original point
// order (both are valid outcomes of "ANY"; see DESIGN.md, raycasting section).
#pragma once

#include "fdm_device.hpp"

namespace fdm {

constexpr uint32_t kRayEmpty = 0xFFFFFFFFu;       // ord() of no float that can occur (NaN pattern)
constexpr int kRayBatchLarge = 32;
constexpr int kRayBatch = 8;                      // cells a ray walks between two rounds of loads
constexpr uint64_t kInvalidVoxel = ~0ull;         // voxel::INVALID_KEY: sorts behind every real key

struct RayParams {
  float ox, oy, oz;     // sensor origin, map frame (scalars: see the DevObst note in fdm_device.hpp)
  float l_obs, l_ghost, l_max, clear_thr, conflict_thr;
  float inv_voxel;      // 1.0f / voxel_size
  float resolution;     // float(map.getResolution())
  unsigned n;           // points (VOXEL: sorted entries)
  int slot;             // geometry ring slot holding the map geometry the stage runs on
  int flag_slot;        // >= 0: DevFlags slot whose ray_any gates the stage (integrate); -1: ungated
  unsigned vis_stamp;   // DevState::vis_ray value if this is the first frame that runs
  int dbg;              // measurement only: 1 = no atomics in k_ray, 2 = no loads either
  int by_sector;        // 1: queue ordered (sector of equal true angle, coarse length class) for k_ray_wedge (fdm_raywedge.hpp)
  // Two stages in flight (option "ray_overlap"): everything of a stage but k_ray_resolve may run BEFORE the scan's update,
  // beside the next scans' launches.  The geometry it runs on is then derived as the update will commit it (make_ctx):
  // geom[pre_slot], moved to cand[pre_slot] iff pre_do_move and (not pre_gate or flags[pre_slot].any_pass).
  int ctx;              // which of the two queue counters (DevState::ray_count / ray_count_b) and buffer sets
  int pre_slot;         // >= 0: derive the geometry from this ring slot (the scan's own); -1: geom[slot] is there
  int pre_do_move, pre_gate;
};
__device__ __forceinline__ DevGeom ray_geom(const DevState* __restrict__ st, const RayParams& Q) {
  if (Q.pre_slot < 0) return st->geom[Q.slot];
  DevGeom g = st->geom[Q.pre_slot];
  if (Q.pre_do_move && (!Q.pre_gate || st->flags[Q.pre_slot].any_pass != 0u)) {
    const DevCand c = st->cand[Q.pre_slot];
    g.px = c.px; g.py = c.py; g.sr = c.sr; g.sc = c.sc;
  }
  return g;
}
__device__ __forceinline__ unsigned* ray_counter(DevState* __restrict__ st, const RayParams& Q) {
  return Q.ctx ? &st->ray_count_b : &st->ray_count;
}

// voxel::pack.  float -> int32 outside the int range is UB in C++; the reference's x86 build
// (cvttss2si) produces INT_MIN for both signs, restated here explicitly.
__device__ __forceinline__ int32_t cvt_x86(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT32_MIN;
  return static_cast<int32_t>(v);
}
__device__ __forceinline__ uint64_t voxel_pack(float x, float y, float z, float inv) {
  constexpr int32_t kOff = 1 << 20, kMin = -kOff, kMax = kOff - 1;
  int32_t ix = cvt_x86(floorf(x * inv)), iy = cvt_x86(floorf(y * inv)), iz = cvt_x86(floorf(z * inv));
  ix = ix < kMin ? kMin : (ix > kMax ? kMax : ix);
  iy = iy < kMin ? kMin : (iy > kMax ? kMax : iy);
  iz = iz < kMin ? kMin : (iz > kMax ? kMax : iz);
  return (uint64_t(iz + kOff) << 42) | (uint64_t(iy + kOff) << 21) | uint64_t(ix + kOff);
}

// Order-preserving compact key for scans whose extent the host can bound (cropRange with a finite
// range_max keeps every point within range_max of the base origin): the three voxel indices are
// rebased to the corner of that box and packed into 3*bits <= 31 bits of a uint32, [z][y][x] like the
// reference key, so the sorted order, the runs and therefore `start` / `count` are the ones of the
// 63-bit key — at half the key traffic and 28 instead of 64 sorted bits.  Larger boxes (3*bits <= 62)
// use the same packing in a uint64 and still sort only 3*bits + 1 bits.
struct VoxelCompact {
  int x0, y0, z0;  // voxel index of the box corner
  int bits;        // per horizontal axis; 0 = use the full 63-bit key
  int zbits;       // vertical axis: the z crop usually bounds it far tighter than the range ball does
};
constexpr uint32_t kInvalidVoxel32 = 0xFFFFFFFFu;
template <typename KEY>
__device__ __forceinline__ KEY voxel_pack_compact(float x, float y, float z, float inv, const VoxelCompact& C) {
  const int hi = (1 << C.bits) - 1, hiz = (1 << C.zbits) - 1;
  int fx = cvt_x86(floorf(x * inv)) - C.x0, fy = cvt_x86(floorf(y * inv)) - C.y0, fz = cvt_x86(floorf(z * inv)) - C.z0;
  fx = fx < 0 ? 0 : (fx > hi ? hi : fx);  // cannot trigger inside the box the host derived (2-cell margin)
  fy = fy < 0 ? 0 : (fy > hi ? hi : fy);
  fz = fz < 0 ? 0 : (fz > hiz ? hiz : fz);
  return (KEY(fz) << (2 * C.bits)) | (KEY(fy) << C.bits) | KEY(fx);
}
template <typename KEY>
struct VoxelKeyTraits;
template <>
struct VoxelKeyTraits<unsigned long long> {
  static constexpr unsigned long long invalid = ~0ull;
};
template <>
struct VoxelKeyTraits<uint32_t> {
  static constexpr uint32_t invalid = 0xFFFFFFFFu;
};

// (key, index) per point; non-finite points (and points the crops dropped, stored with x = NaN)
// get the invalid key and therefore sort to the tail (voxel_grid_impl.hpp:50-55).
// One block per sort tile (TILE = fdm_rsort.hpp's rs_tile(n) points): besides the keys it leaves the
// FIRST pass's digit histogram of its tile (hist[bin][tile], as k_rs_hist would: one launch and one pass over the keys
// less), and the point indices are not written at all — the first scatter pass takes "position" for them.
template <typename KEY, unsigned TILE>
__global__ __launch_bounds__(256) void k_voxel_keys(unsigned n, float inv_voxel, int flag_slot,
                                                    const VoxelCompact C, DevState* __restrict__ st,
                                                    const float* __restrict__ x,
                                                    const float* __restrict__ y,
                                                    const float* __restrict__ z,
                                                    KEY* __restrict__ keys,
                                                    uint32_t* __restrict__ sel,
                                                    unsigned ntiles, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  bool any = false;
#pragma unroll 4
  for (unsigned r = 0; r < TILE / 256u; ++r) {
    const unsigned i = blockIdx.x * TILE + r * 256u + threadIdx.x;
    if (i < n) {
      const float a = x[i], b = y[i], c = z[i];
      const bool valid = isfinite(a) && isfinite(b) && isfinite(c);
      KEY k = VoxelKeyTraits<KEY>::invalid;
      if (valid) k = C.bits > 0 ? voxel_pack_compact<KEY>(a, b, c, inv_voxel, C) : KEY(voxel_pack(a, b, c, inv_voxel));
      keys[i] = k;
      sel[i] = 0u;  // k_voxel_mark sets the representatives
      atomicAdd(&h[unsigned(k) & 255u], 1u);
      any = any || valid;
    }
  }
  if (flag_slot >= 0 && __ballot(any) && (threadIdx.x & 63) == 0) st->flags[flag_slot].ray_any = 1u;
  __syncthreads();
  hist[size_t(threadIdx.x) * ntiles + blockIdx.x] = h[threadIdx.x];
}

__device__ __forceinline__ bool map_contains(double x, double y, const DevGeom& g, const GeomConst& G) {
  const double tx = -((x - g.px) - G.half_x), ty = -((y - g.py) - G.half_y);  // GridMap::isInside
  return tx >= 0.0 && tx < G.len_x && ty >= 0.0 && ty < G.len_y;
}

// stage preconditions (raycasting.cpp:207-220), identical in every thread of every stage kernel
__device__ __forceinline__ bool ray_stage_runs(const RayParams& Q, const DevState* __restrict__ st,
                                               const DevGeom& g, const GeomConst& G) {
  if (Q.flag_slot >= 0 && st->flags[Q.flag_slot].ray_any == 0u) return false;  // scan.empty()
  return map_contains(double(Q.ox), double(Q.oy), g, G);
}

// storage-linear cell of buffer index (mr, mc) if this engine owns it, else -1
__device__ __forceinline__ int owned_storage(int mr, int mc, const GeomConst& G) {
  const int lr = mr - G.o_r0, lc = mc - G.o_c0;
  if (lr < 0 || lc < 0 || lr >= G.o_rows || lc >= G.o_cols) return -1;
  return (mc - G.s_c0) * G.s_rows + (mr - G.s_r0);
}

// The voxel filter's output as a flag per ORIGINAL point index: sel[p] = 1 iff point p represents
// its voxel.  processScan is order-free, so the rays are then traced in the scan's own order:
// consecutive points of a LiDAR firing sequence / an image row share their 2-D direction (nearly),
// so the lanes of a wavefront walk the same cells at the same step and their loads coalesce into
// a few L2 requests instead of 64.
// Run lengths without a serial walk: a block looks at the boundary bits (key differs from its
// predecessor) of its own 256 sorted positions and of the 256 that follow, taken with two ballots
// per wave; a run head finds its end with a find-first-set over those 512 bits (2 * THREADS).  All loads are
// independent, so the kernel is two memory round trips (keys, then the picked index) instead of up
// to 16 dependent ones (7.7 -> ~3 us on a VLP-16 scan).  Only a run reaching more than 256 positions
// past its block bisects for its end.
template <typename KEY, unsigned THREADS = 256>
__device__ __forceinline__ uint32_t voxel_pick_block(const KEY* __restrict__ keys,
                                                     const uint32_t* __restrict__ idx, unsigned n,
                                                     bool& head, const unsigned base = blockIdx.x * THREADS) {
  constexpr unsigned kWords = THREADS / 64u;  // boundary words per half of the 2 * THREADS window
  __shared__ unsigned long long s_b[2 * kWords];
  const unsigned t = threadIdx.x;
  const unsigned i0 = base + t, i1 = i0 + THREADS;
  constexpr KEY inv = VoxelKeyTraits<KEY>::invalid;
  const KEY k0 = i0 < n ? keys[i0] : inv;
  const KEY p0 = (i0 > 0 && i0 < n) ? keys[i0 - 1] : inv;
  const KEY k1 = i1 < n ? keys[i1] : inv;
  const KEY p1 = i1 < n ? keys[i1 - 1] : inv;
  const bool b0 = i0 >= n || i0 == 0 || p0 != k0;
  const bool b1 = i1 >= n || p1 != k1;
  const unsigned long long m0 = __ballot(b0), m1 = __ballot(b1);
  if ((t & 63u) == 0u) { s_b[t >> 6] = m0; s_b[kWords + (t >> 6)] = m1; }
  __syncthreads();
  head = i0 < n && b0 && k0 != inv;
  if (!head) return kNoIdx;
  unsigned word = (t + 1u) >> 6;
  unsigned long long m = s_b[word] & (~0ull << ((t + 1u) & 63u));  // word <= kWords < 2 * kWords
  while (!m && ++word < 2u * kWords) m = s_b[word];
  unsigned end;  // first sorted position after the run
  if (m) {
    end = base + word * 64u + unsigned(__ffsll((long long)m) - 1);
  } else {       // positions i0+1 .. base + 2 * THREADS - 1 all continue the run
    unsigned lo = base + 2u * THREADS - 1u, hi = n;
    while (hi - lo > 1u) {
      const unsigned mid = lo + (hi - lo) / 2u;
      if (keys[mid] == k0) lo = mid; else hi = mid;
    }
    end = hi;
  }
  const unsigned long long c = end - i0, s0 = i0;  // size_t arithmetic in the reference
  return idx[i0 + unsigned((c * 7ull + s0 * 13ull) % c)];
}

template <typename KEY>
__global__ __launch_bounds__(256) void k_voxel_mark(unsigned n, const KEY* __restrict__ keys,
                                                    const uint32_t* __restrict__ idx,
                                                    uint32_t* __restrict__ sel) {
  bool head;
  const uint32_t pick = voxel_pick_block(keys, idx, n, head);
  if (head) sel[pick] = 1u;
}

// ---------------------------------------------------------------------------------------------
// VoxelMode::ANY without a sort, for small scans (a VLP-16 sweep: the library sort was 8 launches, 45 of the stage's
// 70 us).  What the filter needs of a point is (a) its voxel's run in the sorted array — `start` = how many valid
// points have a smaller key, `count` — and (b) its own position in that run under the stable order (= how many points
// of the same voxel have a smaller index).  Bucketing by a PREFIX of the key gives all three without ordering anything
// (bucket order = key order).  The key is [z][y][x] and a scan's points sit in a few z levels, so the buckets have to
// be fine — one (z, y) row of voxels, up to 2^18 of them — and their first positions come from two levels: 32 fine
// buckets to a coarse one, <= 8192 coarse counters scanned by every block, <= 31 fine counters (one cache line) summed
// per point.
//   k_vs_count    key per point (k_voxel_keys' arithmetic); the returning atomicAdd on the fine counter is the
//                 point's (arbitrary, unique) place inside its bucket; one more add on the coarse counter
//   k_vs_scatter  {key, index, bucket start, bucket size} of every valid point at start + place
//   k_vs_mark     thread per placed point: one walk over ITS bucket's members counts the smaller keys, the equal
//                 keys and the equal keys with a smaller index; the point is its voxel's representative iff the last
//                 count equals (count * 7 + start * 13) % count.  sel[] as k_voxel_mark leaves it.  Then it puts its
//                 bucket's two counters back to zero (nobody reads them any more): no memset launches.
constexpr unsigned kVsFineBits = 18u, kVsCoarse = 1u << (kVsFineBits - 5u);
struct VoxelSmall {
  unsigned shift;          // fine bucket = key >> shift; coarse bucket = fine >> 5
  uint32_t* fine;          // [2^kVsFineBits] zero between scans
  uint32_t* coarse;        // [kVsCoarse]     zero between scans
  uint32_t* total;         // [1] valid points of this scan
  uint32_t* place;         // [n] by point
  uint4* rec;              // [cap] by position: key | point | bucket start | bucket size
  unsigned cap;            // entries of rec (reads beyond the scan's positions stay inside the allocation)
  unsigned ibits;          // bits of a point index
  int dbg;                 // measurement only (wrong results): 1 = no fine atomics, 2 = no coarse atomics, 3 = neither
};

// (bodies take the block number and the word that reports "the scan has a valid point" as arguments: the batch kernels
// of fdm_rbatch.hpp run the same code for scan blockIdx.y of a batch)
__device__ __forceinline__ void vs_count_body(unsigned n, float inv_voxel, const VoxelCompact& C, const VoxelSmall& V,
                                              const float* __restrict__ x, const float* __restrict__ y,
                                              const float* __restrict__ z, uint32_t* __restrict__ keys,
                                              uint32_t* __restrict__ sel, unsigned* __restrict__ any_word,
                                              unsigned any_val, unsigned blk) {
  const unsigned i = blk * 256u + threadIdx.x;
  const unsigned lane = threadIdx.x & 63u;
  bool valid = false;
  uint32_t k = kInvalidVoxel32;
  if (i < n) {
    const float a = x[i], b = y[i], c = z[i];
    valid = isfinite(a) && isfinite(b) && isfinite(c);
    if (valid) k = voxel_pack_compact<uint32_t>(a, b, c, inv_voxel, C);
    keys[i] = k;
    sel[i] = 0u;
  }
  // Same-address atomics serialise at the memory side (a ring that runs along a row of voxels puts a hundred
  // consecutive points into one counter; the coarse counters of the ground's z levels collect thousands: 45 us for a
  // VLP-16 sweep with one atomic per point).  One add per distinct counter and wavefront, fine and coarse.
  const unsigned f = valid ? k >> V.shift : 0xFFFFFFFFu;
  {  // one returning add per distinct fine bucket of the wavefront, all of them in flight together; the bucket's lanes
     // take consecutive places behind their leader's base
    unsigned long long todo = __ballot(valid);
    int my_lead = int(lane);
    unsigned my_rank = 0u, my_size = 0u;
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const unsigned f0 = unsigned(__builtin_amdgcn_readlane(int(f), lead));
      const unsigned long long same = __ballot(valid && f == f0);
      if (f == f0) {
        my_lead = lead;
        my_rank = unsigned(__popcll(same & ((1ull << lane) - 1ull)));
        my_size = unsigned(__popcll(same));
      }
      todo &= ~same;
    }
    unsigned base = 0u;
    if (valid && my_lead == int(lane) && !(V.dbg & 1)) base = atomicAdd(&V.fine[f], my_size);
    base = unsigned(__shfl(int(base), my_lead));
    if (valid) V.place[i] = base + my_rank;
  }
  {  // coarse counters: per wavefront one add per distinct counter into a small LDS table of the block (direct-mapped;
     // a collision goes to memory at once), per block one add per occupied entry — the ground's z levels put more than
     // a thousand points of a sweep into ONE counter, and same-address atomics serialise at the memory side (5 of this
     // kernel's 11 us with one add per wavefront and counter)
    __shared__ uint32_t s_id[256], s_n[256];
    s_id[threadIdx.x] = 0xFFFFFFFFu;
    s_n[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned g = f >> 5;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int lead = __ffsll((long long)todo) - 1;
      const unsigned g0 = unsigned(__builtin_amdgcn_readlane(int(g), lead));
      const unsigned long long same = __ballot(valid && g == g0);
      if (int(lane) == lead && !(V.dbg & 2)) {
        const unsigned slot = g0 & 255u, cnt = unsigned(__popcll(same));
        const uint32_t prev = atomicCAS(&s_id[slot], 0xFFFFFFFFu, g0);
        if (prev == 0xFFFFFFFFu || prev == g0) atomicAdd(&s_n[slot], cnt);
        else atomicAdd(&V.coarse[g0], cnt);
      }
      todo &= ~same;
    }
    __syncthreads();
    if (s_n[threadIdx.x]) atomicAdd(&V.coarse[s_id[threadIdx.x]], s_n[threadIdx.x]);
  }
  if (any_word && __ballot(valid) && (threadIdx.x & 63) == 0) *any_word = any_val;
}
inline __global__ __launch_bounds__(256) void k_vs_count(unsigned n, float inv_voxel, int flag_slot, const VoxelCompact C,
                                                  const VoxelSmall V, DevState* __restrict__ st,
                                                  const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ z, uint32_t* __restrict__ keys,
                                                  uint32_t* __restrict__ sel) {
  vs_count_body(n, inv_voxel, C, V, x, y, z, keys, sel, flag_slot >= 0 ? &st->flags[flag_slot].ray_any : nullptr, 1u,
                blockIdx.x);
}

__device__ __forceinline__ void vs_scatter_body(unsigned n, const VoxelSmall& V, const uint32_t* __restrict__ keys,
                                                unsigned blk) {
  __shared__ uint32_t s_start[kVsCoarse];
  __shared__ uint32_t s_wave[4];
  const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  constexpr unsigned per = kVsCoarse / 256u;  // coarse counters per thread, consecutive
  uint32_t c_[per];
  uint32_t mine = 0u;
#pragma unroll
  for (unsigned j = 0; j < per; ++j) { c_[j] = V.coarse[t * per + j]; mine += c_[j]; }
  uint32_t inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d);
    if (int(lane) >= d) inc += o;
  }
  if (lane == 63u) s_wave[wave] = inc;
  __syncthreads();
  uint32_t run = inc - mine;
  for (unsigned w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
  for (unsigned j = 0; j < per; ++j) { s_start[t * per + j] = run; run += c_[j]; }
  if (blk == 0 && t == 255u) *V.total = run;
  __syncthreads();
  const unsigned i = blk * 256u + t;
  if (i >= n) return;
  const uint32_t k = keys[i];
  if (k == kInvalidVoxel32) return;  // (dropped / non-finite points sort behind every voxel: never a representative)
  const unsigned f = k >> V.shift, f0 = f & ~31u, fl = f & 31u;
  uint32_t s = s_start[f >> 5], m = 0u;
  const uint4* const grp = reinterpret_cast<const uint4*>(V.fine + f0);  // the group's 32 counters: one 128-byte line
  uint4 c4[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) c4[q] = grp[q];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const uint32_t w[4] = {c4[q].x, c4[q].y, c4[q].z, c4[q].w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned j = unsigned(q * 4 + r);
      s += j < fl ? w[r] : 0u;
      m = j == fl ? w[r] : m;
    }
  }
  V.rec[s + V.place[i]] = make_uint4(k, i, s, m);
}
inline __global__ __launch_bounds__(256) void k_vs_scatter(unsigned n, const VoxelSmall V, const uint32_t* __restrict__ keys) {
  vs_scatter_body(n, V, keys, blockIdx.x);
}

__device__ __forceinline__ void vs_mark_body(const VoxelSmall& V, uint32_t* __restrict__ sel, unsigned blk) {
  const unsigned p = blk * 256u + threadIdx.x, lane = threadIdx.x & 63u, w0 = p - lane;
  const unsigned total = *V.total;
  // a bucket's members are neighbours: the wavefront's own 64 positions and the 128 on either side are compared in
  // registers (five independent loads, one round trip); a per-lane walk over memory is one dependent round trip per
  // step (a single 148-point row of a VLP-16 sweep: 20 us) and is left to what reaches beyond that window
  const uint4 dead = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
  auto at = [&](unsigned q) { return V.rec[min(q, V.cap - 1u)]; };  // (q - 128 wraps to a huge value: clamped)
  uint4 me = at(p);
  uint4 nb[4] = {at(p - 128u), at(p - 64u), at(p + 64u), at(p + 128u)};
  const bool live = p < total;
  if (!live) me = dead;
  if (p < 128u) nb[0] = dead;
  if (p < 64u) nb[1] = dead;
  if (p + 64u >= total) nb[2] = dead;
  if (p + 128u >= total) nb[3] = dead;
  const uint32_t k = me.x, i = me.y;
  unsigned lower = 0u, equal = 0u, before = 0u;
  // One window of 64 positions against this lane's point, bit-serially: the bucket's members are the lanes
  // [from, to) of the window (positions are contiguous); walking the key's low bits from the top, `pre` keeps the
  // members equal to the lane's key so far and `less` collects those that turn smaller; the same walk over the
  // point index among the equal keys.  ~25 ballots instead of 64 x three lane reads.
  const unsigned kbits = V.shift;  // keys of one bucket differ in their low `shift` bits only
  auto window = [&](const uint4& w, unsigned wpos) {
    const unsigned a = me.z > wpos ? me.z - wpos : 0u;
    const unsigned b = live ? (me.z + me.w > wpos ? min(me.z + me.w - wpos, 64u) : 0u) : 0u;
    unsigned long long pre = (a < b) ? (((b >= 64u) ? ~0ull : ((1ull << b) - 1ull)) & ~((1ull << a) - 1ull)) : 0ull;
    unsigned long long less = 0ull;
    for (int bit = int(kbits) - 1; bit >= 0; --bit) {  // (uniform trip count)
      const unsigned long long ones = __ballot((w.x >> bit) & 1u);
      const bool mine = ((k >> bit) & 1u) != 0u;
      less |= mine ? (pre & ~ones) : 0ull;
      pre &= mine ? ones : ~ones;
    }
    lower += unsigned(__popcll(less));
    equal += unsigned(__popcll(pre));
    unsigned long long lessi = 0ull;
    for (int bit = int(V.ibits) - 1; bit >= 0; --bit) {
      const unsigned long long ones = __ballot((w.y >> bit) & 1u);
      const bool mine = ((i >> bit) & 1u) != 0u;
      lessi |= mine ? (pre & ~ones) : 0ull;
      pre &= mine ? ones : ~ones;
    }
    before += unsigned(__popcll(lessi));
  };
  const unsigned end = live ? me.z + me.w : 0u;
  window(me, w0);
  if (__ballot(live && me.z + 64u < w0)) window(nb[0], w0 - 128u);  // (wave-uniform conditions)
  if (__ballot(live && me.z < w0)) window(nb[1], w0 - 64u);
  if (__ballot(live && end > w0 + 64u)) window(nb[2], w0 + 64u);
  if (__ballot(live && end > w0 + 128u)) window(nb[3], w0 + 128u);
  if (!live) return;
  // buckets of more than ~130 points: what lies beyond the window, eight independent loads per step
  const unsigned wlo = w0 >= 128u ? w0 - 128u : 0u, whi = w0 + 192u;
  auto beyond = [&](unsigned q0, unsigned q1) {
    for (unsigned q = q0; q < q1; q += 8u) {
      uint2 o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = *reinterpret_cast<const uint2*>(V.rec + min(q + unsigned(r), q1 - 1u));
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bool on = q + unsigned(r) < q1;
        lower += (on && o[r].x < k) ? 1u : 0u;
        equal += (on && o[r].x == k) ? 1u : 0u;
        before += (on && o[r].x == k && o[r].y < i) ? 1u : 0u;
      }
    }
  };
  if (me.z < wlo) beyond(me.z, min(end, wlo));
  if (end > whi) beyond(max(me.z, whi), end);
  // (count * 7 + start * 13) % count, size_t arithmetic in the reference (voxel_grid_impl.hpp:171-173); with at most
  // 2^20 points neither product leaves 32 bits
  const unsigned s0 = me.z + lower;
  if (before == (equal * 7u + s0 * 13u) % equal) sel[i] = 1u;
  if (p == me.z) {  // the bucket's first position puts its counters back to zero
    const unsigned f = k >> V.shift;
    V.fine[f] = 0u;
    V.coarse[f >> 5] = 0u;
  }
}
inline __global__ __launch_bounds__(256) void k_vs_mark(const VoxelSmall V, uint32_t* __restrict__ sel) {
  vs_mark_body(V, sel, blockIdx.x);
}

// processScan (raycasting.cpp:142-173), first half: one ray-scan point per thread.
//   VOXEL = true : point i counts if the voxel filter kept it (sel[i] != 0)
//   VOXEL = false: every finite point of the caller's cloud (applyRaycasting called directly)
// Observed evidence is counted here; the downward rays are queued DENSELY for k_ray (order inside a
// block = scan order, blocks land in whatever order their atomicAdd does — the result is order-free),
// so k_ray's wavefronts are full and neighbouring lanes hold neighbouring rays of the scan.
// PTS points per thread: the queue tail is ONE same-address returning atomic per block, and those
// serialise in L2 (8192 blocks of 256 points took 100 us at C4 for this reason alone).
//
// Large scans (bin_cnt != nullptr) bucket the queue by (direction wedge of 0.18 deg, length class) before
// the walk.  k_ray's loop runs until the longest ray of a wavefront ends; 64 beams of one firing step reach
// from 1.5 m to the far wall (lanes ~40 % busy), 64 rays of one wedge and similar length end together and
// still stand in the same few cells at every step.  The direction is a diamond angle (monotone in the
// azimuth, no atan2), the length the Manhattan cell count in classes of 16 cells.  The order INSIDE a bucket
// is irrelevant (every step of the stage is order-free), so this is a counting sort without a sort: the
// returning atomic that counts the bucket is the ray's rank in it; k_ray_bin_sum / k_ray_bin_scan turn
// the counts into offsets and k_ray_scatter places the rays.
constexpr unsigned kRayWedges = 2048u, kRayLenClasses = 128u, kRayBins = kRayWedges * kRayLenClasses;
// Length classes (round 3): finer for the short rays (a third of a 2 M-point sweep ends within 5 m), and the queue is
// ordered (length group, wedge, class inside the group) with TWO groups — rays below / beyond 224 cells — instead of
// (wedge, class): the wavefronts of the short half hold rays of neighbouring wedges whose lengths differ by a few cells,
// they end together.  configs[3] stage by number of groups: 1 (wedge-major) 0.832 ms, 2: 0.774, 4: 0.790, 8: 0.910,
// 16: 1.20; the 272 K-point RGB-D scan: 0.150 / 0.150 / 0.167 / 0.170 / 0.174.  Also measured: ONE k_ray LAUNCH PER
// LENGTH GROUP (four), shortest first, so that the long rays find the near cells already lowered and their visits are
// settled by the read (each (wavefront, cell) pair is otherwise one lowering event: 4.1 M events for 1.0 M cells) —
// 0.92 ms: every launch pays the whole latency chain of a walk with a fraction of the wavefronts.  Removed.
constexpr unsigned kRayGroups = 2u, kRayGroupClasses = kRayLenClasses / kRayGroups;  // 64 classes per group: rays below / beyond 224 cells
__device__ __forceinline__ unsigned ray_len_key(unsigned cells) {  // group * 32 + class inside the group
  // Manhattan length in cells: [0, 96) in classes of 3, [96, 224) of 4, [224, 480) of 8, beyond in classes of 16
  if (cells < 96u) return cells / 3u;
  if (cells < 224u) return 32u + (cells - 96u) / 4u;
  if (cells < 480u) return 64u + (cells - 224u) / 8u;
  return 96u + min(31u, (cells - 480u) / 16u);
}
constexpr unsigned kRayBinBlock = 1024u;  // bins per block of the two scan kernels
// the order k_ray_wedge (fdm_raywedge.hpp) wants: 256 sectors of EQUAL true angle (a spinning LiDAR puts the same number
// of rays into each — diamond-angle sectors are twice as wide on the diagonals as on the axes) x 32 length classes
constexpr unsigned kRaySectors = 256u, kRaySectorClasses = 32u;
__device__ __forceinline__ unsigned ray_sector(float dx, float dy) {
  const float turn = (atan2f(dy, dx) + 3.14159265f) * (1.0f / 6.28318531f);  // [0, 1]
  return min(kRaySectors - 1u, unsigned(fmaxf(turn, 0.0f) * float(kRaySectors)));
}

template <bool VOXEL, int PTS>
__device__ __forceinline__ void ray_compact_body(const RayParams& Q, const GeomConst& G, const DevGeom& g,
                                                 const float* __restrict__ x,
                                                 const float* __restrict__ y,
                                                 const float* __restrict__ z,
                                                 const uint32_t* __restrict__ sel,
                                                 uint32_t* __restrict__ rc_cnt,
                                                 uint32_t* __restrict__ ray_list,
                                                 uint32_t* __restrict__ ray_key,
                                                 uint32_t* __restrict__ ray_rank,
                                                 uint32_t* __restrict__ bin_cnt,
                                                 unsigned* __restrict__ ray_count, uint32_t* __restrict__ blk_cnt,
                                                 const unsigned blk) {
  __shared__ unsigned s_wave[PTS][4];
  __shared__ unsigned s_base;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned idx[PTS], key[PTS], rank[PTS];
  unsigned long long mask[PTS];
  // the block's loads in two round trips (selection flags, then the selected points), whatever PTS is
  bool take[PTS];
  float ex_[PTS], ey_[PTS], ez_[PTS];
#pragma unroll
  for (int k = 0; k < PTS; ++k) {
    const unsigned i = (blk * unsigned(PTS) + unsigned(k)) * 256u + threadIdx.x;
    take[k] = i < Q.n && (!VOXEL || sel[i] != 0u);
  }
#pragma unroll
  for (int k = 0; k < PTS; ++k) {
    const unsigned i = (blk * unsigned(PTS) + unsigned(k)) * 256u + threadIdx.x;
    ex_[k] = ey_[k] = ez_[k] = 0.0f;
    if (take[k]) { ex_[k] = x[i]; ey_[k] = y[i]; ez_[k] = z[i]; }
  }
#pragma unroll
  for (int k = 0; k < PTS; ++k) {
    const unsigned i = (blk * unsigned(PTS) + unsigned(k)) * 256u + threadIdx.x;
    idx[k] = i;
    key[k] = 0u;
    rank[k] = 0u;
    bool ray = false;
    int o = -1;
    if (take[k]) {
      const float ex = ex_[k], ey = ey_[k], ez = ez_[k];
      if (VOXEL || (isfinite(ex) && isfinite(ey) && isfinite(ez))) {
        DevCand c;  // observed evidence: the point's own cell (nanoGrid getIndex, fp64)
        c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
        o = owned_cell(ex, ey, c, G);
        ray = ez < Q.oz;  // upward rays are skipped (raycasting.cpp:168)
        if (ray && bin_cnt) {
          const float dx = ex - Q.ox, dy = ey - Q.oy;
          const float ax = fabsf(dx), ay = fabsf(dy), sum = ax + ay;
          const float p = sum > 0.0f ? dy / sum : 0.0f;                        // [-1, 1]
          const float a = dx >= 0.0f ? (dy >= 0.0f ? p : 4.0f + p) : 2.0f - p;  // [0, 4)
          // (2048 wedges; 1024: +1 %, 512: +20 %; length classes of 2..16 cells measured the same)
          const unsigned wedge = min(kRayWedges - 1u, unsigned(a * float(kRayWedges / 4u)));
          const unsigned lk = ray_len_key(unsigned(min(sum / Q.resolution, 1.0e6f)));
          // (measurement override: dbg bits 16..19 = log2(groups) + 1)
          unsigned cpg = kRayGroupClasses;
          if ((Q.dbg >> 16) & 15) cpg = kRayLenClasses >> (((Q.dbg >> 16) & 15) - 1);
          key[k] = (lk / cpg) * (kRayWedges * cpg) + wedge * cpg + (lk % cpg);
          if (Q.by_sector)  // (which sector a ray lands in decides nothing but speed: atan2f need not be exact)
            key[k] = ray_sector(dx, dy) * kRaySectorClasses + lk / (kRayLenClasses / kRaySectorClasses);
          rank[k] = atomicAdd(&bin_cnt[key[k]], 1u);
        }
      }
    }
    // the evidence count: neighbouring lanes are neighbouring points of the scan (the beams of one firing step up a wall
    // stand in ONE cell), so a run of equal cells is one atomic with the run's length — one memory-side atomic per point
    // was half of this kernel at configs[3] (1 M atomics, 38 -> 19 us without them)
    {
      const int prev = __builtin_amdgcn_update_dpp(-2, o, 0x138, 0xF, 0xF, false);  // wave_shr:1 (lane 0 keeps -2)
      const unsigned long long edge = __ballot(o != prev);                          // a run starts here (lane 0 always)
      if (o >= 0 && o != prev && !(Q.dbg & 128)) {  // (dbg 128, measurement only: no evidence count)
        const unsigned long long later = lane == 63 ? 0ull : (edge >> (lane + 1));
        const unsigned run = later ? unsigned(__ffsll((long long)later)) : 64u - unsigned(lane);
        atomicAdd(&rc_cnt[o], run);
      }
    }
    mask[k] = __ballot(ray);
    if (lane == 0) s_wave[k][w] = unsigned(__popcll(mask[k]));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned tot = 0u;
#pragma unroll
    for (int k = 0; k < PTS; ++k) tot += s_wave[k][0] + s_wave[k][1] + s_wave[k][2] + s_wave[k][3];
    // large scans (blk_cnt != nullptr): the block's rays go to the block's OWN region of the queue, [blk * points per
    // block, ...), and k_ray_scatter walks the regions — the queue tail was one same-address returning atomic per block
    if (blk_cnt) {
      blk_cnt[blk] = tot;
      s_base = blk * unsigned(PTS) * 256u;
    } else {
      s_base = tot ? atomicAdd(ray_count, tot) : 0u;
    }
  }
  __syncthreads();
  unsigned off = s_base;
#pragma unroll
  for (int k = 0; k < PTS; ++k) {
    unsigned mine = off + unsigned(__popcll(mask[k] & ((1ull << lane) - 1ull)));
    for (int v = 0; v < w; ++v) mine += s_wave[k][v];
    if ((mask[k] >> lane) & 1ull) {
      ray_list[mine] = idx[k];
      if (bin_cnt) {
        ray_key[mine] = key[k];
        ray_rank[mine] = rank[k];
      }
    }
    off += s_wave[k][0] + s_wave[k][1] + s_wave[k][2] + s_wave[k][3];
  }
}
template <bool VOXEL, int PTS>
__global__ __launch_bounds__(256) void k_ray_compact(const RayParams Q, const GeomConst G,
                                                     DevState* __restrict__ st,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z,
                                                     const uint32_t* __restrict__ sel,
                                                     uint32_t* __restrict__ rc_cnt,
                                                     uint32_t* __restrict__ ray_list,
                                                     uint32_t* __restrict__ ray_key,
                                                     uint32_t* __restrict__ ray_rank,
                                                     uint32_t* __restrict__ bin_cnt,
                                                     uint32_t* __restrict__ blk_cnt) {
  const DevGeom g = ray_geom(st, Q);
  if (!ray_stage_runs(Q, st, g, G)) return;
  if (blockIdx.x == 0 && threadIdx.x == 0 && st->vis_ray == 0u)
    st->vis_ray = Q.vis_stamp;  // the three layers become visible (raycasting.cpp:223-226)
  ray_compact_body<VOXEL, PTS>(Q, G, g, x, y, z, sel, rc_cnt, ray_list, ray_key, ray_rank, bin_cnt, ray_counter(st, Q),
                               bin_cnt ? blk_cnt : nullptr, blockIdx.x);
}

// bucket counts -> offsets, in two launches of kRayBins / kRayBinBlock blocks: per-block sums, then every block
// adds up the sums in front of it and scans its own kRayBinBlock counts (4 per thread).  The counts are left
// at zero for the next scan.
inline __global__ __launch_bounds__(256) void k_ray_bin_sum(const RayParams Q, const GeomConst G,
                                                     DevState* __restrict__ st,
                                                     const uint32_t* __restrict__ bin_cnt,
                                                     uint32_t* __restrict__ bin_part) {
  __shared__ unsigned s_w[4];
  const DevGeom g = ray_geom(st, Q);
  if (!ray_stage_runs(Q, st, g, G)) return;
  const uint4 c = reinterpret_cast<const uint4*>(bin_cnt)[blockIdx.x * 256u + threadIdx.x];
  unsigned v = c.x + c.y + c.z + c.w;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) bin_part[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

inline __global__ __launch_bounds__(256) void k_ray_bin_scan(const RayParams Q, const GeomConst G,
                                                      DevState* __restrict__ st,
                                                      uint32_t* __restrict__ bin_cnt,
                                                      const uint32_t* __restrict__ bin_part,
                                                      uint32_t* __restrict__ bin_start) {
  __shared__ unsigned s_w[4], s_p[4];
  const DevGeom g = ray_geom(st, Q);
  if (!ray_stage_runs(Q, st, g, G)) return;
  static_assert(kRayBins / kRayBinBlock == 256u, "one partial sum per thread");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned pre = threadIdx.x < blockIdx.x ? bin_part[threadIdx.x] : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) pre += __shfl_xor(pre, d);
  uint4* cnt4 = reinterpret_cast<uint4*>(bin_cnt) + blockIdx.x * 256u + threadIdx.x;
  const uint4 c = *cnt4;
  *cnt4 = make_uint4(0u, 0u, 0u, 0u);
  const unsigned mine = c.x + c.y + c.z + c.w;
  unsigned inc = mine;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_w[w] = inc;
  if (lane == 0) s_p[w] = pre;
  __syncthreads();
  unsigned base = s_p[0] + s_p[1] + s_p[2] + s_p[3] + inc - mine;
  for (int v = 0; v < w; ++v) base += s_w[v];
  uint4 o;
  o.x = base;
  o.y = o.x + c.x;
  o.z = o.y + c.y;
  o.w = o.z + c.z;
  reinterpret_cast<uint4*>(bin_start)[blockIdx.x * 256u + threadIdx.x] = o;
  if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 255u) *ray_counter(st, Q) = o.w + c.w;  // (the queue's length)
}

// the same in ONE launch for the (sector, length class) order of k_ray_wedge, whose 8 K buckets one workgroup scans:
// thread t owns buckets [8 t, 8 t + 8)
constexpr unsigned kRayScan1Threads = 1024u, kRayScan1Per = 8u;
inline __global__ __launch_bounds__(kRayScan1Threads) void k_ray_bin_scan1(const RayParams Q, const GeomConst G,
                                                                    DevState* __restrict__ st,
                                                                    uint32_t* __restrict__ bin_cnt,
                                                                    uint32_t* __restrict__ bin_start) {
  __shared__ unsigned s_w[kRayScan1Threads / 64u];
  const DevGeom g = ray_geom(st, Q);
  if (!ray_stage_runs(Q, st, g, G)) return;
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint4* const cnt4 = reinterpret_cast<uint4*>(bin_cnt) + threadIdx.x * (kRayScan1Per / 4u);
  uint4 c[kRayScan1Per / 4u];
  unsigned mine = 0u;
#pragma unroll
  for (unsigned j = 0; j < kRayScan1Per / 4u; ++j) {
    c[j] = cnt4[j];
    cnt4[j] = make_uint4(0u, 0u, 0u, 0u);  // (left at zero for the next scan)
    mine += c[j].x + c[j].y + c[j].z + c[j].w;
  }
  unsigned inc = mine;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(inc, d);
    if (int(lane) >= d) inc += o;
  }
  if (lane == 63u) s_w[w] = inc;
  __syncthreads();
  unsigned run = inc - mine;
  for (unsigned v = 0; v < w; ++v) run += s_w[v];
  uint4* const out4 = reinterpret_cast<uint4*>(bin_start) + threadIdx.x * (kRayScan1Per / 4u);
#pragma unroll
  for (unsigned j = 0; j < kRayScan1Per / 4u; ++j) {
    uint4 o;
    o.x = run;
    o.y = o.x + c[j].x;
    o.z = o.y + c[j].y;
    o.w = o.z + c[j].z;
    run = o.w + c[j].w;
    out4[j] = o;
  }
  if (threadIdx.x == kRayScan1Threads - 1u) *ray_counter(st, Q) = run;  // (the queue's length: the sum of all buckets)
}

inline __global__ __launch_bounds__(256) void k_ray_scatter(const RayParams Q, const GeomConst G,
                                                     DevState* __restrict__ st,
                                                     const uint32_t* __restrict__ ray_list,
                                                     const uint32_t* __restrict__ ray_key,
                                                     const uint32_t* __restrict__ ray_rank,
                                                     const uint32_t* __restrict__ bin_start,
                                                     const uint32_t* __restrict__ blk_cnt, const unsigned block_points,
                                                     uint32_t* __restrict__ ray_sorted) {
  const DevGeom g = ray_geom(st, Q);
  if (!ray_stage_runs(Q, st, g, G)) return;
  // block b of the queue builder left blk_cnt[b] rays at the start of its region
  const unsigned cnt = blk_cnt[blockIdx.x], base = blockIdx.x * block_points;
  for (unsigned j = threadIdx.x; j < cnt; j += 256u) {
    const unsigned q = base + j;
    ray_sorted[bin_start[ray_key[q]] + ray_rank[q]] = ray_list[q];
  }
}

// Minimum over the 64 lanes with DPP row operations (VALU rate; a __shfl butterfly is six trips through
// the LDS crossbar).  gfx9 encodings: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror =
// 0x141, row_mirror = 0x140, row_bcast15 = 0x142 (rows 1,3), row_bcast31 = 0x143 (rows 2,3).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  constexpr int kId = -1;  // 0xFFFFFFFF: identity of the unsigned minimum, kept by lanes a step does not write
  uint32_t o;
  o = uint32_t(__builtin_amdgcn_update_dpp(kId, int(v), 0xB1, 0xF, 0xF, false)); v = o < v ? o : v;
  o = uint32_t(__builtin_amdgcn_update_dpp(kId, int(v), 0x4E, 0xF, 0xF, false)); v = o < v ? o : v;
  o = uint32_t(__builtin_amdgcn_update_dpp(kId, int(v), 0x141, 0xF, 0xF, false)); v = o < v ? o : v;
  o = uint32_t(__builtin_amdgcn_update_dpp(kId, int(v), 0x140, 0xF, 0xF, false)); v = o < v ? o : v;
  o = uint32_t(__builtin_amdgcn_update_dpp(kId, int(v), 0x142, 0xA, 0xF, false)); v = o < v ? o : v;
  o = uint32_t(__builtin_amdgcn_update_dpp(kId, int(v), 0x143, 0xC, 0xF, false)); v = o < v ? o : v;
  return uint32_t(__builtin_amdgcn_readlane(int(v), 63));
}

// processScan, second half — traceRay (raycasting.cpp:46-140) for the queued rays, one per lane.
// SEG > 1 (small scans: a few hundred wavefronts, each a chain of ~L/8 dependent L2 round trips):
// a ray is walked by SEG lanes, lane k covering steps [k*S, (k+1)*S) (the last one to the end).  The
// DDA state of step k*S is reached by replaying the walk from the start WITHOUT touching memory —
// the same float operations in the same order, so every visited cell and height is bit-identical —
// which trades cheap ALU work for an SEG times shorter dependent chain.  The SEG copies of a ray
// sit in different wavefronts (thread = seg * padded_rays + ray), so the lanes of a wavefront are
// still neighbouring rays at the same step and the segmented min-scan keeps working.
template <bool TILED, int SEG>
__device__ __forceinline__ void ray_walk_body(const RayParams& Q, const GeomConst& G, const DevGeom& g,
                                              const unsigned n_rays,
                                              const float* __restrict__ x, const float* __restrict__ y,
                                              const float* __restrict__ z,
                                              const uint32_t* __restrict__ ray_list,
                                              uint32_t* __restrict__ rc_min, const unsigned gid) {
  // large scans (one lane per ray, queue sorted by wedge and length) walk 32 cells between two rounds of
  // loads (C4 stage: 8 cells 1.16 ms, 16: 1.10, 32: 1.05, 64: 1.16); the segmented small-scan variants stay at 8
  constexpr int kB = SEG == 1 ? kRayBatchLarge : kRayBatch;
  const unsigned n_pad = (n_rays + 63u) & ~63u;  // whole wavefronts per segment
  unsigned i = gid, seg = 0;
  if (SEG > 1) {
    if (n_pad == 0u) return;
    seg = gid / n_pad;
    i = gid - seg * n_pad;
    if (seg >= unsigned(SEG)) return;  // wave-uniform: n_pad is a multiple of 64
  } else if ((gid & ~63u) >= n_rays) {
    return;  // whole wavefront beyond the queue
  }
  const bool have = i < n_rays;
  const unsigned pi = have ? ray_list[i] : 0u;
  const float ex = have ? x[pi] : 0.f, ey = have ? y[pi] : 0.f, ez = have ? z[pi] : 0.f;

  // fp32 exactly as written in traceRay
  const float sx = Q.ox, sy = Q.oy, sz = Q.oz;
  const float dx = ex - sx, dy = ey - sy;
  const float ray_len_2d = sqrtf(dx * dx + dy * dy);
  bool alive = have && !(ray_len_2d < 1e-4f);  // kMinRayLength
  const float dz = ez - sz;
  const float res = Q.resolution;
  const int nrows = G.rows, ncols = G.cols;
  const float origin_x = static_cast<float>(g.px) + float(nrows) * res * 0.5f;
  const float origin_y = static_cast<float>(g.py) + float(ncols) * res * 0.5f;
  const float gr0 = (origin_x - sx) / res, gc0 = (origin_y - sy) / res;
  const float gr1 = (origin_x - ex) / res, gc1 = (origin_y - ey) / res;
  const float dr = gr1 - gr0, dc = gc1 - gc0;
  int r = static_cast<int>(floorf(gr0));
  int c = static_cast<int>(floorf(gc0));
  constexpr float kInf = 1e30f;
  int step_r = 0, step_c = 0;
  float t_max_r = kInf, t_max_c = kInf, t_delta_r = kInf, t_delta_c = kInf;
  if (fabsf(dr) > 1e-8f) {
    step_r = dr > 0 ? 1 : -1;
    const float boundary = step_r > 0 ? (float(r) + 1.0f) : float(r);
    t_max_r = (boundary - gr0) / dr;
    t_delta_r = float(step_r) / dr;
  }
  if (fabsf(dc) > 1e-8f) {
    step_c = dc > 0 ? 1 : -1;
    const float boundary = step_c > 0 ? (float(c) + 1.0f) : float(c);
    t_max_c = (boundary - gc0) / dc;
    t_delta_c = float(step_c) / dc;
  }
  // The walk: a step is straight-line predicated code (in-map test by unsigned compare, wrap by one
  // conditional subtract, DDA advance by selects).  kB cells are walked in registers, their
  // loads go out together (one L2 round trip), then the atomics — which are what this kernel costs
  // (memory-side, ~1 ns each when lanes hit one address):
  //   * rc_min only ever decreases, so a value read from L2 that is already <= h settles the visit
  //     (a stale, larger value merely costs a redundant atomic);
  //   * neighbouring lanes are neighbouring rays of the scan and mostly stand in the SAME cell at
  //     the same step: a segmented min-scan over runs of equal cell leaves one atomic per run.
  // The loop is wave-uniform (all lanes stay in until the longest ray of the wavefront has ended)
  // because of the cross-lane scan; a finished ray keeps stepping with its visits masked off.
  const int lane = threadIdx.x & 63;
  const int max_steps = nrows + ncols;
  int s = 0;
  int s_end = max_steps;
  if (SEG > 1) {
    // cells the ray crosses: one per row / column boundary between the two end cells (+ slack; an
    // underestimate only makes the last segment longer, coverage stays exact)
    const float est_f = fabsf(floorf(gr1) - float(r)) + fabsf(floorf(gc1) - float(c)) + 2.0f;
    const int est = est_f < float(max_steps) ? int(est_f) : max_steps;
    int seg_len = (est + SEG - 1) / SEG;
    seg_len = (seg_len + kB - 1) / kB * kB;
    const int s_begin = int(seg) * seg_len;
    if (int(seg) != SEG - 1) s_end = s_begin + seg_len;
    for (; s < s_begin && alive; ++s) {  // replay: the walk without the visits
      const bool row = t_max_r < t_max_c;
      const float t_exit = row ? t_max_r : t_max_c;
      alive = alive && !(t_exit >= 1.0f) && (s + 1 < max_steps);
      r += row ? step_r : 0;
      c += row ? 0 : step_c;
      t_max_r = row ? t_max_r + t_delta_r : t_max_r;
      t_max_c = row ? t_max_c : t_max_c + t_delta_c;
    }
    alive = alive && s == s_begin && s < s_end;
  }
  while (__ballot(alive)) {
    int cell[kB];
    uint32_t hh[kB];
#pragma unroll
    for (int j = 0; j < kB; ++j) {
      const bool row = t_max_r < t_max_c;
      // == std::min(t_max_r, t_max_c): on a tie both hold the same value
      const float t_exit = row ? t_max_r : t_max_c;
      const bool in_map = alive && unsigned(r) < unsigned(nrows) && unsigned(c) < unsigned(ncols);
      int mr = r + g.sr, mc = c + g.sc;  // (r + start) % size with both operands in [0, size)
      mr -= mr >= nrows ? nrows : 0;
      mc -= mc >= ncols ? ncols : 0;
      int o;
      if (TILED) {
        o = owned_storage(mr, mc, G);
      } else {
        o = mc * nrows + mr;
      }
      const float height = sz + ((1.0f < t_exit) ? 1.0f : t_exit) * dz;
      cell[j] = in_map ? o : -1;
      hh[j] = ord(height);
      alive = alive && !(t_exit >= 1.0f) && (s + j + 1 < s_end);
      r += row ? step_r : 0;
      c += row ? 0 : step_c;
      t_max_r = row ? t_max_r + t_delta_r : t_max_r;
      t_max_c = row ? t_max_c : t_max_c + t_delta_c;
    }
    s += kB;
    // (the loads are unconditional — a masked step reads cell 0 — so that nothing but the loads
    // sits between them and they leave back to back)
    uint32_t seen[kB];
    if (Q.dbg == 3) {  // measurement only: no loads, every visit settled
#pragma unroll
      for (int j = 0; j < kB; ++j) seen[j] = uint32_t(cell[j] & 1);
    } else {
#pragma unroll
      for (int j = 0; j < kB; ++j)
        seen[j] = __hip_atomic_load(&rc_min[cell[j] >= 0 ? cell[j] : 0], __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
      if (Q.dbg == 2) {  // measurement only: every visit settled by the read
#pragma unroll
        for (int j = 0; j < kB; ++j) seen[j] &= 1u;
      }
    }
#pragma unroll
    for (int j = 0; j < kB; ++j) {
      const bool need = cell[j] >= 0 && hh[j] < seen[j];
      unsigned long long todo = __ballot(need);  // wave-uniform loop: one round per distinct cell that gets lowered
      if (SEG == 16) {  // small scans keep the segmented shuffle scan (C2: 70 vs 75 us; C3 / C4: 275 -> 244 us, 1.64 -> 1.58 ms the other way)
        if (todo == 0ull) continue;
        uint32_t v = cell[j] >= 0 ? hh[j] : 0xFFFFFFFFu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int ocell = __shfl_up(cell[j], d);
          const uint32_t ov = __shfl_up(v, d);
          if (lane >= d && ocell == cell[j]) v = ov < v ? ov : v;
        }
        const int ncell = __shfl_down(cell[j], 1);
        const bool tail = cell[j] >= 0 && (lane == 63 || ncell != cell[j]);
        if (tail && v < seen[j] && Q.dbg != 1) atomicMin(&rc_min[cell[j]], v);
        continue;
      }
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int c = __builtin_amdgcn_readlane(cell[j], leader);
        const bool mine = cell[j] == c;  // every lane standing in that cell joins the minimum
        const uint32_t v = wave_min_u32(mine ? hh[j] : 0xFFFFFFFFu);
        if (lane == leader && Q.dbg != 1) atomicMin(&rc_min[c], v);
        todo &= ~__ballot(mine);
      }
    }
  }
}
template <bool TILED, int SEG>
__global__ __launch_bounds__(256) void k_ray(const RayParams Q, const GeomConst G,
                                             DevState* __restrict__ st,
                                             const float* __restrict__ x, const float* __restrict__ y,
                                             const float* __restrict__ z,
                                             const uint32_t* __restrict__ ray_list,
                                             uint32_t* __restrict__ rc_min) {
  const unsigned n_rays = *ray_counter(st, Q);
  const DevGeom g = ray_geom(st, Q);
  ray_walk_body<TILED, SEG>(Q, G, g, n_rays, x, y, z, ray_list, rc_min, blockIdx.x * 256u + threadIdx.x);
}

// voxelGrid(ANY) on its own (fdm_engine_voxel_any): sel[i] = picked point index if sorted position
// i heads a run, else kNoIdx.  Output order of the filter == ascending i.
inline __global__ __launch_bounds__(256) void k_voxel_select(unsigned n,
                                                      const unsigned long long* __restrict__ keys,
                                                      const uint32_t* __restrict__ idx,
                                                      uint32_t* __restrict__ sel) {
  bool head;
  const uint32_t pick = voxel_pick_block(keys, idx, n, head);
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) sel[i] = head ? pick : kNoIdx;
}

struct RayLayers {
  const float* elevation;  // base + stride (record field or own array)
  int elevation_stride;
  float* logodds;          // _visibility_logodds
  float* ray_min;          // raycasting (per-frame min ray height)
  float* ghost;            // ghost_removal
  float* rec;              // cell records (nullable) and their size in floats: clearAt wipes them
  int rec_floats;
};

// One thread per stored cell: fold the observed evidence, publish the frame's min ray height,
// resolve the ghost decision (raycasting.cpp:175-202), and leave the two scratch arrays clean.
inline __global__ __launch_bounds__(256) void k_ray_resolve(const RayParams Q, const GeomConst G,
                                                     DevState* __restrict__ st, const RayLayers L,
                                                     float* const* __restrict__ layer_ptrs,
                                                     int n_layer_ptrs, uint32_t* __restrict__ rc_cnt,
                                                     uint32_t* __restrict__ rc_min, unsigned ncell) {
  const DevGeom g = st->geom[Q.slot];
  if (!ray_stage_runs(Q, st, g, G)) return;
  const unsigned o = blockIdx.x * 256u + threadIdx.x;
  if (o == 0) *ray_counter(st, Q) = 0u;  // the queue is consumed
  if (o >= ncell) return;
  const float nanv = __uint_as_float(0x7FC00000u);
  const uint32_t cnt = rc_cnt[o];
  const uint32_t hmin = rc_min[o];
  if (cnt) rc_cnt[o] = 0u;
  if (hmin != kRayEmpty) rc_min[o] = kRayEmpty;
  const bool visited = hmin != kRayEmpty;
  float lo = L.logodds[o];
  bool lo_dirty = false;
  if (cnt) {
    if (isnan(lo)) lo = 0.0f;
    for (uint32_t k = 0; k < cnt; ++k) {
      const float a = lo + Q.l_obs;
      const float nx = (Q.l_max < a) ? Q.l_max : a;  // std::min(a, l_max)
      if (nx == lo) break;  // fixed point reached: the remaining folds change nothing
      lo = nx;
    }
    lo_dirty = true;
  }
  float ray = nanv;  // map.clear(raycasting) + this frame's rays
  if (visited) {
    ray = unord(hmin);
    const float elev = L.elevation[size_t(o) * L.elevation_stride];
    if (!isnan(elev) && elev > ray + Q.conflict_thr) {
      if (isnan(lo)) lo = 0.0f;
      lo -= Q.l_ghost;
      lo_dirty = true;
      if (lo < Q.clear_thr) {  // ElevationMap::clearAt: NaN in every layer, then the marker
        for (int k = 0; k < n_layer_ptrs; ++k) layer_ptrs[k][o] = nanv;
        if (L.rec)
          for (int f = 0; f < L.rec_floats; ++f) L.rec[size_t(o) * L.rec_floats + f] = nanv;
        L.ghost[o] = 1.0f;
        return;
      }
    }
  }
  L.ray_min[o] = ray;
  if (lo_dirty) L.logodds[o] = lo;
}

}  // namespace fdm
