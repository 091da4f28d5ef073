// fdm_rbatch.hpp — the raycasting stage (fdm_raycast.hpp) for a BATCH of small scans: five launches per batch of up
// to kMaxBatch scans instead of seven per scan (SURVEY.md §8 row f1 inside the batch pipeline of fdm_multi.hpp).
//
// Reference order (fastdem.cpp:125-159): per scan  crops -> move -> rasterize + update -> voxelGrid(ANY) ->
// applyRaycasting (processScan, resolveGhostCells).  What processScan computes for scan k — the observed-evidence count
// and the minimum ray height per cell — depends on the scan's points, the sensor origin and the map GEOMETRY after the
// scan's move, not on the map's contents; only resolveGhostCells reads the elevation scan k's update left.  So the
// batch pipeline runs, between the k_mbatch launch that bins batch b and the one that updates it,
//   k_rb_count -> k_rb_scatter -> k_rb_mark   voxelGrid(ANY) of every scan of the batch (the sort-free filter k_vs_*)
//   k_rb_compact                               evidence counts + the queue of downward rays, per scan
//   k_rb_ray_lds                               traceRay for every queued ray: per-scan minimum-height images (built per
//                                              quadrant in LDS; k_rb_ray = the global-atomic walk for quadrants beyond it)
// (blockIdx.y = scan; the bodies are the single-scan kernels' own), and the update half of k_mbatch resolves the images
// cell by cell, scan k's behind scan k's observation (mupdate_body's ray events).  The geometry of scan k comes from the
// batch's state (MState::E / C and the scouted pass bits), exactly what the bin half indexed scan k's points with.
#pragma once

#include "fdm_multi.hpp"
#include "fdm_raycast.hpp"

namespace fdm {

static_assert(kMRayEmpty == kRayEmpty, "the update half's empty marker is the ray stage's");

struct RBatch {
  unsigned count, stride, ncell, stamp;  // scans | points per scan slot | cells | this batch's RState::any value
  int do_move, gate_on_filter, dbg, pad;
  const MState* ms;                      // the batch's state: geometry before / after every scan's move, pass bits
  RState* rs;
  float inv_voxel, resolution;           // 1 / voxel size (= map resolution, fastdem.cpp:155) | float(map.getResolution())
  unsigned n[kMaxBatch];
  float ox[kMaxBatch], oy[kMaxBatch], oz[kMaxBatch];  // sensor origin in the map frame (fastdem.cpp:153-154)
  VoxelCompact C[kMaxBatch];
  unsigned shift[kMaxBatch], ibits[kMaxBatch];        // VoxelSmall::shift / ibits of scan k
  const float* cap;                      // [3][kMaxBatch][stride] the preprocessed clouds (MBin::cap)
  uint32_t* keys;                        // [kMaxBatch][stride] each of these
  uint32_t* place;
  uint32_t* sel;
  uint32_t* ray_list;                    // [4][kMaxBatch][stride]: one queue per quadrant
  uint4* rec;
  uint32_t* fine;                        // [kMaxBatch][2^kVsFineBits] zero between batches
  uint32_t* coarse;                      // [kMaxBatch][kVsCoarse]
  uint32_t* rc_cnt;                      // [kMaxBatch][ncell]
  uint32_t* rc_min;
};

__device__ __forceinline__ VoxelSmall rb_voxel(const RBatch& R, const unsigned k) {
  VoxelSmall V;
  V.shift = R.shift[k];
  V.fine = R.fine + (size_t(k) << kVsFineBits);
  V.coarse = R.coarse + size_t(k) * kVsCoarse;
  V.total = &R.rs->total[k];
  V.place = R.place + size_t(k) * R.stride;
  V.rec = R.rec + size_t(k) * R.stride;
  V.cap = R.stride;
  V.ibits = R.ibits[k];
  V.dbg = (R.dbg >> 8) & 3;  // (measurement only, as the single-scan path's dbg_ray bits 8..9)
  return V;
}
// map geometry after scan k's move (what the bin half binned the scan against)
__device__ __forceinline__ DevGeom rb_geom(const RBatch& R, const unsigned k) {
  DevGeom g = R.ms->E[k];
  if (R.do_move && (!R.gate_on_filter || ((R.ms->flags[0] >> k) & 1u) != 0u)) {
    const DevCand c = R.ms->C[k];
    g.px = c.px; g.py = c.py; g.sr = c.sr; g.sc = c.sc;
  }
  return g;
}
__device__ __forceinline__ RayParams rb_params(const RBatch& R, const unsigned k) {
  RayParams Q{};
  Q.ox = R.ox[k]; Q.oy = R.oy[k]; Q.oz = R.oz[k];
  Q.inv_voxel = R.inv_voxel;
  Q.resolution = R.resolution;
  Q.n = R.n[k];
  Q.slot = 0; Q.flag_slot = -1; Q.vis_stamp = 0u;
  Q.pre_slot = -1;
  Q.dbg = R.dbg;
  return Q;
}
__device__ __forceinline__ bool rb_runs(const RBatch& R, const unsigned k) {  // raycasting.cpp:207-220
  return R.rs->any[k] == R.stamp && R.rs->origin_in[k] != 0u;
}

inline __global__ __launch_bounds__(256) void k_rb_count(const RBatch R, const GeomConst G) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k]) return;
  if (blockIdx.x == 0u && threadIdx.x == 0u) {  // (read by the launches behind this one)
    const DevGeom g = rb_geom(R, k);
    R.rs->origin_in[k] = map_contains(double(R.ox[k]), double(R.oy[k]), g, G) ? 1u : 0u;
    R.rs->qcount[k][0] = R.rs->qcount[k][1] = R.rs->qcount[k][2] = R.rs->qcount[k][3] = 0u;
  }
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  vs_count_body(R.n[k], R.inv_voxel, R.C[k], rb_voxel(R, k), R.cap + at, R.cap + plane + at, R.cap + 2u * plane + at,
                R.keys + at, R.sel + at, &R.rs->any[k], R.stamp, blockIdx.x);
}

inline __global__ __launch_bounds__(256) void k_rb_scatter(const RBatch R) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k]) return;
  vs_scatter_body(R.n[k], rb_voxel(R, k), R.keys + size_t(k) * R.stride, blockIdx.x);
}

inline __global__ __launch_bounds__(256) void k_rb_mark(const RBatch R) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k]) return;
  vs_mark_body(rb_voxel(R, k), R.sel + size_t(k) * R.stride, blockIdx.x);
}

// traceRay's grid-frame set-up (raycasting.cpp:46-75), shared by the queue builder and the walk: the map corner the grid
// coordinates count from, and the sensor's (fractional) grid coordinates — the same fp32 operations in the same order
struct RayFrame {
  float origin_x, origin_y, gr0, gc0;
  int r0, c0;
};
__device__ __forceinline__ RayFrame ray_frame(const DevGeom& g, const GeomConst& G, float res, float sx, float sy) {
  RayFrame F;
  F.origin_x = static_cast<float>(g.px) + float(G.rows) * res * 0.5f;
  F.origin_y = static_cast<float>(g.py) + float(G.cols) * res * 0.5f;
  F.gr0 = (F.origin_x - sx) / res;
  F.gc0 = (F.origin_y - sy) / res;
  F.r0 = static_cast<int>(floorf(F.gr0));
  F.c0 = static_cast<int>(floorf(F.gc0));
  return F;
}
// the quadrant a ray leaves the sensor's cell into: bit 0 = towards smaller rows, bit 1 = towards smaller columns (a
// direction the DDA does not step in counts as "larger")
__device__ __forceinline__ unsigned ray_quadrant(const RayFrame& F, float res, float ex, float ey) {
  const float gr1 = (F.origin_x - ex) / res, gc1 = (F.origin_y - ey) / res;
  const float dr = gr1 - F.gr0, dc = gc1 - F.gc0;
  return ((fabsf(dr) > 1e-8f && !(dr > 0)) ? 1u : 0u) | ((fabsf(dc) > 1e-8f && !(dc > 0)) ? 2u : 0u);
}

// processScan's first half for scan blockIdx.y (as k_ray_compact): observed evidence counted, the downward rays queued —
// in FOUR queues, by quadrant (k_rb_ray_lds keeps a quadrant's min-height image in LDS)
inline __global__ __launch_bounds__(256) void k_rb_compact(const RBatch R, const GeomConst G) {
  __shared__ unsigned s_cnt[4][4];  // [wavefront][quadrant]
  __shared__ unsigned s_base[4];
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k] || !rb_runs(R, k)) return;
  const DevGeom g = rb_geom(R, k);
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  const float* __restrict__ const x = R.cap + at;
  const float* __restrict__ const y = R.cap + plane + at;
  const float* __restrict__ const z = R.cap + 2u * plane + at;
  const RayFrame F = ray_frame(g, G, R.resolution, R.ox[k], R.oy[k]);
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  bool ray = false;
  unsigned q = 0u;
  if (i < R.n[k] && R.sel[at + i] != 0u) {
    const float ex = x[i], ey = y[i], ez = z[i];
    DevCand c;  // observed evidence: the point's own cell (nanoGrid getIndex, fp64)
    c.px = g.px; c.py = g.py; c.sr = g.sr; c.sc = g.sc; c.shr = 0; c.shc = 0;
    const int o = owned_cell(ex, ey, c, G);
    if (o >= 0) atomicAdd(&R.rc_cnt[size_t(k) * R.ncell + unsigned(o)], 1u);
    ray = ez < R.oz[k];  // upward rays are skipped (raycasting.cpp:168)
    if (ray) q = ray_quadrant(F, R.resolution, ex, ey);
  }
  unsigned long long mine_mask = 0ull;
#pragma unroll
  for (unsigned qq = 0; qq < 4u; ++qq) {
    const unsigned long long m = __ballot(ray && q == qq);
    if (q == qq) mine_mask = m;
    if (lane == 0u) s_cnt[w][qq] = unsigned(__popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < 4u) {
    const unsigned tot = s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
    s_base[threadIdx.x] = tot ? atomicAdd(&R.rs->qcount[k][threadIdx.x], tot) : 0u;
  }
  __syncthreads();
  if (ray) {
    unsigned pos = s_base[q] + unsigned(__popcll(mine_mask & ((1ull << lane) - 1ull)));
    for (unsigned v = 0; v < w; ++v) pos += s_cnt[v][q];
    R.ray_list[(size_t(q) * kMaxBatch + k) * R.stride + pos] = i;
  }
}

// traceRay for the queued rays with one lane per ray and global atomics (the single-scan kernel's body, k_ray<., SEG>):
// maps whose quadrants do not fit the LDS of k_rb_ray_lds.  blockIdx.z = quadrant queue.
template <int SEG>
__global__ __launch_bounds__(256) void k_rb_ray(const RBatch R, const GeomConst G) {
  const unsigned k = blockIdx.y, q = blockIdx.z;
  const unsigned n_rays = R.rs->qcount[k][q];
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  if ((gid & ~63u) >= ((n_rays + 63u) & ~63u) * unsigned(SEG)) return;  // (whole wavefront beyond the queue)
  const DevGeom g = rb_geom(R, k);
  const RayParams Q = rb_params(R, k);
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  ray_walk_body<false, SEG>(Q, G, g, n_rays, R.cap + at, R.cap + plane + at, R.cap + 2u * plane + at,
                            R.ray_list + (size_t(q) * kMaxBatch + k) * R.stride, R.rc_min + size_t(k) * R.ncell, gid);
}

// traceRay with the minimum-height image in LDS.  A ray stays inside the rectangle spanned by the sensor's cell and its
// end point, i.e. inside ONE quadrant of the map around the sensor's cell; a workgroup takes a share of one quadrant's
// queue (blockIdx.x = quadrant * parts + part) and keeps that quadrant's image in LDS: the walk is arithmetic and LDS
// traffic only — no dependent global round trips, no memory-side atomic per (wavefront, cell, step), which is what the
// global-atomic walk costs (88 us for sixteen VLP-16 scans; the arithmetic of their 19 M steps is ~3 us of the chip) —
// and flushes the cells it lowered with one memory-side atomicMin each (<= `parts` workgroups share a cell, the
// quadrants' common row / column twice that).  Same float operations per step as ray_walk_body, in the same order.
// A quadrant larger than the LDS (a sensor far off the centre of a large map) takes the global atomics directly.
constexpr unsigned kRbRayThreads = 1024u;
#ifndef FDM_RB_STEPS
#define FDM_RB_STEPS 4
#endif
constexpr int kRbSteps = FDM_RB_STEPS;  // cells a ray walks between two rounds of LDS reads
// FIMG (round 6): the image holds the heights as FLOATS (+inf = not visited; a lowering visit is ds_min_f32, ord() once per
// lowered cell at the flush), a visit that must not store carries NaN (`NaN < seen` is false) instead of an index of -1,
// the image word is addressed by a running byte offset instead of (c - c_lo) * qrows + (r - r_lo) per visit, the two t
// updates are one v_pk_add_f32, and a ray stops where it leaves the map: a quadrant walk is monotone and its sensor lies
// in the map (RState::origin_in), so it never comes back and nothing it would do out there has an effect — which also
// makes traceRay's max_steps bound redundant (rows + cols monotone steps from inside the map end outside it).  The
// arithmetic of the 19 M steps of a 16-scan VLP-16 batch was 23 of this kernel's 38 us at ~45 instructions per step.
template <bool FIMG>
__global__ __launch_bounds__(kRbRayThreads) void k_rb_ray_lds(const RBatch R, const GeomConst G, const unsigned parts,
                                                              const unsigned lds_words) {
  extern __shared__ uint32_t s_img[];
  const unsigned k = blockIdx.y, q = blockIdx.x / parts, part = blockIdx.x - q * parts;
  const unsigned n_rays = R.rs->qcount[k][q];
  if (part * kRbRayThreads >= n_rays) return;  // (block-uniform; an empty queue or a share beyond it)
  const DevGeom g = rb_geom(R, k);
  const float res = R.resolution;
  const float sx = R.ox[k], sy = R.oy[k], sz = R.oz[k];
  const RayFrame F = ray_frame(g, G, res, sx, sy);
  const int nrows = G.rows, ncols = G.cols;
  // the quadrant's rectangle in (unwrapped) grid coordinates, clipped to the map
  const bool up = (q & 1u) != 0u, left = (q & 2u) != 0u;
  const int r_lo = up ? 0 : max(F.r0, 0), r_hi = up ? min(F.r0, nrows - 1) : nrows - 1;
  const int c_lo = left ? 0 : max(F.c0, 0), c_hi = left ? min(F.c0, ncols - 1) : ncols - 1;
  const int qrows = r_hi - r_lo + 1, qcols = c_hi - c_lo + 1;
  const bool fits = qrows > 0 && qcols > 0 && unsigned(qrows) * unsigned(qcols) <= lds_words;
  const unsigned words = fits ? unsigned(qrows) * unsigned(qcols) : 0u;
  constexpr uint32_t kEmptyImg = FIMG ? 0x7F800000u : kRayEmpty;  // +inf | ord() of no float
  for (unsigned j = threadIdx.x; j < words; j += kRbRayThreads) s_img[j] = kEmptyImg;
  __syncthreads();
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  const float* __restrict__ const x = R.cap + at;
  const float* __restrict__ const y = R.cap + plane + at;
  const float* __restrict__ const z = R.cap + 2u * plane + at;
  const uint32_t* __restrict__ const list = R.ray_list + (size_t(q) * kMaxBatch + k) * R.stride;
  uint32_t* __restrict__ const rc_min = R.rc_min + size_t(k) * R.ncell;
  const int max_steps = nrows + ncols;
  for (unsigned i = part * kRbRayThreads + threadIdx.x; i < n_rays && !(R.dbg & 8192); i += parts * kRbRayThreads) {  // (dbg: measurement only)
    const unsigned pi = list[i];
    const float ex = x[pi], ey = y[pi], ez = z[pi];
    // fp32 exactly as written in traceRay (ray_walk_body)
    const float dx = ex - sx, dy = ey - sy;
    const float ray_len_2d = sqrtf(dx * dx + dy * dy);
    bool alive = !(ray_len_2d < 1e-4f);  // kMinRayLength
    const float dz = ez - sz;
    const float gr1 = (F.origin_x - ex) / res, gc1 = (F.origin_y - ey) / res;
    const float dr = gr1 - F.gr0, dc = gc1 - F.gc0;
    int r = F.r0, c = F.c0;
    constexpr float kInf = 1e30f;
    int step_r = 0, step_c = 0;
    float t_max_r = kInf, t_max_c = kInf, t_delta_r = kInf, t_delta_c = kInf;
    if (fabsf(dr) > 1e-8f) {
      step_r = dr > 0 ? 1 : -1;
      const float boundary = step_r > 0 ? (float(r) + 1.0f) : float(r);
      t_max_r = (boundary - F.gr0) / dr;
      t_delta_r = float(step_r) / dr;
    }
    if (fabsf(dc) > 1e-8f) {
      step_c = dc > 0 ? 1 : -1;
      const float boundary = step_c > 0 ? (float(c) + 1.0f) : float(c);
      t_max_c = (boundary - F.gc0) / dc;
      t_delta_c = float(step_c) / dc;
    }
    if (fits && FIMG) {
      typedef float rb_v2f __attribute__((ext_vector_type(2)));
      const float nanv = __uint_as_float(0x7FC00000u);
      float* const img = reinterpret_cast<float*>(s_img);
      const uint32_t last = (words - 1u) * 4u;
      // byte offset of the ray's cell in the image, and what a step along either axis adds to it
      uint32_t off = (unsigned(c - c_lo) * unsigned(qrows) + unsigned(r - r_lo)) * 4u;
      const uint32_t add_r = uint32_t(step_r * 4), add_c = uint32_t(step_c * qrows * 4);
      while (alive) {
        uint32_t at_b[kRbSteps];
        float hk[kRbSteps], seen[kRbSteps];
#pragma unroll
        for (int j = 0; j < kRbSteps; ++j) {
          const bool row = t_max_r < t_max_c;
          const float t_exit = row ? t_max_r : t_max_c;  // == std::min(t_max_r, t_max_c): on a tie both hold the same value
          const bool ok = alive && unsigned(r) < unsigned(nrows) && unsigned(c) < unsigned(ncols);
          const float height = sz + fminf(t_exit, 1.0f) * dz;  // (t_exit is never NaN: std::min(t_exit, 1.0f))
          hk[j] = ok ? height : nanv;
          at_b[j] = min(off, last);  // (a dead lane's offset may point anywhere)
          alive = ok && !(t_exit >= 1.0f);
          r += row ? step_r : 0;
          c += row ? 0 : step_c;
          off += row ? add_r : add_c;
          rb_v2f t2 = {t_max_r, t_max_c};
          t2 += rb_v2f{row ? t_delta_r : 0.0f, row ? 0.0f : t_delta_c};  // (t >= 0: + 0.0f is the same float)
          t_max_r = t2.x;
          t_max_c = t2.y;
        }
#pragma unroll
        for (int j = 0; j < kRbSteps; ++j) seen[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + at_b[j]);
#pragma unroll
        for (int j = 0; j < kRbSteps; ++j)
          if (hk[j] < seen[j])
            (void)__hip_atomic_fetch_min(reinterpret_cast<float*>(reinterpret_cast<char*>(img) + at_b[j]), hk[j],
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (fits) {
      // kRbSteps steps walked in registers (straight-line predicated code), their LDS reads leave together, then the
      // atomics of the visits the reads did not settle: one LDS round trip per kRbSteps cells instead of one per cell
      int s = 0;
      while (alive) {
        int idx[kRbSteps];
        uint32_t hh[kRbSteps];
#pragma unroll
        for (int j = 0; j < kRbSteps; ++j) {
          const bool row = t_max_r < t_max_c;
          const float t_exit = row ? t_max_r : t_max_c;  // == std::min(t_max_r, t_max_c): on a tie both hold the same value
          const bool in_map = alive && unsigned(r) < unsigned(nrows) && unsigned(c) < unsigned(ncols);
          const float height = sz + ((1.0f < t_exit) ? 1.0f : t_exit) * dz;
          // (inside the quadrant's rectangle by construction — the queue's quadrant is the walk's step signs; the clamp
          // only keeps a surprise inside the image)
          const unsigned at_img = unsigned(c - c_lo) * unsigned(qrows) + unsigned(r - r_lo);
          idx[j] = in_map ? int(min(at_img, words - 1u)) : -1;
          hh[j] = ord(height);
          alive = alive && !(t_exit >= 1.0f) && (s + j + 1 < max_steps);
          r += row ? step_r : 0;
          c += row ? 0 : step_c;
          t_max_r = row ? t_max_r + t_delta_r : t_max_r;
          t_max_c = row ? t_max_c : t_max_c + t_delta_c;
        }
        s += kRbSteps;
        uint32_t seen[kRbSteps];
#pragma unroll
        for (int j = 0; j < kRbSteps; ++j) seen[j] = (R.dbg & 32768) ? uint32_t(idx[j]) : s_img[idx[j] >= 0 ? idx[j] : 0];
#pragma unroll
        for (int j = 0; j < kRbSteps; ++j)
          if (idx[j] >= 0 && hh[j] < seen[j] && !(R.dbg & 16384)) atomicMin(&s_img[idx[j]], hh[j]);
        if ((R.dbg & 49152) && seen[0] == 0x12345u) s_img[1] = seen[1] + seen[2] + seen[3];  // (measurement variants: keep the values alive)
      }
    } else {  // a quadrant beyond the LDS: memory-side atomics, one per visit
      for (int s = 0; alive; ++s) {
        const bool row = t_max_r < t_max_c;
        const float t_exit = row ? t_max_r : t_max_c;
        if (unsigned(r) < unsigned(nrows) && unsigned(c) < unsigned(ncols)) {
          const float height = sz + ((1.0f < t_exit) ? 1.0f : t_exit) * dz;
          int mr = r + g.sr, mc = c + g.sc;
          mr -= mr >= nrows ? nrows : 0;
          mc -= mc >= ncols ? ncols : 0;
          atomicMin(&rc_min[mc * nrows + mr], ord(height));
        }
        alive = !(t_exit >= 1.0f) && (s + 1 < max_steps);
        r += row ? step_r : 0;
        c += row ? 0 : step_c;
        t_max_r = row ? t_max_r + t_delta_r : t_max_r;
        t_max_c = row ? t_max_c : t_max_c + t_delta_c;
      }
    }
  }
  __syncthreads();
  for (unsigned j = threadIdx.x; j < words && !(R.dbg & 4096); j += kRbRayThreads) {
    const uint32_t v = s_img[j];
    if (v == kEmptyImg) continue;
    int mr = r_lo + int(j % unsigned(qrows)) + g.sr, mc = c_lo + int(j / unsigned(qrows)) + g.sc;
    mr -= mr >= nrows ? nrows : 0;
    mc -= mc >= ncols ? ncols : 0;
    atomicMin(&rc_min[mc * nrows + mr], FIMG ? ord(__uint_as_float(v)) : v);
  }
}

}  // namespace fdm
