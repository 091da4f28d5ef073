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
//   k_rb_ray                                   traceRay for every queued ray: per-scan minimum-height images
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
  uint32_t* ray_list;
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
  V.dbg = 0;
  return V;
}
// map geometry after scan k's move (what the bin half binned the scan against)
__device__ __forceinline__ DevGeom rb_geom(const RBatch& R, const unsigned k) {
  DevGeom g = R.ms->E[k];
  if (R.do_move && (!R.gate_on_filter || ((R.ms->flags[0] >> (16u + k)) & 1u) != 0u)) {
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
  Q.dbg = R.dbg;
  return Q;
}
__device__ __forceinline__ bool rb_runs(const RBatch& R, const unsigned k) {  // raycasting.cpp:207-220
  return R.rs->any[k] == R.stamp && R.rs->origin_in[k] != 0u;
}

__global__ __launch_bounds__(256) void k_rb_count(const RBatch R, const GeomConst G) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k]) return;
  if (blockIdx.x == 0u && threadIdx.x == 0u) {  // (read by the launches behind this one)
    const DevGeom g = rb_geom(R, k);
    R.rs->origin_in[k] = map_contains(double(R.ox[k]), double(R.oy[k]), g, G) ? 1u : 0u;
    R.rs->ray_count[k] = 0u;
  }
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  vs_count_body(R.n[k], R.inv_voxel, R.C[k], rb_voxel(R, k), R.cap + at, R.cap + plane + at, R.cap + 2u * plane + at,
                R.keys + at, R.sel + at, &R.rs->any[k], R.stamp, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_rb_scatter(const RBatch R) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k]) return;
  vs_scatter_body(R.n[k], rb_voxel(R, k), R.keys + size_t(k) * R.stride, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_rb_mark(const RBatch R) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k]) return;
  vs_mark_body(rb_voxel(R, k), R.sel + size_t(k) * R.stride, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_rb_compact(const RBatch R, const GeomConst G) {
  const unsigned k = blockIdx.y;
  if (blockIdx.x * 256u >= R.n[k] || !rb_runs(R, k)) return;
  const DevGeom g = rb_geom(R, k);
  const RayParams Q = rb_params(R, k);
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  ray_compact_body<true, 1>(Q, G, g, R.cap + at, R.cap + plane + at, R.cap + 2u * plane + at, R.sel + at,
                            R.rc_cnt + size_t(k) * R.ncell, R.ray_list + at, nullptr, nullptr, nullptr,
                            &R.rs->ray_count[k], blockIdx.x);
}

// SEG lanes per ray, as the single-scan launch of a small scan (k_ray<., 16>): sixteen scans' walks share the chip, so
// the dependent chain of a walk may be longer
template <int SEG>
__global__ __launch_bounds__(256) void k_rb_ray(const RBatch R, const GeomConst G) {
  const unsigned k = blockIdx.y;
  const unsigned n_rays = R.rs->ray_count[k];
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  if ((gid & ~63u) >= ((n_rays + 63u) & ~63u) * unsigned(SEG)) return;  // (whole wavefront beyond the queue)
  const DevGeom g = rb_geom(R, k);
  const RayParams Q = rb_params(R, k);
  const size_t at = size_t(k) * R.stride, plane = size_t(kMaxBatch) * R.stride;
  ray_walk_body<false, SEG>(Q, G, g, n_rays, R.cap + at, R.cap + plane + at, R.cap + 2u * plane + at, R.ray_list + at,
                            R.rc_min + size_t(k) * R.ncell, gid);
}

}  // namespace fdm
