// fdm_kernels.hpp — the two kernels of one scan (gfx950 / CDNA4, wave64).
//
//   k_bin    : one thread per input point.  Fused preprocessScan (fastdem.cpp:164-190:
//              T_base_sensor, cropRange, cropZ, T_world_base) + LOCAL-mode move arithmetic
//              + nanogrid getIndex + per-cell reduction (elevation_mapping.cpp:41-92) into a
//              device-resident scratch (one (z,index) key per cell).  SoA point reads are
//              fully coalesced; same-cell runs of neighbouring lanes are merged inside the
//              wavefront (segmented scan over DPP-free __shfl_up) so only run heads issue
//              the atomics.
//   k_update : one thread per map cell, dense and coalesced down the column-major layers.
//              Applies the rolling-window strip clear (GridMap::move), the per-cell Kalman or
//              P2 update (elevation_mapping.cpp:94-108), min/max, obstacle, intensity, colour
//              (elevation_mapping.cpp:127-175), resets the scratch, and commits the geometry.
//
// Roofline: both are HBM-bound (no contraction => MFMA is irrelevant).  Algorithmic bytes
// (SURVEY.md §8d): 12 B per input point (+4 intensity, +4 colour); per touched cell
// 72 B (Kalman) / 124 B (P2); 4 B per cell per scan for the obstacle clear.
#pragma once

#include "fdm_device.hpp"

namespace fdm {

struct Scratch {
  unsigned long long* key;  // (ord(z) << 32 | point index), min-reduced; kEmptyKey = untouched
  uint32_t* zmax;           // ord(max z), 0 = none
  uint32_t* imax;           // ord(max non-NaN intensity), 0 = none
  uint32_t* first;          // lowest point index in the cell (intensity NaN-first rule)
  uint32_t* last;           // highest point index in the cell (colour = last point wins)
};

struct KalmanLayers {
  float *elevation, *elevation_min, *elevation_max, *variance, *n_points, *kalman_p, *sample_mean,
      *sample_m2, *upper, *lower, *obstacle, *intensity, *color;
  float min_var, max_var, q;
};
struct P2Layers {
  float *elevation, *elevation_min, *elevation_max, *variance, *n_points, *upper, *lower, *obstacle,
      *intensity, *color;
  float* q[5];
  float* n[5];
  P2Params p;
};

// Bring one input point into the map frame; returns whether it survived the crops.
__device__ __forceinline__ bool preprocess_point(const ScanParams& P, float& x, float& y, float& z) {
  if (!P.integrate_mode) return true;
  float w = 1.0f;
  transform4(P.Tbs, x, y, z, w);
  const float d2 = sum3(x * x, y * y, z * z);
  bool pass = (d2 >= P.min_sq) && (d2 <= P.max_sq);
  pass = pass && (z >= P.z_min) && (z <= P.z_max);
  transform4(P.Twb, x, y, z, w);
  return pass;
}

template <bool WAVE_MERGE>
__global__ __launch_bounds__(256) void k_bin(const ScanParams P, const GeomConst G,
                                             DevState* __restrict__ st,
                                             const float* __restrict__ px,
                                             const float* __restrict__ py,
                                             const float* __restrict__ pz,
                                             const float* __restrict__ pint, const Scratch S,
                                             int32_t* __restrict__ cell_ids) {
  const DevGeom g = st->geom[P.slot];
  DevCand cand;
  if (P.do_move) {
    cand = move_candidate(g, G, P.robot_x, P.robot_y);
  } else {
    cand.px = g.px; cand.py = g.py; cand.sr = g.sr; cand.sc = g.sc; cand.shr = 0; cand.shc = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) st->cand[P.slot] = cand;

  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  bool pass = false, inside = false;
  int cell = -1;
  float z = 0.0f;
  if (i < P.n) {
    float x = px[i], y = py[i];
    z = pz[i];
    pass = preprocess_point(P, x, y, z);
    if (pass) {
      int r, c;
      if (cell_of(x, y, cand, G, r, c)) {
        // owned window of this tile (whole map when untiled)
        const int lr = r - G.o_r0, lc = c - G.o_c0;
        if (lr >= 0 && lc >= 0 && lr < G.o_rows && lc < G.o_cols) {
          inside = true;
          cell = (c - G.s_c0) * G.s_rows + (r - G.s_r0);
        }
      }
    }
    if (cell_ids) cell_ids[i] = inside ? cell : (pass ? -2 : -1);
  }

  // ---- per-cell reduction ----
  unsigned long long key = kEmptyKey;
  uint32_t zmx = 0, imx = 0, fst = kNoIdx, lst = 0;
  if (inside) {
    const float zc = (z == 0.0f) ? 0.0f : z;  // -0 and +0 tie, first index wins
    // strict "z < min_z" from FLT_MAX: NaN / +inf / FLT_MAX never become the minimum
    key = (z < kFltMax) ? ((unsigned long long)ord(zc) << 32) | i
                        : ((unsigned long long)ord(kFltMax) << 32) | kNoIdx;
    zmx = (z > -kFltMax) ? ord(zc) : 0u;
    if (P.has_intensity) {
      const float v = pint[i];
      imx = isnan(v) ? 0u : ord(v);
      fst = i;
    }
    lst = i;
  }

  if (WAVE_MERGE) {
    // Segmented inclusive scan over runs of equal `cell` in neighbouring lanes; the LAST lane
    // of each run then holds the run's reduction and is the only one to touch memory.
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ocell = __shfl_up(cell, d);
      const unsigned long long okey = __shfl_up(key, d);
      const uint32_t ozmx = __shfl_up(zmx, d);
      const uint32_t oimx = __shfl_up(imx, d);
      const uint32_t ofst = __shfl_up(fst, d);
      // lanes lane-d..lane all share `cell` iff the lane d below does (runs are contiguous)
      if (lane >= d && ocell == cell && inside) {
        key = okey < key ? okey : key;
        zmx = ozmx > zmx ? ozmx : zmx;
        imx = oimx > imx ? oimx : imx;
        fst = ofst < fst ? ofst : fst;
      }
    }
    const int ncell = __shfl_down(cell, 1);
    const bool tail = inside && (lane == 63 || ncell != cell);
    if (tail) {
      atomicMin(&S.key[cell], key);
      if (zmx) atomicMax(&S.zmax[cell], zmx);
      if (P.has_intensity) {
        if (imx) atomicMax(&S.imax[cell], imx);
        atomicMin(&S.first[cell], fst);
      }
      if (P.has_color) atomicMax(&S.last[cell], lst);  // tail lane has the highest index of the run
    }
  } else if (inside) {
    atomicMin(&S.key[cell], key);
    if (zmx) atomicMax(&S.zmax[cell], zmx);
    if (P.has_intensity) {
      if (imx) atomicMax(&S.imax[cell], imx);
      atomicMin(&S.first[cell], fst);
    }
    if (P.has_color) atomicMax(&S.last[cell], lst);
  }

  // ---- scan-level facts: flags by plain store (benign race), counts sharded ----
  const unsigned long long mp = __ballot(pass), mi = __ballot(inside);
  __shared__ unsigned s_pass[4], s_in[4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_pass[wave] = __popcll(mp);
    s_in[wave] = __popcll(mi);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned np = s_pass[0] + s_pass[1] + s_pass[2] + s_pass[3];
    const unsigned ni = s_in[0] + s_in[1] + s_in[2] + s_in[3];
    if (np) {
      st->flags[P.slot].any_pass = 1u;
      if (ni) st->flags[P.slot].any_inside = 1u;
      atomicAdd(&st->pass_inside[P.slot][blockIdx.x & (kShards - 1)],
                (unsigned long long)np | ((unsigned long long)ni << 32));
    }
  }
}

// Common per-cell prologue of k_update.  Returns false if the thread has nothing to do.
struct CellCtx {
  int o;          // storage-linear cell
  bool in_strip;  // vacated by the move: previous state is NaN in every layer
  bool touched;
  float min_z, min_z_var, max_z;
  uint32_t first, last;
  uint32_t imax;
};

__device__ __forceinline__ void commit_geometry(const ScanParams& P, DevState* st, bool applied) {
  const int slot = P.slot, nxt = (slot + 1) & 3, nn = (slot + 2) & 3;
  DevGeom g = st->geom[slot];
  if (applied) {
    const DevCand c = st->cand[slot];
    g.px = c.px; g.py = c.py; g.sr = c.sr; g.sc = c.sc;
  }
  st->geom[nxt] = g;
  st->flags[nn].any_pass = 0u;
  st->flags[nn].any_inside = 0u;
  for (int k = 0; k < kShards; ++k) {
    st->pass_inside[nn][k] = 0ull;
    st->touched[nn][k] = 0u;
  }
}

template <typename LAYERS>
__device__ __forceinline__ bool cell_prologue(const ScanParams& P, const GeomConst& G,
                                              DevState* __restrict__ st, const LAYERS& L,
                                              float* const* __restrict__ all_layers, int n_layers,
                                              const Scratch& S, const float* __restrict__ px,
                                              const float* __restrict__ py,
                                              const float* __restrict__ pz,
                                              const float* __restrict__ pvar, unsigned ncell,
                                              CellCtx& cx) {
  const int slot = P.slot;
  const bool any_pass = st->flags[slot].any_pass != 0u;
  const bool any_inside = st->flags[slot].any_inside != 0u;
  const bool applied = P.do_move && (!P.gate_on_filter || any_pass);
  const bool do_update = any_inside;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    commit_geometry(P, st, applied);
    if (do_update) {
      unsigned f = 0;
      if (P.has_intensity) f |= 1u;
      if (P.has_color) f |= 2u;
      if (f) st->sticky |= f;
    }
  }
  if (!applied && !do_update) return false;
  const unsigned o = blockIdx.x * 256u + threadIdx.x;
  if (o >= ncell) return false;
  cx.o = int(o);
  cx.in_strip = false;
  if (applied) {
    const DevGeom E = st->geom[slot];
    const DevCand C = st->cand[slot];
    const int r = int(o % unsigned(G.s_rows)) + G.s_r0;
    const int c = int(o / unsigned(G.s_rows)) + G.s_c0;
    cx.in_strip = in_cleared_strip(r, E.sr, C.shr, G.rows) || in_cleared_strip(c, E.sc, C.shc, G.cols);
  }
  const unsigned long long key = do_update ? S.key[o] : kEmptyKey;
  cx.touched = key != kEmptyKey;
  if (cx.in_strip) {
    const float nanv = __uint_as_float(0x7FC00000u);
    for (int l = 0; l < n_layers; ++l) all_layers[l][o] = nanv;
  }
  if (!cx.touched) {
    if (do_update && !cx.in_strip) L.obstacle[o] = __uint_as_float(0x7FC00000u);
    return false;
  }
  // ---- decode the scan's observation of this cell (CellObservation) ----
  const uint32_t idx = uint32_t(key);
  cx.min_z = kFltMax;
  cx.min_z_var = 0.0f;
  if (idx != kNoIdx) {
    float x = px[idx], y = py[idx], z = pz[idx];
    if (P.has_var) {
      cx.min_z_var = pvar[idx];
    } else if (P.integrate_mode) {
      cx.min_z_var = sigma_z2(P, x, y, z);
    }
    preprocess_point(P, x, y, z);
    cx.min_z = z;
  }
  const uint32_t zm = S.zmax[o];
  cx.max_z = zm ? unord(zm) : -kFltMax;
  S.key[o] = kEmptyKey;
  S.zmax[o] = 0u;
  cx.first = kNoIdx;
  cx.last = 0u;
  cx.imax = 0u;
  if (P.has_intensity) {
    cx.first = S.first[o];
    cx.imax = S.imax[o];
    S.first[o] = kNoIdx;
    S.imax[o] = 0u;
  }
  if (P.has_color) {
    cx.last = S.last[o];
    S.last[o] = 0u;
  }
  return true;
}

// updateMinMax / updateObstacle / updateIntensity / updateColor (elevation_mapping.cpp:127-175)
template <typename LAYERS>
__device__ __forceinline__ void cell_epilogue(const ScanParams& P, const LAYERS& L, const CellCtx& cx,
                                              const float* __restrict__ pint,
                                              const uint32_t* __restrict__ prgb) {
  const int o = cx.o;
  const float nanv = __uint_as_float(0x7FC00000u);
  const float smin = cx.in_strip ? nanv : L.elevation_min[o];
  const float smax = cx.in_strip ? nanv : L.elevation_max[o];
  if (isnan(smin) || cx.min_z < smin) L.elevation_min[o] = cx.min_z;
  if (isnan(smax) || cx.max_z > smax) L.elevation_max[o] = cx.max_z;
  L.obstacle[o] = (cx.max_z > cx.min_z) ? cx.max_z : nanv;
  if (P.has_intensity) {
    const float vf = pint[cx.first];
    const float obs = isnan(vf) ? vf : unord(cx.imax);
    const float stored = cx.in_strip ? nanv : L.intensity[o];
    if (isnan(stored) || obs > stored) L.intensity[o] = obs;
  }
  if (P.has_color) {
    reinterpret_cast<uint32_t*>(L.color)[o] = prgb[cx.last] & 0x00FFFFFFu;
  }
}

__device__ __forceinline__ void count_touched(const ScanParams& P, DevState* st, bool touched) {
  const unsigned long long m = __ballot(touched);
  if ((threadIdx.x & 63) == 0 && m)
    atomicAdd(&st->touched[P.slot][(blockIdx.x * 4 + (threadIdx.x >> 6)) & (kShards - 1)],
              unsigned(__popcll(m)));
}

__global__ __launch_bounds__(256) void k_update_kalman(
    const ScanParams P, const GeomConst G, DevState* __restrict__ st, const KalmanLayers L,
    float* const* __restrict__ all_layers, int n_layers, const Scratch S,
    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
    const float* __restrict__ pint, const uint32_t* __restrict__ prgb,
    const float* __restrict__ pvar, unsigned ncell) {
  CellCtx cx;
  cx.touched = false;
  const bool work = cell_prologue(P, G, st, L, all_layers, n_layers, S, px, py, pz, pvar, ncell, cx);
  if (work) {
    const int o = cx.o;
    const float nanv = __uint_as_float(0x7FC00000u);
    KalmanState s;
    if (cx.in_strip) {
      s.x = s.P = s.count = s.mean = s.var = s.m2 = nanv;
    } else {
      s.x = L.elevation[o];
      s.P = L.kalman_p[o];
      s.count = L.n_points[o];
      s.mean = L.sample_mean[o];
      s.var = L.variance[o];
      s.m2 = L.sample_m2[o];
    }
    kalman_step(s, cx.min_z, cx.min_z_var, L.min_var, L.max_var, L.q);
    L.elevation[o] = s.x;
    L.kalman_p[o] = s.P;
    L.n_points[o] = s.count;
    L.sample_mean[o] = s.mean;
    L.variance[o] = s.var;
    L.sample_m2[o] = s.m2;
    L.upper[o] = s.upper;
    L.lower[o] = s.lower;
    cell_epilogue(P, L, cx, pint, prgb);
  }
  count_touched(P, st, work);
}

__global__ __launch_bounds__(256) void k_update_p2(
    const ScanParams P, const GeomConst G, DevState* __restrict__ st, const P2Layers L,
    float* const* __restrict__ all_layers, int n_layers, const Scratch S,
    const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
    const float* __restrict__ pint, const uint32_t* __restrict__ prgb,
    const float* __restrict__ pvar, unsigned ncell) {
  CellCtx cx;
  cx.touched = false;
  const bool work = cell_prologue(P, G, st, L, all_layers, n_layers, S, px, py, pz, pvar, ncell, cx);
  if (work) {
    const int o = cx.o;
    const float nanv = __uint_as_float(0x7FC00000u);
    P2State s;
    if (cx.in_strip) {
      s.count = nanv;
#pragma unroll
      for (int k = 0; k < 5; ++k) s.q[k] = s.n[k] = nanv;
    } else {
      s.count = L.n_points[o];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        s.q[k] = L.q[k][o];
        s.n[k] = L.n[k][o];
      }
    }
    p2_step(s, cx.min_z, L.p);
    L.n_points[o] = s.count;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      L.q[k][o] = s.q[k];
      L.n[k][o] = s.n[k];
    }
    L.elevation[o] = s.elevation;
    L.variance[o] = s.variance;
    L.upper[o] = s.upper;
    L.lower[o] = s.lower;
    cell_epilogue(P, L, cx, pint, prgb);
  }
  count_touched(P, st, work);
}

// ---- small utility kernels ----
__global__ void k_fill(float* __restrict__ p, float v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ void k_fill_u64(unsigned long long* __restrict__ p, unsigned long long v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
// rectangle <-> contiguous buffer (halo exchange); thread = (row within rect), blockIdx.y = col
__global__ void k_region_copy(float* __restrict__ layer, float* __restrict__ buf, int s_rows, int r0,
                              int c0, int nr, int nc, int to_buf) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (r >= nr || c >= nc) return;
  float* a = layer + size_t(c0 + c) * s_rows + (r0 + r);
  float* b = buf + size_t(c) * nr + r;
  if (to_buf) *b = *a; else *a = *b;
}

}  // namespace fdm
