// fdm_device.hpp — device-side arithmetic of the integrate() path (gfx950 only).
//
// Every function states the float/double evaluation ORDER it commits to, because
// cell indices must be bit-exact against the reference semantics and the whole
// translation unit is compiled with -ffp-contract=off (no FMA contraction;
// HIP's default correctly-rounded fp32 div/sqrt and f32 denormals are kept).
// Reference arithmetic being reproduced (file:line under /root/reference):
//   Matrix4f * Vector4f      fastdem/lib/nanoPCL/include/nanopcl/core/transform.hpp:19-37
//   cropRange / cropZ        fastdem/lib/nanoPCL/include/nanopcl/filters/impl/crop_impl.hpp:79-96,167-178
//   sensor models            fastdem/include/fastdem/sensors/{sensor_model,lidar_model,rgbd_model}.hpp
//   R*Sigma*R^T              fastdem/src/fastdem.cpp:182-187
//   getIndex / move          nanoGrid (grid_map_core lineage), call sites elevation_mapping.cpp:55,113
//   Kalman / P2              fastdem/include/fastdem/mapping/{kalman,quantile}_estimation.hpp
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Measurement builds only (make -C fastdem_amd/csrc phases): three intermediate time stamps per block of the fused
// launches, packed into the timeline's end word (scripts/phases_batch.py, scripts/phases_tiled.py).
#ifndef FDM_MB_PHASES
#define FDM_MB_PHASES 0
#endif
#if FDM_MB_PHASES
#define FDM_PHASE(i) do { if (threadIdx.x == 0) fdm::g_phase[i] = unsigned(wall_clock64()); } while (0)
#else
#define FDM_PHASE(i) do { } while (0)
#endif

namespace fdm {

#if FDM_MB_PHASES
__shared__ unsigned g_phase[4];
// end | phase stamps, 16 bits each, 10 ns ticks after t0
__device__ __forceinline__ unsigned long long phase_word(unsigned long long t0) {
  const unsigned base = unsigned(t0);
  auto d16 = [&](unsigned v) { return (unsigned long long)(min(v - base, 0xFFFFu)); };
  return d16(unsigned(wall_clock64())) | (d16(g_phase[0]) << 16) | (d16(g_phase[1]) << 32) | (d16(g_phase[2]) << 48);
}
#endif

constexpr float kFltMax = 3.402823466e+38f;
constexpr uint64_t kEmptyKey = ~0ull;
constexpr uint32_t kNoIdx = 0xFFFFFFFFu;

// ---- device-resident map geometry: host never needs it between scans ----
struct DevGeom {  // nanogrid position + circular-buffer start index
  double px, py;
  int sr, sc;
  int pad0, pad1;
};
struct DevCand {  // geometry after the LOCAL-mode move of this scan + the index shift
  double px, py;
  int sr, sc;
  int shr, shc;
};
struct DevFlags {
  unsigned any_pass;    // some point survived cropRange+cropZ (fastdem.cpp:138)
  unsigned any_inside;  // some point landed in the map (elevation_mapping.cpp:118)
  unsigned ray_any;     // the voxel-downsampled ray scan is not empty (raycasting.cpp:207)
  unsigned pad1;
};
struct DevObst {       // the last scan that updated the map (observed >= 1 cell): the tiles it
  unsigned scan;       // stamped are the only ones whose obstacle layer can hold non-NaN cells
  unsigned pad0, pad1, pad2;  // scalars on purpose: an array member here made hipcc spill the
};                     // struct copy to a promoted-LDS alloca addressed through the AQL dispatch
                       // packet (host-visible memory): +10 us on every k_update launch
struct DevState {
  DevGeom geom[4];   // ring: scan t reads slot t&3, its update kernel writes slot (t+1)&3
  DevCand cand[4];
  DevFlags flags[4];
  DevObst obst[4];
  // When a lazily created layer first became visible (0 = not yet), as an order stamp
  // 3*scan_no + {2: updateIntensity/updateColor, 3: the scan's raycasting stage, 1: applyRaycasting
  // called between scans}: GridMap::getLayers() lists layers in creation order, and the reference
  // creates these on first use (elevation_mapping.cpp:154-175, raycasting.cpp:223-226).
  unsigned vis_int;
  unsigned ray_count;  // rays queued by k_ray_compact for k_ray; k_ray_resolve puts it back to 0
  unsigned vis_col, vis_ray;
  unsigned fault;      // a device-side invariant failed: reported once by the next sync / statistics read-back as FDM_ERR_HIP.
                       // (Nothing sets it since round 4: no kernel of the engine waits for another block any more.)
  // The whole-layer obstacle clear a scan owes when something other than its own pipeline's books may hold obstacle
  // cells (a pipeline switch, a host write of the layer): `dense_owed` is the host's sequence number of the latest such
  // event, `dense_paid` the number the last scan that OBSERVED a cell cleared for (k_obstacle_dense_clear /
  // k_obstacle_dense_paid) — a scan that observes nothing clears nothing, like the reference, and the debt stays.
  unsigned dense_owed, dense_paid;
  unsigned ray_count_b;  // the same counter for a raycasting stage of the second context (RayParams::ctx 1: two stages in flight)
};

struct GeomConst {
  double len_x, len_y, half_x, half_y, res, inv_res;
  double inv_res_k;                    // 2^idx_shift / res (fixed-point index estimate of axis_fast)
  int idx_shift, idx_pad;              // as many fraction bits as the map size leaves in an int32 (<= 20)
  int rows, cols;                      // global buffer size
  int s_r0, s_c0, s_rows, s_cols;      // stored window (tile incl. halo)
  int o_r0, o_c0, o_rows, o_cols;      // owned window (cells this engine updates)
};

struct ScanParams {
  float Tbs[16], Twb[16];  // column-major, Isometry3d::matrix().cast<float>()
  float R[9];              // column-major (T_wb*T_bs).rotation().cast<float>()
  float min_sq, max_sq, z_min, z_max;
  float sp[4];             // sensor-model parameters
  double robot_x, robot_y;
  float ray_ox, ray_oy, ray_oz;  // sensor origin in the map frame (raycasting stage, integrate only)
  double base_x, base_y, base_z;  // T_world_base translation: centre of the cropRange ball (host-side use)
  unsigned n;
  unsigned scan_no;
  int slot;
  int integrate_mode;  // 1 = FastDEM::integrate (transforms + crops), 0 = ElevationMapping::update
  int do_move;         // LOCAL mode / explicit move
  int gate_on_filter;  // integrate(): nothing happens when every point is filtered
  int sensor_type;     // 0 Constant, 1 LiDAR, 2 RGBD
  int has_intensity, has_color, has_var;
  int chain_prev;      // 1: the previous scan's update may still be running on the update stream:
  int prev_do_move;    //    derive this scan's base geometry from the PREVIOUS slot (geom, cand, any_pass)
  int prev_gate;       //    exactly as that update will commit it, instead of reading the committed slot
  int bin_table;       // k_bin: fold the block's run tails into an LDS table before going to memory
  int dbg_no_atomics;  // experiment switch (bench A/B only): skip the scratch atomics
  int drop_nonfinite;  // 1: a point with a non-finite coordinate does not exist (PointCloud2 ingest, from_impl)
  int dbg_upd;         // experiment switch: 1 = k_update returns after the context, 2 = after round 1
  int force_inside;    // 1: "some point of the scan landed in the map" holds whatever this engine's points say (a
                       //    routed slice of a larger scan: the fact is global, fdm_engine_integrate_points4_device)
  int move_basic;      // 1: the strips a move shorter than the map vacates clear the basic layers {elevation, elevation_min,
};                     //    elevation_max} only (option "move_clear_basic": the other reading of nanoGrid's move())

// ---- helpers ----
// uniform value that came out of LDS / a ballot: tell the compiler (everything derived from it — tile number,
// row pointer, loop bounds — then lives in scalar registers instead of one vector register each)
__device__ __forceinline__ unsigned uni(unsigned v) { return unsigned(__builtin_amdgcn_readfirstlane(int(v))); }

__device__ __forceinline__ float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

__device__ __forceinline__ void wrap_index(int& index, int size) {  // grid_map wrapIndexToRange
  if (index < size) {
    if (index >= 0) return;
    if (index >= -size) { index += size; return; }
    index = index % size;
    index += size;
  } else if (index < size * 2) {
    index -= size;
  } else {
    index = index % size;
  }
}

// monotone float -> uint32 (total order of finite floats; -0 < +0)
__device__ __forceinline__ uint32_t ord(float f) {
  const uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unord(uint32_t u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// p <- T * p, Eigen SSE packet order: r = c0*x; r = c1*y + r; r = c2*z + r; r = c3*w + r.
__device__ __forceinline__ void transform4(const float* __restrict__ T, float& x, float& y, float& z,
                                           float& w) {
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float acc = T[0 + r] * x;
    acc = T[4 + r] * y + acc;
    acc = T[8 + r] * z + acc;
    acc = T[12 + r] * w + acc;
    o[r] = acc;
  }
  x = o[0]; y = o[1]; z = o[2]; w = o[3];
}

// ps / res, correctly rounded, without the divide sequence: with inv = RN(1 / res), q0 = RN(ps * inv) is within one ulp
// of the quotient, r = ps - q0 * res is exact in one FMA, and RN(q0 + r * inv) IS RN(ps / res) (Markstein's theorem on
// division by a correctly rounded reciprocal).  scripts/ubench/div_markstein.c: 1.56e8 quotients — around every
// half-integer and integer multiple of 13 resolutions +- 8 ulps, and random — bit-identical to the IEEE divide.  Values
// the theorem's premises do not cover (huge, tiny, non-finite) take the divide.
__device__ __forceinline__ double div_by_res(double ps, double res, double inv) {
  const double q0 = ps * inv;
  if (!(fabs(q0) < 1.0e15) || !(fabs(ps) > 1.0e-280)) return ps / res;
  const double r = fma(-q0, res, ps);
  return fma(r, inv, q0);
}

// nanogrid::GridMap::move arithmetic on the position / start index (no layer access).
__device__ __forceinline__ DevCand move_candidate(const DevGeom& g, const GeomConst& G, double x,
                                                  double y) {
  DevCand c;
  const double ps[2] = {x - g.px, y - g.py};
  int sh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double t = ps[i] / G.res;
    const int v = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
    sh[i] = -v;
  }
  c.sr = g.sr + sh[0];
  c.sc = g.sc + sh[1];
  wrap_index(c.sr, G.rows);
  wrap_index(c.sc, G.cols);
  c.px = g.px + double(-sh[0]) * G.res;
  c.py = g.py + double(-sh[1]) * G.res;
  c.shr = sh[0];
  c.shc = sh[1];
  return c;
}

// The same move() arithmetic without the two IEEE fp64 divides on the common path (a batch of scans walks a chain
// of up to kMaxBatch moves per block, fdm_multi.hpp): t = ps / res is estimated as ps * (1 / res); the estimate is
// off by a few ulp, so trunc(t + 0.5 sign) can only differ from the reference where t + 0.5 sign lies within that
// distance of an integer — those cases (and anything huge) take the exact divide.  Result bit-identical to
// move_candidate.
__device__ __forceinline__ DevCand move_candidate_fast(const DevGeom& g, const GeomConst& G, double x, double y) {
  DevCand c;
  const double ps[2] = {x - g.px, y - g.py};
  int sh[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double te = ps[i] * G.inv_res;
    int v = 0;
    // |t| < ~1e-4 of a cell (an axis the robot does not move along): t +- 0.5 truncates to 0 whatever the sign of
    // t — no divide, no sign question.  Otherwise sign(te) == sign(t) and the half-step estimate decides unless it
    // lies within 1e-4 of an integer (or is huge / NaN): those take the exact divide.
    if (!(fabs(te) <= 1e-4)) {
      const double ue = te + 0.5 * (te > 0 ? 1 : -1);
      v = static_cast<int>(ue);
      const double f = fabs(ue - double(v));
      const bool sure = f > 1e-4 && f < 1.0 - 1e-4 && fabs(ue) < 1.0e6;
      if (!sure) {
        const double t = div_by_res(ps[i], G.res, G.inv_res);
        v = static_cast<int>(t + 0.5 * (t > 0 ? 1 : -1));
      }
    }
    sh[i] = -v;
  }
  c.sr = g.sr + sh[0];
  c.sc = g.sc + sh[1];
  wrap_index(c.sr, G.rows);
  wrap_index(c.sc, G.cols);
  c.px = g.px + double(-sh[0]) * G.res;
  c.py = g.py + double(-sh[1]) * G.res;
  c.shr = sh[0];
  c.shc = sh[1];
  return c;
}

// nanogrid::GridMap::getIndex: fp64, truncation, circular-buffer wrap.
//   inside test : t = -((pos - center) - half);  0 <= t < length      (checkIfPositionWithinMap)
//   index       : trunc(-(((pos - half) - center) / res)), + start, wrapped
// The two expressions associate differently, so both are evaluated where they can disagree; for
// an index 1..size-2 the inside test is implied (they differ by rounding, ~1e-13 of a cell) and
// skipped.  r lies in [0, size] and start in [0, size), so the wrap is one conditional subtract
// (identical to wrapIndexToRange on that range).
__device__ __forceinline__ bool axis_index(double pos, double center, double half, double len,
                                           double res, double inv_res, int start, bool any_start,
                                           int size, int& out) {
  const double v = (pos - half) - center;
  const double t = -v * inv_res;  // fast estimate of -v/res
  int k = static_cast<int>(t);
  const double f = t - double(k);
  const bool sure = f > 1e-4 && f < 1.0 - 1e-4 && t >= 1.0 && t < double(size - 1);
  if (!sure) {  // near a cell edge / map border / outside: the reference arithmetic decides
    const double tt = -((pos - center) - half);
    if (!(tt >= 0.0 && tt < len)) return false;
    k = static_cast<int>(-(v / res));
  }
  k += start;
  if (any_start && k >= size) k -= size;  // getBufferIndexFromIndex wraps BOTH axes if either start != 0
  out = k;
  return k >= 0 && k < size;
}
__device__ __forceinline__ bool cell_of(float xf, float yf, const DevCand& g, const GeomConst& G,
                                        int& r, int& c) {
  const bool any_start = g.sr != 0 || g.sc != 0;
  const bool okr = axis_index(double(xf), g.px, G.half_x, G.len_x, G.res, G.inv_res, g.sr, any_start, G.rows, r);
  const bool okc = axis_index(double(yf), g.py, G.half_y, G.len_y, G.res, G.inv_res, g.sc, any_start, G.cols, c);
  return okr && okc;
}

// The same getIndex, split for branch-lean callers (k_tbin evaluates four points per thread):
//   axis_fast  : k = trunc of a fixed-point estimate of -v/res with `shift` fraction bits.  `sure` when the
//                estimate lies at least 2^-shift cell away from a cell edge and one cell away from the map
//                border: the estimate is off by ~1e-12 cells at most, so trunc(-(v/res)) is k.
//   axis_exact : the reference arithmetic (inside test + IEEE divide) for the lanes that were not sure.
//   axis_wrap  : start index + circular wrap (getBufferIndexFromIndex), range check.
__device__ __forceinline__ int axis_fast(double pos, double off_k, double inv_res_k, int shift, int size,
                                         bool& sure) {
  // -((pos - half) - center) * 2^shift / res in ONE rounding: off_k = (half + center) * inv_res_k per block
  const int ki = static_cast<int>(fma(pos, -inv_res_k, off_k));  // saturating convert; NaN -> 0
  const int mask = (1 << shift) - 1;
  const int k = ki >> shift, fr = ki & mask;
  // fr in [1, mask - 1] and k in [1, size - 2], one unsigned compare each
  sure = unsigned(fr - 1) < unsigned(mask - 1) && unsigned(k - 1) < unsigned(size - 2);
  return k;
}
__device__ __forceinline__ bool axis_exact(double pos, double center, double half, double len, double res,
                                           int& k) {
  const double tt = -((pos - center) - half);
  if (!(tt >= 0.0 && tt < len)) return false;
  const double v = (pos - half) - center;
  k = static_cast<int>(-(v / res));
  return true;
}
__device__ __forceinline__ bool axis_wrap(int& k, int start, bool any_start, int size) {
  k += start;
  if (any_start && k >= size) k -= size;
  return unsigned(k) < unsigned(size);
}

// is buffer index `b` on `axis` inside the strip GridMap::move vacates? (E = geometry
// before the move, sh = index shift).  |sh| >= size -> everything.
__device__ __forceinline__ bool in_cleared_strip(int b, int start, int sh, int size) {
  if (sh == 0) return false;
  const int n = sh > 0 ? sh : -sh;
  if (n >= size) return true;
  int index = sh > 0 ? start : start + sh;
  wrap_index(index, size);
  int d = b - index;
  if (d < 0) d += size;
  return d < n;
}

// Sigma_sensor of one point in the SENSOR frame (column-major 3x3): SensorModel::computeCovariance.
// (PT: ScanParams, or any view with the same member names — fdm_multi.hpp reads them from a device table)
template <class PT>
__device__ __forceinline__ void sensor_cov(const PT& P, float x, float y, float z, float* S) {
  if (P.sensor_type == 1) {  // LiDAR, lidar_model.hpp:64-89
    const float dist_sq = sum3(x * x, y * y, z * z);
    if (dist_sq < 1e-6f) {
      const float v = 0.01f;
#pragma unroll
      for (int k = 0; k < 9; ++k) S[k] = 0.0f * v;
      S[0] = v; S[4] = v; S[8] = v;
    } else {
      const float distance = sqrtf(dist_sq);
      const float dir[3] = {x / distance, y / distance, z / distance};
      const float rn2 = P.sp[0] * P.sp[0];
      const float var_radial = (rn2 < 1e-6f) ? 1e-6f : rn2;
      const float da = distance * P.sp[1];
      const float la2 = da * da;
      const float var_lateral = (la2 < 1e-6f) ? 1e-6f : la2;
#pragma unroll
      for (int k = 0; k < 9; ++k) S[k] = 0.0f * var_lateral;
      S[0] = var_lateral; S[4] = var_lateral; S[8] = var_lateral;
      const float s = var_radial - var_lateral;
      const float t[3] = {s * dir[0], s * dir[1], s * dir[2]};  // (s*A)*B rewrite
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) S[c * 3 + r] = S[c * 3 + r] + dir[c] * t[r];
    }
  } else if (P.sensor_type == 2) {  // RGB-D, rgbd_model.hpp:82-101
#pragma unroll
    for (int k = 0; k < 9; ++k) S[k] = 0.0f;
    const float depth = z;
    if (depth <= 0.0f) {
      const float v = 0.01f;
#pragma unroll
      for (int k = 0; k < 9; ++k) S[k] = 0.0f * v;
      S[0] = v; S[4] = v; S[8] = v;
    } else {
      const float diff = depth - P.sp[2];
      const float sigma_norm = P.sp[0] + P.sp[1] * diff * diff;
      const float sigma_lat = P.sp[3] * depth;
      S[0] = sigma_lat * sigma_lat;
      S[4] = S[0];
      S[8] = sigma_norm * sigma_norm;
    }
  } else {  // Constant, sensor_model.hpp:87-93
    const float v = P.sp[0] * P.sp[0];
#pragma unroll
    for (int k = 0; k < 9; ++k) S[k] = 0.0f * v;
    S[0] = v; S[4] = v; S[8] = v;
  }
}

// sigma_z^2 = (R * Sigma_sensor * R^T)(2,2) for one point in the SENSOR frame.
template <class PT>
__device__ __forceinline__ float sigma_z2(const PT& P, float x, float y, float z) {
  float S[9];  // column-major
  sensor_cov(P, x, y, z, S);
  // M = R*S (row 2 only), out(2,2) = M(2,:) . R(2,:)   — 3-term dots a0b0 + (a1b1 + a2b2)
  const float* R = P.R;
  float M2[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    M2[j] = sum3(R[0 * 3 + 2] * S[j * 3 + 0], R[1 * 3 + 2] * S[j * 3 + 1], R[2 * 3 + 2] * S[j * 3 + 2]);
  return sum3(M2[0] * R[0 * 3 + 2], M2[1] * R[1 * 3 + 2], M2[2] * R[2 * 3 + 2]);
}

// The whole R * Sigma_sensor * R^T (column-major), fastdem.cpp:182-187: M = R*Sigma to a temporary, then
// M*R^T, every coefficient a 3-term dot a0b0 + (a1b1 + a2b2).  Only evaluated for the preprocessed-scan
// callback's cloud, which carries the covariance channel (nanopcl/core/point_cloud.hpp:126-147).
template <class PT>
__device__ __forceinline__ void cov_full(const PT& P, float x, float y, float z, float* out) {
  float S[9], M[9];
  sensor_cov(P, x, y, z, S);
  const float* R = P.R;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      M[j * 3 + i] = sum3(R[0 * 3 + i] * S[j * 3 + 0], R[1 * 3 + i] * S[j * 3 + 1], R[2 * 3 + i] * S[j * 3 + 2]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      out[j * 3 + i] = sum3(M[0 * 3 + i] * R[0 * 3 + j], M[1 * 3 + i] * R[1 * 3 + j], M[2 * 3 + i] * R[2 * 3 + j]);
}

// ---- estimators (one update per touched cell) ----
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
  return (v < lo) ? lo : (hi < v) ? hi : v;
}

struct KalmanState {
  float x, P, count, mean, var, m2, upper, lower;
};
// Kalman::update + computeBounds, kalman_estimation.hpp:98-153
__device__ __forceinline__ void kalman_step(KalmanState& s, float z, float meas_var, float min_var,
                                            float max_var, float q) {
  const float R = (meas_var > 0.0f) ? meas_var : max_var;
  if (isnan(s.x)) {
    s.x = z;
    s.P = R;
    s.count = 1.0f;
  } else {
    s.P += q;
    const float K = s.P / (s.P + R);
    s.x = s.x + K * (z - s.x);
    s.P = (1.0f - K) * s.P;
    s.P = clampf(s.P, min_var, max_var);
    s.count += 1.0f;
  }
  if (isnan(s.mean)) {
    s.mean = z;
    s.var = 0.0f;
    s.m2 = 0.0f;
  } else {
    const float delta = z - s.mean;
    const float new_mean = s.mean + (delta / s.count);
    const float delta2 = z - new_mean;
    s.m2 += delta * delta2;
    s.var = (s.count > 1.0f) ? s.m2 / (s.count - 1.0f) : 0.0f;
    s.mean = new_mean;
  }
  const float sigma = sqrtf((0.0f < s.var) ? s.var : 0.0f);
  s.upper = s.x + 2.0f * sigma;
  s.lower = s.x - 2.0f * sigma;
}

// The same update for a cell that takes SEVERAL observations before its state is stored (fdm_multi.hpp): the sample
// variance and the bounds are functions of (x, count, m2) alone — no later step reads them — so kalman_core leaves them
// out and kalman_finish computes them once, after the last observation, with the operations kalman_step uses (its
// first-sample branch sets var = 0 with m2 = 0: the same value the formula gives).
__device__ __forceinline__ void kalman_core(KalmanState& s, float z, float meas_var, float min_var, float max_var, float q) {
  const float R = (meas_var > 0.0f) ? meas_var : max_var;
  if (isnan(s.x)) {
    s.x = z;
    s.P = R;
    s.count = 1.0f;
  } else {
    s.P += q;
    const float K = s.P / (s.P + R);
    s.x = s.x + K * (z - s.x);
    s.P = (1.0f - K) * s.P;
    s.P = clampf(s.P, min_var, max_var);
    s.count += 1.0f;
  }
  if (isnan(s.mean)) {
    s.mean = z;
    s.m2 = 0.0f;
  } else {
    const float delta = z - s.mean;
    const float new_mean = s.mean + (delta / s.count);
    const float delta2 = z - new_mean;
    s.m2 += delta * delta2;
    s.mean = new_mean;
  }
}
__device__ __forceinline__ void kalman_finish(KalmanState& s) {
  s.var = (s.count > 1.0f) ? s.m2 / (s.count - 1.0f) : 0.0f;
  const float sigma = sqrtf((0.0f < s.var) ? s.var : 0.0f);
  s.upper = s.x + 2.0f * sigma;
  s.lower = s.x - 2.0f * sigma;
}

struct P2Params {
  float dn[5];
  int marker;
  float max_count;
};
struct P2State {
  float elevation, variance, count, upper, lower;
  float q[5], n[5];
};

__device__ __forceinline__ float p2_parabolic(float qm, float q0, float qp, float nm, float n0,
                                              float np, int sign) {
  const float d_right = np - n0;
  const float d_left = n0 - nm;
  const float d_span = np - nm;
  if (d_right == 0.0f || d_left == 0.0f || d_span == 0.0f) return q0;
  const float s = static_cast<float>(sign);
  const float t1 = (d_left + s) * (qp - q0) / d_right;
  const float t2 = (d_right - s) * (q0 - qm) / d_left;
  return q0 + s * (t1 + t2) / d_span;
}
__device__ __forceinline__ float p2_linear(float q0, float qj, float n0, float nj, int sign) {
  const float dn = nj - n0;
  if (dn == 0.0f) return q0;
  return q0 + static_cast<float>(sign) * (qj - q0) / dn;
}

// P2Quantile::update + computeBounds, quantile_estimation.hpp:140-258.
// All marker indexing is static (select chains) so q[]/n[] stay in registers.
__device__ __forceinline__ void p2_step(P2State& s, float x, const P2Params& p) {
  float* q = s.q;
  float* n = s.n;
  float count = s.count;
  if (isnan(count) || count < 0.0f) count = 0.0f;
  if (count < 5.0f) {
    const int ic = static_cast<int>(count);
#pragma unroll
    for (int k = 0; k < 5; ++k) q[k] = (ic == k) ? x : q[k];
    count += 1.0f;
    if (count >= 5.0f) {
      // std::sort on 5 elements IS libstdc++'s __insertion_sort (quantile_estimation.hpp:150; no introsort loop below 16
      // elements), restated step for step: an element smaller than the FIRST goes to the front without a look at the
      // ones in between, any other one walks down until the first neighbour it is not smaller than — and stops there.
      // For five finite values that is any sort.  It is not when a marker is NaN — a cell whose n_points another
      // estimator advanced (setEstimatorType at run time, fastdem.cpp:34-38: the layers are shared) skips a slot, and
      // `NaN < x` is false both ways: rounds 1-5 ran adjacent swaps all the way down instead and left the NaN (and with it
      // every later quantile of the cell) somewhere else than the reference does.  Found by the round-6 oracle soak
      // (scripts/soak_oracle.py, seed 1098 after 784 K scans; profiles/r06/soak_oracle.txt).
#pragma unroll
      for (int i = 1; i < 5; ++i) {
        const float val = q[i];
        if (val < q[0]) {  // std::move_backward(first, i, i + 1); *first = val
#pragma unroll
          for (int j = i; j > 0; --j) q[j] = q[j - 1];
          q[0] = val;
        } else {           // __unguarded_linear_insert
          bool moving = true;
#pragma unroll
          for (int j = i; j > 0; --j) {
            const bool shift = moving && (val < q[j - 1]);
            q[j] = shift ? q[j - 1] : (moving ? val : q[j]);
            moving = shift;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) n[i] = static_cast<float>(i);
    }
  } else {
    int k;
    if (x < q[0]) {
      q[0] = x;
      k = 0;
    } else if (x < q[1]) {
      k = 0;
    } else if (x < q[2]) {
      k = 1;
    } else if (x < q[3]) {
      k = 2;
    } else if (x <= q[4]) {
      k = 3;
    } else {
      q[4] = x;
      k = 3;
    }
#pragma unroll
    for (int i = 1; i < 5; ++i)
      if (i > k) n[i] += 1.0f;
    float n_prime[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) n_prime[i] = p.dn[i] * count;
    count += 1.0f;
    if (p.max_count > 0.0f && count > p.max_count) {
      const float scale = p.max_count / count;
#pragma unroll
      for (int i = 0; i < 5; ++i) n[i] *= scale;
      count = p.max_count;
    }
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float d = n_prime[i] - n[i];
      if ((d >= 1.0f && n[i + 1] - n[i] > 1.0f) || (d <= -1.0f && n[i - 1] - n[i] < -1.0f)) {
        const int sign = (d >= 0.0f) ? 1 : -1;
        const float q_new = p2_parabolic(q[i - 1], q[i], q[i + 1], n[i - 1], n[i], n[i + 1], sign);
        const float qj = (sign > 0) ? q[i + 1] : q[i - 1];
        const float nj = (sign > 0) ? n[i + 1] : n[i - 1];
        q[i] = (q[i - 1] < q_new && q_new < q[i + 1]) ? q_new : p2_linear(q[i], qj, n[i], nj, sign);
        n[i] += static_cast<float>(sign);
      }
    }
  }
  s.count = count;
  // update() writes (count>=5 ? q[m] : x); computeBounds() then overwrites with q[m]
  float qm = q[0];
#pragma unroll
  for (int k = 1; k < 5; ++k) qm = (p.marker == k) ? q[k] : qm;
  s.elevation = qm;
  const float sigma = (q[3] - q[1]) / 2.0f;
  s.variance = sigma * sigma;
  s.lower = q[0];
  s.upper = q[4];
}

}  // namespace fdm
