// fdm_raywedge.hpp — traceRay (raycasting.cpp:46-140) for LARGE scans with the minimum-height image of an angular
// SECTOR in LDS.  gfx950 only.
//
// What k_ray<., 1> (fdm_raycast.hpp) costs at configs[3] is not the DDA (0.18 of its 0.58 ms) but what a visit does:
// a coherent L2 read, a per-wavefront merge of the lanes standing in one cell and a memory-side atomic per (wavefront,
// cell) — 4.1 M lowering events for 1.0 M cells.  Here the ray queue is ordered (sector of 1.4 deg, length class) and a
// workgroup takes ONE sector (a share of one: blockIdx = sector * parts + part) and keeps its footprint in LDS: a
// visit is a window read and, where that does not settle it, one ds_min_u32 — no memory-side atomic, no merge; the
// cells a workgroup lowered are flushed once, one memory-side atomicMin each.
//
// The window.  Seen from the sensor's cell a sector is a thin band: with the MAJOR axis the one the sector runs
// along (u = cells walked along it, always forwards) and v the signed offset along the other one, every cell of a
// ray with slope m = dv/du lies within |m - m0| * u + |m0| + 3 of m0 * u, m0 the slope of the sector's centre line.
// The window has one row per u and kRwCols columns addressed CYCLICALLY (column = v mod kRwCols): cells within
// kRwCols / 2 of m0 * u never alias, whatever the band's direction — no shear tables, no bounding box that a
// diagonal sector would blow up.  Each ray knows the row up to which it provably stays that close (all rows for a ray
// near the middle of a dense sector; few for a stray ray of a sparse one, none for a ray against the major direction);
// beyond that row, and beyond the window's last row, it walks on with memory-side atomics from the state the window
// walk left: the window only ever decides HOW a visit is stored, never which cells are visited.  The per-ray float
// operations are traceRay's, in its order: every visited cell and height is bit-identical.
#pragma once

#include "fdm_raycast.hpp"

namespace fdm {

constexpr unsigned kRwThreads = 1024u;  // rays per workgroup round
constexpr unsigned kRwCols = 64u;       // window columns
constexpr unsigned kRwRowsMax = 636u;   // 636 * 64 * 4 B = 159 KB: one workgroup per CU
constexpr unsigned kRwChunk = 16u;      // visits between two looks at the window's end
constexpr unsigned kRwRead = 4u;        // visits between two rounds of window reads

__device__ __forceinline__ void rw_store(const GeomConst& G, const DevGeom& g, const bool tiled, int r, int c,
                                         uint32_t* __restrict__ rc_min, uint32_t key) {
  if (unsigned(r) >= unsigned(G.rows) || unsigned(c) >= unsigned(G.cols)) return;
  int mr = r + g.sr, mc = c + g.sc;  // (r + start) % size with both operands in [0, size)
  mr -= mr >= G.rows ? G.rows : 0;
  mc -= mc >= G.cols ? G.cols : 0;
  const int o = tiled ? owned_storage(mr, mc, G) : mc * G.rows + mr;
  if (o >= 0) atomicMin(&rc_min[o], key);
}

// traceRay's set-up for the ray to (ex, ey): grid-frame deltas and the DDA's start state (raycasting.cpp:46-106)
struct RwRay {
  float dr, dc, gr1, gc1;
  float t_max_r, t_max_c, t_delta_r, t_delta_c;
  int step_r, step_c;
};
__device__ __forceinline__ RwRay rw_setup(float ex, float ey, float origin_x, float origin_y, float res, float gr0,
                                          float gc0, int r0, int c0) {
  RwRay R;
  R.gr1 = (origin_x - ex) / res;
  R.gc1 = (origin_y - ey) / res;
  R.dr = R.gr1 - gr0;
  R.dc = R.gc1 - gc0;
  constexpr float kInf = 1e30f;
  R.step_r = 0; R.step_c = 0;
  R.t_max_r = kInf; R.t_max_c = kInf; R.t_delta_r = kInf; R.t_delta_c = kInf;
  if (fabsf(R.dr) > 1e-8f) {
    R.step_r = R.dr > 0 ? 1 : -1;
    const float boundary = R.step_r > 0 ? (float(r0) + 1.0f) : float(r0);
    R.t_max_r = (boundary - gr0) / R.dr;
    R.t_delta_r = float(R.step_r) / R.dr;
  }
  if (fabsf(R.dc) > 1e-8f) {
    R.step_c = R.dc > 0 ? 1 : -1;
    const float boundary = R.step_c > 0 ? (float(c0) + 1.0f) : float(c0);
    R.t_max_c = (boundary - gc0) / R.dc;
    R.t_delta_c = float(R.step_c) / R.dc;
  }
  return R;
}

// FWIN (round 6): the window holds the heights as FLOATS (+inf = not visited) and a visit that lowers a cell is a
// ds_min_f32 — no ord() per visit (three instructions of twenty), and a ray that has ended walks on with dz = NaN: its
// heights are NaN, `NaN < seen` is false, nothing is stored.  ord() is taken once per lowered cell, at the flush.
// (Among equal heights the integer window preferred -0 to +0 and this one keeps the first: the reference keeps the
// first too, raycasting.cpp:131 `height < cur_min`.)
// The window starts at LDS address 0 (the kernel has no static LDS; checked once per block): its words are addressed by
// their byte offset alone — through the generic `s_win` pointer every access paid one more add for a base that is zero.
typedef __attribute__((address_space(3))) float rw_lds_f32;
__device__ __forceinline__ void rw_fmin(uint32_t byte_off, float h) {
  (void)__hip_atomic_fetch_min(reinterpret_cast<rw_lds_f32*>(uintptr_t(byte_off)), h, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float rw_read(uint32_t byte_off) { return *reinterpret_cast<rw_lds_f32*>(uintptr_t(byte_off)); }
constexpr uint32_t kRwEmptyF = 0x7F800000u;  // +inf
template <bool TILED, bool FWIN = true>
__global__ __launch_bounds__(kRwThreads) void k_ray_wedge(const RayParams Q, const GeomConst G,
                                                          DevState* __restrict__ st,
                                                          const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ z,
                                                          const uint32_t* __restrict__ ray_list,
                                                          const uint32_t* __restrict__ bin_start,
                                                          uint32_t* __restrict__ rc_min, const unsigned H,
                                                          const unsigned parts) {
  extern __shared__ uint32_t s_win[];  // [H][kRwCols] (the window starts at LDS address 0)
  const unsigned n_rays = *ray_counter(st, Q);
  if (n_rays == 0u) return;
  if (FWIN && uint32_t(uintptr_t((__attribute__((address_space(3))) uint32_t*)s_win)) != 0u) __builtin_trap();
  // the sector's stretch of the queue (bin_start: first queue position of every (sector, length class) bucket)
  const unsigned sector = blockIdx.x / parts, part = blockIdx.x - sector * parts;
  const unsigned q_lo = bin_start[sector * kRaySectorClasses];
  const unsigned q_hi = sector + 1u < kRaySectors ? bin_start[(sector + 1u) * kRaySectorClasses] : n_rays;
  if (q_lo + part * kRwThreads >= q_hi) return;  // (an empty sector, or a share beyond its rays)
  const DevGeom g = ray_geom(st, Q);
  const float sx = Q.ox, sy = Q.oy, sz = Q.oz;
  const float res = Q.resolution;
  const int nrows = G.rows, ncols = G.cols;
  // fp32 exactly as written in traceRay (ray_walk_body)
  const float origin_x = static_cast<float>(g.px) + float(nrows) * res * 0.5f;
  const float origin_y = static_cast<float>(g.py) + float(ncols) * res * 0.5f;
  const float gr0 = (origin_x - sx) / res, gc0 = (origin_y - sy) / res;
  const int r0 = static_cast<int>(floorf(gr0)), c0 = static_cast<int>(floorf(gc0));
  const int max_steps = nrows + ncols;
  constexpr uint32_t kColMask = (kRwCols - 1u) * 4u, kRowBytes = kRwCols * 4u;

  for (unsigned j = threadIdx.x; j < H * (kRwCols / 4u); j += kRwThreads)
    reinterpret_cast<uint4*>(s_win)[j] = FWIN ? make_uint4(kRwEmptyF, kRwEmptyF, kRwEmptyF, kRwEmptyF)
                                              : make_uint4(kRayEmpty, kRayEmpty, kRayEmpty, kRayEmpty);
  // the sector's band: major axis, direction and slope of its CENTRE line, from the sector's number alone (ray_sector:
  // equal angles counted from -pi; grid rows and columns run against x and y)
  bool major_r, band_ok;
  int dir;
  float m0;
  {
    const float ang = (float(sector) + 0.5f) * (6.28318531f / float(kRaySectors)) - 3.14159265f;
    const float cdr = -cosf(ang), cdc = -sinf(ang);
    major_r = fabsf(cdr) >= fabsf(cdc);
    const float DMr = major_r ? cdr : cdc, DNr = major_r ? cdc : cdr;
    dir = DMr > 0 ? 1 : -1;
    m0 = DNr / fabsf(DMr);
    band_ok = !(Q.dbg & 64);  // (dbg 64, measurement only: no ray takes the window)
  }
  __syncthreads();

  // wavefronts take the round's 64-ray groups in snake order (0 1 2 3 | 7 6 5 4 | 8 ...): the queue is sorted by length
  // inside a sector and wavefront w runs on SIMD w % 4, so every SIMD gets short and long groups alike
  const unsigned wv = threadIdx.x >> 6;
  const unsigned slot = (((wv >> 2) & 1u) ? (wv ^ 3u) : wv) * 64u + (threadIdx.x & 63u);
  for (unsigned base = q_lo + part * kRwThreads; base < q_hi; base += parts * kRwThreads) {
    const unsigned i = base + slot;
    const bool have = i < q_hi;
    const unsigned pi = have ? ray_list[i] : 0u;
    const float ex = have ? x[pi] : sx, ey = have ? y[pi] : sy, ez = have ? z[pi] : sz;
    const float dx = ex - sx, dy = ey - sy;
    const float ray_len_2d = sqrtf(dx * dx + dy * dy);
    bool alive = have && !(ray_len_2d < 1e-4f);  // kMinRayLength
    const float dz = ez - sz;
    const RwRay R = rw_setup(ex, ey, origin_x, origin_y, res, gr0, gc0, r0, c0);
    // the walk in (major, minor) terms: the same variables under other names
    float tM = major_r ? R.t_max_r : R.t_max_c, tN = major_r ? R.t_max_c : R.t_max_r;
    const float dM = major_r ? R.t_delta_r : R.t_delta_c, dN = major_r ? R.t_delta_c : R.t_delta_r;
    const int sM = major_r ? R.step_r : R.step_c, sN = major_r ? R.step_c : R.step_r;
    const float DM = major_r ? R.dr : R.dc, DN = major_r ? R.dc : R.dr;
    const float m = DN / fabsf(DM);
    // rows up to which the ray's cells provably stay within kRwCols / 2 - 1 of m0 * u (2 rows of slack for the DDA's
    // float drift against the straight line; written so that a NaN yields no row at all)
    const float room = float(kRwCols / 2u - 1u) - 3.0f - fabsf(m0);
    float rows_f = room / fabsf(m - m0) - 2.0f;  // (+inf for the middle ray itself)
    rows_f = rows_f >= 0.0f ? fminf(rows_f, float(H)) : 0.0f;
    const uint32_t row_limit = (alive && band_ok && sM == dir) ? uint32_t(rows_f) * kRowBytes : 0u;
    uint32_t rowoff = 0u;  // u * kRowBytes
    int vb = 0;            // v * 4
    const int sNb = sN * 4;
    int s = 0;
    uint32_t dead = 0u;
    float dzw = dz;  // FWIN: NaN once the ray has ended
    const bool no_min = (Q.dbg & (1 << 21)) != 0;  // (dbg_ray 2097152, measurement only: the window is read, never lowered)
    auto walk_window = [&](auto MAJOR_R) {
      // kRwChunk visits between two looks at the window's end: u grows by at most one per step, so a lane with
      // kRwChunk rows (and steps) to spare cannot leave its rows inside a chunk; one without them leaves the window
      // for good.  Inside a chunk there is NO lane-dependent control flow: a ray that has ended keeps stepping with its
      // keys forced to all ones (a minimum with kRayEmpty changes nothing, wherever it lands), which costs one compare,
      // one select and one OR per step — the compiler's masks for a per-lane exit cost thirteen scalar instructions.
      bool in_win = rowoff + (kRwChunk + 1u) * kRowBytes <= row_limit && 1 + int(kRwChunk) <= max_steps;
      if (in_win) {
        // the FIRST visit as traceRay writes it: only here can both t_max be zeros of different sign (a sensor on a cell
        // corner), where `t_max_r < t_max_c ? t_max_r : t_max_c` and a hardware minimum may pick different zeros
        const bool stepM = decltype(MAJOR_R)::value ? (tM < tN) : !(tN < tM);
        const float t_exit = stepM ? tM : tN;
        const float height = sz + ((1.0f < t_exit) ? 1.0f : t_exit) * dz;
        if (FWIN)
          rw_fmin(rowoff | (uint32_t(vb) & kColMask), height);
        else
          atomicMin(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(s_win) + (rowoff | (uint32_t(vb) & kColMask))),
                    ord(height));
        dead = (t_exit >= 1.0f) ? 0xFFFFFFFFu : 0u;
        dzw = (t_exit >= 1.0f) ? __uint_as_float(0x7FC00000u) : dz;
        rowoff += stepM ? kRowBytes : 0u;
        vb += stepM ? 0 : sNb;
        tM = stepM ? tM + dM : tM;
        tN = stepM ? tN : tN + dN;
        s = 1;
      }
      while (true) {
        if (FWIN) dead = (dzw != dzw) ? 0xFFFFFFFFu : 0u;
        in_win = in_win && dead == 0u && rowoff + kRwChunk * kRowBytes <= row_limit && s + int(kRwChunk) <= max_steps;
        if (__ballot(in_win) == 0ull) break;
        if (in_win) {
          s += int(kRwChunk);
#pragma unroll 1
          for (unsigned k = 0; FWIN && k < kRwChunk; k += kRwRead) {
            uint32_t at[kRwRead];
            float hk[kRwRead], seen[kRwRead];
#pragma unroll
            for (unsigned j = 0; j < kRwRead; ++j) {
              const bool stepM = decltype(MAJOR_R)::value ? (tM < tN) : !(tN < tM);  // traceRay's `t_max_r < t_max_c`
              float t_cl;
              asm("v_min3_f32 %0, %1, %2, 1.0" : "=v"(t_cl) : "v"(tM), "v"(tN));
              hk[j] = sz + t_cl * dzw;
              at[j] = rowoff | (uint32_t(vb) & kColMask);
              dzw = (t_cl >= 1.0f) ? __uint_as_float(0x7FC00000u) : dzw;  // t_exit >= 1: traceRay leaves the loop behind this visit
              rowoff += stepM ? kRowBytes : 0u;
              vb += stepM ? 0 : sNb;
              // the axis that steps takes its delta, the other one + 0.0f (t >= 0: the same float): one v_pk_add_f32
              typedef float rw_v2f __attribute__((ext_vector_type(2)));
              rw_v2f t2 = {tM, tN};
              t2 += rw_v2f{stepM ? dM : 0.0f, stepM ? 0.0f : dN};
              tM = t2.x;
              tN = t2.y;
            }
#pragma unroll
            for (unsigned j = 0; j < kRwRead; ++j)
              seen[j] = rw_read(at[j]);
#pragma unroll
            for (unsigned j = 0; j < kRwRead; ++j)
              if (hk[j] < seen[j] && !no_min) rw_fmin(at[j], hk[j]);
          }
#pragma unroll 1
          for (unsigned k = 0; !FWIN && k < kRwChunk; k += kRwRead) {
            // kRwRead steps walked in registers, their window words read together (lanes reading one word share the
            // read), then a ds_min only where the read did not settle the visit
            uint32_t at[kRwRead], key[kRwRead], seen[kRwRead];
#pragma unroll
            for (unsigned j = 0; j < kRwRead; ++j) {
              const bool stepM = decltype(MAJOR_R)::value ? (tM < tN) : !(tN < tM);  // traceRay's `t_max_r < t_max_c`
              // min(std::min(t_max_r, t_max_c), 1.0f) in one instruction (behind the first visit no two zeros meet)
              float t_cl;
              asm("v_min3_f32 %0, %1, %2, 1.0" : "=v"(t_cl) : "v"(tM), "v"(tN));
              const float height = sz + t_cl * dz;
              const int hb = __float_as_int(height);
              key[j] = uint32_t(hb ^ ((hb >> 31) | int(0x80000000u))) | dead;  // ord(height)
              at[j] = rowoff | (uint32_t(vb) & kColMask);
              dead = (t_cl >= 1.0f) ? 0xFFFFFFFFu : dead;  // t_exit >= 1: traceRay leaves the loop behind this visit
              rowoff += stepM ? kRowBytes : 0u;
              vb += stepM ? 0 : sNb;
              tM = stepM ? tM + dM : tM;
              tN = stepM ? tN : tN + dN;
            }
#pragma unroll
            for (unsigned j = 0; j < kRwRead; ++j)
              seen[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_win) + at[j]);
#pragma unroll
            for (unsigned j = 0; j < kRwRead; ++j)
              if (key[j] < seen[j])
                atomicMin(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(s_win) + at[j]), key[j]);
          }
        }
      }
    };
    if (Q.dbg & 16384) alive = false;  // (measurement only: no walk at all)
    if (alive) {
      if (major_r) walk_window(std::true_type{}); else walk_window(std::false_type{});
    }
    if (FWIN ? (dzw != dzw) : (dead != 0u)) alive = false;
    if (alive && !(Q.dbg & 8192)) {  // (dbg 8192, measurement only: no walk outside the window)
      // outside its rows of the window (or never in it): memory-side atomics, one per visit
      const int u = int(rowoff / kRowBytes) * sM, v = vb / 4;
      int r = major_r ? r0 + u : r0 + v, c = major_r ? c0 + v : c0 + u;
      float t_max_r = major_r ? tM : tN, t_max_c = major_r ? tN : tM;
      for (; s < max_steps; ++s) {
        const bool row = t_max_r < t_max_c;
        const float t_exit = row ? t_max_r : t_max_c;
        const float height = sz + ((1.0f < t_exit) ? 1.0f : t_exit) * dz;
        rw_store(G, g, TILED, r, c, rc_min, ord(height));
        if (t_exit >= 1.0f) break;
        r += row ? R.step_r : 0;
        c += row ? 0 : R.step_c;
        t_max_r = row ? t_max_r + R.t_delta_r : t_max_r;
        t_max_c = row ? t_max_c : t_max_c + R.t_delta_c;
      }
    }
  }
  __syncthreads();
  // flush: the cell of window word (u, column) is the one within kRwCols / 2 of floor(m0 * u) in that column
  for (unsigned j = threadIdx.x; j < H * (kRwCols / 4u) && !(Q.dbg & 4096); j += kRwThreads) {  // (dbg 4096, measurement only: no flush)
    const uint4 q = reinterpret_cast<const uint4*>(s_win)[j];
    constexpr uint32_t kEmpty = FWIN ? kRwEmptyF : kRayEmpty;
    if (q.x == kEmpty && q.y == kEmpty && q.z == kEmpty && q.w == kEmpty) continue;
    const int u = int(j / (kRwCols / 4u)), col0 = int(j % (kRwCols / 4u)) * 4;
    const int lo = int(floorf(m0 * float(u))) - int(kRwCols / 2u);
    const int M = dir * u;
    const uint32_t keys[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (keys[k] == kEmpty) continue;
      const int v = lo + ((col0 + k - lo) & int(kRwCols - 1u));
      rw_store(G, g, TILED, major_r ? r0 + M : r0 + v, major_r ? c0 + v : c0 + M, rc_min,
               FWIN ? ord(__uint_as_float(keys[k])) : keys[k]);
    }
  }
}

}  // namespace fdm
