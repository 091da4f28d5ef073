// How many cycles does one wave64 vector instruction cost a SIMD on gfx950?  (round 5: the large-scan kernels are
// issue-bound, so the price list of the instruction classes they are made of decides what a rewrite can gain.)
// Each kernel runs ITER x 64 independent instructions of one class per wave; W waves per SIMD (256-thread blocks,
// W blocks per CU, 256 CUs).  Prints cycles per instruction per SIMD = clock * t * 1024 SIMDs / (waves * instr).
// hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue.bin && ./valu_issue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body, decl, sink)                                                            \
  __global__ __launch_bounds__(256) void name(float* out, int iters, float seed) {                \
    decl;                                                                                         \
    for (int i = 0; i < iters; ++i) { asm volatile(REP8(body) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); } \
    if (sink == 123.456f) out[0] = sink;                                                          \
  }

#define DECLF float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5f
#define DECLU unsigned a0 = unsigned(seed), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 * 3u
#define SINKF (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define SINKU float(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

// 8 independent instructions per body (one per accumulator), REP8 -> 64 per loop iteration
KERNEL(k_add_f32, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n", DECLF, SINKF)
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n", DECLF, SINKF)
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n", DECLF, SINKF)
KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n", DECLU, SINKU)
KERNEL(k_and_b32, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n", DECLU, SINKU)
KERNEL(k_mov_b32, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n", DECLU, SINKU)
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n", DECLU, SINKU)
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8\n", DECLU, SINKU)
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", DECLU, SINKU)
KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8\n", DECLU, SINKU)
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n", DECLF, SINKF)
KERNEL(k_cvt_f64, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n", DECLF, SINKF)

// packed fp32 and fp64 need register pairs
__global__ __launch_bounds__(256) void k_pk_mul_f32(float* out, int iters, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {seed, seed + 1}, a1 = {seed + 2, seed + 3}, a2 = {seed + 4, seed + 5}, a3 = {seed + 6, seed + 7};
  f2 a4 = a0 * 2.f, a5 = a1 * 2.f, a6 = a2 * 2.f, a7 = a3 * 2.f, b = {seed * 0.5f, seed * 0.25f};
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
  }
  const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s.x + s.y == 123.456f) out[0] = s.x;
}
__global__ __launch_bounds__(256) void k_fma_f64(float* out, int iters, float seed) {
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
  }
  const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 123.456) out[0] = float(s);
}
__global__ __launch_bounds__(256) void k_add_f64(float* out, int iters, float seed) {
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
  }
  const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 123.456) out[0] = float(s);
}
// a VALU stream with one SALU instruction after every vector instruction (do they share an issue slot?)
__global__ __launch_bounds__(256) void k_valu_salu(float* out, int iters, float seed) {
  unsigned a0 = unsigned(seed), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 * 3u;
  unsigned s0 = __builtin_amdgcn_readfirstlane(a0);
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8("v_add_u32 %0, %0, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %1, %1, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %2, %2, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %3, %3, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %4, %4, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %5, %5, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %6, %6, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %7, %7, %9\n s_add_u32 %8, %8, 1\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "v"(b) : "scc");
  }
  if (float(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0) == 123.456f) out[0] = 1.f;
}
// LDS atomics on distinct addresses per lane (ds_max_u32 non-returning), and ds_read/ds_write
__global__ __launch_bounds__(256) void k_ds_max(float* out, int iters, float seed) {
  __shared__ unsigned tab[4096];
  for (int k = threadIdx.x; k < 4096; k += 256) tab[k] = 0u;
  __syncthreads();
  unsigned v = unsigned(seed) + threadIdx.x;
  unsigned addr = (threadIdx.x * 4u) & 16383u;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP64("ds_max_u32 %0, %1\n") :: "v"(addr), "v"(v) : "memory");
  }
  __syncthreads();
  if (float(tab[threadIdx.x]) == 123.456f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_ds_min64(float* out, int iters, float seed) {
  __shared__ unsigned long long tab[4096];
  for (int k = threadIdx.x; k < 4096; k += 256) tab[k] = ~0ull;
  __syncthreads();
  unsigned long long v = (unsigned long long)(seed) + threadIdx.x;
  unsigned addr = (threadIdx.x * 8u) & 32767u;
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP64("ds_min_u64 %0, %1\n") :: "v"(addr), "v"(v) : "memory");
  }
  __syncthreads();
  if (float(tab[threadIdx.x]) == 123.456f) out[0] = 1.f;
}

typedef void (*kern_t)(float*, int, float);
struct K { const char* name; kern_t fn; double per_iter; };

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
  const K ks[] = {
    {"v_add_f32", k_add_f32, 64}, {"v_mul_f32", k_mul_f32, 64}, {"v_fma_f32", k_fma_f32, 64}, {"v_pk_mul_f32", k_pk_mul_f32, 64},
    {"v_add_u32", k_add_u32, 64}, {"v_and_b32", k_and_b32, 64}, {"v_mov_b32", k_mov_b32, 64}, {"v_mul_lo_u32", k_mul_lo_u32, 64},
    {"v_lshl_add_u32", k_lshl_add, 64}, {"v_cndmask_b32", k_cndmask, 64}, {"v_cmp_lt_u32", k_cmp_u32, 64},
    {"v_rcp_f32", k_rcp_f32, 64}, {"v_cvt_i32_f32", k_cvt_f64, 64}, {"v_fma_f64", k_fma_f64, 64}, {"v_add_f64", k_add_f64, 64},
    {"v_add_u32+s_add_u32", k_valu_salu, 64}, {"ds_max_u32", k_ds_max, 64}, {"ds_min_u64", k_ds_min64, 64},
  };
  const int iters = 2000;
  for (const K& k : ks) {
    for (int w : {1, 2, 4, 8}) {
      const int grid = 256 * w;  // 256-thread blocks: one wave per SIMD each
      float best = 1e30f;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.fn, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
      }
      // instructions per SIMD = w waves x iters x per_iter; cycles at the nominal clock
      const double instr = double(w) * iters * k.per_iter;
      const double cyc = double(best) * 1e-3 * double(clk_khz) * 1e3 / instr;
      printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_instr_per_simd_at_%dMHz\": %.2f}\n",
             k.name, w, best, clk_khz / 1000, cyc);
    }
  }
  return 0;
}
