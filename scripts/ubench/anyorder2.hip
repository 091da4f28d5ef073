// Follow-up of anyorder.hip: per-XCD dispatch order of two any-order kernels of one stream, and which pairs overlap.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Rec { unsigned long long t0, t1; unsigned xcc, pad; };
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
__global__ void k_spin(long long ticks, Rec* rec, unsigned* done) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) {
    if (rec) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].xcc = xcc_id(); }
    if (done) atomicAdd(done, 1u);
  }
}
__global__ void k_spin_b(long long ticks, Rec* rec, unsigned* done) {   // (a different kernel object)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks + 1) {}
  if (threadIdx.x == 0) {
    if (rec) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].xcc = xcc_id(); }
    if (done) atomicAdd(done, 1u);
  }
}
__global__ void k_wait(const unsigned* done, unsigned need, long long max_ticks, Rec* rec, unsigned* timeouts) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    bool ok = false;
    while (wall_clock64() - t0 < max_ticks) {
      if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) atomicAdd(timeouts, 1u);
    rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); rec[blockIdx.x].xcc = xcc_id();
  }
  __syncthreads();
}
template <typename... A>
static hipError_t launch(void (*k)(A...), dim3 g, dim3 b, hipStream_t s, int flags, A... a) {
  void* args[] = {(void*)&a...};
  return hipExtLaunchKernel((const void*)k, g, b, args, 0, s, nullptr, nullptr, flags);
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Rec *dA, *dB; unsigned *d_done, *d_to;
  const int cap = 16384;
  CK(hipMalloc(&dA, sizeof(Rec) * cap)); CK(hipMalloc(&dB, sizeof(Rec) * cap));
  CK(hipMalloc(&d_done, 256)); CK(hipMalloc(&d_to, 256));
  for (int i = 0; i < 10; ++i) CK(launch(k_spin, dim3(1), dim3(64), s, 0, 100LL, (Rec*)nullptr, (unsigned*)nullptr));
  CK(hipStreamSynchronize(s));
  // ---- alternate two kernels, small grids ----
  for (int flags : {0, 1}) {
    const int K = 64;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; ++i) {
      if (i & 1) CK(launch(k_spin_b, dim3(64), dim3(256), s, flags, 2000LL, (Rec*)nullptr, (unsigned*)nullptr));
      else CK(launch(k_spin, dim3(64), dim3(256), s, flags, 2000LL, (Rec*)nullptr, (unsigned*)nullptr));
    }
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("alt flags=%d: %d kernels (two alternating) of 20 us: %.2f us per kernel\n", flags, K, us / K);
  }
  // ---- pairs ----
  struct Cfg { int nA, tA; long long spinA; int nB, tB; };
  const Cfg cfgs[] = {{2048, 256, 1000, 1444, 256}, {2048, 256, 1000, 1, 64}, {2048, 256, 1000, 8, 256}, {2048, 256, 1000, 64, 256},
                      {8192, 256, 500, 1444, 256}, {256, 256, 1000, 256, 256}, {4096, 256, 1000, 4096, 256}};
  for (const Cfg& c : cfgs) for (int flags : {0, 1}) {
    CK(hipMemsetAsync(dA, 0, sizeof(Rec) * cap, s)); CK(hipMemsetAsync(dB, 0, sizeof(Rec) * cap, s));
    CK(hipMemsetAsync(d_done, 0, 4, s)); CK(hipMemsetAsync(d_to, 0, 4, s));
    CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    CK(launch(k_spin, dim3(c.nA), dim3(c.tA), s, flags, c.spinA, dA, d_done));
    CK(launch(k_wait, dim3(c.nB), dim3(c.tB), s, flags, (const unsigned*)d_done, (unsigned)c.nA, 200000LL, dB, d_to));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned to = 0; CK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
    std::vector<Rec> a(c.nA), b(c.nB);
    CK(hipMemcpy(a.data(), dA, sizeof(Rec) * c.nA, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dB, sizeof(Rec) * c.nB, hipMemcpyDeviceToHost));
    unsigned long long a0 = ~0ull, a1s = 0, a1e = 0, b0 = ~0ull, b1 = 0, wmax = 0;
    unsigned long long xa_last[8] = {0}, xb_first[8]; int xa_n[8] = {0}, xb_n[8] = {0};
    for (auto& v : xb_first) v = ~0ull;
    for (auto& r : a) { a0 = std::min(a0, r.t0); a1s = std::max(a1s, r.t0); a1e = std::max(a1e, r.t1); xa_last[r.xcc & 7] = std::max(xa_last[r.xcc & 7], r.t0); xa_n[r.xcc & 7]++; }
    for (auto& r : b) { b0 = std::min(b0, r.t0); b1 = std::max(b1, r.t1); wmax = std::max(wmax, r.t1 - r.t0); xb_first[r.xcc & 7] = std::min(xb_first[r.xcc & 7], r.t0); xb_n[r.xcc & 7]++; }
    // how many B blocks started before A's last block had STARTED / before A had ENDED
    int early_s = 0, early_e = 0;
    for (auto& r : b) { early_s += r.t0 < a1s; early_e += r.t0 < a1e; }
    printf("pair A=%dx%d(%lld us) B=%dx%d flags=%d: host %.1f us, timeouts %u | A starts over %.1f us, ends at %.1f | B first start %.1f, last end %.1f, longest wait %.1f | B blocks started before A's last start: %d, before A's end: %d\n",
           c.nA, c.tA, c.spinA / 100, c.nB, c.tB, flags, us, to, (a1s - a0) / 100.0, (a1e - a0) / 100.0,
           (double)(long long)(b0 - a0) / 100.0, (double)(long long)(b1 - a0) / 100.0, wmax / 100.0, early_s, early_e);
    if (flags == 1 && c.nB >= 64) {
      printf("   per XCD: blocks A/B, A's last start, B's first start (us after A's first):");
      for (int x = 0; x < 8; ++x) printf("  [%d] %d/%d %.1f %.1f", x, xa_n[x], xb_n[x], (double)(long long)(xa_last[x] - a0) / 100.0, xb_n[x] ? (double)(long long)(xb_first[x] - a0) / 100.0 : -1.0);
      printf("\n");
    }
  }
  return 0;
}
