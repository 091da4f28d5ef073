// Can two kernels of ONE stream overlap on this machine, and in which order are their blocks dispatched?
//   1. K one-block kernels that spin 20 us each, back to back on one stream: default launch vs
//      hipExtLaunchKernel(..., hipExtAnyOrderLaunch) (the dispatch packet without the barrier bit).
//   2. dispatch order: kernel A = many more blocks than the chip holds (each spins 5 us and stamps its start), then
//      kernel B (one block, stamps its start) launched any-order behind it: does B start before A's LAST block has started?
//   3. producer / consumer across the two launches: B's blocks wait (bounded) for a counter every A block adds to
//      when it ends.  If B's blocks could be dispatched ahead of A's they would hold the slots A needs: the bounded
//      spin turns that into a reported timeout instead of a hang.
//   hipcc --offload-arch=gfx950 -O2 anyorder.hip -o anyorder.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_spin(long long ticks, unsigned long long* start, unsigned* done) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && start) start[blockIdx.x] = (unsigned long long)t0;
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && done) atomicAdd(done, 1u);
}
// waits until *done >= need (bounded), then stamps
__global__ void k_wait(const unsigned* done, unsigned need, long long max_ticks, unsigned long long* stamp, unsigned* timeouts) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    stamp[2 * blockIdx.x] = (unsigned long long)t0;
    bool ok = false;
    while (wall_clock64() - t0 < max_ticks) {
      if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = true; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) atomicAdd(timeouts, 1u);
    stamp[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64();
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
  unsigned long long* d_start; unsigned* d_done; unsigned long long* d_stamp; unsigned* d_to;
  const int nA = 8192;
  CK(hipMalloc(&d_start, sizeof(unsigned long long) * nA));
  CK(hipMalloc(&d_done, 256));
  CK(hipMalloc(&d_stamp, sizeof(unsigned long long) * 2 * 4096));
  CK(hipMalloc(&d_to, 256));
  // warm
  for (int i = 0; i < 10; ++i) CK(launch(k_spin, dim3(1), dim3(64), s, 0, 100LL, (unsigned long long*)nullptr, (unsigned*)nullptr));
  CK(hipStreamSynchronize(s));
  // ---- 1 ----
  for (int g1 : {1, 256})
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    for (int rep = 0; rep < 2; ++rep) {
      const int K = 64;
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < K; ++i) CK(launch(k_spin, dim3(g1), dim3(256), s, flags, 2000LL, (unsigned long long*)nullptr, (unsigned*)nullptr));
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("test1 grid=%d flags=%d: %d kernels of 20 us: %.1f us total = %.2f us per kernel\n", g1, flags, K, us, us / K);
    }
  }
  // ---- 2 ----
  for (int flags : {0, (int)hipExtAnyOrderLaunch}) {
    CK(hipMemsetAsync(d_start, 0, sizeof(unsigned long long) * nA, s));
    CK(hipMemsetAsync(d_stamp, 0, 16, s));
    CK(hipMemsetAsync(d_done, 0, 4, s));
    CK(hipMemsetAsync(d_to, 0, 4, s));
    CK(hipStreamSynchronize(s));
    CK(launch(k_spin, dim3(nA), dim3(256), s, flags, 500LL, d_start, d_done));
    CK(launch(k_wait, dim3(1), dim3(64), s, flags, (const unsigned*)d_done, 0u, 100LL, d_stamp, d_to));
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> st(nA);
    unsigned long long stampB[2];
    CK(hipMemcpy(st.data(), d_start, sizeof(unsigned long long) * nA, hipMemcpyDeviceToHost));
    CK(hipMemcpy(stampB, d_stamp, 16, hipMemcpyDeviceToHost));
    const unsigned long long a0 = *std::min_element(st.begin(), st.end()), a1 = *std::max_element(st.begin(), st.end());
    printf("test2 flags=%d: A's blocks start over %.1f us; B starts %.1f us after A's first, %.1f us after A's LAST start\n",
           flags, (a1 - a0) / 100.0, (double)(long long)(stampB[0] - a0) / 100.0, (double)(long long)(stampB[0] - a1) / 100.0);
  }
  // ---- 3 ----
  for (int f3 : {0, (int)hipExtAnyOrderLaunch})
  for (int nB : {256, 2048}) {
    CK(hipMemsetAsync(d_start, 0, sizeof(unsigned long long) * nA, s));
    CK(hipMemsetAsync(d_stamp, 0, sizeof(unsigned long long) * 2 * 4096, s));
    CK(hipMemsetAsync(d_done, 0, 4, s));
    CK(hipMemsetAsync(d_to, 0, 4, s));
    CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    CK(launch(k_spin, dim3(nA), dim3(256), s, f3, 500LL, d_start, d_done));
    CK(launch(k_wait, dim3(nB), dim3(256), s, f3, (const unsigned*)d_done, (unsigned)nA, 200000LL /* 2 ms */, d_stamp, d_to));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    unsigned to = 0;
    CK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> st(nA), sb(2 * nB);
    CK(hipMemcpy(st.data(), d_start, sizeof(unsigned long long) * nA, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sb.data(), d_stamp, sizeof(unsigned long long) * 2 * nB, hipMemcpyDeviceToHost));
    const unsigned long long a0 = *std::min_element(st.begin(), st.end()), a1 = *std::max_element(st.begin(), st.end());
    unsigned long long b0 = ~0ull, wmax = 0;
    for (int i = 0; i < nB; ++i) { b0 = std::min(b0, sb[2 * i]); wmax = std::max(wmax, sb[2 * i + 1] - sb[2 * i]); }
    printf("test3 flags=%d nB=%d: %.1f us total, timeouts %u, A starts over %.1f us, first B block %.1f us after A's last start, longest wait %.1f us\n",
           f3, nB, us, to, (a1 - a0) / 100.0, (double)(long long)(b0 - a1) / 100.0, wmax / 100.0);
  }
  return 0;
}
