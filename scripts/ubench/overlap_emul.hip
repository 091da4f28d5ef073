// What would overlapping consecutive fused launches buy?  An emulation of the configs[3] launch: 1444 "tile groups"
// (idle chains: s_sleep, 16 KB LDS, mean 10 us, 80 of them 27 us) + 2048 "bin blocks" (3 us idle, ~1100 dependent
// vector instructions per wavefront, 3 us idle; 22 KB LDS), 6 waves per SIMD.  40 launches back to back on one
// stream: in order (barrier bit) vs hipExtAnyOrderLaunch (no dependencies modelled: the upper bound of the gain).
//   hipcc --offload-arch=gfx950 -O2 overlap_emul.hip -o overlap_emul.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256, 6) void k_mix(int n_upd, int upd_first, int fma_iters, float* sink) {
  __shared__ unsigned char lds[22 * 1024];
  unsigned b = blockIdx.x;
  const int n_bin = int(gridDim.x) - n_upd;
  bool upd;
  if (upd_first) { upd = int(b) < n_upd; if (!upd) b -= n_upd; }
  else { upd = int(b) >= n_bin; if (upd) b -= n_bin; }
  float acc = float(threadIdx.x);
  if (upd) {
    // idle chain: ~1 us per 2 x s_sleep(16); mean 10 us, every 18th 27 us, a third 3 us
    int us = 10;
    if (b % 18u == 0u) us = 27; else if (b % 3u == 0u) us = 3;
    for (int i = 0; i < us * 2; ++i) __builtin_amdgcn_s_sleep(16);
    for (int k = 0; k < 460; ++k) acc = acc * 1.0001f + 0.5f;
  } else {
    for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(16);
    for (int i = 0; i < fma_iters; ++i) {
#pragma unroll 16
      for (int k = 0; k < 256; ++k) acc = acc * 1.0001f + 0.5f;
    }
    for (int i = 0; i < 6; ++i) __builtin_amdgcn_s_sleep(16);
  }
  if (acc == 12345.f) { lds[threadIdx.x] = 1; __syncthreads(); sink[0] = lds[(threadIdx.x + 1) % 1024]; }
}
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* sink; CK(hipMalloc(&sink, 256));
  auto go = [&](int n_upd, int n_bin, int upd_first, int fma_iters, int flags) -> double {
    void* args[4]; args[0] = &n_upd; args[1] = &upd_first; args[2] = &fma_iters; args[3] = &sink;
    const int K = 60;
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipStreamSynchronize(s);
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < K; ++i)
        hipExtLaunchKernel((const void*)k_mix, dim3(n_upd + n_bin), dim3(256), args, 0, s, nullptr, nullptr, flags);
      hipStreamSynchronize(s);
      best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K);
    }
    return best;
  };
  for (int fma_iters : {4, 5}) {
    printf("fma_iters %d: update groups alone %.1f us, bin blocks alone %.1f us\n", fma_iters, go(1444, 0, 1, fma_iters, 0), go(0, 2048, 1, fma_iters, 0));
    for (int upd_first : {1, 0})
      printf("  fused, %s first: in order %.1f us per launch, any order %.1f us per launch\n", upd_first ? "updates" : "bins",
             go(1444, 2048, upd_first, fma_iters, 0), go(1444, 2048, upd_first, fma_iters, 1));
  }
  return 0;
}
