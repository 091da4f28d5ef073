// How fast does the chip start workgroups once it is full?  N blocks that idle (s_sleep) for ~X us each, more
// blocks than slots: with a free dispatcher the kernel takes ceil(N / slots) * X.
//   hipcc --offload-arch=gfx950 -O2 dispatch_rate.hip -o dispatch_rate.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Rec { unsigned long long t0, t1; };
template <int LDS>
__global__ __launch_bounds__(256) void k_idle(int iters, int jitter, Rec* rec, int valu) {
  __shared__ unsigned char lds[LDS];
  const unsigned long long t0 = wall_clock64();
  int n = iters;
  if (jitter) n = iters / 2 + int((blockIdx.x * 2654435761u >> 8) % unsigned(iters));
  float acc = float(threadIdx.x);
  for (int i = 0; i < n; ++i) {
    if (valu) {
#pragma unroll 16
      for (int k = 0; k < 256; ++k) acc = acc * 1.0001f + 0.5f;     // ~256 dependent FMAs = ~1024+ cycles
    } else {
      __builtin_amdgcn_s_sleep(16);   // 16 * 64 clocks
    }
  }
  if (acc == 12345.f) { lds[threadIdx.x] = 1; __syncthreads(); rec[1].t1 = lds[(threadIdx.x + 1) % LDS]; }
  if (threadIdx.x == 0) { rec[blockIdx.x].t0 = t0; rec[blockIdx.x].t1 = wall_clock64(); }
}
int main() {
  Rec* d; const int cap = 1 << 16;
  CK(hipMalloc(&d, sizeof(Rec) * cap));
  std::vector<Rec> h(cap);
  auto run = [&](int lds_kb, int threads, int N, int iters, int jitter, int valu) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      if (lds_kb <= 1) hipLaunchKernelGGL(k_idle<1024>, dim3(N), dim3(threads), 0, 0, iters, jitter, d, valu);
      else if (lds_kb <= 22) hipLaunchKernelGGL(k_idle<22 * 1024>, dim3(N), dim3(threads), 0, 0, iters, jitter, d, valu);
      else hipLaunchKernelGGL(k_idle<40 * 1024>, dim3(N), dim3(threads), 0, 0, iters, jitter, d, valu);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), d, sizeof(Rec) * N, hipMemcpyDeviceToHost));
    unsigned long long a0 = ~0ull, a1 = 0, s1 = 0; double dur = 0;
    for (int i = 0; i < N; ++i) { a0 = std::min(a0, h[i].t0); a1 = std::max(a1, h[i].t1); s1 = std::max(s1, h[i].t0); dur += double(h[i].t1 - h[i].t0); }
    // resident blocks at the moment the first block ends
    unsigned long long e0 = ~0ull; for (int i = 0; i < N; ++i) e0 = std::min(e0, h[i].t1);
    int res = 0; for (int i = 0; i < N; ++i) res += h[i].t0 < e0;
    printf("lds %2d KB, %3d thr, N %5d, %s%s: kernel %.1f us, mean block %.2f us, first-wave residents %d, ideal %.1f us, blocks/us after the first fill %.0f\n",
           lds_kb, threads, N, valu ? "valu" : "sleep", jitter ? "+jitter" : "", (a1 - a0) / 100.0, dur / N / 100.0, res,
           dur / N / 100.0 * ((N + res - 1) / res), (N - res) / std::max(0.01, (s1 - e0) / 100.0));
    return 0;
  };
  for (int valu : {0, 1})
    for (int jitter : {0, 1})
      for (int N : {2048, 4096, 8192, 16384})
        run(1, 256, N, valu ? 10 : 10, jitter, valu);
  run(22, 256, 8192, 10, 1, 0);
  run(22, 256, 8192, 10, 1, 1);
  run(40, 256, 8192, 10, 1, 0);
  run(1, 64, 32768, 10, 1, 0);
  run(1, 256, 16384, 3, 1, 0);
  run(1, 256, 16384, 30, 1, 0);
  return 0;
}
