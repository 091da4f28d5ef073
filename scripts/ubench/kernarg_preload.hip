// What does reading the kernel arguments cost a block?  4096 blocks, each: first global load (address from a
// pointer ARGUMENT), stamp when the data is back.  Built twice: plain, and with
// -mllvm -amdgpu-kernarg-preload-count=8 (the first arguments arrive in SGPRs with the wavefront).
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=8] kernarg_preload.hip -o kernarg_preload[_on].bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Pad { float v[200]; };
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, unsigned long long* __restrict__ stamp,
                                         int iters, const Pad pad) {
  const unsigned long long t0 = wall_clock64();
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  float v = in[i];                       // address needs `in` only
  v += 1.0f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = wall_clock64();
  float acc = v;
  for (int j = 0; j < iters; ++j) acc = acc * 1.0001f + pad.v[j & 127];
  out[i] = acc;
  if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = t0; stamp[2 * blockIdx.x + 1] = t1 - t0; }
}
int main() {
  const int nb = 8192;
  float *in, *out; unsigned long long* st;
  CK(hipMalloc(&in, nb * 256 * 4)); CK(hipMalloc(&out, nb * 256 * 4)); CK(hipMalloc(&st, nb * 16));
  CK(hipMemset(in, 0, nb * 256 * 4));
  Pad pad; for (auto& x : pad.v) x = 0.5f;
  std::vector<unsigned long long> h(2 * nb);
  for (int iters : {0, 400}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, in, out, st, iters, pad);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), st, nb * 16, hipMemcpyDeviceToHost));
    std::vector<double> d(nb);
    unsigned long long a0 = ~0ull, a1 = 0;
    for (int b = 0; b < nb; ++b) { d[b] = h[2 * b + 1] / 100.0; a0 = std::min(a0, h[2 * b]); a1 = std::max(a1, h[2 * b]); }
    std::sort(d.begin(), d.end());
    printf("iters %d: first load back after %.2f / %.2f / %.2f us (p10 / p50 / p90), blocks start over %.1f us\n", iters,
           d[nb / 10], d[nb / 2], d[nb * 9 / 10], (a1 - a0) / 100.0);
  }
  return 0;
}
