// Achievable HBM bandwidth on the box ("verify ... and use the measured figure") —
// read-only sum, copy and triad over buffers far larger than the caches, 16 B per lane.
// hipcc --offload-arch=gfx950 -O3 hbm_bw.hip -o hbm_bw.bin && ./hbm_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n4) {
  float s = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += size_t(gridDim.x) * 256ull) {
    const float4 v = a[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;  // keep the loads
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += size_t(gridDim.x) * 256ull) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(const float4* __restrict__ a, const float4* __restrict__ b,
                                               float4* __restrict__ c, size_t n4) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += size_t(gridDim.x) * 256ull) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + 3.f * y.x, x.y + 3.f * y.y, x.z + 3.f * y.z, x.w + 3.f * y.w);
  }
}

int main() {
  const size_t bytes = size_t(2) << 30;  // 2 GiB per buffer
  const size_t n4 = bytes / 16;
  float4 *a, *b, *c;
  float* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {256 * 8, 256 * 16, 256 * 32};
  for (int g : grids) {
    for (int which = 0; which < 3; ++which) {
      float best = 1e30f;
      for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        if (which == 0) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, out, n4);
        if (which == 1) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n4);
        if (which == 2) hipLaunchKernelGGL(k_triad, dim3(g), dim3(256), 0, 0, a, b, c, n4);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
      }
      const double moved = double(bytes) * (which == 0 ? 1 : which == 1 ? 2 : 3);
      printf("{\"kernel\": \"%s\", \"grid\": %d, \"GiB_per_buffer\": 2, \"ms\": %.3f, \"TBps\": %.2f}\n",
             which == 0 ? "read" : which == 1 ? "copy" : "triad", g, best, moved / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
