// Do narrower-scope atomics / loads stay in the XCD's L2 on gfx950?  (design question for k_ray / k_bin)
// hipcc --offload-arch=gfx950 -O3 atomic_scope.hip -o atomic_scope.bin && ./atomic_scope.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int SCOPE>
__global__ void k_min(unsigned* tab, const unsigned* idx, unsigned n) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_fetch_min(&tab[idx[i]], i, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__global__ void k_load(const unsigned* tab, const unsigned* idx, unsigned n, unsigned* out) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = 0;
  if (i < n) {
    // a chain of 8 dependent loads, like the DDA batches
    unsigned a = idx[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) a = (__hip_atomic_load(&tab[a], __ATOMIC_RELAXED, SCOPE) + idx[i] + k * 977u) % 1440000u;
    s = a;
  }
  if (s == 0xdeadbeefu) out[0] = s;
}
__global__ void k_xcc(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID[3:0]
}

int main() {
  const unsigned n = 1u << 22, cells = 1440000;
  unsigned *tab, *d_idx, *out;
  CK(hipMalloc(&tab, cells * 4)); CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&out, 4096 * 4));
  std::vector<unsigned> idx(n);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, auto launch) {
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      CK(hipMemset(tab, 0x7f, cells * 4));
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("%-64s %8.1f us  %7.1f Gop/s\n", name, best * 1e3, n / (best * 1e-3) / 1e9);
  };
  for (int pat = 0; pat < 2; ++pat) {
    srand(1);
    for (unsigned i = 0; i < n; ++i) idx[i] = pat == 0 ? (unsigned)((rand() * 32768ull + rand()) % cells) : (i / 16) % cells;
    CK(hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice));
    printf("--- %s ---\n", pat == 0 ? "random over 1.44M cells" : "runs of 16 lanes per cell");
    timeit("atomic min, agent scope", [&] { hipLaunchKernelGGL(k_min<__HIP_MEMORY_SCOPE_AGENT>, dim3(n / 256), dim3(256), 0, 0, tab, d_idx, n); });
    timeit("atomic min, workgroup scope", [&] { hipLaunchKernelGGL(k_min<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(n / 256), dim3(256), 0, 0, tab, d_idx, n); });
    timeit("atomic min, wavefront scope", [&] { hipLaunchKernelGGL(k_min<__HIP_MEMORY_SCOPE_WAVEFRONT>, dim3(n / 256), dim3(256), 0, 0, tab, d_idx, n); });
    timeit("8 dependent loads, agent scope", [&] { hipLaunchKernelGGL(k_load<__HIP_MEMORY_SCOPE_AGENT>, dim3(n / 256), dim3(256), 0, 0, tab, d_idx, n, out); });
    timeit("8 dependent loads, workgroup scope", [&] { hipLaunchKernelGGL(k_load<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(n / 256), dim3(256), 0, 0, tab, d_idx, n, out); });
  }
  hipLaunchKernelGGL(k_xcc, dim3(32), dim3(64), 0, 0, out);
  unsigned h[32];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  printf("XCC id of blocks 0..31:");
  for (int i = 0; i < 32; ++i) printf(" %u", h[i]);
  printf("\n");
  return 0;
}
