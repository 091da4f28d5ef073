// Microbenchmark of gfx950 global atomics (device scope, no return) — informs the k_bin design.
// hipcc --offload-arch=gfx950 -O3 atomics.hip -o /tmp/atomics && /tmp/atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__global__ void k_atom64(unsigned long long* tab, const unsigned* idx, unsigned n) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMin(&tab[idx[i]], (unsigned long long)i);
}
__global__ void k_atom32(unsigned* tab, const unsigned* idx, unsigned n) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax(&tab[idx[i]], i);
}
__global__ void k_store32(unsigned* tab, const unsigned* idx, unsigned n) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tab[idx[i]] = i;
}
__global__ void k_lds(unsigned* out, const unsigned* idx, unsigned n) {
  __shared__ unsigned long long t[4096];
  for (int k = threadIdx.x; k < 4096; k += blockDim.x) t[k] = ~0ull;
  __syncthreads();
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMin(&t[idx[i] & 4095], (unsigned long long)i);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)t[0];
}

int main() {
  const unsigned n = 1u << 22;        // 4M ops
  const unsigned cells = 1440000;
  unsigned long long* tab64; unsigned* tab32; unsigned* d_idx; unsigned* out;
  CK(hipMalloc(&tab64, cells * 8)); CK(hipMalloc(&tab32, cells * 4)); CK(hipMalloc(&d_idx, n * 4)); CK(hipMalloc(&out, (n/256)*4));
  CK(hipMemset(tab64, 0xff, cells * 8)); CK(hipMemset(tab32, 0, cells * 4));
  std::vector<unsigned> idx(n);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, int which) {
    CK(hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice));
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(a));
      if (which == 0) hipLaunchKernelGGL(k_atom64, dim3(n/256), dim3(256), 0, 0, tab64, d_idx, n);
      if (which == 1) hipLaunchKernelGGL(k_atom32, dim3(n/256), dim3(256), 0, 0, tab32, d_idx, n);
      if (which == 2) hipLaunchKernelGGL(k_store32, dim3(n/256), dim3(256), 0, 0, tab32, d_idx, n);
      if (which == 3) hipLaunchKernelGGL(k_lds, dim3(n/256), dim3(256), 0, 0, out, d_idx, n);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("%-44s %8.1f us  %7.1f Gop/s\n", name, best * 1e3, n / (best * 1e-3) / 1e9);
  };
  const char* kinds[4] = {"atomicMin u64", "atomicMax u32", "plain store u32", "LDS atomicMin u64"};
  for (int which = 0; which < 4; ++which) {
    printf("--- %s, %u ops ---\n", kinds[which], n);
    for (unsigned i = 0; i < n; ++i) idx[i] = i % cells;
    run("sequential distinct (coalesced)", which);
    srand(1); for (unsigned i = 0; i < n; ++i) idx[i] = (unsigned)((rand() * 32768ull + rand()) % cells);
    run("random over 1.44M cells", which);
    for (unsigned i = 0; i < n; ++i) idx[i] = (unsigned)((rand() * 32768ull + rand()) % 160000) * 9 % cells;
    run("random over 160K cells (26 ops/cell)", which);
    for (unsigned i = 0; i < n; ++i) idx[i] = (i / 16) % cells;
    run("runs of 16 same cell (adjacent lanes)", which);
    for (unsigned i = 0; i < n; ++i) idx[i] = (i / 4096) % cells;
    run("runs of 4096 same cell (hot cells)", which);
    for (unsigned i = 0; i < n; ++i) idx[i] = (unsigned)((rand() * 32768ull + rand()) % 1024) * 1400;
    run("random over 1024 hot cells (4096 ops/cell)", which);
    for (unsigned i = 0; i < n; ++i) idx[i] = (i % 64 == 0) ? (unsigned)((rand() * 32768ull + rand()) % cells) : 0xffffffffu;
  }
  // sparse: only 1 lane in 64 issues (emulates run tails): reuse kernels with guard via idx<cells
  return 0;
}
