// Cost of a grid-wide barrier (all blocks co-resident) against the per-kernel floor of a queue.
// A sense-reversing counter barrier: one agent-scope atomic per block to arrive, the last one flips a generation
// word, everybody polls it with agent-scope loads.  hipcc --offload-arch=gfx950 -O2 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
// variant: fire-and-forget arrive, everybody polls the counter itself
__global__ void k_barriers_nr(unsigned* cnt, unsigned* gen, int rounds, unsigned* sink) {
  unsigned g = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      (void)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = gridDim.x * unsigned(r + 1);
      while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {}
      g = r + 1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *sink = g;
  (void)gen;
}
__global__ void k_barriers(unsigned* cnt, unsigned* gen, int rounds, unsigned* sink) {
  unsigned g = 0;
  for (int r = 0; r < rounds; ++r) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned arrived = atomicAdd(cnt, 1u) + 1u;
      if (arrived == gridDim.x * unsigned(r + 1)) {
        __hip_atomic_store(gen, unsigned(r + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < unsigned(r + 1)) {}
      }
      g = r + 1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *sink = g;
}
int main() {
  unsigned *cnt, *gen, *sink;
  CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&gen, 4)); CK(hipMalloc(&sink, 4));
  for (int blocks : {64, 128, 256, 512}) {
    for (int rounds : {1, 1001}) {
      CK(hipMemset(cnt, 0, 4)); CK(hipMemset(gen, 0, 4));
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(256), 0, 0, cnt, gen, rounds, sink);
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("blocks %d rounds %d: %.1f us total\n", blocks, rounds, us);
      CK(hipMemset(cnt, 0, 4));
      CK(hipDeviceSynchronize());
      const auto t1 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_barriers_nr, dim3(blocks), dim3(256), 0, 0, cnt, gen, rounds, sink);
      CK(hipDeviceSynchronize());
      const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
      printf("blocks %d rounds %d: %.1f us total (non-returning arrive)\n", blocks, rounds, us1);
    }
  }
  return 0;
}
