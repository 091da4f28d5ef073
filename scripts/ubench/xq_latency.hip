// Cross-queue dependency latency on one device: two streams ping-pong through events
// (kernel on s1 -> event -> s2 waits -> kernel on s2 -> event -> s1 waits ...).  Prints us per hop for empty
// kernels and for kernels that spin ~20 us (so that the host enqueues ahead and only the device-side hand-over
// is on the critical path).   hipcc --offload-arch=gfx950 -O2 xq_latency.hip -o xq_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int hops = 400;
  hipEvent_t ev[2 * hops];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (long long spin_us : {0LL, 20LL}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int h = 0; h < hops; ++h) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, spin_us * 100);
        CK(hipEventRecord(ev[2 * h], s1));
        CK(hipStreamWaitEvent(s2, ev[2 * h], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, spin_us * 100);
        CK(hipEventRecord(ev[2 * h + 1], s2));
        CK(hipStreamWaitEvent(s1, ev[2 * h + 1], 0));
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      // same-stream reference: 2 * hops kernels back to back on one stream
      const auto t1 = std::chrono::steady_clock::now();
      for (int h = 0; h < 2 * hops; ++h) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, spin_us * 100);
      CK(hipDeviceSynchronize());
      const double us1 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
      printf("spin %lld us: two streams %.2f us per kernel, one stream %.2f us per kernel -> hand-over costs %.2f us\n",
             spin_us, us / (2 * hops), us1 / (2 * hops), (us - us1) / (2 * hops));
    }
  }
  return 0;
}
