// dep_latency.hip — what one DEPENDENT round trip to HBM costs a wavefront (round 6: the update half of the large-scan
// launch is a chain of such trips).  Every lane walks a chain of K random 16-byte loads over a buffer of S bytes;
// `waves` wavefronts (one per block) do so at once.  Prints ns per hop (device clock, median over the waves).
//   hipcc --offload-arch=gfx950 -O3 -o dep_latency.bin dep_latency.hip && ./dep_latency.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_init(uint4* buf, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    buf[i] = make_uint4(unsigned(h % n), unsigned(h >> 32), 0u, 0u);
  }
}
template <bool SAME_LINE>
__global__ void k_chase(const uint4* __restrict__ buf, unsigned n, int hops, unsigned long long* out, unsigned* sink) {
  unsigned lane = threadIdx.x & 63u;
  // SAME_LINE: the 64 lanes read 64 consecutive 16-byte words (a coalesced 1 KB request); else 64 different lines
  unsigned idx = (blockIdx.x * 7919u * 64u + (SAME_LINE ? 0u : lane * 104729u)) % n;
  const unsigned long long t0 = wall_clock64();
  unsigned acc = 0;
  for (int k = 0; k < hops; ++k) {
    const unsigned at = SAME_LINE ? (idx & ~63u) + lane : idx;
    const uint4 v = buf[at % n];
    acc += v.y;
    idx = SAME_LINE ? unsigned(__builtin_amdgcn_readfirstlane(int(v.x))) : v.x;
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc;
}
int main() {
  const size_t sizes[] = {size_t(64) << 20, size_t(1) << 30, size_t(4) << 30};
  const int waves_list[] = {256, 3072, 12288};
  const int hops = 32;
  unsigned long long* d_out; unsigned* d_sink;
  CK(hipMalloc(&d_out, 16384 * 8)); CK(hipMalloc(&d_sink, 4));
  for (size_t S : sizes) {
    uint4* buf; CK(hipMalloc(&buf, S));
    const size_t n = S / 16;
    hipLaunchKernelGGL(k_init, dim3(4096), dim3(256), 0, 0, buf, n);
    CK(hipDeviceSynchronize());
    for (int same = 0; same < 2; ++same)
      for (int w : waves_list) {
        for (int rep = 0; rep < 2; ++rep) {
          if (same) hipLaunchKernelGGL(k_chase<true>, dim3(w), dim3(64), 0, 0, buf, unsigned(n), hops, d_out, d_sink);
          else hipLaunchKernelGGL(k_chase<false>, dim3(w), dim3(64), 0, 0, buf, unsigned(n), hops, d_out, d_sink);
          CK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> t(w);
        CK(hipMemcpy(t.data(), d_out, w * 8, hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end());
        printf("{\"buffer_MB\": %zu, \"pattern\": \"%s\", \"waves\": %d, \"ns_per_hop_p50\": %.0f, \"p90\": %.0f, \"max\": %.0f}\n", S >> 20,
               same ? "coalesced 1 KB" : "64 lines", w, t[w / 2] * 10.0 / hops, t[w * 9 / 10] * 10.0 / hops, t[w - 1] * 10.0 / hops);
      }
    CK(hipFree(buf));
  }
  return 0;
}
