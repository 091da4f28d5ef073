// How many scattered memory requests per second does the chip sustain?  (round 5: the large-scan launch moves ~105 MB
// as ~1.1 M memory-side requests; if scattered 64-byte accesses run at tens of G/s, the REQUEST count is its roofline.)
// Each lane group of 4 / 8 lanes reads or writes one random 64 B / 128 B granule of a 256 MiB region (> L2, < MALL)
// and of a 2 GiB region (> MALL).   hipcc --offload-arch=gfx950 -O3 req_rate.hip -o req_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// LANES lanes share one granule of LANES * 16 bytes
template <int LANES, int MODE>  // MODE 0 read, 1 write, 2 read-modify-write
__global__ __launch_bounds__(256) void k_scatter(uint4* __restrict__ buf, uint32_t granules, uint32_t per_thread, uint32_t seed, uint4* out) {
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t group = gid / LANES, sub = gid % LANES;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (uint32_t i = 0; i < per_thread; ++i) {
    const uint32_t g = mix(group * 7919u + i * 104729u + seed) % granules;
    uint4* p = buf + size_t(g) * LANES + sub;
    if (MODE == 0) { const uint4 v = *p; acc.x ^= v.x; acc.y += v.y; }
    else if (MODE == 1) { *p = make_uint4(gid, i, seed, 1u); }
    else { uint4 v = *p; v.x += 1u; *p = v; }
  }
  if (acc.x == 0x12345678u) out[0] = acc;
}

template <int LANES, int MODE>
double run(uint4* buf, size_t bytes, uint4* out, int blocks, int per_thread) {
  const uint32_t granules = uint32_t(bytes / (size_t(LANES) * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_scatter<LANES, MODE>), dim3(blocks), dim3(256), 0, 0, buf, granules, uint32_t(per_thread), uint32_t(it * 977 + 1), out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  const double req = double(blocks) * 256.0 / LANES * per_thread;
  return req / (double(best) * 1e-3) / 1e9;  // G granule accesses per second
}

int main() {
  uint4 *buf, *out;
  const size_t big = size_t(2) << 30, small = size_t(256) << 20;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, big));
  const int blocks = 256 * 16, per = 16;
  for (size_t bytes : {small, big}) {
    printf("{\"region_MiB\": %zu, \"read64_G\": %.1f, \"write64_G\": %.1f, \"rmw64_G\": %.1f, \"read128_G\": %.1f, \"write128_G\": %.1f, \"read16_G\": %.1f, \"write16_G\": %.1f}\n",
           bytes >> 20, run<4, 0>(buf, bytes, out, blocks, per), run<4, 1>(buf, bytes, out, blocks, per), run<4, 2>(buf, bytes, out, blocks, per),
           run<8, 0>(buf, bytes, out, blocks, per), run<8, 1>(buf, bytes, out, blocks, per), run<1, 0>(buf, bytes, out, blocks, per), run<1, 1>(buf, bytes, out, blocks, per));
  }
  return 0;
}
