// rec_gather.hip — how to read-modify-write scattered 64-byte cell records (round 6): every wavefront handles 64 random
// records per hop of a dependent chain, (A) one lane per record: three 16-byte loads + three 16-byte stores per lane (what the
// large-scan update does: 64 different lines per instruction), (B) four lanes per record: one 16-byte load + store per lane
// and four instructions per 64 records (16 lines per instruction, each line read whole).  `waves` wavefronts at once.
//   hipcc --offload-arch=gfx950 -O3 -o rec_gather.bin rec_gather.hip && ./rec_gather.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned mix(unsigned v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; }
template <int MODE>
__global__ void k_rmw(float4* __restrict__ rec, unsigned n_rec, int hops, unsigned long long* out) {
  const unsigned lane = threadIdx.x & 63u;
  unsigned seed = blockIdx.x * 7919u + 17u;
  const unsigned long long t0 = wall_clock64();
  float carry = 0.f;
  for (int k = 0; k < hops; ++k) {
    if (MODE == 0) {  // one lane per record
      const unsigned r = mix(seed + lane * 104729u + unsigned(__float_as_uint(carry) & 1u)) % n_rec;
      float4* p = rec + size_t(r) * 4;
      const float4 a = p[0], b = p[1], c = p[2];
      const float s = a.x + b.y + c.z;
      p[0] = make_float4(s, a.y, a.z, a.w);
      p[1] = make_float4(b.x, s, b.z, b.w);
      p[2] = make_float4(c.x, c.y, s, c.w);
      carry = s;
    } else {  // four lanes per record, four rounds
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned cell = unsigned(q) * 16u + (lane >> 2);  // which of the wavefront's 64 records
        const unsigned r = mix(seed + cell * 104729u + unsigned(__float_as_uint(carry) & 1u)) % n_rec;
        float4* p = rec + size_t(r) * 4 + (lane & 3u);
        float4 v = *p;
        const float s = v.x + v.y;
        v.x = s;
        if ((lane & 3u) != 3u) *p = v;
        acc += s;
      }
      carry = acc;
    }
    seed = mix(seed + unsigned(k));
    carry = __shfl(carry, 0);  // (the next hop's addresses depend on this hop's data)
  }
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
  const unsigned n_rec = 1440000u;  // the 1200 x 1200 map: 92 MB of records
  float4* rec; unsigned long long* d_out;
  CK(hipMalloc(&rec, size_t(n_rec) * 64)); CK(hipMemset(rec, 0, size_t(n_rec) * 64)); CK(hipMalloc(&d_out, 16384 * 8));
  const int hops = 16;
  for (int mode = 0; mode < 2; ++mode)
    for (int w : {256, 3072, 8192}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k_rmw<0>, dim3(w), dim3(64), 0, 0, rec, n_rec, hops, d_out);
        else hipLaunchKernelGGL(k_rmw<1>, dim3(w), dim3(64), 0, 0, rec, n_rec, hops, d_out);
        CK(hipDeviceSynchronize());
      }
      std::vector<unsigned long long> t(w);
      CK(hipMemcpy(t.data(), d_out, w * 8, hipMemcpyDeviceToHost));
      std::sort(t.begin(), t.end());
      printf("{\"mode\": \"%s\", \"waves\": %d, \"ns_per_hop_of_64_records_p50\": %.0f, \"max\": %.0f}\n",
             mode ? "four lanes per record" : "one lane per record", w, t[w / 2] * 10.0 / hops, t[w - 1] * 10.0 / hops);
    }
  return 0;
}
