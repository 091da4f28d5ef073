#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float v[2040]; };  // 8160 B
__global__ void k(Big b, float* out) { out[threadIdx.x] = b.v[threadIdx.x] + b.v[2039 - threadIdx.x]; }
int main() {
  Big b; for (int i = 0; i < 2040; ++i) b.v[i] = float(i);
  float* d; if (hipMalloc(&d, 256 * 4) != hipSuccess) { printf("no device\n"); return 0; }
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize();
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%s %g %g\n", hipGetErrorString(e), h[0], h[255]);
  return 0;
}
