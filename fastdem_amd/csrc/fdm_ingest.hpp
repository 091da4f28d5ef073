// fdm_ingest.hpp — sensor_msgs/PointCloud2 byte blob -> SoA channels in HBM.
// gfx950 only.
//
// Reference being reproduced: fastdem/lib/nanoPCL/include/nanopcl/bridge/ros/impl.hpp
//   :41-100 field offsets, :104-118 readIntensity (UINT8 / UINT16 / FLOAT32 / FLOAT64, else 0),
//   :163-171 readRgb (0x00RRGGBB from the packed float), :174-246 from_impl — points with a
//   non-finite coordinate are dropped, order kept.
// The reference decodes on the host into an AoS cloud that integrate() then copies twice; here
// the raw message bytes go over PCIe once and are decoded + compacted on the device
// (count per block -> scan -> scatter, so the order is the message's), straight into the SoA
// channels the bin kernel reads.
#pragma once

#include "fdm_device.hpp"

namespace fdm {

struct IngestLayout {
  unsigned point_step;
  int off_x, off_y, off_z;
  int off_intensity, intensity_type;  // PointField datatype code: 2 UINT8, 4 UINT16, 7 FLOAT32, 8 FLOAT64
  int off_rgb;
  int aligned;                        // every 4-byte field sits on a 4-byte boundary of the blob
};

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p, bool aligned) {
  if (aligned) return *reinterpret_cast<const uint32_t*>(p);
  return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24);
}

struct IngestPoint {
  bool valid;
  float x, y, z;
};
__device__ __forceinline__ IngestPoint ingest_xyz(const uint8_t* __restrict__ blob, const IngestLayout& L,
                                                  unsigned long long i, unsigned long long n) {
  IngestPoint p;
  p.valid = false;
  p.x = p.y = p.z = 0.f;
  if (i >= n) return p;
  const uint8_t* pt = blob + i * L.point_step;
  p.x = __uint_as_float(load_u32(pt + L.off_x, L.aligned));
  p.y = __uint_as_float(load_u32(pt + L.off_y, L.aligned));
  p.z = __uint_as_float(load_u32(pt + L.off_z, L.aligned));
  p.valid = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
  return p;
}

inline __global__ __launch_bounds__(256) void k_ingest_count(const uint8_t* __restrict__ blob, const IngestLayout L,
                                                      unsigned long long n, uint32_t* __restrict__ counts) {
  __shared__ unsigned s_w[4];
  const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
  const IngestPoint p = ingest_xyz(blob, L, i, n);
  const unsigned long long m = __ballot(p.valid);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = unsigned(__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

inline __global__ __launch_bounds__(256) void k_ingest_write(const uint8_t* __restrict__ blob, const IngestLayout L,
                                                      unsigned long long n,
                                                      const uint32_t* __restrict__ offsets,
                                                      float* __restrict__ ox, float* __restrict__ oy,
                                                      float* __restrict__ oz, float* __restrict__ oint,
                                                      uint32_t* __restrict__ orgb) {
  __shared__ unsigned s_w[4];
  const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
  const IngestPoint p = ingest_xyz(blob, L, i, n);
  const unsigned long long m = __ballot(p.valid);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) s_w[w] = unsigned(__popcll(m));
  __syncthreads();
  if (!p.valid) return;
  unsigned rank = unsigned(__popcll(m & ((1ull << lane) - 1ull)));
  for (int q = 0; q < w; ++q) rank += s_w[q];
  const size_t o = size_t(offsets[blockIdx.x]) + rank;
  ox[o] = p.x;
  oy[o] = p.y;
  oz[o] = p.z;
  const uint8_t* pt = blob + i * L.point_step;
  if (oint) {
    const uint8_t* q = pt + L.off_intensity;
    float v = 0.0f;
    if (L.intensity_type == 2) {
      v = float(q[0]);
    } else if (L.intensity_type == 4) {
      v = float(unsigned(q[0]) | (unsigned(q[1]) << 8));
    } else if (L.intensity_type == 7) {
      v = __uint_as_float(load_u32(q, L.aligned));
    } else if (L.intensity_type == 8) {
      const unsigned long long lo = load_u32(q, L.aligned), hi = load_u32(q + 4, L.aligned);
      v = static_cast<float>(__longlong_as_double((long long)(lo | (hi << 32))));
    }
    oint[o] = v;
  }
  if (orgb) orgb[o] = load_u32(pt + L.off_rgb, L.aligned) & 0x00FFFFFFu;
}

// One-pass variant for fdm_engine_integrate_cloud2: decode in place (output index = message index),
// no compaction.  Dropping the non-finite points is left to the bin kernel (ScanParams::
// drop_nonfinite) — an order-preserving compaction changes no min / first / last decision, only
// the indices — so the host needs neither the kept count nor a sync before it can size the scan's
// launch.  counts[block] = finite points of the block (summed by k_collect_stats for n_input).
inline __global__ __launch_bounds__(256) void k_ingest_soa(const uint8_t* __restrict__ blob, const IngestLayout L,
                                                    unsigned long long n, uint32_t* __restrict__ counts,
                                                    float* __restrict__ ox, float* __restrict__ oy,
                                                    float* __restrict__ oz, float* __restrict__ oint,
                                                    uint32_t* __restrict__ orgb) {
  __shared__ unsigned s_w[4];
  const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
  const IngestPoint p = ingest_xyz(blob, L, i, n);
  const unsigned long long m = __ballot(p.valid);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = unsigned(__popcll(m));
  if (i < n) {
    ox[i] = p.x;
    oy[i] = p.y;
    oz[i] = p.z;
    const uint8_t* pt = blob + i * L.point_step;
    if (oint) {
      const uint8_t* q = pt + L.off_intensity;
      float v = 0.0f;
      if (L.intensity_type == 2) {
        v = float(q[0]);
      } else if (L.intensity_type == 4) {
        v = float(unsigned(q[0]) | (unsigned(q[1]) << 8));
      } else if (L.intensity_type == 7) {
        v = __uint_as_float(load_u32(q, L.aligned));
      } else if (L.intensity_type == 8) {
        const unsigned long long lo = load_u32(q, L.aligned), hi = load_u32(q + 4, L.aligned);
        v = static_cast<float>(__longlong_as_double((long long)(lo | (hi << 32))));
      }
      oint[i] = v;
    }
    if (orgb) orgb[i] = load_u32(pt + L.off_rgb, L.aligned) & 0x00FFFFFFu;
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

}  // namespace fdm
