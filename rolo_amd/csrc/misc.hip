// Small layout / utility kernels: strided PCL records -> float4, covariance SoA <-> Matrix4d, and the float
// pcl::transformPointCloud used at reference src/lidarOdometry.cpp:459,492 and lsq_registration_impl.hpp:78,178.
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include <cfloat>

namespace rolo {

namespace {

// strided records -> float4 (x, y, z, 1); the same pass leaves the workgroup's partial bounding box (what the neighbour search's key kernel
// folds — a separate bbox launch re-read the cloud for it)
__global__ __launch_bounds__(256) void pack_xyz_kernel(const float* __restrict__ in, int stride, float4* __restrict__ out, int n, int* __restrict__ bbox_part) {
  ROLO_ALL_KERNEL_PRIO();
  __shared__ float smn[4][3], smx[4][3];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < n) {
    const float* p = in + (size_t)i * stride;
    const float x = p[0], y = p[1], z = p[2];
    out[i] = make_float4(x, y, z, 1.0f);  // pcl::PointXYZI data[3] = 1
    mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
  }
  if (!bbox_part) return;   // uniform
#pragma unroll
  for (int d = 0; d < 3; d++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64)); }
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; d++) { smn[wv][d] = mn[d]; smx[wv][d] = mx[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    bbox_part[blockIdx.x * 6 + d] = f2ord(fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d])));
    bbox_part[blockIdx.x * 6 + 3 + d] = f2ord(fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d])));
  }
}

__global__ __launch_bounds__(256) void cov_unpack_kernel(const double* __restrict__ soa, int n, double* __restrict__ m16) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t p = (size_t)n;
  const double xx = soa[i], xy = soa[p + i], xz = soa[2 * p + i], yy = soa[3 * p + i], yz = soa[4 * p + i], zz = soa[5 * p + i];
  double* o = m16 + (size_t)i * 16;
  o[0] = xx; o[1] = xy; o[2] = xz; o[3] = 0;
  o[4] = xy; o[5] = yy; o[6] = yz; o[7] = 0;
  o[8] = xz; o[9] = yz; o[10] = zz; o[11] = 0;
  o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 0;
}

__global__ __launch_bounds__(256) void cov_pack_kernel(const double* __restrict__ m16, int n, double* __restrict__ soa) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t p = (size_t)n;
  const double* o = m16 + (size_t)i * 16;
  soa[i] = o[0]; soa[p + i] = 0.5 * (o[1] + o[4]); soa[2 * p + i] = 0.5 * (o[2] + o[8]);
  soa[3 * p + i] = o[5]; soa[4 * p + i] = 0.5 * (o[6] + o[9]); soa[5 * p + i] = o[10];
}

struct Mat4f { float m[16]; };

// PCL >= 1.10 SSE2 Transformer::se3: p0 + (p1 + (p2 + c3)), no FMA, w <- 1, other fields copied
__global__ __launch_bounds__(256) void transform_cloud_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int stride, Mat4f T) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* s = in + (size_t)i * stride;
  float* d = out + (size_t)i * stride;
  const float x = s[0], y = s[1], z = s[2];
  float o[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
    o[r] = __fadd_rn(__fmul_rn(T.m[r * 4], x), __fadd_rn(__fmul_rn(T.m[r * 4 + 1], y), __fadd_rn(__fmul_rn(T.m[r * 4 + 2], z), T.m[r * 4 + 3])));
  for (int k = 3; k < stride; k++) d[k] = s[k];
  d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
  if (stride >= 8) d[3] = 1.0f;   // pcl::PointXYZI (32-byte records): data[3] = 1; n x 4 clouds (x, y, z, intensity) keep slot 3
}

// Both clouds of a registration in ONE launch, the source moved by T on the way (the odometry driver: *Propagated_cloud = T * featureOld,
// src/lidarOdometry.cpp:459 — the same float operations as transform_cloud_kernel, then the pack): a transform launch and a pack launch less per frame.
__global__ __launch_bounds__(256) void pack_pair_kernel(const float* __restrict__ in0, int stride0, float4* __restrict__ out0, int n0, int* __restrict__ bbox0, int split,
                                                       const float* __restrict__ in1, int stride1, float4* __restrict__ out1, int n1, int* __restrict__ bbox1, Mat4f T, int move0) {
  ROLO_ALL_KERNEL_PRIO();
  __shared__ float smn[4][3], smx[4][3];
  const bool second = (int)blockIdx.x >= split;
  const int blk = (int)blockIdx.x - (second ? split : 0);
  const float* in = second ? in1 : in0; const int stride = second ? stride1 : stride0, n = second ? n1 : n0;
  float4* out = second ? out1 : out0; int* bbox_part = second ? bbox1 : bbox0;
  const int i = blk * 256 + threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < n) {
    const float* p = in + (size_t)i * stride;
    float x = p[0], y = p[1], z = p[2];
    if (!second && move0) {
      float o[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
        o[r] = __fadd_rn(__fmul_rn(T.m[r * 4], x), __fadd_rn(__fmul_rn(T.m[r * 4 + 1], y), __fadd_rn(__fmul_rn(T.m[r * 4 + 2], z), T.m[r * 4 + 3])));
      x = o[0]; y = o[1]; z = o[2];
    }
    out[i] = make_float4(x, y, z, 1.0f);
    mn[0] = mx[0] = x; mn[1] = mx[1] = y; mn[2] = mx[2] = z;
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64)); }
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; d++) { smn[wv][d] = mn[d]; smx[wv][d] = mx[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    bbox_part[blk * 6 + d] = f2ord(fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d])));
    bbox_part[blk * 6 + 3 + d] = f2ord(fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d])));
  }
}

}  // namespace

// debug timeline (ROLO_STAMP=1): one thread leaves the 100 MHz wall clock in buf[slot]; enqueue_frame places five of them per frame
__global__ void stamp_kernel(unsigned long long* buf, int slot) { buf[slot] = wall_clock64(); }
hipError_t launch_stamp(unsigned long long* buf, int slot, hipStream_t s) {
  stamp_kernel<<<1, 1, 0, s>>>(buf, slot);
  return hipGetLastError();
}

hipError_t launch_pack_xyz(const float* in, int stride, float4* out, int n, hipStream_t s, int* bbox_part) {
  if (n > 0) pack_xyz_kernel<<<(n + 255) / 256, 256, 0, s>>>(in, stride, out, n, bbox_part);
  return hipGetLastError();
}
hipError_t launch_pack_pair(const float* in0, int stride0, float4* out0, int n0, int* bbox0, const float* T16_host_or_null,
                            const float* in1, int stride1, float4* out1, int n1, int* bbox1, hipStream_t s) {
  Mat4f T{};
  if (T16_host_or_null) for (int i = 0; i < 16; i++) T.m[i] = T16_host_or_null[i];
  const int g0 = (n0 + 255) / 256, g1 = (n1 + 255) / 256;
  if (g0 + g1 > 0) pack_pair_kernel<<<g0 + g1, 256, 0, s>>>(in0, stride0, out0, n0, bbox0, g0, in1, stride1, out1, n1, bbox1, T, T16_host_or_null ? 1 : 0);
  return hipGetLastError();
}
hipError_t launch_cov_unpack(const double* soa, int n, double* m16, hipStream_t s) {
  if (n > 0) cov_unpack_kernel<<<(n + 255) / 256, 256, 0, s>>>(soa, n, m16);
  return hipGetLastError();
}
hipError_t launch_cov_pack(const double* m16, int n, double* soa, hipStream_t s) {
  if (n > 0) cov_pack_kernel<<<(n + 255) / 256, 256, 0, s>>>(m16, n, soa);
  return hipGetLastError();
}
// rolo_debug_chain (api.hip): a launch that does nothing — its dispatch, its kernel boundary (the L2 write-back / invalidate around it) and nothing else
__global__ void empty_kernel() {}
hipError_t launch_empty(int grid, int threads, hipStream_t s) {
  empty_kernel<<<grid, threads, 0, s>>>();
  return hipGetLastError();
}
hipError_t launch_transform_cloud(const float* in, float* out, int n, int stride, const float*, const float* T16_host, hipStream_t s) {
  Mat4f T;
  for (int i = 0; i < 16; i++) T.m[i] = T16_host[i];
  if (n > 0) transform_cloud_kernel<<<(n + 255) / 256, 256, 0, s>>>(in, out, n, stride, T);
  return hipGetLastError();
}

}  // namespace rolo
