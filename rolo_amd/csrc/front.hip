// K1-K4 — the front end on gfx950.
// Replaces ImageProjection::projectPointCloud / cloudExtraction (reference src/imageProjection.cpp:399-505) and
// FeatureExtraction::calculateSmoothness / markOccludedPoints / extractFeatures incl. pcl::VoxelGrid
// (reference src/featureExtraction.cpp:87-266). De-skew (deskewPoint :368-396, off in every shipped config) is applied when
// armed with rolo_front_set_deskew: per-point times from the caller, from the message, or interpolated from the azimuth
// (deskewCloudInfo :270-327) when the cloud carries none.
//
// MI355X design. The reference's three serial loops become:
//   K1  one thread per raw point; "first point to claim a pixel wins" (imageProjection.cpp:451) is an atomicMin of
//       the raw point index on a per-pixel owner word — the smallest index is exactly the serial winner;
//   K2  row-major stream compaction: one workgroup per ring does an LDS prefix scan of its valid pixels, a second
//       pass adds the ring offsets and scatters (x,y,z,intensity), column and range — coalesced on both sides;
//   K3  a stencil kernel (11-tap float sum in the reference's written order) plus a mark kernel — the marks only
//       ever store 1 and depend only on ranges / columns, so the serial loop is order-free;
//   K4  one workgroup of 1024 threads per ring, everything for the ring staged in LDS: all six sectors in one segmented bitonic sort of
//       packed (curvature bits << 32 | index) keys; the greedy corner / surface picks — a serial walk in the reference — as the equivalent
//       fixed point "picked iff no better-ranked candidate that reaches it is picked", every lane owning a position, a few rounds per
//       sector (only the two sectors touching the cloud ends keep the serial walk); the label<=0 collection is a ballot
//       compaction, and pcl::VoxelGrid is a second bitonic sort of (cell << 32 | order) keys followed by per-run centroids.
// fp32 arithmetic follows the reference expression by expression; this file is compiled with -ffp-contract=off.
#include "rolo_internal.hpp"
#include <atomic>
#include "dev_math.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

namespace rolo {

constexpr int FRONT_GUARD = 8;       // guard cells in front of / behind the per-point arrays (reference reads index -1.. )
constexpr int FRONT_MAX_H = 2048;      // Horizon_SCAN up to here: one ring's working set lives in LDS (shipped configs: 1024, 1800, 2048)
constexpr int FRONT_MAX_H_BIG = 4096;  // ... and up to here in a per-ring scratch area in HBM (the same kernel through global pointers: correct, several times slower;
                                       // no sensor the reference accepts has that many columns — its params.yaml mentions a Livox Horizon at 4000)
// bitonic capacity (elements) for a Horizon_SCAN limit of maxh: 6 sector segments of <= maxh / 4, which also holds one ring's surface scan (<= maxh)
constexpr int front_sort_cap(int maxh) { return 6 * (maxh / 4); }
constexpr size_t front_ring_bytes(int maxh) { return sizeof(unsigned long long) * (size_t)front_sort_cap(maxh) + sizeof(int) * (size_t)(maxh + 32) * 8 + sizeof(int) * (size_t)(maxh + 16); }
constexpr int ST_OUT = 0, ST_UNDECIDED = 1, ST_PICKED = 2;

namespace {

// ---- sensor_msgs/PointCloud2 payload -> x, y, z / ring / per-point time -----------------------------------------------
// What pcl::moveFromROSMsg + the Ouster conversion loop do on the host in ImageProjection::cachePointCloud
// (imageProjection.cpp:188-212): fields are found by their byte offsets in the message, Velodyne "time" is a float in
// seconds, Ouster "t" a uint32 in nanoseconds (dst.time = src.t * 1e-9f), ring is uint16 (Velodyne) or uint8 (Ouster).
// Byte-wise loads: a point_step of 22 (the velodyne driver's packed layout) has no alignment to offer.
ROLO_DEV unsigned load_u32_bytes(const unsigned char* p) { return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24); }

__global__ __launch_bounds__(256) void unpack_cloud_kernel(const unsigned char* __restrict__ data, rolo_cloud_layout L, int n, float* __restrict__ xyz,
                                                          unsigned short* __restrict__ ring, float* __restrict__ rel_time) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned char* p = data + (size_t)i * L.point_step;
  xyz[3 * (size_t)i] = __uint_as_float(load_u32_bytes(p + L.off_x));
  xyz[3 * (size_t)i + 1] = __uint_as_float(load_u32_bytes(p + L.off_y));
  xyz[3 * (size_t)i + 2] = __uint_as_float(load_u32_bytes(p + L.off_z));
  ring[i] = L.ring_bytes == 1 ? (unsigned short)p[L.off_ring] : (unsigned short)(p[L.off_ring] | (p[L.off_ring + 1] << 8));
  float t = 0.f;
  if (L.time_kind == 1) t = __uint_as_float(load_u32_bytes(p + L.off_time));
  else if (L.time_kind == 2) t = (float)load_u32_bytes(p + L.off_time) * 1e-9f;   // :209
  rel_time[i] = fabsf(t);                                                          // deskewCloudInfo :358-359
}

// ---- per-point times of a cloud without a time field: deskewCloudInfo, timeFlag == -1 (imageProjection.cpp:270-327) ----
// The reference interpolates the time from the azimuth in a serial loop with one flag, halfPassed, that flips at the first
// point whose (first-half-adjusted) azimuth is more than pi past the start: a prefix property. Kernel 1 finds that point
// (atomicMin), kernel 2 applies the first-half rule up to and including it and the second-half rule after it.
struct AzimuthRef { float start, end, diff; };
ROLO_DEV AzimuthRef azimuth_ref(const float* __restrict__ pts, int stride, int n) {
  float start = -atan2f(pts[1], pts[0]);                                                           // :272
  float end = (float)((double)(-atan2f(pts[(size_t)(n - 1) * stride + 1], pts[(size_t)(n - 1) * stride])) + 2 * M_PI);   // :273
  if ((double)(end - start) > 3 * M_PI) end = (float)((double)end - 2 * M_PI);                     // :274-277
  else if ((double)(end - start) < M_PI) end = (float)((double)end + 2 * M_PI);
  return AzimuthRef{start, end, end - start};
}
ROLO_DEV float azimuth_first_half(float ori, float start) {
  if ((double)ori < (double)start - M_PI / 2) ori = (float)((double)ori + 2 * M_PI);
  else if ((double)ori > (double)start + M_PI * 3 / 2) ori = (float)((double)ori - 2 * M_PI);
  return ori;
}
__global__ __launch_bounds__(256) void azimuth_flag_kernel(const float* __restrict__ pts, int stride, int n, int* __restrict__ first_passed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AzimuthRef r = azimuth_ref(pts, stride, n);
  const float a = azimuth_first_half(-atan2f(pts[(size_t)i * stride + 1], pts[(size_t)i * stride]), r.start);   // point.x = in.y, point.z = in.x (:303-307)
  if ((double)(a - r.start) > M_PI) atomicMin(first_passed, i);
}
__global__ __launch_bounds__(256) void azimuth_time_kernel(const float* __restrict__ pts, int stride, int n, float scan_period,
                                                          const int* __restrict__ first_passed, float* __restrict__ rel_time) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AzimuthRef r = azimuth_ref(pts, stride, n);
  float ori = -atan2f(pts[(size_t)i * stride + 1], pts[(size_t)i * stride]);
  if (i <= *first_passed) {
    ori = azimuth_first_half(ori, r.start);
  } else {                                                                                          // :316-322
    ori = (float)((double)ori + 2 * M_PI);
    if ((double)ori < (double)r.end - M_PI * 3 / 2) ori = (float)((double)ori + 2 * M_PI);
    else if ((double)ori > (double)r.end + M_PI / 2) ori = (float)((double)ori - 2 * M_PI);
  }
  const float relTime = (ori - r.start) / r.diff;                                                   // :324
  rel_time[i] = scan_period * relTime;                                                              // :325
}

// ---- K1 ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ pts, int stride, const unsigned short* __restrict__ ring,
                                                     int n_raw, rolo_front_params P, int* __restrict__ owner) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_raw) return;
  const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
  const float range = sqrtf(x * x + y * y + z * z);  // utility.h:462-465
  if (range < P.lidar_min_range || range > P.lidar_max_range) return;
  // NaN coordinates (the node refuses clouds that are not is_dense, :226-231): the reference's column index is then the int
  // conversion of NaN — INT_MIN on x86, i.e. the point is dropped; the device conversion would give 0
  if (range != range) return;
  const int rowIdn = ring[i];
  if (rowIdn < 0 || rowIdn >= P.n_scan) return;
  if (rowIdn % P.downsample_rate != 0) return;
  const int H = P.horizon_scan;
  const float horizonAngle = atan2f(x, y) * 180 / M_PI;           // imageProjection.cpp:437
  const float ang_res_x = 360.0 / float(H);                       // :438
  int columnIdn = -round((horizonAngle - 90.0) / ang_res_x) + H / 2;  // :440
  if (columnIdn >= H) columnIdn -= H;
  if (columnIdn < 0 || columnIdn >= H) return;
  atomicMin(&owner[rowIdn * H + columnIdn], i);  // first point wins (:451)
}

// ---- K2 pass A: per ring, local index of every valid pixel + ring count ----
__global__ __launch_bounds__(256) void ring_scan_kernel(const int* __restrict__ owner, int H, int* __restrict__ local_idx, int* __restrict__ ring_count) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < H; base += 256) {
    const int j = base + t;
    const int v = (j < H && owner[row * H + j] != INT_MAX) ? 1 : 0;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; w++) woff += wsum[w];
    const int c = carry;
    if (j < H) local_idx[row * H + j] = v ? (c + woff + inc - 1) : -1;
    __syncthreads();
    if (t == 255) carry = c + woff + inc;
    __syncthreads();
  }
  if (t == 0) ring_count[row] = carry;
}

// ---- K2 pass B: ring offsets, start/end ring index, scatter ----
// ImageProjection::deskewPoint (imageProjection.cpp:368-396), rotation only, for clouds with a per-point time:
// rel_time[i] = fabs(point.time) (what deskewCloudInfo :358-359 leaves in deskewCloud->points[i].intensity)
struct DeskewArgs { const float* rel_time; float incre_r, incre_p, incre_y, scan_period; double odom_time_diff; };

ROLO_DEV void deskew_point(const DeskewArgs& d, float rel_time_f, float& x, float& y, float& z) {
  const double relTime = (double)rel_time_f;
  const float ratio = relTime / d.scan_period;                     // :380 double / float, stored float
  const float s1 = (float)(d.scan_period / d.odom_time_diff);      // :383 float / double; Eigen casts the scalar to float
  const float roll = -(d.incre_r * s1 * ratio), pitch = -(d.incre_p * s1 * ratio), yaw = -(d.incre_y * s1 * ratio);
  // pcl::getTransformation(0, 0, 0, roll, pitch, yaw), float (:386)
  const float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch), E = cosf(roll), F = sinf(roll), DE = D * E, DF = D * F;
  const float nx = (A * C) * x + (A * DF - B * E) * y + (B * F + A * DE) * z + 0.f;   // :389-391
  const float ny = (B * C) * x + (A * E + B * DF) * y + (B * DE - A * F) * z + 0.f;
  const float nz = (-D) * x + (C * F) * y + (C * E) * z + 0.f;
  x = nx; y = ny; z = nz;
}

__global__ __launch_bounds__(256) void ring_scatter_kernel(const float* __restrict__ pts, int stride, const unsigned short* __restrict__ ring,
                                                          const int* __restrict__ owner, const int* __restrict__ local_idx,
                                                          const int* __restrict__ ring_count, int n_scan, int H, float4* __restrict__ extracted,
                                                          int* __restrict__ col_ind, float* __restrict__ point_range, int* __restrict__ start_ring,
                                                          int* __restrict__ end_ring, float* __restrict__ range_mat, int* __restrict__ n_valid,
                                                          DeskewArgs dsk) {
  __shared__ int s_off;
  const int row = blockIdx.x, t = threadIdx.x;
  if (t == 0) {
    int off = 0;
    for (int r = 0; r < row; r++) off += ring_count[r];
    s_off = off;
    start_ring[row] = off - 1 + 5;                     // imageProjection.cpp:485
    end_ring[row] = off + ring_count[row] - 1 - 5;     // :503
    if (row == n_scan - 1) *n_valid = off + ring_count[row];
  }
  __syncthreads();
  const int off = s_off;
  for (int j = t; j < H; j += 256) {
    const int o = owner[row * H + j];
    float rng = FLT_MAX;
    if (o != INT_MAX) {
      const float x = pts[(size_t)o * stride], y = pts[(size_t)o * stride + 1], z = pts[(size_t)o * stride + 2];
      rng = sqrtf(x * x + y * y + z * z);
      const int dst = off + local_idx[row * H + j];
      float sx = x, sy = y, sz = z;
      if (dsk.rel_time) deskew_point(dsk, dsk.rel_time[o], sx, sy, sz);  // :454 — range and pixel come from the raw point
      extracted[dst] = make_float4(sx, sy, sz, ring[o] * z);  // intensity <- ring * z (:410)
      col_ind[dst] = j;
      point_range[dst] = rng;
    }
    if (range_mat) range_mat[row * H + j] = rng;
  }
}

// ---- K3 (calculateSmoothness, markOccludedPoints) lives in the window load of extract_kernel below ----

// ---- K4 ----
// Ascending bitonic sort of every aligned `seg`-element segment of key[0, total) in LDS (seg a power of two, total a
// multiple of seg; NT threads). Exchange distances >= 64 are block-wide steps with a barrier each; the distances
// 32..1 that close every merge stage stay inside an aligned 64-element chunk, so one wavefront takes the chunk into
// registers and finishes the stage with cross-lane exchanges (DPP / permlane swaps) — one barrier per stage instead of
// one per step.
template <int J2>
ROLO_DEV void bitonic_lane_step(unsigned long long& v, int lane, bool up) {
  const unsigned long long o = lane_xor_u64<J2>(v);   // DPP / v_permlane*_swap: no LDS crossbar (dev_math.hpp)
  const bool take_min = ((lane & J2) == 0) == up;     // the lane with the smaller index of the pair keeps the minimum when ascending
  v = take_min ? (o < v ? o : v) : (o > v ? o : v);
}
template <int NT>
ROLO_DEV void bitonic_sort_lds_seg(unsigned long long* key, int total, int seg) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n_chunks = (total + 63) >> 6;
  for (int k = 2; k <= seg; k <<= 1) {
    int jj = k >> 1;
    for (; jj >= 64; jj >>= 1) {
      for (int i = t; i < total; i += NT) {
        const int ixj = i ^ jj;
        if (ixj > i) {
          const unsigned long long a = key[i], b = key[ixj];
          const bool up = ((i & (seg - 1)) & k) == 0;
          if ((a > b) == up) { key[i] = b; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
    for (int c = wv; c < n_chunks; c += NT / 64) {
      const int i = (c << 6) + lane;
      unsigned long long v = i < total ? key[i] : ~0ull;
      const bool up = ((i & (seg - 1)) & k) == 0;
      if (jj >= 32) bitonic_lane_step<32>(v, lane, up);
      if (jj >= 16) bitonic_lane_step<16>(v, lane, up);
      if (jj >= 8) bitonic_lane_step<8>(v, lane, up);
      if (jj >= 4) bitonic_lane_step<4>(v, lane, up);
      if (jj >= 2) bitonic_lane_step<2>(v, lane, up);
      bitonic_lane_step<1>(v, lane, up);
      if (i < total) key[i] = v;
    }
    __syncthreads();
  }
}
template <int NT>
ROLO_DEV void bitonic_sort_lds(unsigned long long* key, int n_pow2) { bitonic_sort_lds_seg<NT>(key, n_pow2, n_pow2); }

struct FeatArgs {
  const float4* extracted; const int* col; const float* range; float* curv; int* picked; int* label;  // global, guard-offset pointers
  const int* start_ring; const int* end_ring;
  const int* n_ptr; int n_scan; float edge_threshold, surf_threshold, leaf;
  float4* corner_stage; int* corner_cnt;  // [n_scan][6][20], [n_scan][6]
  float4* surf_stage; int* surf_cnt;      // [n_scan][MAXH], [n_scan]
  unsigned char* big;                     // Horizon_SCAN > 2048: front_ring_bytes(MAXH) of scratch per ring
};

// One workgroup of XT = 1024 threads per ring: 128 rings are only 128 workgroups, so the parallelism has to come from inside —
// 16 wavefronts (4 per SIMD) hide the LDS / cross-lane latency of the sorts that 4 wavefronts (1 per SIMD) exposed.
#ifndef ROLO_XT
#define ROLO_XT 1024
#endif
constexpr int XT = ROLO_XT, XW = XT / 64;   // threads, wavefronts
#ifdef ROLO_XT_STATS
__device__ unsigned long long g_xt[128][8];   // per ring: phase time stamps of extract_kernel (shader clock)
#define XT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 128) g_xt[blockIdx.x][k] = clock64(); } while (0)
extern "C" int rolo_debug_extract_times(unsigned long long* out /* 128 x 8 */) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xt), sizeof(unsigned long long) * 128 * 8) == hipSuccess ? 0 : -1;
}
#else
#define XT_STAMP(k)
#endif
template <int MAXH, bool GLOBAL>
__global__ __launch_bounds__(XT) void extract_kernel(FeatArgs A) {
  extern __shared__ unsigned char smem_lds[];
  constexpr int SORT_CAP = front_sort_cap(MAXH), SEGMAX = MAXH / 4, UP = (SEGMAX + XT - 1) / XT;   // UP: sector positions per thread
  unsigned char* smem_raw = GLOBAL ? A.big + (size_t)blockIdx.x * front_ring_bytes(MAXH) : smem_lds;
  // layout: keys[SORT_CAP] u64 | l_picked | l_col | l_label | l_curv (ring window) | list[MAXH+16] | l_brk | l_rank | l_stat | l_reach (window)
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);
  const int WIN = MAXH + 2 * 16;
  int* l_picked = reinterpret_cast<int*>(keys + SORT_CAP);
  int* l_col = l_picked + WIN;
  int* l_label = l_col + WIN;
  float* l_curv = reinterpret_cast<float*>(l_label + WIN);
  int* list = reinterpret_cast<int*>(l_curv + WIN);
  int* l_brk = list + (MAXH + 16);
  int* l_rank = l_brk + WIN;
  int* l_stat = l_rank + WIN;
  int* l_reach = l_stat + WIN;   // how far a pick's suppression marks go from this cell: forward | backward << 4 (0..5 each)
  __shared__ int s_heads, s_ep;
  __shared__ int s_pk[UP][XW], s_cw[2][XW];
  __shared__ float s_red[2][3][XW];
  __shared__ int s_wsum[XW];

  const int ring = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  XT_STAMP(0);
  const int s = A.start_ring[ring], e = A.end_ring[ring];
  const int n = *A.n_ptr;
  // ring window in LDS: global indices [w0, w0 + wlen)
  const int w0 = s - 16;
  int wlen = (e - s) + 32 + 1;
  if (wlen < 0) wlen = 0;
  if (wlen > WIN) wlen = WIN;  // guarded by the host (ring population <= MAXH)
  // K3 inside this kernel (was: smoothness_kernel + occlusion_kernel, two launches on the frame's critical path): curvature
  // (calculateSmoothness, featureExtraction.cpp:87-110) and the occlusion / parallel-beam marks (markOccludedPoints, :112-149) of the window's
  // cells straight from range / col. The reference SCATTERS the marks (point j marks j-5..j or j+1..j+6); here every cell GATHERS the three
  // conditions of the points that could mark it — the same set, no write shared between workgroups.
  {
    int* l_flag = reinterpret_cast<int*>(keys);   // scratch (the sector sort fills keys later): conditions of the points [w0 - 6, w0 + wlen + 6)
    for (int i = t; i < wlen + 12; i += XT) {
      const int j = w0 - 6 + i;
      int fl = 0;
      if (j >= 5 && j < n - 6) {   // :116
        const float depth1 = A.range[j], depth2 = A.range[j + 1];
        const int columnDiff = abs(A.col[j + 1] - A.col[j]);
        if (columnDiff < 10) {
          if (depth1 - depth2 > 0.3) fl |= 1;        // marks j-5 .. j
          else if (depth2 - depth1 > 0.3) fl |= 2;   // marks j+1 .. j+6
        }
        const float diff1 = fabsf(float(A.range[j - 1] - A.range[j]));
        const float diff2 = fabsf(float(A.range[j + 1] - A.range[j]));
        if (diff1 > 0.02 * A.range[j] && diff2 > 0.02 * A.range[j]) fl |= 4;   // marks j
      }
      l_flag[i] = fl;
    }
    __syncthreads();
    for (int i = t; i < wlen; i += XT) {
      const int gi = w0 + i;
      const bool in = gi >= -FRONT_GUARD && gi < n + FRONT_GUARD;
      float c = 0.f;
      if (gi >= 5 && gi < n - 5) {  // :91-99, float sum in the written order
        const float* range = A.range;
        const float diffRange = range[gi - 5] + range[gi - 4] + range[gi - 3] + range[gi - 2] + range[gi - 1] - range[gi] * 10
                              + range[gi + 1] + range[gi + 2] + range[gi + 3] + range[gi + 4] + range[gi + 5];
        c = diffRange * diffRange;
      }
      int pk = (l_flag[i + 6] >> 2) & 1;
#pragma unroll
      for (int k = 0; k <= 5; k++) pk |= l_flag[i + 6 + k] & 1;
#pragma unroll
      for (int k = 1; k <= 6; k++) pk |= (l_flag[i + 6 - k] >> 1) & 1;
      l_picked[i] = pk;
      l_col[i] = in ? A.col[gi] : 0;
      l_label[i] = 0;
      l_curv[i] = c;
      if (gi >= s - 4 && gi <= e + 5 && gi >= 0 && gi < n) A.curv[gi] = c;   // the ring's own cells (rings tile [0, n)): cloudCurvature as the reference leaves it
    }
    __syncthreads();   // keys is scratch no longer
  }
  int scan_cnt = 0, par = 0;  // surface-scan length so far (every thread keeps the same count)
  // brk[i]: the suppression marks of a pick stop between window cells i and i + 1 (|columnDiff| > 10, :199-210)
  for (int i = t; i < wlen; i += XT) { l_rank[i] = INT_MAX; l_stat[i] = ST_OUT; }
  __syncthreads();
  for (int i = t; i + 1 < wlen; i += XT) l_brk[i] = abs(l_col[i + 1] - l_col[i]) > 10 ? 1 : 0;
  __syncthreads();
  for (int i = t; i < wlen; i += XT) {   // once per ring: the rounds below then read their neighbours without a dependent chain of break tests
    int fr = 0, br = 0;
    if (i >= 5 && i + 5 < wlen) {
      for (int d = 1; d <= 5; d++) { if (l_brk[i + d - 1]) break; fr = d; }
      for (int d = 1; d <= 5; d++) { if (l_brk[i - d]) break; br = d; }
    }
    l_reach[i] = fr | (br << 4);
  }

  // ---- all six sector sorts at once (they depend on the curvatures only): a segmented bitonic sort, 6 x seg keys ----
  int seg = 2;
  {
    int maxlen = 0;
    for (int j = 0; j < 6; j++) {
      const int sp = (s * (6 - j) + e * j) / 6, ep = (s * (5 - j) + e * (j + 1)) / 6 - 1;
      maxlen = max(maxlen, ep - sp);
    }
    while (seg < maxlen) seg <<= 1;  // <= SEGMAX: a sector holds at most Horizon_SCAN / 6 + 1 points
  }
  for (int i = t; i < 6 * seg; i += XT) {
    const int j = i / seg, li = i - j * seg;
    const int sp = (s * (6 - j) + e * j) / 6, ep = (s * (5 - j) + e * (j + 1)) / 6 - 1;
    unsigned long long kv = ~0ull;
    if (li < ep - sp) {
      const int k = sp + li;
      // cloudSmoothness[k] = {curvature, k} for k in [5, n-5), else the zero-initialised {0, 0}
      const bool live = k >= 5 && k < n - 5;
      const float cv = live ? l_curv[k - w0] : 0.f;
      const int ind = live ? k : 0;
      kv = ((unsigned long long)__float_as_uint(cv) << 32) | (unsigned)ind;
    }
    keys[i] = kv;
  }
  __syncthreads();
  XT_STAMP(1);
  bitonic_sort_lds_seg<XT>(keys, 6 * seg, seg);   // std::sort(begin+sp, begin+ep) with the (value, ind) tie order
  XT_STAMP(2);

  // ---- all twelve pick stages of the ring as ONE fixed point (rings away from the cloud's two ends, at most 20 corners per sector) ----
  // The reference walks sector 0 corners, sector 0 surfaces, sector 1 corners, ... (featureExtraction.cpp:168-252); a pick marks its +-5
  // neighbours (up to a column break) and later candidates that are marked are skipped. Order every candidate of the ring by (sector, pass,
  // rank in its walk): a candidate is picked iff no candidate EARLIER in that order that reaches it is picked — the stage-by-stage rule
  // below with the order extended across stages, so the chains of twelve stages resolve together (a stage costs ~4 us of barriers with
  // a third of the threads busy). Not covered here and left to the staged code: sectors at the cloud's ends (stale smoothness entries),
  // thresholds that let a cell be corner AND surface candidate, and the cap — the corner walk stops after its 20th pick
  // (largestPickedNum, :186-193), so a sector with more corner picks than that is redone stage by stage (checked before anything is applied).
  bool global_done = false;
#ifndef ROLO_XT_STAGED
  {
    constexpr int UPG = (6 * SEGMAX + XT - 1) / XT;   // entries (sector, position) per thread
    __shared__ int s_gpk[UPG][XW], s_gep[6], s_gbad;
    bool eligible = A.surf_threshold > 0.f && A.edge_threshold >= A.surf_threshold && seg >= 64 && e - s >= 12;
    int sps[6], eps[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      sps[j] = (s * (6 - j) + e * j) / 6; eps[j] = (s * (5 - j) + e * (j + 1)) / 6 - 1;
      eligible = eligible && eps[j] > sps[j] && eps[j] - sps[j] < SEGMAX;
    }
    // the cloud's ends (as in the staged code below): the head ring's stale {0, 0} entries name point 0 — inside this ring's window,
    // an ordinary surface candidate of curvature 0; a tail ring's name point 0 too, which the first ring marked long ago: out of the window, skipped
    const bool head = s < 5, tail = eps[5] >= n - 5;
    if (head) eligible = eligible && !tail && eps[0] >= 5 && w0 <= 0 && -w0 < wlen;
    if (tail) eligible = eligible && !head && n >= 64 && A.start_ring[0] == 4 && A.end_ring[0] >= 30;
    if (eligible) {   // uniform
      if (t < 6) s_gep[t] = 0;
      if (t == 0) s_gbad = 0;
      int my_li[UPG], my_rank[UPG];
#pragma unroll
      for (int u = 0; u < UPG; u++) {
        const int i = t + XT * u, j = i / SEGMAX, p = i - j * SEGMAX;
        my_li[u] = -1; my_rank[u] = INT_MAX;
        if (j < 6) {
          int sp = sps[0], ep = eps[0];
#pragma unroll
          for (int jj = 1; jj < 6; jj++) if (j == jj) { sp = sps[jj]; ep = eps[jj]; }
          const int len = ep - sp;
          if (p <= len) {
            const int k = sp + p;
            const int ind = (k == ep) ? ((ep >= 5 && ep < n - 5) ? ep : 0) : (int)(unsigned)(keys[j * seg + p] & 0xffffffffull);   // smooth[ep] is outside the sorted range
            const int li = ind - w0;
            const bool in_win = li >= 0 && li < wlen;   // a stale entry of a tail ring names point 0, far outside this ring's window: a no-op
            const float cv = (ind == 0 || !in_win) ? 0.f : l_curv[li];   // stale entry: curvature 0 whatever the window holds
            const bool corner = cv > A.edge_threshold, surf = cv < A.surf_threshold;   // never both: edge_threshold >= surf_threshold
            if (in_win && (corner || surf)) {
              const int rank = corner ? ((k == ep) ? 0 : ep - k) : p;   // corners walk k = ep .. sp, surfaces k = sp .. ep
              my_li[u] = li; my_rank[u] = ((2 * j + (corner ? 0 : 1)) << 12) | rank;
              l_rank[li] = my_rank[u];
              l_stat[li] = l_picked[li] == 0 ? ST_UNDECIDED : ST_OUT;
            }
          }
        }
      }
      __syncthreads();
      unsigned my_nb[UPG];
#pragma unroll
      for (int u = 0; u < UPG; u++) {
        my_nb[u] = 0u;
        const int li = my_li[u];
        if (li < 0) continue;
        const int rc = l_reach[li], fr = rc & 15, br = rc >> 4;
#pragma unroll
        for (int d = 1; d <= 5; d++) {
          const int rf = l_rank[li + d], rb = l_rank[li - d];
          if (d <= fr && rf < my_rank[u]) my_nb[u] |= 1u << (d - 1);
          if (d <= br && rb < my_rank[u]) my_nb[u] |= 1u << (4 + d);
        }
      }
      while (true) {
        int undecided = 0;
#pragma unroll
        for (int u = 0; u < UPG; u++) {
          const int li = my_li[u];
          if (li < 0 || l_stat[li] != ST_UNDECIDED) continue;
          bool any_sel = false, any_und = false;
#pragma unroll
          for (int d = 1; d <= 5; d++) {
            const int sf = l_stat[li + d], sb = l_stat[li - d];
            if (my_nb[u] & (1u << (d - 1))) { any_sel |= sf == ST_PICKED; any_und |= sf == ST_UNDECIDED; }
            if (my_nb[u] & (1u << (4 + d))) { any_sel |= sb == ST_PICKED; any_und |= sb == ST_UNDECIDED; }
          }
          if (any_sel) l_stat[li] = ST_OUT;
          else if (!any_und) l_stat[li] = ST_PICKED;
          else undecided = 1;
        }
        if (!__syncthreads_or(undecided)) break;
      }
      // corner picks per (entry slot, wavefront); the k == ep entry of a sector leads its walk
      bool pk[UPG];
#pragma unroll
      for (int u = 0; u < UPG; u++) {
        const int i = t + XT * u, j = i / SEGMAX, p = i - j * SEGMAX;
        const bool is_corner = my_li[u] >= 0 && ((my_rank[u] >> 12) & 1) == 0;
        pk[u] = is_corner && l_stat[my_li[u]] == ST_PICKED;
        bool is_ep = false;
        if (j < 6 && is_corner) { int len = eps[0] - sps[0];
#pragma unroll
          for (int jj = 1; jj < 6; jj++) if (j == jj) len = eps[jj] - sps[jj];
          is_ep = p == len; }
        if (is_ep && pk[u]) s_gep[j] = 1;
        const unsigned long long bal = __ballot(pk[u] && !is_ep);
        if (lane == 0) s_gpk[u][wv] = __popcll(bal);
        // ordinal inside the slot: picks at higher positions of this wavefront's 64
        my_rank[u] = (my_rank[u] & ~0xfff) | (is_ep ? 0xfff : __popcll(bal & ~((2ull << lane) - 1ull)));   // rank no longer needed: low 12 bits = picks above in the wave (0xfff: the ep entry)
      }
      __syncthreads();
      // a sector's total and, per entry, the picks at higher positions in other wavefront slots of the same sector
      int higher[UPG];
#pragma unroll
      for (int u = 0; u < UPG; u++) {
        higher[u] = 0;
        const int i0 = wv * 64 + XT * u, j = i0 / SEGMAX;
        int total = j < 6 ? s_gep[min(j, 5)] : 0;
#pragma unroll
        for (int u2 = 0; u2 < UPG; u2++)
#pragma unroll
          for (int w2 = 0; w2 < XW; w2++) {
            const int i2 = w2 * 64 + XT * u2;
            if (i2 / SEGMAX == j) { const int c = s_gpk[u2][w2]; total += c; if (i2 > i0) higher[u] += c; }
          }
        if (j < 6 && total > 20) s_gbad = 1;
        if (j < 6 && lane == 0 && (i0 - j * SEGMAX) == 0) A.corner_cnt[ring * 6 + j] = total;   // (rewritten by the staged code if the cap strikes)
      }
      __syncthreads();
      if (s_gbad == 0) {
#pragma unroll
        for (int u = 0; u < UPG; u++) {
          const int li = my_li[u];
          if (li < 0 || l_stat[li] != ST_PICKED) continue;
          const int j = (t + XT * u) / SEGMAX;
          if (((my_rank[u] >> 12) & 1) == 0) {
            const int within = my_rank[u] & 0xfff;
            const int ordinal = within == 0xfff ? 0 : s_gep[j] + higher[u] + within;
            l_label[li] = 1;
            A.corner_stage[(ring * 6 + j) * 20 + ordinal] = A.extracted[li + w0];
          } else {
            l_label[li] = -1;
          }
          l_picked[li] = 1;
          const int rc = l_reach[li], fr = rc & 15, br = rc >> 4;
#pragma unroll
          for (int d = 1; d <= 5; d++) { if (d <= fr) l_picked[li + d] = 1; if (d <= br) l_picked[li - d] = 1; }
        }
        __syncthreads();
        // every k of a sector with label <= 0 joins the ring's surface scan, in k order (:240-252); the sectors tile [s, e - 1]
        const int e_last = eps[5];
        for (int base = s; base <= e_last; base += XT) {
          const int k = base + t;
          const bool v = k <= e_last && l_label[k - w0] <= 0;
          const unsigned long long bal = __ballot(v);
          if (lane == 0) s_cw[par][wv] = __popcll(bal);
          __syncthreads();
          int woff = 0, tot = 0;
#pragma unroll
          for (int w = 0; w < XW; w++) { const int c = s_cw[par][w]; tot += c; if (w < wv) woff += c; }
          if (v) list[scan_cnt + woff + __popcll(bal & ((1ull << lane) - 1ull))] = k;
          scan_cnt += tot;
          par ^= 1;
        }
        global_done = true;
      } else {
        // the cap strikes somewhere on this ring: nothing has been applied; neutral rank / state cells for the staged code
#pragma unroll
        for (int u = 0; u < UPG; u++) if (my_li[u] >= 0) { l_rank[my_li[u]] = INT_MAX; l_stat[my_li[u]] = ST_OUT; }
        __syncthreads();
      }
    }
  }
#endif

  if (!global_done)
  for (int j = 0; j < 6; j++) {
    const int sp = (s * (6 - j) + e * j) / 6;
    const int ep = (s * (5 - j) + e * (j + 1)) / 6 - 1;
    if (sp >= ep) { if (t == 0) A.corner_cnt[ring * 6 + j] = 0; continue; }  // uniform
    const unsigned long long* skeys = keys + j * seg;
    const int len = ep - sp;  // sorted range [sp, ep); positions sp..ep take part in the picks
    // Sectors at the two ends of the cloud hold the reference's stale {0, 0} smoothness entries: entries that all name POINT 0,
    // with curvature 0. With sane thresholds (0 is a surface candidate, not a corner) they amount to "point 0 is the best-ranked surface
    // candidate": in the sector that holds point 0 the parallel picks below handle that as it stands (the duplicates fold onto one window
    // cell); in the last sector of the cloud point 0 was picked long ago — the ring that holds it ran first in the reference — so the
    // entries do nothing and are skipped. Anything odder (tiny clouds, both ends in one sector, thresholds that flip the argument) takes the
    // serial walk as written. (Both end sectors on the serial walk: 155 us of this kernel's 170 on the OS1-128 frame.)
    const bool head_sp = sp < 5, tail_sp = ep >= n - 5;
    const bool thr_ok = A.surf_threshold > 0.f && A.edge_threshold >= 0.f;
    const bool head_ok = !head_sp || (thr_ok && !tail_sp && ep >= 5 && w0 <= 0 && -w0 < wlen);
    const bool tail_ok = !tail_sp || (thr_ok && !head_sp && n >= 64 && A.start_ring[0] == 4 && A.end_ring[0] >= 30);   // ring 0 starts at 0 - 1 + 5 and its stale entry marked point 0
    if (head_ok && tail_ok && len < SEGMAX) {
      // ---- parallel greedy picks ----
      // The reference walks the sector in curvature order and a pick marks its +-5 neighbours (up to a column break) as
      // taken. Equivalent fixed point: a candidate is picked iff no better-ranked candidate that reaches it is picked.
      // Every round decides the candidates whose better-ranked reaching candidates are all decided; the chains are a
      // handful of links long, so a sector takes a few rounds of ~20 LDS reads instead of ~10^3 dependent LDS steps
      // on one lane.
      int my_li[UP], my_rank[UP];
      for (int pass = 0; pass < 2; pass++) {   // 0: corners (:181-211), 1: surfaces (:213-238)
        const float thr = pass == 0 ? A.edge_threshold : A.surf_threshold;
#pragma unroll
        for (int u = 0; u < UP; u++) {
          const int p = t + XT * u;
          my_li[u] = -1; my_rank[u] = INT_MAX;
          if (p <= len) {
            const int k = sp + p;
            const int ind = (k == ep) ? ((ep >= 5 && ep < n - 5) ? ep : 0) : (int)(unsigned)(skeys[p] & 0xffffffffull);   // smooth[ep] is outside the sorted range
            const int li = ind - w0;
            if (li >= 0 && li < wlen) {   // a stale entry of the cloud's last sector names point 0, far outside this ring's window: a no-op (above)
              const bool stale = ind == 0;   // (head sector) curvature 0 whatever l_curv holds
              const int rank = pass == 0 ? ((k == ep) ? 0 : ep - k) : p;   // corners walk k = ep .. sp, surfaces k = sp .. ep
              const float cv = stale ? 0.f : l_curv[li];
              const bool cand = l_picked[li] == 0 && (pass == 0 ? cv > thr : cv < thr);
              my_li[u] = li; my_rank[u] = rank;
              l_rank[li] = rank;
              l_stat[li] = cand ? ST_UNDECIDED : ST_OUT;
            }
          }
        }
        __syncthreads();
        // which of the ten neighbours can suppress this candidate: within reach and better ranked — fixed for the stage
        unsigned my_nb[UP];
#pragma unroll
        for (int u = 0; u < UP; u++) {
          my_nb[u] = 0u;
          const int li = my_li[u];
          if (li < 0) continue;
          const int rc = l_reach[li], fr = rc & 15, br = rc >> 4;
#pragma unroll
          for (int d = 1; d <= 5; d++) {
            const int rf = l_rank[li + d], rb = l_rank[li - d];
            if (d <= fr && rf < my_rank[u]) my_nb[u] |= 1u << (d - 1);
            if (d <= br && rb < my_rank[u]) my_nb[u] |= 1u << (4 + d);
          }
        }
        while (true) {   // states only ever move UNDECIDED -> final, so a neighbour's state read mid-update is either still valid
          int undecided = 0;
#pragma unroll
          for (int u = 0; u < UP; u++) {
            const int li = my_li[u];
            if (li < 0 || l_stat[li] != ST_UNDECIDED) continue;
            bool any_sel = false, any_und = false;
#pragma unroll
            for (int d = 1; d <= 5; d++) {   // ten independent LDS reads
              const int sf = l_stat[li + d], sb = l_stat[li - d];
              if (my_nb[u] & (1u << (d - 1))) { any_sel |= sf == ST_PICKED; any_und |= sf == ST_UNDECIDED; }
              if (my_nb[u] & (1u << (4 + d))) { any_sel |= sb == ST_PICKED; any_und |= sb == ST_UNDECIDED; }
            }
            if (any_sel) l_stat[li] = ST_OUT;
            else if (!any_und) l_stat[li] = ST_PICKED;
            else undecided = 1;
          }
          if (!__syncthreads_or(undecided)) break;
        }
        // apply the picks; corners: only the first 20 in walk order exist (largestPickedNum, :186-193)
        int ordinal[UP] = {};
        if (pass == 0) {
          // walk order = position len first, then len - 1 .. 0: a pick's ordinal is the number of picks at higher positions
          int higher[UP];
#pragma unroll
          for (int u = 0; u < UP; u++) {
            const int p = t + XT * u;
            const bool pk = my_li[u] >= 0 && l_stat[my_li[u]] == ST_PICKED;
            if (p == len) s_ep = pk ? 1 : 0;
            const unsigned long long bal = __ballot(pk && p != len);
            higher[u] = __popcll(bal & ~((2ull << lane) - 1ull));
            if (lane == 0) s_pk[u][wv] = __popcll(bal);
          }
          __syncthreads();
          int total = s_ep;
#pragma unroll
          for (int u = 0; u < UP; u++) {
#pragma unroll
            for (int w = 0; w < XW; w++) {
              total += s_pk[u][w];
#pragma unroll
              for (int u2 = 0; u2 < UP; u2++) if (u > u2 || (u == u2 && w > wv)) higher[u2] += s_pk[u][w];
            }
          }
#pragma unroll
          for (int u = 0; u < UP; u++) ordinal[u] = (t + XT * u == len) ? 0 : s_ep + higher[u];
          if (t == 0) A.corner_cnt[ring * 6 + j] = min(total, 20);
        }
#pragma unroll
        for (int u = 0; u < UP; u++) {
          const int li = my_li[u];
          if (li < 0 || l_stat[li] != ST_PICKED) continue;
          if (pass == 0) {
            if (ordinal[u] >= 20) continue;
            l_label[li] = 1;
            A.corner_stage[(ring * 6 + j) * 20 + ordinal[u]] = A.extracted[li + w0];
          } else {
            l_label[li] = -1;
          }
          l_picked[li] = 1;
          const int rc = l_reach[li], fr = rc & 15, br = rc >> 4;
#pragma unroll
          for (int d = 1; d <= 5; d++) { if (d <= fr) l_picked[li + d] = 1; if (d <= br) l_picked[li - d] = 1; }
        }
        __syncthreads();
      }
      // leave the rank / state cells of this sector neutral for the next one
#pragma unroll
      for (int u = 0; u < UP; u++) if (my_li[u] >= 0) { l_rank[my_li[u]] = INT_MAX; l_stat[my_li[u]] = ST_OUT; }
    } else if (t == 0) {
      // sector at a cloud end (stale {0, 0} smoothness entries): the reference's serial walk as written
      // smooth[ep] is outside the sorted range but inside both loops
      const bool live_ep = ep >= 5 && ep < n - 5;
      const int ind_ep = live_ep ? ep : 0;
      // corners: featureExtraction.cpp:181-211
      int largestPickedNum = 0, ncorner = 0;
      for (int k = ep; k >= sp; k--) {
        const int ind = (k == ep) ? ind_ep : (int)(unsigned)(skeys[k - sp] & 0xffffffffull);
        const int li = ind - w0;
        const bool in_win = li >= 0 && li < wlen;
        const int pk = in_win ? l_picked[li] : A.picked[ind];
        const float cv = in_win ? l_curv[li] : A.curv[ind];
        if (pk == 0 && cv > A.edge_threshold) {
          largestPickedNum++;
          if (largestPickedNum <= 20) {
            if (in_win) l_label[li] = 1; else A.label[ind] = 1;
            A.corner_stage[(ring * 6 + j) * 20 + ncorner] = A.extracted[ind];
            ncorner++;
          } else break;
          if (in_win) l_picked[li] = 1; else A.picked[ind] = 1;
          for (int l = 1; l <= 5; l++) {
            const int a = ind + l, b = ind + l - 1;
            const int ca = (a - w0 >= 0 && a - w0 < wlen) ? l_col[a - w0] : A.col[a];
            const int cb = (b - w0 >= 0 && b - w0 < wlen) ? l_col[b - w0] : A.col[b];
            if (abs(ca - cb) > 10) break;
            if (a - w0 >= 0 && a - w0 < wlen) l_picked[a - w0] = 1; else A.picked[a] = 1;
          }
          for (int l = -1; l >= -5; l--) {
            const int a = ind + l, b = ind + l + 1;
            const int ca = (a - w0 >= 0 && a - w0 < wlen) ? l_col[a - w0] : A.col[a];
            const int cb = (b - w0 >= 0 && b - w0 < wlen) ? l_col[b - w0] : A.col[b];
            if (abs(ca - cb) > 10) break;
            if (a - w0 >= 0 && a - w0 < wlen) l_picked[a - w0] = 1; else A.picked[a] = 1;
          }
        }
      }
      A.corner_cnt[ring * 6 + j] = ncorner;
      // surfaces: :213-238
      for (int k = sp; k <= ep; k++) {
        const int ind = (k == ep) ? ind_ep : (int)(unsigned)(skeys[k - sp] & 0xffffffffull);
        const int li = ind - w0;
        const bool in_win = li >= 0 && li < wlen;
        const int pk = in_win ? l_picked[li] : A.picked[ind];
        const float cv = in_win ? l_curv[li] : A.curv[ind];
        if (pk == 0 && cv < A.surf_threshold) {
          if (in_win) { l_label[li] = -1; l_picked[li] = 1; } else { A.label[ind] = -1; A.picked[ind] = 1; }
          for (int l = 1; l <= 5; l++) {
            const int a = ind + l, b = ind + l - 1;
            const int ca = (a - w0 >= 0 && a - w0 < wlen) ? l_col[a - w0] : A.col[a];
            const int cb = (b - w0 >= 0 && b - w0 < wlen) ? l_col[b - w0] : A.col[b];
            if (abs(ca - cb) > 10) break;
            if (a - w0 >= 0 && a - w0 < wlen) l_picked[a - w0] = 1; else A.picked[a] = 1;
          }
          for (int l = -1; l >= -5; l--) {
            const int a = ind + l, b = ind + l + 1;
            const int ca = (a - w0 >= 0 && a - w0 < wlen) ? l_col[a - w0] : A.col[a];
            const int cb = (b - w0 >= 0 && b - w0 < wlen) ? l_col[b - w0] : A.col[b];
            if (abs(ca - cb) > 10) break;
            if (a - w0 >= 0 && a - w0 < wlen) l_picked[a - w0] = 1; else A.picked[a] = 1;
          }
        }
      }
    }
    __syncthreads();
    // every k in [sp, ep] with label <= 0 joins the ring's surface scan, in k order (:240-252): ballot compaction
    for (int base = sp; base <= ep; base += XT) {
      const int k = base + t;
      const bool v = k <= ep && l_label[k - w0] <= 0;
      const unsigned long long bal = __ballot(v);
      if (lane == 0) s_cw[par][wv] = __popcll(bal);
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < XW; w++) { const int c = s_cw[par][w]; tot += c; if (w < wv) woff += c; }
      if (v) list[scan_cnt + woff + __popcll(bal & ((1ull << lane) - 1ull))] = k;
      scan_cnt += tot;
      par ^= 1;
    }
  }
  XT_STAMP(3);
  // write the ring's picked / label window back (the oracle's arrays after extraction)
  for (int i = t; i < wlen; i += XT) {
    const int gi = w0 + i;
    if (gi >= s - 10 && gi <= e + 10 && gi >= -FRONT_GUARD && gi < n + FRONT_GUARD) {
      if (l_picked[i]) A.picked[gi] = 1;
      if (gi >= s && gi <= e) A.label[gi] = l_label[i];
      else if (l_label[i] != 0) A.label[gi] = l_label[i];
    }
  }
  __syncthreads();

  // ---- pcl::VoxelGrid on the ring's surface scan (featureExtraction.cpp:254-258) ----
  const int m = scan_cnt;
  if (m == 0) { if (t == 0) A.surf_cnt[ring] = 0; return; }
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = t; i < m; i += XT) {
    const float4 p = A.extracted[list[i]];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64)); }
    if (lane == 0) { s_red[0][d][wv] = mn[d]; s_red[1][d][wv] = mx[d]; }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; d++) {
    float lo = s_red[0][d][0], hi = s_red[1][d][0];
#pragma unroll
    for (int w = 1; w < XW; w++) { lo = fminf(lo, s_red[0][d][w]); hi = fmaxf(hi, s_red[1][d][w]); }
    mn[d] = lo; mx[d] = hi;
  }
  const float inv = 1.0f / A.leaf;
  const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1, dz = (long long)((mx[2] - mn[2]) * inv) + 1;
  float4* out = A.surf_stage + (size_t)ring * MAXH;
  if (dx * dy * dz > (long long)INT_MAX) {  // PCL: warn and copy the input through
    for (int i = t; i < m; i += XT) out[i] = A.extracted[list[i]];
    if (t == 0) A.surf_cnt[ring] = m;
    return;
  }
  const int min_b0 = (int)floorf(mn[0] * inv), min_b1 = (int)floorf(mn[1] * inv), min_b2 = (int)floorf(mn[2] * inv);
  const int div_b0 = (int)floorf(mx[0] * inv) - min_b0 + 1, div_b1 = (int)floorf(mx[1] * inv) - min_b1 + 1;
  int np2 = 2; while (np2 < m) np2 <<= 1;
  for (int i = t; i < np2; i += XT) {
    unsigned long long kv = ~0ull;
    if (i < m) {
      const float4 p = A.extracted[list[i]];
      const int ijk0 = (int)floorf(p.x * inv) - min_b0, ijk1 = (int)floorf(p.y * inv) - min_b1, ijk2 = (int)floorf(p.z * inv) - min_b2;
      const int cell = ijk0 + ijk1 * div_b0 + ijk2 * div_b0 * div_b1;
      kv = ((unsigned long long)(unsigned)cell << 32) | (unsigned)i;
    }
    keys[i] = kv;
  }
  __syncthreads();
  XT_STAMP(4);
  bitonic_sort_lds<XT>(keys, np2);  // std::sort by cell index, ties by point order
  XT_STAMP(5);
  // run heads -> output slot; each head accumulates its run in order with float accumulators (pcl::CentroidPoint)
  if (t == 0) s_heads = 0;
  __syncthreads();
  for (int base = 0; base < m; base += XT) {
    const int i = base + t;
    const bool head = i < m && (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32));
    int inc = head ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; w++) woff += s_wsum[w];
    const int c = s_heads;
    if (head) {
      const unsigned cell = (unsigned)(keys[i] >> 32);
      float sx = 0, sy = 0, sz = 0, si = 0;
      int cnt = 0;
      for (int r = i; r < m && (unsigned)(keys[r] >> 32) == cell; r++) {
        const float4 p = A.extracted[list[(int)(unsigned)(keys[r] & 0xffffffffull)]];
        sx += p.x; sy += p.y; sz += p.z; si += p.w; cnt++;
      }
      const float fc = (float)cnt;
      out[c + woff + inc - 1] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
    }
    __syncthreads();
    if (t == XT - 1) s_heads = c + woff + inc;
    __syncthreads();
  }
  XT_STAMP(6);
  if (t == 0) A.surf_cnt[ring] = s_heads;
}

// concatenate per-ring / per-sector staging areas in order
__global__ __launch_bounds__(256) void concat_kernel(const float4* __restrict__ stage, const int* __restrict__ cnt, int n_groups, int group_cap,
                                                    float4* __restrict__ out, int* __restrict__ total) {
  __shared__ int s_off, s_part[4];
  const int g = blockIdx.x, t = threadIdx.x;
  // offset of this group = sum of the counts before it: all 256 threads share the (up to n_scan * 6) loads
  int part = 0;
  for (int r = t; r < g; r += 256) part += cnt[r];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((t & 63) == 0) s_part[t >> 6] = part;
  __syncthreads();
  if (t == 0) {
    const int off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    s_off = off;
    if (g == n_groups - 1) *total = off + cnt[g];
  }
  __syncthreads();
  const int c = cnt[g];
  for (int i = t; i < c; i += 256) out[s_off + i] = stage[(size_t)g * group_cap + i];
}

// fused frame path: corner groups (n_scan x 6 sectors, <= 20 each) then surface groups (one per ring) straight into *featureLast =
// corner ++ surface (lidarOdometry.cpp:521-523) — the two concat launches and feature_concat_kernel in one; counts to counters[1], [2]
__global__ __launch_bounds__(256) void feature_gather_kernel(const float4* __restrict__ corner_stage, const int* __restrict__ corner_cnt, int n_cgroups,
                                                            const float4* __restrict__ surf_stage, const int* __restrict__ surf_cnt, int n_sgroups, int surf_cap,
                                                            float4* __restrict__ out, int* __restrict__ counters, int* __restrict__ pub3) {
  __shared__ int s_part[2][4], s_off;
  const int g = blockIdx.x, t = threadIdx.x;
  const bool is_surf = g >= n_cgroups;
  const int gs = is_surf ? g - n_cgroups : g;
  // corners before this group (every corner for a surface group) + surface points before this group
  int pc = 0, ps = 0;
  for (int r = t; r < (is_surf ? n_cgroups : gs); r += 256) pc += corner_cnt[r];
  if (is_surf) for (int r = t; r < gs; r += 256) ps += surf_cnt[r];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { pc += __shfl_xor(pc, off, 64); ps += __shfl_xor(ps, off, 64); }
  if ((t & 63) == 0) { s_part[0][t >> 6] = pc; s_part[1][t >> 6] = ps; }
  __syncthreads();
  if (t == 0) {
    const int oc = s_part[0][0] + s_part[0][1] + s_part[0][2] + s_part[0][3], os = s_part[1][0] + s_part[1][1] + s_part[1][2] + s_part[1][3];
    s_off = oc + os;
    // pub3: the caller's pinned (n_valid, n_corner, n_surface) — written here instead of by a copy launch behind this kernel
    if (g == n_cgroups - 1) { const int v = oc + corner_cnt[gs]; counters[1] = v; if (pub3) { pub3[1] = v; pub3[0] = counters[0]; } }
    if (is_surf && gs == n_sgroups - 1) { const int v = os + surf_cnt[gs]; counters[2] = v; if (pub3) pub3[2] = v; }
  }
  __syncthreads();
  const int c = is_surf ? surf_cnt[gs] : corner_cnt[gs];
  const float4* __restrict__ src = is_surf ? surf_stage + (size_t)gs * surf_cap : corner_stage + (size_t)gs * 20;
  for (int i = t; i < c; i += 256) out[s_off + i] = src[i];
}

// one launch instead of six fills per frame: owner = INT_MAX (npix), and zeros for col / range / curv / picked / label incl. their guard cells (np)
__global__ __launch_bounds__(256) void front_clear_kernel(int* owner, size_t npix, int* col, float* range, float* curv, int* picked, int* label, size_t np) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  for (size_t i = t; i < np; i += nt) {
    if (i < npix) owner[i] = INT_MAX;
    col[i] = 0; range[i] = 0.f; curv[i] = 0.f; picked[i] = 0; label[i] = 0;
  }
}

__global__ void fill_int_kernel(int* p, int n, int v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}


struct Front {
  int device = 0;
  size_t cap_raw = 0, cap_raw_pts = 0, cap_pix = 0, cap_scan = 0;
  float* raw = nullptr; unsigned short* ring = nullptr;
  int *owner = nullptr, *local_idx = nullptr, *ring_count = nullptr, *start_ring = nullptr, *end_ring = nullptr, *counters = nullptr;
  float4* extracted = nullptr; int* col = nullptr; float* range = nullptr; float* range_mat = nullptr;
  float* curv = nullptr; int *picked = nullptr, *label = nullptr;
  float4 *corner_stage = nullptr, *surf_stage = nullptr, *corner_out = nullptr, *surf_out = nullptr;
  int *corner_cnt = nullptr, *surf_cnt = nullptr;
  unsigned char* big = nullptr; int maxh = FRONT_MAX_H, cap_maxh = 0;   // Horizon_SCAN > 2048: per-ring scratch of the extract kernel, staging pitch
  int n_valid = 0; int n_scan = 0, H = 0;
  bool projected = false;
  bool extract_cleared = false;   // the projection's clear launch already zeroed curv / picked / label for the extraction that follows
  size_t precleared_np = 0;       // fused frames: the per-point arrays (np elements) were cleared BEHIND the previous frame's features, off the next frame's critical path (0: not)
  // rolo_front_set_deskew: armed for the next projection only
  bool deskew_armed = false;
  int deskew_n = 0;   // length of the armed per-point time array
  DeskewArgs deskew{};
  float* rel_time = nullptr; size_t cap_time = 0;   // staging when the times come from the host
  bool deskew_from_msg = false;                      // armed without times: the next message brings them
  // message-level entry: raw payload and what the unpack kernel makes of it
  unsigned char* msg_raw = nullptr; size_t cap_msg = 0;
  float* msg_xyz = nullptr; unsigned short* msg_ring = nullptr; float* msg_time = nullptr; size_t cap_msg_pts = 0;
};

thread_local std::string g_ferr;

template <typename T>
bool dev_alloc(T*& p, size_t count) {
  if (p) { (void)hipFree(p); p = nullptr; }
  return hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)) == hipSuccess;
}

void front_free(Front* f) {
  void* bufs[] = {f->raw, f->ring, f->owner, f->local_idx, f->ring_count, f->start_ring, f->end_ring, f->counters, f->extracted, f->col, f->range,
                  f->range_mat, f->curv, f->picked, f->label, f->corner_stage, f->surf_stage, f->corner_out, f->surf_out, f->corner_cnt, f->surf_cnt, f->big, f->rel_time, f->msg_raw, f->msg_xyz, f->msg_ring, f->msg_time};
  for (void* b : bufs) if (b) (void)hipFree(b);
}

}  // namespace

// accessors implemented in api.hip
void** ctx_front_slot(rolo_ctx* c);
hipStream_t ctx_stream(rolo_ctx* c);
int ctx_device(rolo_ctx* c);
void ctx_set_error(const char* msg);

#define FCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { ctx_set_error((std::string(#x) + ": " + hipGetErrorString(_e)).c_str()); return ROLO_EHIP; } } while (0)

namespace {

// device buffers for frames of up to n_raw points / n_scan x horizon_scan pixels
int front_prepare(rolo_ctx* c, const rolo_front_params* P, int n_raw, int stride, bool stage_raw, Front** out) {
  if (P->n_scan <= 0 || P->horizon_scan <= 0 || P->downsample_rate <= 0) { ctx_set_error("bad front params"); return ROLO_EINVAL; }
  if (P->horizon_scan > FRONT_MAX_H_BIG) { ctx_set_error("Horizon_SCAN above the limit of this build (4096)"); return ROLO_EUNSUPPORTED; }
  FCHK(hipSetDevice(ctx_device(c)));
  void** slot = ctx_front_slot(c);
  if (!*slot) *slot = new Front();
  Front* f = static_cast<Front*>(*slot);
  f->device = ctx_device(c);
  const int NS = P->n_scan, H = P->horizon_scan;
  const size_t npix = (size_t)NS * H;
  bool ok = true;
  // two capacities: floats of the staged records (n_raw * stride) and points (ring) — a later frame with a smaller stride can hold
  // more points inside the same float capacity
  if (stage_raw && ((size_t)n_raw * stride > f->cap_raw || !f->raw)) {
    ok = ok && dev_alloc(f->raw, (size_t)n_raw * stride);
    f->cap_raw = ok ? (size_t)n_raw * stride : 0;
  }
  if (stage_raw && ((size_t)n_raw > f->cap_raw_pts || !f->ring)) {
    ok = ok && dev_alloc(f->ring, (size_t)n_raw);
    f->cap_raw_pts = ok ? (size_t)n_raw : 0;
  }
  if (npix > f->cap_pix || !f->owner) {
    const size_t np = npix + 2 * FRONT_GUARD;
    ok = ok && dev_alloc(f->owner, npix) && dev_alloc(f->local_idx, npix) && dev_alloc(f->extracted, np) && dev_alloc(f->col, np) && dev_alloc(f->range, np) &&
         dev_alloc(f->range_mat, npix) && dev_alloc(f->curv, np) && dev_alloc(f->picked, np) && dev_alloc(f->label, np) && dev_alloc(f->corner_out, npix) &&
         dev_alloc(f->surf_out, npix);
    f->cap_pix = npix; f->precleared_np = 0;   // fresh buffers: not cleared
  }
  const int maxh = H > FRONT_MAX_H ? FRONT_MAX_H_BIG : FRONT_MAX_H;   // pitch of the per-ring staging rows, and which extract kernel runs
  if ((size_t)NS > f->cap_scan || !f->ring_count || maxh > f->cap_maxh) {
    ok = ok && dev_alloc(f->ring_count, NS) && dev_alloc(f->start_ring, NS) && dev_alloc(f->end_ring, NS) && dev_alloc(f->counters, 8) &&
         dev_alloc(f->corner_stage, (size_t)NS * 6 * 20) && dev_alloc(f->corner_cnt, (size_t)NS * 6) && dev_alloc(f->surf_stage, (size_t)NS * maxh) &&
         dev_alloc(f->surf_cnt, NS);
    if (maxh > FRONT_MAX_H) ok = ok && dev_alloc(f->big, (size_t)NS * front_ring_bytes(FRONT_MAX_H_BIG));
    f->cap_scan = ok ? NS : 0; f->cap_maxh = ok ? maxh : 0;
  }
  f->maxh = maxh;
  if (!ok) { ctx_set_error("hipMalloc failed (front end)"); return ROLO_EHIP; }
  f->n_scan = NS; f->H = H; f->projected = false;
  *out = f;
  return ROLO_OK;
}

// K1 + K2 on raw points that are on the device; N lands in counters[0]. No host synchronisation.
int front_project_enqueue(Front* f, const rolo_front_params* P, const float* d_pts, int stride, const unsigned short* d_ring, int n_raw, bool want_range_mat,
                          hipStream_t s) {
  const int NS = f->n_scan, H = f->H;
  const size_t npix = (size_t)NS * H;
  if (f->deskew_from_msg) {   // armed without times and no time field came with the points: interpolate them from the azimuth
    f->deskew_from_msg = false;
    if (n_raw > 0) {
      if ((size_t)n_raw > f->cap_time || !f->rel_time) {
        if (!dev_alloc(f->rel_time, (size_t)n_raw)) { ctx_set_error("hipMalloc failed (de-skew times)"); return ROLO_EHIP; }
        f->cap_time = (size_t)n_raw;
      }
      fill_int_kernel<<<1, 64, 0, s>>>(f->counters + 4, 1, INT_MAX);
      azimuth_flag_kernel<<<(n_raw + 255) / 256, 256, 0, s>>>(d_pts, stride, n_raw, f->counters + 4);
      azimuth_time_kernel<<<(n_raw + 255) / 256, 256, 0, s>>>(d_pts, stride, n_raw, f->deskew.scan_period, f->counters + 4, f->rel_time);
      f->deskew.rel_time = f->rel_time;
      f->deskew_armed = true; f->deskew_n = n_raw;
    }
  }
  if (f->deskew_armed && f->deskew_n < n_raw) {   // ring_scatter_kernel reads rel_time[o] for every raw point of this frame
    f->deskew_armed = false;
    ctx_set_error("de-skew armed with fewer per-point times than the frame has points");
    return ROLO_EINVAL;
  }
  // guard cells of the per-point arrays are zero; the three arrays of the extraction stage are cleared in the same launch
  if (f->precleared_np != npix + 2 * FRONT_GUARD)   // (a fused frame leaves the arrays cleared for its successor)
    front_clear_kernel<<<512, 256, 0, s>>>(f->owner, npix, f->col, f->range, f->curv, f->picked, f->label, npix + 2 * FRONT_GUARD);
  f->precleared_np = 0;
  f->extract_cleared = true;
  if (n_raw > 0) project_kernel<<<(n_raw + 255) / 256, 256, 0, s>>>(d_pts, stride, d_ring, n_raw, *P, f->owner);
  ring_scan_kernel<<<NS, 256, 0, s>>>(f->owner, H, f->local_idx, f->ring_count);
  ring_scatter_kernel<<<NS, 256, 0, s>>>(d_pts, stride, d_ring, f->owner, f->local_idx, f->ring_count, NS, H, f->extracted + FRONT_GUARD,
                                        f->col + FRONT_GUARD, f->range + FRONT_GUARD, f->start_ring, f->end_ring, want_range_mat ? f->range_mat : nullptr,
                                        f->counters, f->deskew_armed ? f->deskew : DeskewArgs{});
  f->deskew_armed = false;
  FCHK(hipGetLastError());
  return ROLO_OK;
}

// K3 + K4 on what front_project_enqueue left on the device; n_corner / n_surface land in counters[1] / [2]
int front_extract_enqueue(Front* f, const rolo_front_params* P, hipStream_t s, float4* fused_out = nullptr, int* pub3 = nullptr) {
  const int NS = f->n_scan;
  const size_t npix = f->cap_pix;
  // guards of curvature / picked / label are zero; live entries are written by the smoothness kernel
  const size_t np = npix + 2 * FRONT_GUARD;
  if (!f->extract_cleared) {   // a projection loaded from the host (rolo_front_load_projection): nobody cleared them yet
    FCHK(hipMemsetAsync(f->curv, 0, sizeof(float) * np, s));
    FCHK(hipMemsetAsync(f->picked, 0, sizeof(int) * np, s));
    FCHK(hipMemsetAsync(f->label, 0, sizeof(int) * np, s));
  }
  f->extract_cleared = false;
  // (K3 — calculateSmoothness + markOccludedPoints — runs inside extract_kernel; curv / picked / label are zero at this point)
  FeatArgs A;
  A.extracted = f->extracted + FRONT_GUARD; A.col = f->col + FRONT_GUARD; A.range = f->range + FRONT_GUARD; A.curv = f->curv + FRONT_GUARD; A.picked = f->picked + FRONT_GUARD;
  A.label = f->label + FRONT_GUARD; A.start_ring = f->start_ring; A.end_ring = f->end_ring; A.n_ptr = f->counters; A.n_scan = NS;
  A.edge_threshold = P->edge_threshold; A.surf_threshold = P->surf_threshold; A.leaf = P->odometry_surf_leaf_size;
  A.corner_stage = f->corner_stage; A.corner_cnt = f->corner_cnt; A.surf_stage = f->surf_stage; A.surf_cnt = f->surf_cnt;
  A.big = reinterpret_cast<unsigned char*>(f->big);
  if (f->maxh > FRONT_MAX_H) {
    extract_kernel<FRONT_MAX_H_BIG, true><<<NS, XT, 0, s>>>(A);   // the ring's working set in HBM scratch
  } else {
    constexpr size_t lds = front_ring_bytes(FRONT_MAX_H);
    static std::atomic<unsigned long long> attr_set{0};   // per device ordinal (bit d): the attribute belongs to the device's code object
    const unsigned long long dev_bit = 1ull << (f->device & 63);
    if (!(attr_set.load() & dev_bit)) { FCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(extract_kernel<FRONT_MAX_H, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr_set.fetch_or(dev_bit); }
    extract_kernel<FRONT_MAX_H, false><<<NS, XT, lds, s>>>(A);
  }
  if (fused_out) {
    feature_gather_kernel<<<NS * 7, 256, 0, s>>>(f->corner_stage, f->corner_cnt, NS * 6, f->surf_stage, f->surf_cnt, NS, f->maxh, fused_out, f->counters, pub3);
  } else {
    concat_kernel<<<NS * 6, 256, 0, s>>>(f->corner_stage, f->corner_cnt, NS * 6, 20, f->corner_out, f->counters + 1);
    concat_kernel<<<NS, 256, 0, s>>>(f->surf_stage, f->surf_cnt, NS, f->maxh, f->surf_out, f->counters + 2);
  }
  FCHK(hipGetLastError());
  return ROLO_OK;
}

}  // namespace

// Fused K1-K4 for the device-resident pipeline (rolo_odom_frame): raw frame -> *featureLast = corner ++ surface in
// d_feat (capacity front_feature_capacity floats4), one host synchronisation for the three counts.
size_t front_feature_capacity(const rolo_front_params* P) { return (size_t)P->n_scan * P->horizon_scan + (size_t)P->n_scan * 6 * 20; }

// No host synchronisation: `done` is recorded on the context's stream behind the read-back of the three counts into
// h_counts3 (pinned host memory), so the caller can keep another stream busy meanwhile.
int front_frame_features_enqueue(rolo_ctx* c, const rolo_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                                 bool on_device, float4* d_feat, int* h_counts3, hipEvent_t done) {
  Front* f = nullptr;
  int rc = front_prepare(c, P, n_raw, stride, !on_device, &f);
  if (rc) return rc;
  hipStream_t s = ctx_stream(c);
  const float* d_pts = pts; const unsigned short* d_ring = ring;
  if (!on_device) {
    FCHK(hipMemcpyAsync(f->raw, pts, sizeof(float) * (size_t)n_raw * stride, hipMemcpyHostToDevice, s));
    FCHK(hipMemcpyAsync(f->ring, ring, sizeof(uint16_t) * (size_t)n_raw, hipMemcpyHostToDevice, s));
    d_pts = f->raw; d_ring = f->ring;
  }
  if ((rc = front_project_enqueue(f, P, d_pts, stride, d_ring, n_raw, false, s))) return rc;
  if ((rc = front_extract_enqueue(f, P, s, d_feat, h_counts3))) return rc;   // h_counts3 is pinned: the gather kernel writes the counts there itself
  FCHK(hipGetLastError());
  if (done) FCHK(hipEventRecord(done, s));
  // the next frame's clear, behind this frame's features: nobody reads the per-point arrays of a fused frame any more (its outputs are d_feat
  // and the three counts), and the front-end stream is idle until the next frame arrives — 3.6 us and a kernel boundary off its critical path
  {
    const size_t npix = f->cap_pix;
    front_clear_kernel<<<512, 256, 0, s>>>(f->owner, npix, f->col, f->range, f->curv, f->picked, f->label, npix + 2 * FRONT_GUARD);
    FCHK(hipGetLastError());
    f->precleared_np = npix + 2 * FRONT_GUARD;
  }
  f->projected = false;  // the staged rolo_extract_features must not run on top of a fused frame
  return ROLO_OK;
}

// Message-level entry of the fused path: the PointCloud2 payload goes to the device as it is (one copy), the field
// extraction the node does on the host runs as a kernel, and the frame continues device-resident.
int front_frame_features_from_msg(rolo_ctx* c, const rolo_front_params* P, const unsigned char* data, const rolo_cloud_layout* L, int n_raw,
                                  bool on_device, float4* d_feat, int* h_counts3, hipEvent_t done) {
  if (L->point_step <= 0 || (L->ring_bytes != 1 && L->ring_bytes != 2) || L->time_kind < 0 || L->time_kind > 2) { ctx_set_error("bad cloud layout"); return ROLO_EINVAL; }
  const int need = std::max(std::max(L->off_x, L->off_y), L->off_z) + 4;
  if (L->off_x < 0 || L->off_y < 0 || L->off_z < 0 || L->off_ring < 0 || need > L->point_step || L->off_ring + L->ring_bytes > L->point_step ||
      (L->time_kind && (L->off_time < 0 || L->off_time + 4 > L->point_step))) { ctx_set_error("cloud layout: field outside the point record"); return ROLO_EINVAL; }
  Front* f = nullptr;
  int rc = front_prepare(c, P, n_raw, 3, false, &f);
  if (rc) return rc;
  hipStream_t s = ctx_stream(c);
  const size_t bytes = (size_t)n_raw * L->point_step;
  if ((size_t)n_raw > f->cap_msg_pts || !f->msg_xyz) {
    if (!dev_alloc(f->msg_xyz, 3 * (size_t)n_raw) || !dev_alloc(f->msg_ring, (size_t)n_raw) || !dev_alloc(f->msg_time, (size_t)n_raw)) { ctx_set_error("hipMalloc failed (message buffers)"); return ROLO_EHIP; }
    f->cap_msg_pts = (size_t)n_raw;
  }
  const unsigned char* d_data = data;
  if (!on_device) {
    if (bytes > f->cap_msg || !f->msg_raw) { if (!dev_alloc(f->msg_raw, bytes)) { ctx_set_error("hipMalloc failed (message payload)"); return ROLO_EHIP; } f->cap_msg = bytes; }
    FCHK(hipMemcpyAsync(f->msg_raw, data, bytes, hipMemcpyHostToDevice, s));
    d_data = f->msg_raw;
  }
  if (n_raw > 0) unpack_cloud_kernel<<<(n_raw + 255) / 256, 256, 0, s>>>(d_data, *L, n_raw, f->msg_xyz, f->msg_ring, f->msg_time);
  FCHK(hipGetLastError());
  if (f->deskew_from_msg && L->time_kind != 0) {   // without a time field the projection interpolates the times from the azimuth
    f->deskew_from_msg = false;
    f->deskew.rel_time = f->msg_time;
    f->deskew_armed = true; f->deskew_n = n_raw;
  }
  return front_frame_features_enqueue(c, P, f->msg_xyz, 3, f->msg_ring, n_raw, true, d_feat, h_counts3, done);
}

}  // namespace rolo

using namespace rolo;

namespace rolo {
// rolo_ctx_release: a recycled context is a fresh object — a projection, an armed de-skew or cleared arrays of its previous owner must not carry over
// (a staged rolo_extract_features on a recycled context returns ROLO_ESTATE until its new owner has projected a frame)
void front_reset_object_state(rolo_ctx* c) {
  void** slot = ctx_front_slot(c);
  if (!*slot) return;
  Front* f = static_cast<Front*>(*slot);
  f->projected = false; f->extract_cleared = false; f->precleared_np = 0;
  f->deskew_armed = false; f->deskew_from_msg = false; f->deskew_n = 0;
  f->n_valid = 0;
}
}  // namespace rolo

extern "C" {

void rolo_front_destroy(rolo_ctx* c) {
  void** slot = ctx_front_slot(c);
  if (*slot) { Front* f = static_cast<Front*>(*slot); front_free(f); delete f; *slot = nullptr; }
}

int rolo_front_set_deskew(rolo_ctx* c, const rolo_deskew* d, const float* rel_time, int n_raw, int on_device) {
  if (!c || !d) return ROLO_EINVAL;
  FCHK(hipSetDevice(ctx_device(c)));
  void** slot = ctx_front_slot(c);
  if (!*slot) *slot = new Front();
  Front* f = static_cast<Front*>(*slot);
  f->deskew_armed = false; f->deskew_from_msg = false;
  if (!d->enabled) return ROLO_OK;   // deskewPoint returns the point as is (:371-372)
  if (!(d->odom_time_diff != 0.0) || !(d->scan_period != 0.f)) { ctx_set_error("de-skew: zero scan period / odometry time difference"); return ROLO_EINVAL; }
  if (!rel_time) {   // the times come with the next message (rolo_odom_submit_msg)
    f->deskew = DeskewArgs{nullptr, d->odom_incre_rpy[0], d->odom_incre_rpy[1], d->odom_incre_rpy[2], d->scan_period, d->odom_time_diff};
    f->deskew_from_msg = true;
    return ROLO_OK;
  }
  if (n_raw < 0) { ctx_set_error("de-skew needs the per-point times"); return ROLO_EINVAL; }
  const float* dt = rel_time;
  if (!on_device) {
    if ((size_t)n_raw > f->cap_time || !f->rel_time) {
      if (!dev_alloc(f->rel_time, (size_t)n_raw)) { ctx_set_error("hipMalloc failed (de-skew times)"); return ROLO_EHIP; }
      f->cap_time = (size_t)n_raw;
    }
    FCHK(hipMemcpyAsync(f->rel_time, rel_time, sizeof(float) * (size_t)n_raw, hipMemcpyHostToDevice, ctx_stream(c)));
    dt = f->rel_time;
  }
  f->deskew = DeskewArgs{dt, d->odom_incre_rpy[0], d->odom_incre_rpy[1], d->odom_incre_rpy[2], d->scan_period, d->odom_time_diff};
  f->deskew_armed = true; f->deskew_n = n_raw;
  return ROLO_OK;
}

void rolo_front_default_params(rolo_front_params* p) {  // config/params.yaml:20-36
  p->n_scan = 32; p->horizon_scan = 1024; p->downsample_rate = 1;
  p->lidar_min_range = 2.0f; p->lidar_max_range = 1000.0f;
  p->edge_threshold = 0.8f; p->surf_threshold = 0.1f; p->odometry_surf_leaf_size = 0.4f;
}

int rolo_project_frame(rolo_ctx* c, const rolo_front_params* P, const float* pts, int stride, const uint16_t* ring, int n_raw,
                       float* extracted, int32_t* point_col_ind, float* point_range, int32_t* start_ring, int32_t* end_ring,
                       float* range_mat, int* n_valid) {
  if (!c || !P || !pts || !ring || stride < 3 || n_raw < 0 || !n_valid) return ROLO_EINVAL;
  Front* f = nullptr;
  int rc = front_prepare(c, P, n_raw, stride, true, &f);
  if (rc) return rc;
  hipStream_t s = ctx_stream(c);
  const int NS = P->n_scan;
  const size_t npix = (size_t)NS * P->horizon_scan;
  FCHK(hipMemcpyAsync(f->raw, pts, sizeof(float) * (size_t)n_raw * stride, hipMemcpyHostToDevice, s));
  FCHK(hipMemcpyAsync(f->ring, ring, sizeof(uint16_t) * (size_t)n_raw, hipMemcpyHostToDevice, s));
  if ((rc = front_project_enqueue(f, P, f->raw, stride, f->ring, n_raw, true, s))) return rc;
  int nv = 0;
  FCHK(hipMemcpyAsync(&nv, f->counters, sizeof(int), hipMemcpyDeviceToHost, s));
  FCHK(hipStreamSynchronize(s));
  f->n_valid = nv;
  *n_valid = nv;
  if (extracted && nv) FCHK(hipMemcpyAsync(extracted, f->extracted + FRONT_GUARD, sizeof(float4) * (size_t)nv, hipMemcpyDeviceToHost, s));
  if (point_col_ind && nv) FCHK(hipMemcpyAsync(point_col_ind, f->col + FRONT_GUARD, sizeof(int) * (size_t)nv, hipMemcpyDeviceToHost, s));
  if (point_range && nv) FCHK(hipMemcpyAsync(point_range, f->range + FRONT_GUARD, sizeof(float) * (size_t)nv, hipMemcpyDeviceToHost, s));
  if (start_ring) FCHK(hipMemcpyAsync(start_ring, f->start_ring, sizeof(int) * (size_t)NS, hipMemcpyDeviceToHost, s));
  if (end_ring) FCHK(hipMemcpyAsync(end_ring, f->end_ring, sizeof(int) * (size_t)NS, hipMemcpyDeviceToHost, s));
  if (range_mat) FCHK(hipMemcpyAsync(range_mat, f->range_mat, sizeof(float) * npix, hipMemcpyDeviceToHost, s));
  FCHK(hipStreamSynchronize(s));
  f->projected = true;
  return ROLO_OK;
}

// FeatureExtraction::laserCloudInfoHandler's input when the node runs in its own process: the arrays of the received
// rolo/cloud_info (fromROSMsg(cloud_projected) + pointColInd / pointRange / start- / endRingIndex) go to the device buffers
// rolo_project_frame would have left there; rolo_extract_features then runs as usual.
int rolo_front_load_projection(rolo_ctx* c, const rolo_front_params* P, const float* extracted, const int32_t* point_col_ind, const float* point_range,
                               const int32_t* start_ring, const int32_t* end_ring, int n_valid) {
  if (!c || !P || n_valid < 0 || !start_ring || !end_ring || (n_valid > 0 && (!extracted || !point_col_ind || !point_range))) return ROLO_EINVAL;
  if ((size_t)n_valid > (size_t)P->n_scan * P->horizon_scan) { ctx_set_error("more valid points than range-image pixels"); return ROLO_EINVAL; }
  // The kernels use these values as array indices: a malformed message (or an N_SCAN / Horizon_SCAN mismatch between the projection and the
  // feature process) must not become an out-of-bounds device access that takes the whole GPU context down. cloudExtraction writes
  // startRingIndex[i] = count - 1 + 5 before ring i and endRingIndex[i] = count - 1 - 5 after it (imageProjection.cpp:481-503), count running
  // from 0 to N; pointColInd is a column of the range image.
  {
    int prev = 0;
    for (int i = 0; i < P->n_scan; i++) {
      const long long before = (long long)start_ring[i] - 4, after = (long long)end_ring[i] + 6;
      if (before < prev || after < before || after > n_valid) { ctx_set_error("startRingIndex / endRingIndex are not the running counts of a projection of n_valid points"); return ROLO_EINVAL; }
      prev = (int)after;
    }
    for (int i = 0; i < n_valid; i++)
      if (point_col_ind[i] < 0 || point_col_ind[i] >= P->horizon_scan) { ctx_set_error("pointColInd outside [0, Horizon_SCAN)"); return ROLO_EINVAL; }
  }
  Front* f = nullptr;
  int rc = front_prepare(c, P, 0, 4, false, &f);
  if (rc) return rc;
  hipStream_t s = ctx_stream(c);
  const int NS = P->n_scan;
  const size_t np = (size_t)NS * P->horizon_scan + 2 * FRONT_GUARD;
  // guard cells and everything behind N are zero, as after a projection
  FCHK(hipMemsetAsync(f->col, 0, sizeof(int) * np, s));
  FCHK(hipMemsetAsync(f->range, 0, sizeof(float) * np, s));
  if (n_valid) {
    FCHK(hipMemcpyAsync(f->extracted + FRONT_GUARD, extracted, sizeof(float4) * (size_t)n_valid, hipMemcpyHostToDevice, s));
    FCHK(hipMemcpyAsync(f->col + FRONT_GUARD, point_col_ind, sizeof(int) * (size_t)n_valid, hipMemcpyHostToDevice, s));
    FCHK(hipMemcpyAsync(f->range + FRONT_GUARD, point_range, sizeof(float) * (size_t)n_valid, hipMemcpyHostToDevice, s));
  }
  FCHK(hipMemcpyAsync(f->start_ring, start_ring, sizeof(int) * (size_t)NS, hipMemcpyHostToDevice, s));
  FCHK(hipMemcpyAsync(f->end_ring, end_ring, sizeof(int) * (size_t)NS, hipMemcpyHostToDevice, s));
  FCHK(hipMemcpyAsync(f->counters, &n_valid, sizeof(int), hipMemcpyHostToDevice, s));
  FCHK(hipStreamSynchronize(s));   // n_valid is a stack variable; pageable copies are staged anyway
  f->n_valid = n_valid;
  f->projected = true;
  f->extract_cleared = false; f->precleared_np = 0;   // the arrays hold this projection now
  return ROLO_OK;
}

int rolo_extract_features(rolo_ctx* c, const rolo_front_params* P, float* corner, int* n_corner, float* surface, int* n_surface,
                          float* curvature, int32_t* neighbor_picked, int32_t* label) {
  if (!c || !P || !n_corner || !n_surface) return ROLO_EINVAL;
  void** slot = ctx_front_slot(c);
  Front* f = static_cast<Front*>(*slot);
  if (!f || !f->projected) { ctx_set_error("rolo_extract_features needs a preceding rolo_project_frame"); return ROLO_ESTATE; }
  FCHK(hipSetDevice(ctx_device(c)));
  hipStream_t s = ctx_stream(c);
  const int n = f->n_valid;
  int rc = front_extract_enqueue(f, P, s);
  if (rc) return rc;
  int cnts[3] = {0, 0, 0};
  FCHK(hipMemcpyAsync(cnts, f->counters, sizeof(int) * 3, hipMemcpyDeviceToHost, s));
  FCHK(hipStreamSynchronize(s));
  *n_corner = cnts[1]; *n_surface = cnts[2];
  if (corner && cnts[1]) FCHK(hipMemcpyAsync(corner, f->corner_out, sizeof(float4) * (size_t)cnts[1], hipMemcpyDeviceToHost, s));
  if (surface && cnts[2]) FCHK(hipMemcpyAsync(surface, f->surf_out, sizeof(float4) * (size_t)cnts[2], hipMemcpyDeviceToHost, s));
  if (curvature && n) FCHK(hipMemcpyAsync(curvature, f->curv + FRONT_GUARD, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, s));
  if (neighbor_picked && n) FCHK(hipMemcpyAsync(neighbor_picked, f->picked + FRONT_GUARD, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
  if (label && n) FCHK(hipMemcpyAsync(label, f->label + FRONT_GUARD, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
  FCHK(hipStreamSynchronize(s));
  return ROLO_OK;
}

}  // extern "C"
