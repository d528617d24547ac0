// K6 — target voxel map on gfx950.
// Replaces VmfVoxelMap::create_voxelmap / polar_coord / voxel_coord / AdditiveVmfVoxel / lookup_voxel
// (reference include/rot_gicp/gicp/vmp_voxel.hpp:93-108,167-233): a serial std::unordered_map<Vector3i,
// shared_ptr<VmfVoxel>> insert loop.
//
// MI355X design: an open-addressing hash table in HBM whose slot is ONE 64-bit word — the three voxel
// coordinates packed 3 x 21 bits — so a single atomicCAS both claims a slot and publishes the whole key. The
// winner of a slot takes a compact voxel id from a counter and zeroes that voxel's 96-byte record. A second
// kernel accumulates points into the records: consecutive lanes that hit the same voxel (targets arrive in scan
// order, so runs are long) are combined inside the wavefront first, so HBM sees one atomic add per run
// instead of one per point. The sums are 64-bit FIXED-POINT integers (scale = the power of two that keeps n * max|value| below
// 2^62: float coordinates are then represented exactly, covariances to ~6e-14): integer addition is associative, so every
// voxel's mean and covariance come out bit-identical run after run and in either point order (fp64 atomics summed in arrival
// order). A third kernel finalises (mean /= n, cov /= n, w = sqrt(n)). Lookups in the pass
// kernels are then: pack key -> hash -> probe 8-byte slots -> 96-byte record gather.
// Voxel coordinates are evaluated in fp64 with true divisions, exactly the reference's expressions, so the
// integer keys are bit-identical to the CPU path (up to libm ulps of atan2/acos at bin edges).
#include "rolo_internal.hpp"
#include "lm_begin.hpp"
#include "dev_math.hpp"
#include "voxel_dev.hpp"
#include <climits>

namespace rolo {

namespace {

// clears the hash table and the four counters in one launch (two memset launches and a one-thread kernel before); with the bounding box
// at hand counters[3] gets max |coordinate| right here
__global__ __launch_bounds__(256) void voxel_clear_kernel(unsigned long long* __restrict__ keys, size_t n_slots, const int* __restrict__ bbox6, int* counters) {
  voxel_clear_body(keys, n_slots, bbox6, counters, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}
__global__ __launch_bounds__(256) void maxabs_kernel(const float4* __restrict__ pts, int n, int* counters) {
  __shared__ int sm[4];
  float m = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const float4 p = pts[i]; m = fmaxf(m, fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z))); }
  int mb = __float_as_int(m);   // non-negative floats order like their bit patterns
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mb = max(mb, __shfl_xor(mb, off, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mb;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&counters[3], max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

__global__ __launch_bounds__(256) void voxel_insert_kernel(const float4* __restrict__ pts, int n, VoxelTable tab,
                                                          unsigned long long* tgt_keys, int* tgt_slot, int* counters) {
  voxel_insert_point(tab, pts, n, blockIdx.x * blockDim.x + threadIdx.x, tgt_keys, tgt_slot, counters);
}

// wave-level segmented combine: lanes holding the same voxel id as their predecessor fold into the run head
__global__ __launch_bounds__(256) void voxel_accum_kernel(const float4* __restrict__ pts, const double* __restrict__ cov, int n,
                                                         VoxelTable tab, const int* __restrict__ tgt_slot, int* counters, int fixed_cov) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int id = -1;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  double c[6] = {0, 0, 0, 0, 0, 0};
  if (i < n) {
    const int slot = tgt_slot[i];
    if (slot >= 0) {
      id = slot_id(tab, (unsigned)slot);
      p = pts[i];
#pragma unroll
      for (int d = 0; d < 6; d++) c[d] = cov[(size_t)d * n + i];
    }
  }
  // (fixed_cov is a kernel argument: uniform — every thread of the workgroup takes the same call; the workgroup form meets the runs' sums in LDS first, voxel_dev.hpp)
  if (fixed_cov) accumulate_point_wg<256>(tab, id, p, c, fix_scales(n, counters), counters + 1);
  else accumulate_point(tab, id, p, c, fix_scales(n, counters), false, counters + 1);
}

// ---- the same two kernels over the target in MORTON order (CloudDev::sorted of the neighbour search) -----------------
// Spatially adjacent points are adjacent lanes: a wavefront sees a handful of voxels instead of ~30, so (a) only the
// first lane of every run of equal keys probes / claims the hash slot and hands it to the run, and (b) the wave-level
// combine folds many more points per fp64 atomic. Pays when voxels are coarse (the production POLAR grid: ~30 points per
// voxel, frame latency 0.91 -> 0.85 ms); with ~8 points per voxel (0.5 m uniform leaves on a dense frame) the scattered
// covariance gathers cost more than the atomics saved (76 -> 80 us), so the caller picks by points per voxel.
__global__ __launch_bounds__(256) void voxel_insert_sorted_kernel(const float4* __restrict__ sorted, int n_sorted, VoxelTable tab,
                                                                 int* __restrict__ slot_sorted, int* counters) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  unsigned long long key = KEY_EMPTY;
  bool valid = false;
  {
    float4 p = j < n_sorted ? sorted[j] : make_float4(1.f, 1.f, 1.f, __int_as_float(INT_MAX));
    const bool real = __float_as_int(p.w) != INT_MAX;
    if (!real) { p.x = 1.f; p.y = 1.f; p.z = 1.f; }   // padding holds +inf
    int kx, ky, kz; bool near_edge;
    voxel_coord_dev_edge(tab, (double)p.x, (double)p.y, (double)p.z, kx, ky, kz, near_edge);
    note_point(real, near_edge, counters);
    if (real) {
      if (pack_key(kx, ky, kz, key)) valid = true;
      else { atomicMin(&counters[1], ROLO_EKEYRANGE); key = KEY_EMPTY; }
    }
  }
  const unsigned long long prev_key = ((unsigned long long)(unsigned)__shfl_up((int)(key >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(key & 0xffffffffull), 1, 64);
  const bool head = valid && (lane == 0 || prev_key != key);
  int h = -1;
  if (head) {
    unsigned hh = hash_key(key) & tab.mask;
    while (true) {
      const unsigned long long prev = atomicCAS(slot_key(tab, hh), KEY_EMPTY, key);
      if (prev == KEY_EMPTY) {
        const int id = atomicAdd(&counters[0], 1);
        set_slot_id(tab, hh, id);
        tab.id_keys[id] = key;
        double* r = tab.rec + (size_t)id * REC_DOUBLES;
#pragma unroll
        for (int d = 0; d < REC_DOUBLES; d++) r[d] = 0.0;
        break;
      }
      if (prev == key) break;
      hh = (hh + 1) & tab.mask;
    }
    h = (int)hh;
  }
  // every lane of a run takes the slot of the run's head (the nearest head at or below it)
  const unsigned long long heads = __ballot(head);
  const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int head_lane = below ? 63 - __clzll(below) : lane;
  const int hs = __shfl(h, head_lane, 64);
  if (j < n_sorted) slot_sorted[j] = valid ? hs : -1;
}

__global__ __launch_bounds__(256) void voxel_accum_sorted_kernel(const float4* __restrict__ sorted, const double* __restrict__ cov, int n, int n_sorted,
                                                                VoxelTable tab, const int* __restrict__ slot_sorted, int* counters, int fixed_cov) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int id = -1;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  double c[6] = {0, 0, 0, 0, 0, 0};
  if (j < n_sorted) {
    const int slot = slot_sorted[j];
    if (slot >= 0) {
      id = slot_id(tab, (unsigned)slot);   // written by another lane / workgroup of the insert kernel: visible after the kernel boundary
      p = sorted[j];
      const int i = __float_as_int(p.w);
#pragma unroll
      for (int d = 0; d < 6; d++) c[d] = cov[(size_t)d * n + i];
    }
  }
  if (fixed_cov) accumulate_point_wg<256>(tab, id, p, c, fix_scales(n, counters), counters + 1);
  else accumulate_point(tab, id, p, c, fix_scales(n, counters), false, counters + 1);
}

__global__ __launch_bounds__(256) void voxel_finalize_kernel(VoxelTable tab, const int* counters, int n_pts, int fixed_cov, int* pub_counters, LmState* lm_state, const FrameArgs* lm_args) {
  ROLO_ALL_KERNEL_PRIO();
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (lm_state && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) frame_begin_dev(lm_state, lm_args);   // the frame's LM state (was: frame_begin_kernel, its own launch); a lane of the last, mostly idle workgroup
  if (pub_counters && id < 4) pub_counters[id] = counters[id];   // the counters are final before this launch
  if (id >= counters[0]) return;
  double* r = tab.rec + (size_t)id * REC_DOUBLES;
  const long long* q = reinterpret_cast<const long long*>(r);
  const FixScale S = fix_scales(n_pts, counters);
  const double n = (double)q[10];
  double v[9];
#pragma unroll
  for (int d = 0; d < 3; d++) v[d] = ((double)q[d] / S.pos) / n;                                // vmp_voxel.hpp:102 (mean_dir /= n)
#pragma unroll
  for (int d = 3; d < 9; d++) v[d] = (fixed_cov ? (double)q[d] / S.cov : r[d]) / n;            // :107 (cov /= n)
#pragma unroll
  for (int d = 0; d < 9; d++) r[d] = v[d];
  r[9] = sqrt(n);                               // w = sqrt(num_points), rot_vgicp_impl.hpp:336
  r[10] = n;
}

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float4* __restrict__ pts, int n, VoxelTable tab, int32_t* keys3) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int kx, ky, kz; bool near_edge;
  voxel_coord_dev_edge(tab, (double)p.x, (double)p.y, (double)p.z, kx, ky, kz, near_edge);   // the keys the map build files the target under
  keys3[3 * (size_t)i] = kx; keys3[3 * (size_t)i + 1] = ky; keys3[3 * (size_t)i + 2] = kz;
}

}  // namespace

hipError_t launch_voxel_build(const CloudDev& tgt, VoxelTable tab, unsigned long long* tgt_keys, int* tgt_slot, int* counters, bool morton_order, bool fixed_cov,
                              const int* bbox6, bool prefused, hipStream_t s, int* pub_counters, LmState* lm_state, const FrameArgs* lm_args) {
  static_assert(KEY_EMPTY == ~0ull, "voxel_clear_kernel and the 0xFF memsets elsewhere agree on the empty key");
  const int grid = (tgt.n + 255) / 256;
  if (prefused) {
    voxel_finalize_kernel<<<grid, 256, 0, s>>>(tab, counters, tgt.n, 1, pub_counters, lm_state, lm_args);
    return hipGetLastError();
  }
  voxel_clear_kernel<<<256, 256, 0, s>>>(tab.keys, (size_t)tab.mask + 1, bbox6, counters);
  if (!bbox6) maxabs_kernel<<<64, 256, 0, s>>>(tgt.xyz, tgt.n, counters);
  if (morton_order && tgt.have_sorted) {   // Morton order of the neighbour search: tgt_slot (>= 8 * n_leaves entries) is indexed by sorted position
    const int n_sorted = KNN_LEAF * tgt.n_leaves, gs = (n_sorted + 255) / 256;
    voxel_insert_sorted_kernel<<<gs, 256, 0, s>>>(tgt.sorted, n_sorted, tab, tgt_slot, counters);
    voxel_accum_sorted_kernel<<<gs, 256, 0, s>>>(tgt.sorted, tgt.cov, tgt.n, n_sorted, tab, tgt_slot, counters, fixed_cov ? 1 : 0);
  } else {                                 // input order
    voxel_insert_kernel<<<grid, 256, 0, s>>>(tgt.xyz, tgt.n, tab, tgt_keys, tgt_slot, counters);
    voxel_accum_kernel<<<grid, 256, 0, s>>>(tgt.xyz, tgt.cov, tgt.n, tab, tgt_slot, counters, fixed_cov ? 1 : 0);
  }
  voxel_finalize_kernel<<<grid, 256, 0, s>>>(tab, counters, tgt.n, fixed_cov ? 1 : 0, pub_counters, lm_state, lm_args);
  return hipGetLastError();
}

hipError_t launch_voxel_keys(const float4* pts, int n, VoxelTable tab, int32_t* keys3, hipStream_t s) {
  voxel_keys_kernel<<<(n + 255) / 256, 256, 0, s>>>(pts, n, tab, keys3);
  return hipGetLastError();
}

}  // namespace rolo
