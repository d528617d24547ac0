// K6 — target voxel map on gfx950.
// Replaces VmfVoxelMap::create_voxelmap / polar_coord / voxel_coord / AdditiveVmfVoxel / lookup_voxel
// (reference include/rot_gicp/gicp/vmp_voxel.hpp:93-108,167-233): a serial std::unordered_map<Vector3i,
// shared_ptr<VmfVoxel>> insert loop.
//
// MI355X design: an open-addressing hash table in HBM whose slot is ONE 64-bit word — the three voxel
// coordinates packed 3 x 21 bits — so a single atomicCAS both claims a slot and publishes the whole key. The
// winner of a slot takes a compact voxel id from a counter and zeroes that voxel's 96-byte record. A second
// kernel accumulates points into the records: consecutive lanes that hit the same voxel (targets arrive in scan
// order, so runs are long) are combined inside the wavefront first, so HBM sees one fp64 atomic add per run
// instead of one per point. A third kernel finalises (mean /= n, cov /= n, w = sqrt(n)). Lookups in the pass
// kernels are then: pack key -> hash -> probe 8-byte slots -> 96-byte record gather.
// Voxel coordinates are evaluated in fp64 with true divisions, exactly the reference's expressions, so the
// integer keys are bit-identical to the CPU path (up to libm ulps of atan2/acos at bin edges).
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include "voxel_dev.hpp"
#include <climits>

namespace rolo {

namespace {

__global__ __launch_bounds__(256) void voxel_insert_kernel(const float4* __restrict__ pts, int n, VoxelTable tab,
                                                          unsigned long long* tgt_keys, int* tgt_slot, int* counters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int kx, ky, kz;
  voxel_coord_dev(tab, (double)p.x, (double)p.y, (double)p.z, kx, ky, kz);
  unsigned long long key;
  if (!pack_key(kx, ky, kz, key)) { atomicExch(&counters[1], ROLO_EKEYRANGE); tgt_slot[i] = -1; tgt_keys[i] = KEY_EMPTY; return; }
  unsigned h = hash_key(key) & tab.mask;
  while (true) {
    unsigned long long prev = atomicCAS(&tab.keys[h], KEY_EMPTY, key);
    if (prev == KEY_EMPTY) {
      int id = atomicAdd(&counters[0], 1);
      tab.ids[h] = id;
      tab.id_keys[id] = key;
      double* r = tab.rec + (size_t)id * REC_DOUBLES;
#pragma unroll
      for (int d = 0; d < REC_DOUBLES; d++) r[d] = 0.0;
      break;
    }
    if (prev == key) break;
    h = (h + 1) & tab.mask;
  }
  tgt_slot[i] = (int)h;
  tgt_keys[i] = key;
}

// wave-level segmented combine: lanes holding the same voxel id as their predecessor fold into the run head
__global__ __launch_bounds__(256) void voxel_accum_kernel(const float4* __restrict__ pts, const double* __restrict__ cov, int n,
                                                         VoxelTable tab, const int* __restrict__ tgt_slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int id = -1;
  double v[10];
#pragma unroll
  for (int d = 0; d < 10; d++) v[d] = 0.0;
  if (i < n) {
    const int slot = tgt_slot[i];
    if (slot >= 0) {
      id = tab.ids[slot];
      const float4 p = pts[i];
      v[0] = (double)p.x; v[1] = (double)p.y; v[2] = (double)p.z;
#pragma unroll
      for (int d = 0; d < 6; d++) v[3 + d] = cov[(size_t)d * n + i];
      v[9] = 1.0;
    }
  }
  // inclusive segmented scan (Hillis-Steele) over runs of equal id; the last lane of each run holds the run total
  // after summing leftwards, so instead fold rightwards: each lane adds the value `off` lanes to its left when the
  // whole span [lane-off, lane] belongs to one run.
  const int prev_id = __shfl_up(id, 1, 64);
  const bool head = (lane == 0) || (prev_id != id);
  // distance to the run head
  unsigned long long head_mask = __ballot(head);
  const unsigned long long below = head_mask & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int head_lane = 63 - __clzll(below);
  const int dist = lane - head_lane;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
    for (int d = 0; d < 10; d++) {
      double o = __shfl_up(v[d], off, 64);
      if (dist >= off) v[d] += o;
    }
  }
  const int next_id = __shfl_down(id, 1, 64);
  const bool tail = (lane == 63) || (next_id != id);
  if (tail && id >= 0) {
    double* r = tab.rec + (size_t)id * REC_DOUBLES;
#pragma unroll
    for (int d = 0; d < 9; d++) atomicAdd(&r[d], v[d]);
    atomicAdd(&r[10], v[9]);
  }
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
  if (j < n_sorted) {
    const float4 p = sorted[j];
    if (__float_as_int(p.w) != INT_MAX) {
      int kx, ky, kz;
      voxel_coord_dev(tab, (double)p.x, (double)p.y, (double)p.z, kx, ky, kz);
      if (pack_key(kx, ky, kz, key)) valid = true;
      else { atomicExch(&counters[1], ROLO_EKEYRANGE); key = KEY_EMPTY; }
    }
  }
  const unsigned long long prev_key = ((unsigned long long)(unsigned)__shfl_up((int)(key >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(key & 0xffffffffull), 1, 64);
  const bool head = valid && (lane == 0 || prev_key != key);
  int h = -1;
  if (head) {
    unsigned hh = hash_key(key) & tab.mask;
    while (true) {
      const unsigned long long prev = atomicCAS(&tab.keys[hh], KEY_EMPTY, key);
      if (prev == KEY_EMPTY) {
        const int id = atomicAdd(&counters[0], 1);
        tab.ids[hh] = id;
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
                                                                VoxelTable tab, const int* __restrict__ slot_sorted) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int id = -1;
  double v[10];
#pragma unroll
  for (int d = 0; d < 10; d++) v[d] = 0.0;
  if (j < n_sorted) {
    const int slot = slot_sorted[j];
    if (slot >= 0) {
      id = tab.ids[slot];   // written by another lane / workgroup of the insert kernel: visible after the kernel boundary
      const float4 p = sorted[j];
      const int i = __float_as_int(p.w);
      v[0] = (double)p.x; v[1] = (double)p.y; v[2] = (double)p.z;
#pragma unroll
      for (int d = 0; d < 6; d++) v[3 + d] = cov[(size_t)d * n + i];
      v[9] = 1.0;
    }
  }
  const int prev_id = __shfl_up(id, 1, 64);
  const bool head = (lane == 0) || (prev_id != id);
  const unsigned long long head_mask = __ballot(head);
  const unsigned long long below = head_mask & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int head_lane = 63 - __clzll(below);
  const int dist = lane - head_lane;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
    for (int d = 0; d < 10; d++) {
      double o = __shfl_up(v[d], off, 64);
      if (dist >= off) v[d] += o;
    }
  }
  const int next_id = __shfl_down(id, 1, 64);
  const bool tail = (lane == 63) || (next_id != id);
  if (tail && id >= 0) {
    double* r = tab.rec + (size_t)id * REC_DOUBLES;
#pragma unroll
    for (int d = 0; d < 9; d++) atomicAdd(&r[d], v[d]);
    atomicAdd(&r[10], v[9]);
  }
}

__global__ __launch_bounds__(256) void voxel_finalize_kernel(VoxelTable tab, const int* counters) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= counters[0]) return;
  double* r = tab.rec + (size_t)id * REC_DOUBLES;
  const double n = r[10];
#pragma unroll
  for (int d = 0; d < 9; d++) r[d] = r[d] / n;  // vmp_voxel.hpp:102,107 (mean_dir /= n ; cov /= n)
  r[9] = sqrt(n);                               // w = sqrt(num_points), rot_vgicp_impl.hpp:336
}

__global__ __launch_bounds__(256) void voxel_keys_kernel(const float4* __restrict__ pts, int n, VoxelTable tab, int32_t* keys3) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  int kx, ky, kz;
  voxel_coord_dev(tab, (double)p.x, (double)p.y, (double)p.z, kx, ky, kz);
  keys3[3 * (size_t)i] = kx; keys3[3 * (size_t)i + 1] = ky; keys3[3 * (size_t)i + 2] = kz;
}

}  // namespace

hipError_t launch_voxel_build(const CloudDev& tgt, VoxelTable tab, unsigned long long* tgt_keys, int* tgt_slot, int* counters, bool morton_order, hipStream_t s) {
  hipError_t e = hipMemsetAsync(tab.keys, 0xFF, sizeof(unsigned long long) * ((size_t)tab.mask + 1), s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(counters, 0, 2 * sizeof(int), s);
  if (e != hipSuccess) return e;
  const int grid = (tgt.n + 255) / 256;
  if (morton_order && tgt.have_sorted) {   // Morton order of the neighbour search: tgt_slot (>= 8 * n_leaves entries) is indexed by sorted position
    const int n_sorted = KNN_LEAF * tgt.n_leaves, gs = (n_sorted + 255) / 256;
    voxel_insert_sorted_kernel<<<gs, 256, 0, s>>>(tgt.sorted, n_sorted, tab, tgt_slot, counters);
    voxel_accum_sorted_kernel<<<gs, 256, 0, s>>>(tgt.sorted, tgt.cov, tgt.n, n_sorted, tab, tgt_slot);
  } else {                                 // input order
    voxel_insert_kernel<<<grid, 256, 0, s>>>(tgt.xyz, tgt.n, tab, tgt_keys, tgt_slot, counters);
    voxel_accum_kernel<<<grid, 256, 0, s>>>(tgt.xyz, tgt.cov, tgt.n, tab, tgt_slot);
  }
  voxel_finalize_kernel<<<grid, 256, 0, s>>>(tab, counters);
  return hipGetLastError();
}

hipError_t launch_voxel_keys(const float4* pts, int n, VoxelTable tab, int32_t* keys3, hipStream_t s) {
  voxel_keys_kernel<<<(n + 255) / 256, 256, 0, s>>>(pts, n, tab, keys3);
  return hipGetLastError();
}

}  // namespace rolo
