// K5 — per-point k-nearest-neighbour covariances on gfx950.
// Replaces RotVGICP::calculate_covariances (reference include/rot_gicp/gicp/impl/rot_vgicp_impl.hpp:421-496):
// pcl::search::KdTree::nearestKSearch (FLANN, exact, float L2, query included) + centred 4x20 outer product +
// Eigen::JacobiSVD + regularisation.
//
// MI355X design: no pointer-chasing kd-tree. The cloud is sorted along a Hilbert curve (hand-written LSD radix sort of the
// 27-bit keys, below), groups of KNN_LEAF = 16 consecutive points become the leaves of an *implicit* complete
// binary BVH stored in heap order (children of h are 2h, 2h+1; both child boxes sit in one 64-byte line), built
// bottom-up in LDS. The search itself is a wavefront-wide packet traversal (knn_walk.hpp): 64 curve-adjacent
// queries share one walk. Exactness: distances are float ((dx*dx)+(dy*dy))+(dz*dz) with FMA contraction off,
// box bounds use the same operation order so they are true lower bounds, ties are explored (<=) and broken by
// original point index — the neighbour set is the same pure function of the cloud the oracle computes.
// Two kernels: the walk (58 VGPRs: 8 wavefronts per SIMD, so searches of several contexts share the chip) hands the
// neighbour indices to the covariance kernel (whose 3x3 fp64 SVD needs ~150 VGPRs); fused they ran at 4 per SIMD.
#include <cstdio>
#include <cstring>
#include <string.h>
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include "voxel_dev.hpp"
#ifdef ROLO_KNN_ROCPRIM_SORT
#include <rocprim/rocprim.hpp>   // only the A/B build of the key sort uses a library kernel
#endif
#include <cfloat>
#include <climits>

namespace rolo {

namespace {


// Every kernel of the search takes a KnnPair: the source and the target cloud of a registration go through ONE chain of
// launches (workgroups [0, split) belong to cloud 0, the rest to cloud 1) — half the launches of two separate chains,
// both searches start together and fill the chip that one search alone leaves half empty. A single cloud is a pair
// with n_clouds = 1.
constexpr int BBOX_BLOCKS = 128;   // per cloud; each writes its partial box to bbox[BBOX_PART + ...] — no atomics, no init launch
constexpr int BBOX_PART = 16;

__global__ __launch_bounds__(256) void bbox_kernel(KnnPair A, int* bbox) {
  __shared__ float smn[4][3], smx[4][3];
  const int which = blockIdx.x / BBOX_BLOCKS, blk = blockIdx.x - which * BBOX_BLOCKS;
  const float4* __restrict__ p = A.c[which].xyz;
  const int n = A.c[which].n;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = blk * blockDim.x + threadIdx.x; i < n; i += BBOX_BLOCKS * blockDim.x) {
    float4 q = p[i];
    mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
    mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
  }
#pragma unroll
  for (int d = 0; d < 3; d++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64));
      mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64));
    }
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; d++) { smn[wv][d] = mn[d]; smx[wv][d] = mx[d]; }
  }
  __syncthreads();
  // the block's partial box; the key kernel folds the BBOX_BLOCKS partials of a cloud in its prologue (a per-wave atomic on 6 hot words
  // serialised 2048 waves: 141 us; one atomic per block: 7 us and an init launch)
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    const float a = fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d]));
    const float b = fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]));
    int* part = bbox + BBOX_PART + (which * BBOX_BLOCKS + blk) * 6;
    part[d] = f2ord(a);
    part[3 + d] = f2ord(b);
  }
}

ROLO_DEV uint32_t expand10(uint32_t v) {
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// 3-D Hilbert index of a 10-bit cell (Skilling's transpose algorithm): consecutive indices are face-adjacent cells, so 64 consecutive
// points of the sorted cloud always form ONE connected blob — a Morton run breaks into separated pieces at every power-of-two boundary,
// and a wavefront whose 64 queries straddle such a jump walks two neighbourhoods.
ROLO_DEV uint32_t hilbert30(uint32_t x, uint32_t y, uint32_t z) {
  uint32_t X[3] = {x, y, z};
  for (uint32_t Q = 512; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (X[i] & Q) X[0] ^= P;
      else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  X[1] ^= X[0]; X[2] ^= X[1];
  uint32_t t = 0;
  for (uint32_t Q = 512; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
  X[0] ^= t; X[1] ^= t; X[2] ^= t;
  return (expand10(X[0]) << 2) | (expand10(X[1]) << 1) | expand10(X[2]);
}

// keys: curve index (Hilbert; Morton in the -DROLO_KNN_MORTON A/B build) + the cloud number in the top bit, so one sort of both clouds leaves each cloud sorted in its own
// range [0, n0) / [n0, n0 + n1) of the arrays
// ---- key sort: a stable LSD radix sort in three 9-bit passes, written for this path (round 1 used rocPRIM's merge sort: 17 launches) --------
// The keys are 27 bits (the top 26 bits of the 30-bit curve index + the cloud bit: the walk does not care about the low bits of the index)
// and there are at most a few hundred thousand of them, so the array is cut into SORT_NB = 128 tiles (64 tiles x 4 elements per thread: build
// 0.090 ms, 128 x 2: 0.076, 256 x 2 on 512 threads: 0.078), one 1024-thread workgroup each. A pass is a histogram launch (per tile: 512
// digit counts by LDS atomics, plain stores; the first pass's are counted by the key kernel, which uses the same tiling) and a scatter
// launch: the workgroup of tile b reads the 128 x 512 counts (their scan is a prologue, not a kernel), ranks its elements stably — waves own
// consecutive 64-element runs, equal digits inside a run are matched with 9 ballots, runs are ordered by a 16-step prefix per digit in LDS —
// and scatters keys and values. 1 + 5 launches (four 8-bit passes of 31-bit keys: 1 + 7, build 0.071 against 0.065 ms), the result identical
// to a stable sort by key (ties by original index) — what rocPRIM's radix_sort_pairs returns. (Counting the next pass's digits inside the
// scatter, one global integer atomic per element on the counter of its destination tile, saved the histogram launches and cost 18 us per
// pass: 262 k device-scope atomics cross the fabric.)
#ifndef ROLO_SORT_NB
#define ROLO_SORT_NB 128
#endif
#ifndef ROLO_SORT_T
#define ROLO_SORT_T 1024
#endif
#ifndef ROLO_SORT_EPT
#define ROLO_SORT_EPT 2
#endif
#ifndef ROLO_SORT_BITS
#define ROLO_SORT_BITS 9   // digit width; SORT_PASSES x SORT_BITS >= key bits. 3 x 9 = 27: the curve index is cut to 26 bits + the cloud bit (the walk is
#endif                     // indifferent to the low bits of the curve index: 30 / 27 / 24 bits 0.1917 / 0.1917 / 0.1920 ms); 4 x 8 bits before
#ifndef ROLO_SORT_PASSES
#define ROLO_SORT_PASSES 3
#endif
constexpr int SORT_BITS = ROLO_SORT_BITS, SORT_PASSES = ROLO_SORT_PASSES, SORT_D = 1 << SORT_BITS, KEY_BITS = SORT_BITS * SORT_PASSES;
static_assert(KEY_BITS >= 25 && KEY_BITS <= 31, "curve index + cloud bit");
constexpr int SORT_NB = ROLO_SORT_NB, SORT_T = ROLO_SORT_T;
static_assert(SORT_T % SORT_D == 0 || SORT_D % SORT_T == 0, "threads share the count prologue by digit");
static_assert(SORT_T >= 12 * 64, "the key kernel folds the 12 box components with one wavefront each");
ROLO_DEV int sort_tile(int n_total) { return ((n_total + SORT_NB - 1) / SORT_NB + SORT_T - 1) / SORT_T * SORT_T; }

constexpr int SORT_EPT = ROLO_SORT_EPT;   // elements per thread and round: a wave ranks 256 consecutive elements between two workgroup barriers
__global__ __launch_bounds__(SORT_T) void sort_scatter_kernel(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout,
                                                             uint32_t* __restrict__ vout, int n, int tile, int pass, int* __restrict__ cnt /* [SORT_PASSES][SORT_NB][SORT_D] */,
                                                             VoxelFuse vf) {
  ROLO_ALL_KERNEL_PRIO();
  if ((int)blockIdx.x >= SORT_NB) {   // VoxelFuse: this pass's share of the target points goes into the voxel hash table on the CUs the sort leaves idle
    const int nt = vf.n_tgt, quarter = (nt + SORT_PASSES - 1) / SORT_PASSES;
    const int i = pass * quarter + ((int)blockIdx.x - SORT_NB) * SORT_T + (int)threadIdx.x;
    voxel_insert_point(vf.tab, vf.tgt_xyz, i < (pass + 1) * quarter ? nt : 0, i, vf.tgt_keys, vf.tgt_slot, vf.counters);
    return;
  }
  __shared__ int wcnt[SORT_T / 64][SORT_D];
  __shared__ int base[SORT_D], run[SORT_D], wsum[SORT_D / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, blk = blockIdx.x;
  const int* __restrict__ c = cnt + (size_t)pass * SORT_NB * SORT_D;
  for (int k = tid; k < (SORT_T / 64) * SORT_D; k += SORT_T) (&wcnt[0][0])[k] = 0;
  // digit totals over all tiles and over the tiles before this one: every thread sums a share of the tiles, the shares meet in LDS
  constexpr int Q = SORT_T / SORT_D;
  static_assert(Q >= 1, "at least one thread per digit");
  __shared__ int part[2][Q][SORT_D];
  {
    const int d = tid & (SORT_D - 1), q = tid / SORT_D;
    int tot = 0, bef = 0;
#pragma unroll 16
    for (int b = q; b < SORT_NB; b += Q) { const int v = c[b * SORT_D + d]; tot += v; bef += b < blk ? v : 0; }
    part[0][q][d] = tot; part[1][q][d] = bef;
  }
  __syncthreads();
  int t = 0;
  if (tid < SORT_D) {
    int before = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) { t += part[0][q][tid]; before += part[1][q][tid]; }
    run[tid] = before;   // elements with this digit in earlier tiles (becomes the running offset inside the tile below)
    // exclusive scan of the digit totals: inside the wave by shuffles, across the waves through LDS
    int incl = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) wsum[wv] = incl;
    t = incl - t;   // exclusive inside the wave
  }
  __syncthreads();
  if (tid < SORT_D) { int add = 0; for (int w = 0; w < wv; w++) add += wsum[w]; base[tid] = t + add; }
  __syncthreads();
  const int shift = SORT_BITS * pass;
  for (int r0 = 0; r0 < tile; r0 += SORT_T * SORT_EPT) {
    // the wave's run of this round: 256 consecutive elements, lane-contiguous in SORT_EPT sub-rounds of 64
    uint32_t key[SORT_EPT]; int off[SORT_EPT], e[SORT_EPT];
#pragma unroll
    for (int j = 0; j < SORT_EPT; j++) {
      e[j] = blk * tile + r0 + (wv * SORT_EPT + j) * 64 + lane;
      const bool valid = e[j] < n && (r0 + (wv * SORT_EPT + j) * 64 + lane) < tile;
      key[j] = valid ? kin[e[j]] : 0u;
      if (!valid) e[j] = -1;
    }
#pragma unroll
    for (int j = 0; j < SORT_EPT; j++) {
      const bool valid = e[j] >= 0;
      const int d = (int)((key[j] >> shift) & (uint32_t)(SORT_D - 1));
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < SORT_BITS; bit++) {
        const bool one = (d >> bit) & 1;
        const unsigned long long bl = __ballot(one);
        m &= one ? bl : ~bl;
      }
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      int old = 0;
      if (valid && rank == 0) { old = wcnt[wv][d]; wcnt[wv][d] = old + __popcll(m); }   // one leader per (wave, digit): no atomics needed
      const int leader = valid ? __ffsll((long long)m) - 1 : lane;
      old = __shfl(old, leader, 64);
      off[j] = old + rank;   // position among this wave's elements of digit d so far
    }
    __syncthreads();
    if (tid < SORT_D) {
      int acc = run[tid];
#pragma unroll
      for (int w = 0; w < SORT_T / 64; w++) { const int v = wcnt[w][tid]; wcnt[w][tid] = acc; acc += v; }
      run[tid] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_EPT; j++) {
      if (e[j] >= 0) {
        const int d = (int)((key[j] >> shift) & (uint32_t)(SORT_D - 1));
        const int dest = base[d] + wcnt[wv][d] + off[j];
        kout[dest] = key[j]; vout[dest] = vin[e[j]];
      }
    }
    if (r0 + SORT_T * SORT_EPT < tile) {
      __syncthreads();
      for (int k = tid; k < (SORT_T / 64) * SORT_D; k += SORT_T) (&wcnt[0][0])[k] = 0;
      __syncthreads();
    }
  }
}

// keys of both clouds of the pair (cloud number = bit 30), one workgroup per sort tile; also the tile's digit counts of the first sort pass
constexpr int VF_CLEAR_BLOCKS = 64;   // VoxelFuse: workgroups behind the key kernel's that clear the target's voxel table
__global__ __launch_bounds__(SORT_T) void morton_kernel(KnnPair A, int* __restrict__ bbox, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int n_total, int tile,
                                                       int* __restrict__ cnt, VoxelFuse vf) {
  ROLO_ALL_KERNEL_PRIO();
  if ((int)blockIdx.x >= SORT_NB) {
    const size_t n_words = 2 * ((size_t)vf.tab.mask + 1);   // 16 bytes per slot: key + id (voxel_dev.hpp)
    for (size_t k = (size_t)((int)blockIdx.x - SORT_NB) * SORT_T + threadIdx.x; k < n_words; k += (size_t)VF_CLEAR_BLOCKS * SORT_T) vf.tab.keys[k] = KEY_EMPTY;
    if (blockIdx.x == SORT_NB && threadIdx.x < 3) vf.counters[threadIdx.x] = 0;   // counters[3] (max |coordinate|) is block 0's
    return;
  }
  __shared__ int hist[SORT_D];
  __shared__ float par[2][4];   // per cloud: min x, y, z and the scale
  const int tid = threadIdx.x, blk = blockIdx.x;
  if (tid < SORT_D) hist[tid] = 0;
  __shared__ int fin[12];
  {  // fold the partial boxes: wave w < 12 owns component w % 6 of cloud w / 6
    const int w = tid >> 6, lane = tid & 63;
    if (w < 6 * A.n_clouds) {
      const bool is_min = (w % 6) < 3;
      int v = is_min ? INT_MAX : INT_MIN;
      const int* __restrict__ bp = A.c[w / 6].bpart; const int nbp = A.c[w / 6].n_bpart;
      for (int b = lane; b < nbp; b += 64) { const int o = bp[b * 6 + (w % 6)]; v = is_min ? min(v, o) : max(v, o); }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = is_min ? min(v, o) : max(v, o); }
      if (lane == 0) { fin[w] = v; if (blk == 0) bbox[w] = v; }   // the final box, for the voxel map's scales (separate map build)
    }
  }
  __syncthreads();
  if (vf.enabled && blk == 0 && tid == 0) {   // VoxelFuse: max |coordinate| of the target for the fixed-point scales (voxel_clear_body's job otherwise)
    float m = 0.f;
    for (int k = 0; k < 6; k++) { const int o = fin[6 * vf.which + k]; m = fmaxf(m, fabsf(__int_as_float(o >= 0 ? o : o ^ 0x7fffffff))); }
    vf.counters[3] = __float_as_int(m);
  }
  if (tid < A.n_clouds) {
    const int* bb = fin + 6 * tid;
    const float mnx = ord2f(bb[0]), mny = ord2f(bb[1]), mnz = ord2f(bb[2]);
    const float ext = fmaxf(fmaxf(ord2f(bb[3]) - mnx, ord2f(bb[4]) - mny), ord2f(bb[5]) - mnz);
    par[tid][0] = mnx; par[tid][1] = mny; par[tid][2] = mnz; par[tid][3] = ext > 0.f ? 1024.0f / ext : 0.f;
  }
  __syncthreads();
  const int n0 = A.c[0].n;
  for (int r = tid; r < tile; r += SORT_T) {
    const int e = blk * tile + r;
    if (e >= n_total) break;
    const int which = e >= n0 ? 1 : 0;
    const int i = e - (which ? n0 : 0);
    const float4 q = A.c[which].xyz[i];
    const float mnx = par[which][0], mny = par[which][1], mnz = par[which][2], sc = par[which][3];
    const int ix = min(1023, max(0, (int)((q.x - mnx) * sc)));
    const int iy = min(1023, max(0, (int)((q.y - mny) * sc)));
    const int iz = min(1023, max(0, (int)((q.z - mnz) * sc)));
#ifndef ROLO_KNN_MORTON   // A/B builds: the Z-order curve this replaced (walk 0.274 ms against 0.230 ms for the 2 x 131 072-point pair)
    const uint32_t key = (hilbert30((uint32_t)ix, (uint32_t)iy, (uint32_t)iz) >> (31 - KEY_BITS)) | ((uint32_t)which << (KEY_BITS - 1));   // the top KEY_BITS - 1 bits of the curve index
#else
    const uint32_t key = ((expand10(ix) | (expand10(iy) << 1) | (expand10(iz) << 2)) >> (31 - KEY_BITS)) | ((uint32_t)which << (KEY_BITS - 1));
#endif
    keys[e] = key;
    vals[e] = (uint32_t)i;
    atomicAdd(&hist[key & (uint32_t)(SORT_D - 1)], 1);
  }
  __syncthreads();
  if (cnt && tid < SORT_D) cnt[blk * SORT_D + tid] = hist[tid];
}

// digit counts of sort pass `pass` for every tile of the (partially sorted) keys: LDS atomics, plain stores — counting the next pass's digits
// with global atomics inside the scatter cost 18 us per pass (262 k device-scope atomics cross the fabric), this launch costs 4
__global__ __launch_bounds__(SORT_T) void sort_hist_kernel(const uint32_t* __restrict__ keys, int n, int tile, int pass, int* __restrict__ cnt) {
  ROLO_ALL_KERNEL_PRIO();
  __shared__ int hist[SORT_D];
  const int tid = threadIdx.x, blk = blockIdx.x;
  if (tid < SORT_D) hist[tid] = 0;
  __syncthreads();
  const int shift = SORT_BITS * pass;
  for (int r = tid; r < tile; r += SORT_T) {
    const int e = blk * tile + r;
    if (e >= n) break;
    atomicAdd(&hist[(keys[e] >> shift) & (uint32_t)(SORT_D - 1)], 1);
  }
  __syncthreads();
  if (tid < SORT_D) cnt[(size_t)pass * SORT_NB * SORT_D + blk * SORT_D + tid] = hist[tid];
}

#ifdef ROLO_KNN_KD_REFINE
// 256 threads, one point each (padding: +inf coordinates, sorts last on every axis): returns the point of this thread's slot after the splits
ROLO_DEV float4 kd_refine_block(float4 cur) {
  __shared__ float4 pts[256];
  __shared__ unsigned long long key[256];
  __shared__ float w_lo[4][3], w_hi[4][3];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int seg = 256; seg >= 2 * KNN_LEAF; seg >>= 1) {
    const bool pad = __float_as_int(cur.w) == INT_MAX;
    float lo[3] = {pad ? INFINITY : cur.x, pad ? INFINITY : cur.y, pad ? INFINITY : cur.z};
    float hi[3] = {pad ? -INFINITY : cur.x, pad ? -INFINITY : cur.y, pad ? -INFINITY : cur.z};
    const int in_wave = seg < 64 ? seg : 64;
#pragma unroll
    for (int d = 0; d < 3; d++)
      for (int m = 1; m < in_wave; m <<= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], m)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], m)); }
    if (seg > 64) {
      if (lane == 0) for (int d = 0; d < 3; d++) { w_lo[wv][d] = lo[d]; w_hi[wv][d] = hi[d]; }
      __syncthreads();
      const int w0 = (t / seg) * (seg / 64);
      for (int d = 0; d < 3; d++) { lo[d] = w_lo[w0][d]; hi[d] = w_hi[w0][d]; for (int w = 1; w < seg / 64; w++) { lo[d] = fminf(lo[d], w_lo[w0 + w][d]); hi[d] = fmaxf(hi[d], w_hi[w0 + w][d]); } }
    }
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const int axis = (ey > ex && ey >= ez) ? 1 : ((ez > ex && ez > ey) ? 2 : 0);
    const float c = axis == 0 ? cur.x : (axis == 1 ? cur.y : cur.z);
    unsigned u = __float_as_uint(c); u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    pts[t] = cur;
    key[t] = ((unsigned long long)u << 32) | (unsigned)t;
    __syncthreads();
    for (int k = 2; k <= seg; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        const int ixj = t ^ j;
        if (ixj > t) {
          const unsigned long long a = key[t], b = key[ixj];
          const bool up = ((t & (seg - 1)) & k) == 0;
          if ((a > b) == up) { key[t] = b; key[ixj] = a; }
        }
        __syncthreads();
      }
    cur = pts[(int)(unsigned)(key[t] & 0xffffffffull)];
    __syncthreads();
  }
  return cur;
}
#endif

// one thread per slot of the sorted copy: gather the point in curve order, write it + (first lane of a leaf) the leaf box
__global__ __launch_bounds__(256) void leaf_kernel(KnnPair A, int split, const uint32_t* __restrict__ order) {
  ROLO_ALL_KERNEL_PRIO();
  // one thread per slot of the sorted copy (a thread per leaf gathered its 16 points one after the other: 13 us); the leaf box is a min / max
  // over the 16 lanes of the leaf — exact, so the order of the reduction does not matter
  const int which = (int)blockIdx.x >= split ? 1 : 0;
  const float4* __restrict__ p = A.c[which].xyz;
  const int n = A.c[which].n, n_leaves = A.c[which].n_leaves, P = A.c[which].P;
  float4* sorted = A.c[which].sorted; float4* boxes = A.c[which].boxes;
  order += which ? A.c[0].n : 0;
  const int s = ((int)blockIdx.x - (which ? split : 0)) * blockDim.x + threadIdx.x;
  const int g = s / KNN_LEAF;
  float4 o = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(INT_MAX));   // padding: never a neighbour, never inside a box
  if (g < P && s < n) {
    const uint32_t idx = order[s];
    const float4 q = p[idx];
    o = make_float4(q.x, q.y, q.z, __int_as_float((int)idx));
  }
#ifdef ROLO_KNN_KD_REFINE
  // (experiment) the 256 curve-consecutive points of this workgroup re-ordered by four median splits on the widest axis (256 -> 16 x 16): the curve
  // keeps the block together in space, the splits make its packets (64) and leaves (16) compact boxes instead of stretches of a curve that
  // enters and leaves a surface. Padding sorts last at every level, so the real points stay a prefix of the block. (P >= 16: whole blocks.)
  if (P >= 16) o = kd_refine_block(o);
#endif
  if (g >= P) return;   // whole leaves only: KNN_LEAF divides the block size
  const bool real = __float_as_int(o.w) != INT_MAX;
  float lox = real ? o.x : INFINITY, loy = real ? o.y : INFINITY, loz = real ? o.z : INFINITY;
  float hix = real ? o.x : -INFINITY, hiy = real ? o.y : -INFINITY, hiz = real ? o.z : -INFINITY;
  if (g < n_leaves) sorted[s] = o;
#pragma unroll
  for (int m = 1; m < KNN_LEAF; m <<= 1) {
    lox = fminf(lox, __shfl_xor(lox, m)); loy = fminf(loy, __shfl_xor(loy, m)); loz = fminf(loz, __shfl_xor(loz, m));
    hix = fmaxf(hix, __shfl_xor(hix, m)); hiy = fmaxf(hiy, __shfl_xor(hiy, m)); hiz = fmaxf(hiz, __shfl_xor(hiz, m));
  }
  if ((s & (KNN_LEAF - 1)) == 0) {
    boxes[2 * (size_t)(P + g)] = make_float4(lox, loy, loz, 0.f);
    boxes[2 * (size_t)(P + g) + 1] = make_float4(hix, hiy, hiz, 0.f);
  }
  // the levels above this workgroup's 256 / KNN_LEAF leaves, up to their common ancestor: one launch of tree_reduce_kernel less per frame
  // (whole blocks only: P >= LPB leaves, which makes the block's leaves an aligned subtree)
  constexpr int LPB = 256 / KNN_LEAF;
  if (P >= LPB) {   // uniform
    __shared__ float4 nlo[LPB], nhi[LPB];
    const int lt = threadIdx.x / KNN_LEAF;   // leaf of this thread inside the block
    if ((s & (KNN_LEAF - 1)) == 0) { nlo[lt] = make_float4(lox, loy, loz, 0.f); nhi[lt] = make_float4(hix, hiy, hiz, 0.f); }
    __syncthreads();
    const size_t g0 = (size_t)P + (size_t)(g - lt);   // heap index of the block's first leaf (all threads: g - lt is the same leaf)
    int m = LPB; size_t first = g0;
    while (m > 1) {   // LPB = 16: four levels, a handful of lanes each
      const int half = m >> 1;
      first >>= 1;
      float4 a, b;
      const int u = threadIdx.x;
      if (u < half) {
        const float4 l0 = nlo[2 * u], l1 = nlo[2 * u + 1], h0 = nhi[2 * u], h1 = nhi[2 * u + 1];
        a = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.f);
        b = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.f);
      }
      __syncthreads();
      if (u < half) { nlo[u] = a; nhi[u] = b; boxes[2 * (first + u)] = a; boxes[2 * (first + u) + 1] = b; }
      __syncthreads();
      m = half;
    }
  }
}

// Builds log2(chunk) levels of the implicit BVH in LDS: inputs are the `count_in` nodes at heap indices
// [count_in, 2*count_in); block b owns inputs [b*chunk, (b+1)*chunk).
__global__ __launch_bounds__(256) void tree_reduce_kernel(float4* boxes0, int count_in0, int chunk0, int split, float4* boxes1, int count_in1, int chunk1) {
  ROLO_ALL_KERNEL_PRIO();
  __shared__ float4 lo[512], hi[512];
  const bool second = (int)blockIdx.x >= split;
  float4* boxes = second ? boxes1 : boxes0;
  const int count_in = second ? count_in1 : count_in0, chunk = second ? chunk1 : chunk0;
  const int b = (int)blockIdx.x - (second ? split : 0), t = threadIdx.x;
  const size_t base_in = (size_t)count_in + (size_t)b * chunk;
  for (int i = t; i < chunk; i += 256) { lo[i] = boxes[2 * (base_in + i)]; hi[i] = boxes[2 * (base_in + i) + 1]; }
  __syncthreads();
  int m = chunk, level_count = count_in;
  while (m > 1) {
    const int half = m >> 1;
    level_count >>= 1;
    float4 nlo, nhi;
    if (t < half) {
      float4 a = lo[2 * t], c = lo[2 * t + 1], e = hi[2 * t], f = hi[2 * t + 1];
      nlo = make_float4(fminf(a.x, c.x), fminf(a.y, c.y), fminf(a.z, c.z), 0.f);
      nhi = make_float4(fmaxf(e.x, f.x), fmaxf(e.y, f.y), fmaxf(e.z, f.z), 0.f);
    }
    __syncthreads();
    if (t < half) {
      lo[t] = nlo; hi[t] = nhi;
      size_t h = (size_t)level_count + (size_t)b * half + t;
      boxes[2 * h] = nlo; boxes[2 * h + 1] = nhi;
    }
    __syncthreads();
    m = half;
  }
}

// ---- Eigen::JacobiSVD<Matrix3d> restated for one thread (rot_vgicp_impl.hpp:468) ---------------------------
struct Rot2 { double c, s; };

ROLO_DEV void make_jacobi(double x, double y, double z, Rot2& r) {
  double deno = 2.0 * fabs(y);
  if (deno < DBL_MIN) { r.c = 1; r.s = 0; return; }
  double tau = (x - z) / deno;
  double w = sqrt(tau * tau + 1.0);
  double t = (tau > 0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
  double sign_t = t > 0 ? 1.0 : -1.0;
  double n = 1.0 / sqrt(t * t + 1.0);
  r.s = -sign_t * (y / fabs(y)) * fabs(t) * n;
  r.c = n;
}

// W, U, V are 3x3 row-major in registers; p,q are compile-time so indexing stays static
template <int P_, int Q_>
ROLO_DEV bool jacobi_pair(double (&W)[9], double (&U)[9], double (&V)[9], double& max_diag) {
  const double threshold = fmax(DBL_MIN, 2.0 * DBL_EPSILON * max_diag);
  if (!(fabs(W[P_ * 3 + Q_]) > threshold || fabs(W[Q_ * 3 + P_]) > threshold)) return false;
  double m00 = W[P_ * 3 + P_], m01 = W[P_ * 3 + Q_], m10 = W[Q_ * 3 + P_], m11 = W[Q_ * 3 + Q_];
  Rot2 rot1;
  double t = m00 + m11, d = m10 - m01;
  if (fabs(d) < DBL_MIN) { rot1.s = 0; rot1.c = 1; }
  else { double u = t / d; double tmp = sqrt(1.0 + u * u); rot1.s = 1.0 / tmp; rot1.c = u / tmp; }
  double n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11, n11 = -rot1.s * m01 + rot1.c * m11;
  Rot2 jr; make_jacobi(n00, n01, n11, jr);
  Rot2 jl; { double c2 = jr.c, s2 = -jr.s; jl.c = rot1.c * c2 - rot1.s * s2; jl.s = rot1.c * s2 + rot1.s * c2; }
  // W.applyOnTheLeft(p,q,jl): rows
#pragma unroll
  for (int k = 0; k < 3; k++) { double x = W[P_ * 3 + k], y = W[Q_ * 3 + k]; W[P_ * 3 + k] = jl.c * x + jl.s * y; W[Q_ * 3 + k] = -jl.s * x + jl.c * y; }
  // U.applyOnTheRight(p,q,jl.transpose()) : transpose = (c,-s); right-apply of (c,s') : colp = c x - s' y ; colq = s' x + c y with s' = -s
#pragma unroll
  for (int k = 0; k < 3; k++) { double x = U[k * 3 + P_], y = U[k * 3 + Q_]; U[k * 3 + P_] = jl.c * x + jl.s * y; U[k * 3 + Q_] = -jl.s * x + jl.c * y; }
  // W.applyOnTheRight(p,q,jr), V.applyOnTheRight(p,q,jr)
#pragma unroll
  for (int k = 0; k < 3; k++) { double x = W[k * 3 + P_], y = W[k * 3 + Q_]; W[k * 3 + P_] = jr.c * x - jr.s * y; W[k * 3 + Q_] = jr.s * x + jr.c * y; }
#pragma unroll
  for (int k = 0; k < 3; k++) { double x = V[k * 3 + P_], y = V[k * 3 + Q_]; V[k * 3 + P_] = jr.c * x - jr.s * y; V[k * 3 + Q_] = jr.s * x + jr.c * y; }
  max_diag = fmax(max_diag, fmax(fabs(W[P_ * 3 + P_]), fabs(W[Q_ * 3 + Q_])));
  return true;
}

ROLO_DEV void swap_cols(double (&M)[9], int a, int b) {
#pragma unroll
  for (int k = 0; k < 3; k++) { double t = M[k * 3 + a]; M[k * 3 + a] = M[k * 3 + b]; M[k * 3 + b] = t; }
}

ROLO_DEV void jacobi_svd3(const double (&A)[9], double (&U)[9], double (&sv)[3], double (&V)[9]) {
  double scale = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) scale = fmax(scale, fabs(A[i]));
  if (!(scale > 0) || !isfinite(scale)) scale = 1.0;
  double W[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { W[i] = A[i] / scale; U[i] = (i % 4 == 0) ? 1.0 : 0.0; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  double max_diag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
  for (int sweep = 0; sweep < 100; sweep++) {
    bool any = false;
    any |= jacobi_pair<1, 0>(W, U, V, max_diag);
    any |= jacobi_pair<2, 0>(W, U, V, max_diag);
    any |= jacobi_pair<2, 1>(W, U, V, max_diag);
    if (!any) break;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double a = fabs(W[i * 4]);
    sv[i] = a * scale;
    if (a != 0) { double sgn = W[i * 4] / a; U[0 * 3 + i] *= sgn; U[1 * 3 + i] *= sgn; U[2 * 3 + i] *= sgn; }
  }
  // sort descending with matching column swaps (3 elements: selection sort as in Eigen)
  if (sv[1] > sv[0] && sv[1] >= sv[2]) { double t = sv[0]; sv[0] = sv[1]; sv[1] = t; swap_cols(U, 0, 1); swap_cols(V, 0, 1); }
  else if (sv[2] > sv[0] && sv[2] > sv[1]) { double t = sv[0]; sv[0] = sv[2]; sv[2] = t; swap_cols(U, 0, 2); swap_cols(V, 0, 2); }
  if (sv[2] > sv[1]) { double t = sv[1]; sv[1] = sv[2]; sv[2] = t; swap_cols(U, 1, 2); swap_cols(V, 1, 2); }
}

ROLO_DEV void inv3(const double (&A)[9], double (&o)[9]) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[2] * A[7] - A[1] * A[8], c02 = A[1] * A[5] - A[2] * A[4];
  double c10 = A[5] * A[6] - A[3] * A[8], c11 = A[0] * A[8] - A[2] * A[6], c12 = A[2] * A[3] - A[0] * A[5];
  double c20 = A[3] * A[7] - A[4] * A[6], c21 = A[1] * A[6] - A[0] * A[7], c22 = A[0] * A[4] - A[1] * A[3];
  double det = A[0] * c00 + A[1] * c10 + A[2] * c20;
  double inv = 1.0 / det;
  o[0] = c00 * inv; o[1] = c01 * inv; o[2] = c02 * inv; o[3] = c10 * inv; o[4] = c11 * inv; o[5] = c12 * inv; o[6] = c20 * inv; o[7] = c21 * inv; o[8] = c22 * inv;
}

// regularisation of the neighbourhood covariance (rot_vgicp_impl.hpp:457-490) and the store of its six unique entries
ROLO_DEV void knn_covariance_finish(double cxx, double cxy, double cxz, double cyy, double cyz, double czz, int n, int qi, int reg,
                                    double* __restrict__ cov, double (&c6)[6], double* __restrict__ nrm = nullptr) {
  double out[9];

  if (reg == ROLO_REG_NONE) {
    out[0] = cxx; out[1] = cxy; out[2] = cxz; out[3] = cxy; out[4] = cyy; out[5] = cyz; out[6] = cxz; out[7] = cyz; out[8] = czz;
  } else if (reg == ROLO_REG_FROBENIUS) {
    double C[9] = {cxx + 1e-3, cxy, cxz, cxy, cyy + 1e-3, cyz, cxz, cyz, czz + 1e-3};
    double Ci[9]; inv3(C, Ci);
    double nrm = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) nrm += Ci[i] * Ci[i];
    nrm = sqrt(nrm);
#pragma unroll
    for (int i = 0; i < 9; i++) Ci[i] /= nrm;
    inv3(Ci, out);
  } else {
    const double A[9] = {cxx, cxy, cxz, cxy, cyy, cyz, cxz, cyz, czz};
    double U[9], V[9], sv[3];
    jacobi_svd3(A, U, sv, V);
    double v0 = 1, v1 = 1, v2 = 1e-3;
    if (reg == ROLO_REG_MIN_EIG) { v0 = fmax(sv[0], 1e-3); v1 = fmax(sv[1], 1e-3); v2 = fmax(sv[2], 1e-3); }
    else if (reg == ROLO_REG_NORMALIZED_MIN_EIG) { double m = fmax(sv[0], fmax(sv[1], sv[2])); v0 = fmax(sv[0] / m, 1e-3); v1 = fmax(sv[1] / m, 1e-3); v2 = fmax(sv[2] / m, 1e-3); }
    else if (reg == ROLO_REG_PLANE_S) { double s = sv[0] + sv[1] + sv[2]; v0 = sv[0] / s; v1 = sv[1] / s; v2 = 1e-3; }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) out[a * 3 + b] = (U[a * 3 + 0] * v0) * V[b * 3 + 0] + (U[a * 3 + 1] * v1) * V[b * 3 + 1] + (U[a * 3 + 2] * v2) * V[b * 3 + 2];
    if (nrm && reg == ROLO_REG_PLANE) {
      // U diag(1, 1, 1e-3) V^T with orthonormal V and U = V up to the sign fix of each column: u0 u0^T + u1 u1^T + s 1e-3 u2 u2^T = I - (1 - s 1e-3) u2 u2^T,
      // s = U's sign on the third column (-1 only for a rank-deficient neighbourhood whose zero singular value came out of the sweep with a minus sign)
      const double s3 = U[0 * 3 + 2] * V[0 * 3 + 2] + U[1 * 3 + 2] * V[1 * 3 + 2] + U[2 * 3 + 2] * V[2 * 3 + 2];   // +-1 to rounding
      const double al = sqrt(1.0 - (s3 < 0 ? -1e-3 : 1e-3));
      // The identity needs u0 = v0 and u1 = v1. A neighbourhood of rank <= 1 (collinear or coincident neighbours: TWO singular values ~ 0) can leave the sweep with the
      // sign fix on the second column as well (or, for coincident points, on all three): the six entries above are then NOT I - m m^T, and they are what the voxel map
      // and every getter see. Such a point is poisoned — NaN in m.x — and the passes take its six entries instead (passes.hip load_pt / rotated_cov): advisor, round 5.
      const double s1 = U[0 * 3 + 0] * V[0 * 3 + 0] + U[1 * 3 + 0] * V[1 * 3 + 0] + U[2 * 3 + 0] * V[2 * 3 + 0];
      const double s2 = U[0 * 3 + 1] * V[0 * 3 + 1] + U[1 * 3 + 1] * V[1 * 3 + 1] + U[2 * 3 + 1] * V[2 * 3 + 1];
      const bool plane_form = s1 > 0 && s2 > 0;
      const size_t pn = (size_t)n;
      nrm[qi] = plane_form ? al * V[0 * 3 + 2] : __builtin_nan(""); nrm[pn + qi] = al * V[1 * 3 + 2]; nrm[2 * pn + qi] = al * V[2 * 3 + 2];
    }
  }
  const size_t pitch = (size_t)n;
  c6[0] = out[0]; c6[1] = 0.5 * (out[1] + out[3]); c6[2] = 0.5 * (out[2] + out[6]); c6[3] = out[4]; c6[4] = 0.5 * (out[5] + out[7]); c6[5] = out[8];
#pragma unroll
  for (int d = 0; d < 6; d++) cov[d * pitch + qi] = c6[d];
}

// covariance of the neighbourhood (rot_vgicp_impl.hpp:438-455), fp64, centred two-pass, the neighbours in list order: m6 = xx xy xz yy yz zz
// the K neighbours are gathered once and stay in registers for both passes
template <int KMAX>
ROLO_DEV void knn_neighbourhood_moments(const int (&ki)[KMAX], int kk, const float4* __restrict__ orig, double (&m6)[6]) {
  float px[KMAX], py[KMAX], pz[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) {
    const float4 p = (u < kk) ? orig[ki[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
    px[u] = p.x; py[u] = p.y; pz[u] = p.z;
  }
  double mx = 0, my = 0, mz = 0;
#pragma unroll
  for (int u = 0; u < KMAX; u++) if (u < kk) { mx += (double)px[u]; my += (double)py[u]; mz += (double)pz[u]; }
  mx /= kk; my /= kk; mz /= kk;
  double cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
#pragma unroll
  for (int u = 0; u < KMAX; u++) if (u < kk) {
    const double ax = (double)px[u] - mx, ay = (double)py[u] - my, az = (double)pz[u] - mz;
    cxx += ax * ax; cxy += ax * ay; cxz += ax * az; cyy += ay * ay; cyz += ay * az; czz += az * az;
  }
  m6[0] = cxx / kk; m6[1] = cxy / kk; m6[2] = cxz / kk; m6[3] = cyy / kk; m6[4] = cyz / kk; m6[5] = czz / kk;
}

// covariance of the neighbourhood + regularisation, one lane per query
template <int KMAX>
ROLO_DEV void knn_covariance_tail(const int (&ki)[KMAX], int kk, const float4* __restrict__ orig, int n, int qi, int reg,
                                  double* __restrict__ cov, double (&c6)[6], double* __restrict__ nrm = nullptr) {
  double m6[6];
  knn_neighbourhood_moments<KMAX>(ki, kk, orig, m6);
  knn_covariance_finish(m6[0], m6[1], m6[2], m6[3], m6[4], m6[5], n, qi, reg, cov, c6, nrm);
}

// the same for any number of neighbours (k_correspondences > 64): the neighbours are gathered twice, slot by slot, in the same order
ROLO_DEV void knn_covariance_tail_loop(const int32_t* __restrict__ nbr, size_t n_sorted, int j, int kk, const float4* __restrict__ orig, int n, int qi, int reg,
                                       double* __restrict__ cov, double (&c6)[6], double* __restrict__ nrm = nullptr) {
  double mx = 0, my = 0, mz = 0;
  for (int u = 0; u < kk; u++) { const float4 p = orig[nbr[(size_t)u * n_sorted + j]]; mx += (double)p.x; my += (double)p.y; mz += (double)p.z; }
  mx /= kk; my /= kk; mz /= kk;
  double cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
  for (int u = 0; u < kk; u++) {
    const float4 p = orig[nbr[(size_t)u * n_sorted + j]];
    const double ax = (double)p.x - mx, ay = (double)p.y - my, az = (double)p.z - mz;
    cxx += ax * ax; cxy += ax * ay; cxz += ax * az; cyy += ay * ay; cyz += ay * az; czz += az * az;
  }
  cxx /= kk; cxy /= kk; cxz /= kk; cyy /= kk; cyz /= kk; czz /= kk;
  knn_covariance_finish(cxx, cxy, cxz, cyy, cyz, czz, n, qi, reg, cov, c6, nrm);
}
}  // namespace
}  // namespace rolo

#include "knn_walk.hpp"

namespace rolo {

bool knn_voxel_fuse_supported() {
#ifdef ROLO_KNN_ROCPRIM_SORT
  return false;   // the insert rides on the hand-written sort's scatter launches
#else
  return true;
#endif
}

size_t knn_bbox_ints() { return BBOX_PART + 2 * BBOX_BLOCKS * 6; }

size_t knn_sort_temp_bytes(int n) {  // for n points in total (one cloud or the sum of a pair), keys of up to 31 bits
#ifdef ROLO_KNN_ROCPRIM_SORT
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (size_t)n, 0, 31, (hipStream_t)0);
  return bytes;
#else
  (void)n;
  return sizeof(int) * SORT_PASSES * SORT_NB * SORT_D;   // digit counters of the passes
#endif
}

// curve sort + implicit BVHs of the pair's clouds. sorted / boxes must be allocated for n_leaves / P of each cloud;
// keys / vals hold n0 + n1 entries, bbox 12 ints.
hipError_t launch_knn_build(const KnnPair& A_in, void* sort_tmp, size_t sort_tmp_bytes, uint32_t* keys0, uint32_t* keys1,
                            uint32_t* vals0, uint32_t* vals1, int* bbox, const VoxelFuse& vf, hipStream_t s) {
  KnnPair A = A_in;
  bool have_parts = true;   // every cloud brings the partial boxes its pack kernel left: no bbox launch
  for (int i = 0; i < A.n_clouds; i++) have_parts = have_parts && A.c[i].bpart != nullptr && A.c[i].n_bpart > 0;
  if (!have_parts) for (int i = 0; i < A.n_clouds; i++) { A.c[i].bpart = bbox + BBOX_PART + i * BBOX_BLOCKS * 6; A.c[i].n_bpart = BBOX_BLOCKS; }
  const int nc = A.n_clouds;
  const int n_total = A.c[0].n + (nc > 1 ? A.c[1].n : 0);
#ifdef ROLO_KNN_ROCPRIM_SORT
  const int tile_ = ((n_total + SORT_NB - 1) / SORT_NB + SORT_T - 1) / SORT_T * SORT_T;
  if (!have_parts) bbox_kernel<<<BBOX_BLOCKS * nc, 256, 0, s>>>(A, bbox);
  morton_kernel<<<SORT_NB + (vf.enabled ? VF_CLEAR_BLOCKS : 0), SORT_T, 0, s>>>(A, bbox, keys0, vals0, n_total, tile_, nullptr, vf);
  hipError_t e = rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys0, keys1, vals0, vals1, (size_t)n_total, 0, nc > 1 ? KEY_BITS : KEY_BITS - 1, s);
  if (e != hipSuccess) return e;
  const uint32_t* order = vals1;
#else
  (void)sort_tmp_bytes;
  int* cnt = static_cast<int*>(sort_tmp);
  const int gvf = vf.enabled ? ((vf.n_tgt + SORT_PASSES - 1) / SORT_PASSES + SORT_T - 1) / SORT_T : 0;   // insert workgroups per scatter launch
  const int tile = ((n_total + SORT_NB - 1) / SORT_NB + SORT_T - 1) / SORT_T * SORT_T;
  if (!have_parts) bbox_kernel<<<BBOX_BLOCKS * nc, 256, 0, s>>>(A, bbox);
  morton_kernel<<<SORT_NB + (vf.enabled ? VF_CLEAR_BLOCKS : 0), SORT_T, 0, s>>>(A, bbox, keys0, vals0, n_total, tile, cnt, vf);
  uint32_t *ki = keys0, *vi = vals0, *ko = keys1, *vo = vals1;
  for (int p = 0; p < SORT_PASSES; p++) {
    if (p > 0) sort_hist_kernel<<<SORT_NB, SORT_T, 0, s>>>(ki, n_total, tile, p, cnt);
    sort_scatter_kernel<<<SORT_NB + gvf, SORT_T, 0, s>>>(ki, vi, ko, vo, n_total, tile, p, cnt, vf);
    uint32_t* t1 = ki; ki = ko; ko = t1; t1 = vi; vi = vo; vo = t1;
  }
  const uint32_t* order = vi;   // wherever the last pass left the values
#endif
  static_assert(256 % KNN_LEAF == 0 && (KNN_LEAF & (KNN_LEAF - 1)) == 0, "leaf_kernel reduces a leaf inside a wavefront");
  const int l0 = (A.c[0].P * KNN_LEAF + 255) / 256, l1 = nc > 1 ? (A.c[1].P * KNN_LEAF + 255) / 256 : 0;
  leaf_kernel<<<l0 + l1, 256, 0, s>>>(A, l0, order);
  // leaf_kernel leaves the 4 levels above its 16 leaves behind: the reduction starts at P / 16 nodes (one launch up to 8192 leaves)
  constexpr int LPB = 256 / KNN_LEAF;
  int count0 = A.c[0].P >= LPB ? A.c[0].P / LPB : A.c[0].P, count1 = nc > 1 ? (A.c[1].P >= LPB ? A.c[1].P / LPB : A.c[1].P) : 1;
  while (count0 > 1 || count1 > 1) {
    const int chunk0 = count0 < 512 ? count0 : 512, chunk1 = count1 < 512 ? count1 : 512;
    const int b0 = count0 > 1 ? count0 / chunk0 : 0, b1 = count1 > 1 ? count1 / chunk1 : 0;
    tree_reduce_kernel<<<b0 + b1, 256, 0, s>>>(A.c[0].boxes, count0, chunk0, b0, nc > 1 ? A.c[1].boxes : nullptr, count1, chunk1);
    if (count0 > 1) count0 /= chunk0;
    if (count1 > 1) count1 /= chunk1;
  }
  return hipGetLastError();
}

#ifdef ROLO_KNN_STATS
extern "C" int rolo_debug_wave_records(unsigned* out /* 16384 x 8 */) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_wave_rec), sizeof(unsigned) * 16384 * 8) == hipSuccess ? 0 : -1;
}
#endif

static inline int slice_blocks(const KnnCloud& c) { return (c.q_end - c.q_begin + 255) / 256; }

// Which walk (k = 20): four lanes per query up to this many 64-query packets in the launch, two lanes per query above (rocprofv3 / bench.py, round 4):
//   one ~48.7 k-point feature cloud (761 packets): 0.138 ms with 64-query packets, 0.064 ms with four lanes; the pipeline's source + target launch
//   (1450-1600 packets): raw frame -> pose 0.489 -> 0.423 ms (two lanes: 0.449);
//   2 x 65 536 points (2048 packets, four contexts, BASELINE configs[4]): 4.40 k scans/s with packets, 4.35 k four lanes, 4.42 k two lanes;
//   2 x 131 072 points (4096 packets): walk over the pool 0.218 / 0.169 / 0.175 ms, single-frame latency 0.79 / - / 0.76 ms, four contexts in flight
//   2.87 / 2.74 / 2.85 k scans/s (paired runs) with packets / four / two lanes — two lanes issue the packets' instruction count (SQ_INSTS_VALU 51.5 M against
//   52.4 M per launch) in chains half as long; four lanes issue 8 % more and twice the scalar instructions.
// ROLO_KNN_SUB=0 keeps the 64-query packets at every size (the A/B, and the kernel of every k other than 20).
constexpr int KNN_SUB_MAX_PACKETS = 1792;   // (the pipeline's pair launch is 1450-1600 packets, frame by frame: the limit sits clear of it, and of configs[4]'s 2048)
// Round 5: above the limit the choice follows the DEVICE'S LOAD. Alone on the chip the two-lane walk finishes sooner (0.175 against 0.22 ms: single-frame latency 0.71
// against 0.76 ms); with other contexts' frames in flight the packets — half the wavefronts, the same instructions — leave the other frames' short LM kernels more of
// every SIMD's issue slots: 3.04 against 2.98 k scans/s, and 3.17 against 3.01 k once those kernels run at raised priority (ROLO_SHORT_PRIO). The caller says which
// case it is (frames in flight on the device when this one is enqueued, api.hip); a captured hipGraph is keyed on it.
hipError_t launch_knn_walk(const KnnPair& A, int k, int regularization_or_minus1, const VoxelFuse& vf, hipStream_t s, int coop_budget, int* lanes_out, bool device_busy, bool moments) {
  if (lanes_out) *lanes_out = 1;
  constexpr int QPB = 256;   // queries per workgroup of the plain walk: four wavefronts of 64
  const int n0 = A.c[0].q_end - A.c[0].q_begin, n1 = A.n_clouds > 1 ? A.c[1].q_end - A.c[1].q_begin : 0;
  const int g0 = (n0 + QPB - 1) / QPB, g1 = (n1 + QPB - 1) / QPB;
  if (g0 + g1 == 0) return hipSuccess;
  (void)vf;   // (insert workgroups appended to THIS launch made its wave-uniform leaf loads vector loads: a store anywhere in the kernel is a potential clobber)
  constexpr int pad = 0;   // (an LDS pad here limited the walk to 3 / 2 workgroups per CU: 0.216 / 0.259 ms against 0.196)
  if (k > 64) {   // any k: rounds of 64 — round r searches the 64 (the last: k - 64 r) nearest ABOVE the previous round's last key (KnnCloud::lower).
                  // ceil(k / 64) full walks: correct, as slow as it sounds; the reference accepts any k, its default is 20 and ROLO never changes it
    if (regularization_or_minus1 >= 0) return hipErrorInvalidValue;   // the covariance tail is its own launch here
    KnnPair R = A;
    for (int r = 0; 64 * r < k; r++) {
      const int kr = k - 64 * r < 64 ? k - 64 * r : 64;
      for (int i = 0; i < R.n_clouds; i++) { R.c[i].slot0 = 64 * r; R.c[i].k_total = k; }
      if (r == 0) knn_walk_kernel<64, false><<<g0 + g1, 256, pad, s>>>(R, g0, kr, -1);
      else knn_walk_kernel<64, false, true><<<g0 + g1, 256, pad, s>>>(R, g0, kr, -1);
    }
    return hipGetLastError();
  }
  if (k == 20) {
    if (regularization_or_minus1 >= 0) knn_walk_kernel<20, true><<<g0 + g1, 256, pad, s>>>(A, g0, k, regularization_or_minus1);
    else if (coop_budget > 0) {
      // the cooperative walk (knn_walk.hpp): NW packets per workgroup, a heavy packet's remaining sub-trees go to the workgroup's idle wavefronts.
      // NW follows the launch size — a workgroup per CU at least: 4 for the pipeline's ~48 k-point feature clouds, 16 for the 2 x 131 072-point frame
      const int packets = (n0 + 63) / 64 + (n1 + 63) / 64;
      static const int force_nw = [] {   // 4 / 8 / 16 wavefronts per workgroup; anything else would run the NW = 4 kernel with a grid sized for another: ignored
        const char* e = getenv("ROLO_KNN_COOP_NW");
        if (!e) return 0;
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16) return v;
        fprintf(stderr, "librolo_hip: ROLO_KNN_COOP_NW=%s is not one of 4 / 8 / 16: ignored\n", e);
        return 0;
      }();
      const int nw = force_nw ? force_nw : (packets >= 16 * 256 ? 16 : (packets >= 8 * 256 ? 8 : 4));
      const int G4 = g0 + g1, G = (G4 + nw / 4 - 1) / (nw / 4);   // a workgroup = nw / 4 runs of four consecutive packets (the plain walk's blocks), strided by G
      if (nw == 16) knn_walk_coop_kernel<16><<<G, 1024, 0, s>>>(A, g0, G4, coop_budget);
      else if (nw == 8) knn_walk_coop_kernel<8><<<G, 512, 0, s>>>(A, g0, G4, coop_budget);
      else knn_walk_coop_kernel<4><<<G, 256, 0, s>>>(A, g0, G4, coop_budget);
    }
    else {
      // small clouds: 16 queries x 4 lanes per wavefront (knn_walk_sub_kernel) — below ~2 packets of 64 per SIMD the walk is a chain of fetches, not
      // inserts. ROLO_KNN_SUB (an A/B switch): 0 = the 64-query packets at every size, 2 = two lanes per query always, 4 (or 1) = four lanes always; unset = by size.
      // Any other value is ignored with a warning instead of silently picking a kernel (advisor, round 4).
      static const int sub_env = [] {
        const char* e = getenv("ROLO_KNN_SUB");
        if (!e) return -1;
        const int v = atoi(e);
        if (v == 0 || v == 1 || v == 2 || v == 4) return v;
        fprintf(stderr, "librolo_hip: ROLO_KNN_SUB=%s is not one of 0 / 1 / 2 / 4: ignored (the walk is picked by size)\n", e);
        return -1;
      }();
      const int packets = (n0 + 63) / 64 + (n1 + 63) / 64;
      const int lanes = sub_env < 0 ? (packets <= KNN_SUB_MAX_PACKETS ? 4 : (device_busy ? 0 : 2)) : (sub_env == 2 ? 2 : (sub_env ? 4 : 0));
      if (lanes_out) *lanes_out = lanes ? lanes : 1;
      // ROLO_KNN_WALK_WGS = workgroups of the walk a CU may hold at a time (through a dynamic-LDS pad; unset / 0: as many as fit): with two lanes per query the
      // dense frame's 8192 wavefronts fill all 8 wave slots of every SIMD, and whatever another context has queued waits for slots until the walk thins out
      static const int walk_pad = [] {
        const char* e = getenv("ROLO_KNN_WALK_WGS"); const int v = e ? atoi(e) : 0;
        const int pad_bytes = (v >= 1 && v <= 8) ? (160 * 1024 / v - 1024 - 512) & ~255 : 0;
        if (pad_bytes > 48 * 1024) {   // above the default limit of dynamic LDS a launch needs the attribute raised (1 or 2 workgroups per CU); if the runtime refuses, the pad is dropped
          bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_walk_kernel<20, false>), hipFuncAttributeMaxDynamicSharedMemorySize, pad_bytes) == hipSuccess;
          ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_walk_sub_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pad_bytes) == hipSuccess;
          ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_walk_sub_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, pad_bytes) == hipSuccess;
          if (!ok) { (void)hipGetLastError(); fprintf(stderr, "librolo_hip: ROLO_KNN_WALK_WGS=%d needs %d bytes of dynamic LDS, which this runtime refuses: ignored\n", v, pad_bytes); return 0; }
        }
        return pad_bytes;
      }();
      // moments: the walk's epilogue gathers each query's twenty winners once and leaves the six centred second moments of the neighbourhood where its covariance
      // will go — no 80 B/pt index array to write, read back and gather through again in the tail (round 6)
      if (lanes == 4) { const int s0 = (n0 + 63) / 64, s1 = (n1 + 63) / 64; if (moments) knn_walk_sub_kernel<4, true><<<s0 + s1, 256, walk_pad, s>>>(A, s0); else knn_walk_sub_kernel<4><<<s0 + s1, 256, walk_pad, s>>>(A, s0); }
      else if (lanes == 2) { const int s0 = (n0 + 127) / 128, s1 = (n1 + 127) / 128; if (moments) knn_walk_sub_kernel<2, true><<<s0 + s1, 256, walk_pad, s>>>(A, s0); else knn_walk_sub_kernel<2><<<s0 + s1, 256, walk_pad, s>>>(A, s0); }
      else if (moments) knn_walk_kernel<20, false, false, true><<<g0 + g1, 256, walk_pad, s>>>(A, g0, k, -1);
      else knn_walk_kernel<20, false><<<g0 + g1, 256, walk_pad, s>>>(A, g0, k, -1);
    }
  }
  else {
    if (k > 32) {   // up to 64 neighbours: 128 key registers per lane — correct, not tuned (the reference accepts any k; its default is 20)
      if (regularization_or_minus1 >= 0) knn_walk_kernel<64, true><<<g0 + g1, 256, pad, s>>>(A, g0, k, regularization_or_minus1);
      else knn_walk_kernel<64, false><<<g0 + g1, 256, pad, s>>>(A, g0, k, -1);
    } else if (regularization_or_minus1 >= 0) knn_walk_kernel<32, true><<<g0 + g1, 256, pad, s>>>(A, g0, k, regularization_or_minus1);
    else knn_walk_kernel<32, false><<<g0 + g1, 256, pad, s>>>(A, g0, k, -1);
  }
  return hipGetLastError();
}

hipError_t launch_knn_unstage(const KnnPair& A, bool own_slice_only, const VoxelFuse& vf, hipStream_t s) {
  const int g0 = own_slice_only ? slice_blocks(A.c[0]) : (A.c[0].n_sorted + 255) / 256;
  const int g1 = A.n_clouds > 1 ? (own_slice_only ? slice_blocks(A.c[1]) : (A.c[1].n_sorted + 255) / 256) : 0;
  if (g0 + g1 == 0) return hipSuccess;
  VoxelFuse v = vf;
  if (own_slice_only) v.enabled = 0;   // a slice alone cannot build the map
  knn_unstage_kernel<<<g0 + g1, 256, 0, s>>>(A, g0, own_slice_only ? 1 : 0, v);
  return hipGetLastError();
}

hipError_t launch_knn_tail(const KnnPair& A, int k, int regularization, const VoxelFuse& vf, hipStream_t s, bool moments) {
  const int g0 = slice_blocks(A.c[0]), g1 = A.n_clouds > 1 ? slice_blocks(A.c[1]) : 0;
  if (g0 + g1 == 0) return hipSuccess;
  if (k > 64) knn_tail_loop_kernel<<<g0 + g1, 256, 0, s>>>(A, g0, k, regularization, vf);
  else if (k == 20 && moments) knn_tail_kernel<20, true><<<g0 + g1, 256, 0, s>>>(A, g0, k, regularization, vf);
  else if (k == 20) knn_tail_kernel<20><<<g0 + g1, 256, 0, s>>>(A, g0, k, regularization, vf);
  else if (k <= 32) knn_tail_kernel<32><<<g0 + g1, 256, 0, s>>>(A, g0, k, regularization, vf);
  else knn_tail_kernel<64><<<g0 + g1, 256, 0, s>>>(A, g0, k, regularization, vf);
  return hipGetLastError();
}

}  // namespace rolo
