// Peer exchange kernels that are not part of the LM controller (passes.hip calls peer_allreduce_block from ctrl_kernel itself):
//   * peer_allreduce_kernel — the 32 fp64 sums of a stage-level evaluation (rolo_so3_linearize, rolo_compute_error, rolo_t3_linearize ...)
//     summed over the ranks in rank order;
//   * peer_cov_push_kernel / peer_cov_wait_kernel — K5's covariance exchange ("K5 shards by query point ... all-gather of
//     <= 4 MB per cloud"): every rank computed the 48-byte covariances of ITS slice of the curve-sorted queries into segment `rank` of its
//     exchange area; the push kernel copies that segment into the same segment of every peer's area (plain 16-byte stores over xGMI /
//     the fabric; two areas alternate by the parity of the exchange's number, taken from the epoch word on the device), each workgroup fences at system scope and takes a ticket, and the LAST workgroup raises this rank's flag (the epoch) in
//     every mailbox; the wait kernel (one wavefront) polls the own mailbox for every rank's flag of this epoch. The scatter kernel that
//     follows in stream order (knn_unstage_kernel) then reads complete segments. Replaces ncclAllGather on the sharded path: no
//     library call, graph-capturable, and testable with two processes on ONE device (which RCCL refuses).
#include "peer_dev.hpp"
#include <algorithm>

namespace rolo {

namespace {

__global__ __launch_bounds__(256) void peer_allreduce_kernel(double* __restrict__ sums_io, PeerArgs pa, int* __restrict__ err_flag) {
  __shared__ double sums[NV_MAX];
  __shared__ unsigned xw[PEER_MAX * PEER_SLOT_WORDS];
  __shared__ int bad;
  if (threadIdx.x < NV_MAX) sums[threadIdx.x] = sums_io[threadIdx.x];
  __syncthreads();
  const bool ok = peer_allreduce_block<256>(sums, xw, &bad, pa);
  if (threadIdx.x < NV_MAX) sums_io[threadIdx.x] = sums[threadIdx.x];
  if (!ok && threadIdx.x == 0 && err_flag) *err_flag = ROLO_ECOMM;
}

// grid-stride copy of the own segment into every peer's area; flag by the last workgroup to arrive
__global__ __launch_bounds__(256) void peer_cov_push_kernel(PeerArgs pa, size_t area_bytes /* of ONE exchange area */, size_t seg_doubles) {
  const int W = pa.world, rank = pa.rank;
  unsigned long long* own = pa.box[rank];
  // the area of exchange number e = (own epoch + 1) is e & 1 — on every rank, in every launch of a replayed hipGraph (the wait kernel behind this one bumps the epoch)
  const size_t area_off = PEER_STAGE_OFFSET + (size_t)((peer_load(own + PEER_W_COV_EPOCH) + 1ull) & 1ull) * area_bytes;
  const size_t n2 = seg_doubles / 2;   // 16-byte units (seg_doubles is a multiple of 6 * 256)
  const double2* __restrict__ src = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(own) + area_off) + (size_t)rank * n2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = src[i];
    for (int p = 1; p < W; p++) {   // start with the next rank: the ranks' bursts fan out over different links
      const int dst = (rank + p) % W;
      reinterpret_cast<double2*>(reinterpret_cast<char*>(pa.box[dst]) + area_off)[(size_t)rank * n2 + i] = v;
    }
  }
  __threadfence_system();   // this thread's stores are visible system-wide before its workgroup takes the ticket
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(own + PEER_W_TICKET, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = t == gridDim.x - 1 ? 1 : 0;   // (the last workgroup resets the ticket below: every launch starts at 0)
  }
  __syncthreads();
  if (!last) return;
  __threadfence_system();
  const unsigned long long e = peer_load(own + PEER_W_COV_EPOCH) + 1ull;
  if ((int)threadIdx.x < W) __hip_atomic_store(pa.box[threadIdx.x] + PEER_W_COV_FLAG + rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x == 0) __hip_atomic_store(own + PEER_W_TICKET, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every workgroup has arrived: reset for the next frame
}

__global__ __launch_bounds__(64) void peer_cov_wait_kernel(PeerArgs pa, int* __restrict__ err_flag) {
  unsigned long long* own = pa.box[pa.rank];
  const unsigned long long e = peer_load(own + PEER_W_COV_EPOCH) + 1ull;
  const long long t0 = wall_clock64();
  bool ok = true;
  if ((int)threadIdx.x < pa.world) {
    const unsigned long long* p = own + PEER_W_COV_FLAG + threadIdx.x;
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
      if ((unsigned long long)(wall_clock64() - t0) > pa.timeout_ticks) { ok = false; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  const bool all_ok = __all(ok);
  if (threadIdx.x == 0) {
    peer_store(own + PEER_W_COV_EPOCH, e);
    if (!all_ok && err_flag) *err_flag = ROLO_ECOMM;
  }
}

// rolo_peer_selftest: known words into the own segment of the area the NEXT exchange uses (word i of rank r = r * 2^32 + i, exact in fp64)
__global__ __launch_bounds__(256) void peer_selftest_fill_kernel(PeerArgs pa, size_t area_bytes, size_t seg_doubles) {
  unsigned long long* own = pa.box[pa.rank];
  const size_t area_off = PEER_STAGE_OFFSET + (size_t)((peer_load(own + PEER_W_COV_EPOCH) + 1ull) & 1ull) * area_bytes;
  double* seg = reinterpret_cast<double*>(reinterpret_cast<char*>(own) + area_off) + (size_t)pa.rank * seg_doubles;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < seg_doubles; i += (size_t)gridDim.x * blockDim.x)
    seg[i] = (double)pa.rank * 4294967296.0 + (double)i;
}
// ... and after the exchange: every rank's segment of the area the LAST exchange used, checked against the same pattern; bad[r] = words of rank r that differ
__global__ __launch_bounds__(256) void peer_selftest_check_kernel(PeerArgs pa, size_t area_bytes, size_t seg_doubles, unsigned* __restrict__ bad) {
  unsigned long long* own = pa.box[pa.rank];
  const size_t area_off = PEER_STAGE_OFFSET + (size_t)(peer_load(own + PEER_W_COV_EPOCH) & 1ull) * area_bytes;
  const double* area = reinterpret_cast<const double*>(reinterpret_cast<const char*>(own) + area_off);
  for (int r = 0; r < pa.world; r++) {
    unsigned n = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < seg_doubles; i += (size_t)gridDim.x * blockDim.x)
      if (area[(size_t)r * seg_doubles + i] != (double)r * 4294967296.0 + (double)i) n++;
    if (n) atomicAdd(bad + r, n);
  }
}

}  // namespace

hipError_t launch_peer_selftest_fill(const PeerArgs& peer, size_t area_bytes, size_t seg_doubles, hipStream_t s) {
  peer_selftest_fill_kernel<<<64, 256, 0, s>>>(peer, area_bytes, seg_doubles);
  return hipGetLastError();
}
hipError_t launch_peer_selftest_check(const PeerArgs& peer, size_t area_bytes, size_t seg_doubles, unsigned* bad, hipStream_t s) {
  peer_selftest_check_kernel<<<64, 256, 0, s>>>(peer, area_bytes, seg_doubles, bad);
  return hipGetLastError();
}

hipError_t launch_peer_allreduce(double* sums, const PeerArgs& peer, int* err_flag, hipStream_t s) {
  peer_allreduce_kernel<<<1, 256, 0, s>>>(sums, peer, err_flag);
  return hipGetLastError();
}

hipError_t launch_peer_cov_exchange(const PeerArgs& peer, size_t area_bytes, size_t seg_doubles, int* err_flag, hipStream_t s) {
  const size_t n2 = seg_doubles / 2;
  const int grid = (int)std::min<size_t>(std::max<size_t>((n2 + 255) / 256, 1), 1024);
  peer_cov_push_kernel<<<grid, 256, 0, s>>>(peer, area_bytes, seg_doubles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  peer_cov_wait_kernel<<<1, 64, 0, s>>>(peer, err_flag);
  return hipGetLastError();
}

}  // namespace rolo
