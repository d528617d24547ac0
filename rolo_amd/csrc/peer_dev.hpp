// Device side of the peer exchange (rolo_internal.hpp PeerArgs): the all-reduce of the 32 fp64 sums of one LM pass over the ranks of a
// node WITHOUT a collective library — SURVEY.md 5(ii) / 8e: "every GPU peer-writes its partial into a slot on each peer, fixed-rank-order
// local sum (deterministic, no RCCL latency)". Splits the loop the reference runs over all correspondences on one CPU
// (include/rot_gicp/gicp/impl/rot_vgicp_impl.hpp:313-382, the per-thread Hs[] / bs[] summed at :377-382) across GPUs.
//
// Protocol (one workgroup of 256 threads, called by the controller between its row sum and the scalar LM step):
//   e = own LM epoch + 1, parity = e & 1;
//   every rank writes the 64 words {e : 32 | half of a double : 32} of its sums into slot (parity, rank) of EVERY rank's mailbox —
//   relaxed 64-bit system-scope atomic stores: write-through, single-copy atomic, each word carries its own validity;
//   every rank polls the W x 64 words of its OWN mailbox (system-scope loads: they miss the XCD's L2 and see what came in over the
//   fabric / xGMI) until all carry epoch e, then adds the slots in rank order: identical bits on every rank.
// A rank can be at most one exchange ahead of a peer (finishing exchange e + 1 needs the peer's e + 1 words, which the peer sends only
// after it has read all of e), so two parities suffice; a stale word of epoch e - 2 never matches e. No fence is needed: nothing but the
// self-validating words crosses.
// A poll that lasts longer than timeout_ticks gives up (returns false): the caller raises ROLO_ECOMM in the LM state and ends the stage,
// so a lost peer costs a bounded wait, never a hung GPU.
#pragma once
#include "rolo_internal.hpp"

namespace rolo {

__device__ __forceinline__ unsigned long long peer_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peer_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sums: NV_MAX doubles in LDS, complete on entry (caller synchronised); on exit (after the internal barriers) the rank-ordered total.
// xw: LDS scratch of PEER_MAX * PEER_SLOT_WORDS 32-bit words. All threads of the workgroup must call; THREADS threads take part.
template <int THREADS>
__device__ __forceinline__ bool peer_allreduce_block(double* __restrict__ sums, unsigned* __restrict__ xw, int* __restrict__ bad /* LDS flag, zeroed here */,
                                                   const PeerArgs& pa) {
  const int W = pa.world, rank = pa.rank;
  unsigned long long* own = pa.box[rank];
  if (threadIdx.x == 0) *bad = 0;
  const unsigned e = (unsigned)peer_load(own + PEER_W_LM_EPOCH) + 1u;   // the same value in every thread and on every rank
  const int par = (int)(e & 1u);
  __syncthreads();   // nobody bumps the epoch before everybody has read it
  // publish: word w of the own slot in every rank's mailbox
  for (int idx = threadIdx.x; idx < W * PEER_SLOT_WORDS; idx += THREADS) {
    const int dst = idx / PEER_SLOT_WORDS, w = idx - dst * PEER_SLOT_WORDS;
    const double v = sums[w >> 1];
    const unsigned half = (w & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
    peer_store(pa.box[dst] + PEER_W_SLOTS + (size_t)(par * PEER_MAX + rank) * PEER_SLOT_WORDS + w, ((unsigned long long)e << 32) | half);
  }
  // collect: the own mailbox, all ranks' slots of this parity
  const long long t0 = wall_clock64();
  for (int idx = threadIdx.x; idx < W * PEER_SLOT_WORDS; idx += THREADS) {
    const int src = idx / PEER_SLOT_WORDS, w = idx - src * PEER_SLOT_WORDS;
    const unsigned long long* p = own + PEER_W_SLOTS + (size_t)(par * PEER_MAX + src) * PEER_SLOT_WORDS + w;
    unsigned long long x = peer_load(p);
    while ((unsigned)(x >> 32) != e) {
      if ((unsigned long long)(wall_clock64() - t0) > pa.timeout_ticks) { *bad = 1; break; }
      __builtin_amdgcn_s_sleep(1);
      x = peer_load(p);
    }
    xw[idx] = (unsigned)x;
  }
  __syncthreads();
  const bool ok = *bad == 0;
  if (threadIdx.x < NV_MAX) {
    double t = 0.0;
    for (int r = 0; r < W; r++) {   // rank order: the same additions on every rank
      const unsigned lo = xw[r * PEER_SLOT_WORDS + 2 * threadIdx.x], hi = xw[r * PEER_SLOT_WORDS + 2 * threadIdx.x + 1];
      t += __hiloint2double((int)hi, (int)lo);
    }
    sums[threadIdx.x] = t;
  }
  if (threadIdx.x == 0) peer_store(own + PEER_W_LM_EPOCH, (unsigned long long)e);   // the exchange is counted whether or not it completed
  __syncthreads();
  return ok;
}

}  // namespace rolo
