// K5 search kernels: exact k-nearest neighbours by *packet traversal* — one wavefront walks the implicit
// BVH once for its 64 curve-adjacent (Hilbert order) queries. Included by knn_cov.hip (needs box_d2 and the covariance tail).
//
// Why a packet: with one independent walk per lane the wavefront executes the union of 64 divergent walks and
// pays the sorted-insert (the expensive part) on almost every visited point because *some* lane accepts it;
// measured 2.45 ms per 131k-point cloud, vs 0.59 ms for the packet. Here every control decision is wave-uniform:
//   * seed: the wavefront's own 64 points (64 / KNN_LEAF = 4 consecutive leaves of 16) are scored first, so every lane starts the walk
//     with a finite search radius;
//   * a node is expanded if ANY lane's search sphere reaches its box (ballot); of two live children the one
//     nearer to the majority of interested lanes goes first, the other is pushed on ONE small per-wave stack in LDS;
//   * node boxes and leaf points are fetched through wave-uniform addresses, so a leaf's 16 points are loaded
//     once per wave, not once per lane;
//   * each lane keeps its k best as packed 64-bit keys  (float_bits(d2) << 32) | index : for non-negative floats
//     the unsigned order of the key IS the (d2, index) lexicographic order of the oracle, and — the patterns being
//     finite positive doubles — a sorted insert is  new[s] = max(K[s-1], min(c, K[s]))  = 2 fp64 VALU ops per
//     slot with no compares, selects or tie special-cases.
#pragma once
#ifdef ROLO_KNN_STATS
#define KNN_STAT(x) x
#else
#define KNN_STAT(x)
#endif
#include "knn_packet.hpp"   // keys, sorted insert, explicit scalar loads, leaf scoring, the packet walk and its work-stealing hooks

namespace rolo {
namespace {

// Sharded K5 with peers (rolo_peer_*): the exchange buffer alternates between two areas of the rank's mailbox. WHICH one a frame uses is the parity
// of the exchange's number, read on the device from the own mailbox's epoch word (peer.hip bumps it once per exchange, in stream order) — a captured
// hipGraph replays the same launches frame after frame, so the host cannot bake the area in. add = 1 before the exchange of this frame (tail), 0 after it (unstage).
ROLO_DEV double* stage_area(const KnnCloud& cl, unsigned add) {
  double* s = cl.stage;
  if (cl.stage_epoch) {
    const unsigned long long e = __hip_atomic_load(cl.stage_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + add;
    if (e & 1ull) s += cl.stage_alt;
  }
  return s;
}

// Instrumented build (-DROLO_KNN_STATS): one record per wavefront of the walk. Everything is counted in scalar registers and stored once at
// the very end; the start time comes from a NON-volatile asm that also produces the root node index, so it cannot move. (A store, an atomic, a
// clock builtin or a volatile asm before the loop is a potential memory clobber to the compiler: after it the wave-uniform box / leaf loads are
// no longer provably unclobbered and turn into vector loads — 64 more VGPRs, spills, a 5-10x slower walk. That is what a first version measured.)
#ifdef ROLO_KNN_STATS
__device__ unsigned g_knn_wave_rec[16384][8];   // nodes, leaves, insert executions, pushes, start, end (100 MHz wall clock), lanes live summed over the
                                                 // insert executions, per-leaf maximum over the lanes of the candidates accepted at leaf entry (summed)
#endif

// neighbour lists of one finished packet -> the debug lists (rolo_get_knn), the next round's lower bound (k > 64), the slot-major index array for knn_tail_kernel
template <int KMAX>
ROLO_DEV void walk_write_lists(const KnnCloud& cl, const double (&K)[KMAX], int kk, double bkey, int qi, int j) {
  int ki[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) ki[u] = key_idx(K[u]);
  const int slot0 = (KMAX == 64) ? cl.slot0 : 0;                          // rounds exist for the 64-slot kernel only
  const int ktot = (KMAX == 64 && cl.k_total) ? cl.k_total : kk;
  if (cl.knn_idx) {
#pragma unroll
    for (int u = 0; u < KMAX; u++) if (u < kk) { cl.knn_idx[(size_t)qi * ktot + slot0 + u] = ki[u]; cl.knn_d2[(size_t)qi * ktot + slot0 + u] = key_d2(K[u]); }
  }
  if (KMAX == 64 && cl.lower) cl.lower[j] = bkey;   // the next round starts above this round's last key
  int32_t* __restrict__ nbr = cl.nbr;
#pragma unroll
  for (int u = 0; u < KMAX; u++) nbr[(size_t)(slot0 + u) * cl.n_sorted + j] = ki[u];
}

// Round 6: a finished query gathers its kk winners ONCE, right here, and leaves the six centred second moments of its neighbourhood (the oracle's order: mean first,
// then the products summed over the list, rot_vgicp_impl.hpp:438-455) in cov[] — 48 B/pt that the tail finishes in place — instead of 80 B/pt of indices that the tail
// read back and gathered through again (70 MB of fabric traffic for 6 MB of algorithmic bytes, 202 VGPRs: round 5's verdict, item 4). The light wavefronts do this while
// the heavy ones still walk; it is the gather and ~250 fp64 instructions, not the SVD (the whole tail inside the walk lost: ROLO_KNN_FUSE_TAIL).
template <int KMAX>
ROLO_DEV void walk_write_moments(const KnnCloud& cl, const double (&K)[KMAX], int kk, int qi, int j, int (*s_ki)[256]) {
  if (cl.knn_idx) {   // the debug lists (rolo_get_knn)
#pragma unroll
    for (int u = 0; u < KMAX; u++) if (u < kk) { cl.knn_idx[(size_t)qi * kk + u] = key_idx(K[u]); cl.knn_d2[(size_t)qi * kk + u] = key_d2(K[u]); }
  }
  // The indices go to this lane's column of an LDS array and the two passes over the list are REAL loops (five gathers in flight): with the list in registers the loops
  // must be unrolled, the scheduler hoists all twenty gathers of a pass, and the walk pays for 60 more registers with half its occupancy (8 % of the headline: DEAD_ENDS,
  // round 6). The second pass finds the points in the caches.
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < KMAX; u++) s_ki[u][tid] = key_idx(K[u]);
  const float4* __restrict__ orig = cl.xyz;
  double mx = 0, my = 0, mz = 0;
#pragma unroll 5
  for (int u = 0; u < kk; u++) { const float4 p = orig[s_ki[u][tid]]; mx += (double)p.x; my += (double)p.y; mz += (double)p.z; }
  mx /= kk; my /= kk; mz /= kk;
  double cxx = 0, cxy = 0, cxz = 0, cyy = 0, cyz = 0, czz = 0;
#pragma unroll 5
  for (int u = 0; u < kk; u++) {
    const float4 p = orig[s_ki[u][tid]];
    const double ax = (double)p.x - mx, ay = (double)p.y - my, az = (double)p.z - mz;
    cxx += ax * ax; cxy += ax * ay; cxz += ax * az; cyy += ay * ay; cyz += ay * az; czz += az * az;
  }
  // by SORTED position (coalesced here and in the tail), in the scratch the index lists would have taken: 6 doubles of the 32 int slots per position
  double* __restrict__ mom = reinterpret_cast<double*>(cl.nbr);
  const size_t pitch = (size_t)cl.n_sorted;
  mom[j] = cxx / kk; mom[pitch + j] = cxy / kk; mom[2 * pitch + j] = cxz / kk; mom[3 * pitch + j] = cyy / kk; mom[4 * pitch + j] = cyz / kk; mom[5 * pitch + j] = czz / kk;
}

// The walk keeps only the 20 packed keys and the query live (58 VGPRs): cut for 8 wavefronts per SIMD. Neighbour indices
// go to A.c[].nbr, slot-major so every store is coalesced; knn_tail_kernel turns them into covariances.
#ifndef ROLO_KNN_WALK_OCC
#define ROLO_KNN_WALK_OCC 4   // 128 VGPRs allowed: the loop needs 61, the slack buys the compiler ~3 % (0.196 -> 0.188 ms); a 2 x 131 072-point pair fills 4 waves per SIMD
#endif
#ifndef ROLO_KNN_PACKET
#define ROLO_KNN_PACKET 64   // queries per wavefront (32- and 16-query packets were measured and lost: DESIGN.md section 9)
#endif
#ifndef ROLO_KNN_SEED_EXTRA
#define ROLO_KNN_SEED_EXTRA 1   // leaves on either side of the wavefront's own ones (along the curve) scored before the tree walk starts: curve neighbours are space neighbours,
                                // so every lane enters the walk with a tighter bound (packets, 0 / 1 / 2 / 4: walk 0.188 / 0.177 / 0.178 / 0.187 ms at 2 x 131 072 points, 0.147 / 0.136 / 0.138 / 0.138 at 2 x 65 536;
                                // the sub-lane walks, 0 / 1 / 2: 0.194 / 0.179 / 0.180 ms over the pool, 0.138 / 0.082 / 0.080 at 2 x 43 776)
#endif
static_assert(ROLO_KNN_PACKET == 64, "one query per lane");

// the seeds of a packet: its own 64 / KNN_LEAF leaves first, then ROLO_KNN_SEED_EXTRA leaves on either side (the own points are the nearer ones, so
// fewer keys are inserted only to be pushed out again: walk 0.1763 -> 0.1725 ms at 2 x 131 072 points, 0.1486 -> 0.1470 at 2 x 43 776)
template <int KMAX, bool LOWER, bool ASM>
ROLO_DEV void walk_seeds(const float4* __restrict__ sorted, int g_mine0, int g_own0, int g_own1, int n_leaves, const float4& q, double (&K)[KMAX], int kk, double& bkey, float& bd, double lo,
                         unsigned& st_leaves, unsigned& st_ins, unsigned& st_lane, unsigned& st_rounds) {
  const int n_own = min(g_mine0 + 64 / KNN_LEAF, n_leaves) - g_mine0, n_before = g_mine0 - g_own0;
  for (int i = 0; i < g_own1 - g_own0; i++) {
    const int g = i < n_own ? g_mine0 + i : (i - n_own < n_before ? g_own0 + (i - n_own) : g_mine0 + (i - n_before));
    knn_score_leaf<KMAX, LOWER, false, ASM>(sorted, g, q, K, kk, bkey, bd, st_ins, st_lane, st_rounds, lo);
    st_leaves++;
  }
}

template <int KMAX, bool FUSE_TAIL, bool LOWER = false, bool MOMENTS = false>
__global__ __launch_bounds__(256, KMAX > 32 ? 2 : (MOMENTS ? 6 : ROLO_KNN_WALK_OCC)) void knn_walk_kernel(KnnPair A, int split, int k, int reg) {   // 64 slots = 128 key registers: 256 VGPRs, 2 waves per SIMD
  __shared__ int stk[4][WALK_STACK];
  __shared__ int s_ki[MOMENTS ? KMAX : 1][256];   // MOMENTS: every lane's neighbour indices for the epilogue's loops (walk_write_moments)
  const int tid = threadIdx.x;
  const int wv = tid >> 6;
  // which cloud of the pair this workgroup searches (wave-uniform: everything below stays in scalar registers)
  const int blk = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x, 4);
  const int which = blk >= split ? 1 : 0;
  const float4* __restrict__ sorted = A.c[which].sorted;
  const float4* __restrict__ boxes = A.c[which].boxes;
  const int n_sorted = A.c[which].n_sorted, P = A.c[which].P;
  const int lane_ = tid & 63;
  const int j = A.c[which].q_begin + (blk - (which ? split : 0)) * 256 + wv * 64 + lane_;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (j < A.c[which].q_end) { q = sorted[j]; qi = __float_as_int(q.w); }
  const bool active = qi != INT_MAX;  // not padding
  const int kk = (KMAX == 20) ? 20 : k;
  const int n_leaves = n_sorted / KNN_LEAF;
  unsigned st_nodes = 0, st_leaves = 0, st_ins = 0, st_lane = 0, st_rounds = 0, st_push = 0;

  // K[0..KMAX) ascending; sentinel = (inf, INT_MAX)
  const double sentinel = key_pack(INFINITY, INT_MAX);
  double K[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) K[u] = sentinel;
  // pruning radius of this lane: d2 of its kk-th best so far; inactive lanes never reach anything
  float bd = active ? INFINITY : -1.0f;
  double bkey = active ? sentinel : key_pack(0.f, 0);

  // ---- seed: the wavefront's own 64 / KNN_LEAF leaves ----
  const int g_mine0 = __builtin_amdgcn_readfirstlane(j / KNN_LEAF);  // lane 0 of the wave: j is a multiple of 64
  const int g_own0 = max(g_mine0 - ROLO_KNN_SEED_EXTRA, 0);
  const int g_own1 = min(g_mine0 + 64 / KNN_LEAF + ROLO_KNN_SEED_EXTRA, n_leaves);
  // rounds of a search for more than 64 neighbours (LOWER): only keys above the previous round's last one count
  double lo = 0.0;
  if (LOWER && active) lo = A.c[which].lower[j];
  (void)lo;
  walk_seeds<KMAX, LOWER, KNN_PLAIN_ASM>(sorted, g_mine0, g_own0, g_own1, n_leaves, q, K, kk, bkey, bd, lo, st_leaves, st_ins, st_lane, st_rounds);

  // ---- packet walk ----
  // (a stack in one vector register — slot i in lane i, v_writelane / v_readlane — measured the same as this LDS stack: 0.202 vs 0.200 ms;
  // gfx950's v_writelane takes its lane select from M0 when the value is an SGPR, which inline asm may not clobber safely)
  int sp = 0;
  int h = 1;
#ifdef ROLO_KNN_STATS
  unsigned long long wt0;
  asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %1, 1" : "=s"(wt0), "=s"(h));
#endif
  { const CoopPub none{}; int n_scored = 0, n_pub = 0;
    packet_walk<KMAX, LOWER, false, false, KNN_PLAIN_ASM>(sorted, boxes, P, g_own0, g_own1, q, K, kk, bkey, bd, lo, 0.0, (lds_int*)&stk[wv][0], sp, h, none, n_scored, n_pub, st_nodes, st_leaves, st_ins, st_lane, st_rounds, st_push); }
#ifdef ROLO_KNN_STATS
  { unsigned long long wt1; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt1));
    unsigned st_rounds_rec = st_rounds;
#ifdef ROLO_KNN_STATS2
    for (int off = 32; off > 0; off >>= 1) st_rounds_rec = max(st_rounds_rec, (unsigned)__shfl_xor((int)st_rounds_rec, off, 64));
#endif
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && wid < 16384) {
      g_knn_wave_rec[wid][0] = st_nodes; g_knn_wave_rec[wid][1] = st_leaves; g_knn_wave_rec[wid][2] = st_ins; g_knn_wave_rec[wid][3] = st_push;
      g_knn_wave_rec[wid][4] = (unsigned)wt0; g_knn_wave_rec[wid][5] = (unsigned)wt1; g_knn_wave_rec[wid][6] = st_lane; g_knn_wave_rec[wid][7] = st_rounds_rec;
    } }
#endif
  (void)st_nodes; (void)st_leaves; (void)st_ins; (void)st_lane; (void)st_rounds; (void)st_push;

  if (!active) return;

  if (MOMENTS) {   // the neighbourhood's moments instead of its indices (walk_write_moments)
    walk_write_moments<KMAX>(A.c[which], K, kk, qi, j, s_ki);
    return;
  }
  if (!FUSE_TAIL) {   // neighbour indices only (slot-major, coalesced): knn_tail_kernel turns them into covariances
    walk_write_lists<KMAX>(A.c[which], K, kk, bkey, qi, j);
    return;
  }
  // FUSE_TAIL (ROLO_KNN_FUSE_TAIL=1, an A/B): covariance + regularisation right here, so that the light wavefronts do theirs while the heavy
  // ones are still walking and the 42 MB of index traffic disappear. Measured: 0.2335 ms against 0.1884 + 0.0394 ms for walk + tail launch,
  // and the 4-context throughput falls from 2750 to 2490 scans/s — the tails' fp64 work competes with the walking wavefronts for issue slots.
  const KnnCloud& cl = A.c[which];
  int ki[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) ki[u] = key_idx(K[u]);
  if (cl.knn_idx) {
#pragma unroll
    for (int u = 0; u < KMAX; u++) if (u < kk) { cl.knn_idx[(size_t)qi * kk + u] = ki[u]; cl.knn_d2[(size_t)qi * kk + u] = key_d2(K[u]); }
  }
  if (cl.stage) {   // multi-GPU: into the exchange buffer, sorted order
    double* o = stage_area(cl, 1) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
    double c6[6]; knn_covariance_tail<KMAX>(ki, kk, cl.xyz, 1, 0, reg, o, c6);
  } else {
    double c6[6]; knn_covariance_tail<KMAX>(ki, kk, cl.xyz, cl.n, qi, reg, cl.cov, c6, cl.nrm);
  }
}

// ---- four lanes per query (round 4): the walk of the SMALL clouds ------------------------------------------------------------------------------------
// The pipeline's feature clouds (~48 k points) are 760 packets of 64: not one wavefront per SIMD, every one a chain of ~100 dependent fetches at
// ~1 us each — the launch (0.137 ms, a quarter of the frame) lasts as long as its longest chain and the chip idles. Here a wavefront carries 16 queries
// (one leaf's worth), four adjacent lanes per query: four times the wavefronts, each with a shorter chain (a quarter packet's frontier), a visited leaf
// costs each lane 4 candidates (point u of the leaf goes to sub-lane u % 4), and where the tree allows it a step takes TWO levels with the four
// grandchild boxes tested one per sub-lane. Each sub-lane keeps the 20 best of ITS candidates; the query's pruning bound is shared by its four lanes:
//     B = min( min_s K_s[19],  max_s K_s[4] )
// — twenty real candidates at or below it either way (twenty in one list; five in each of four), so nothing beyond B can belong to the twenty nearest
// and the walk stays exact. At the end the four lists are merged pairwise (two bitonic merges in registers, the partner's keys through DPP):
// the same 20 (d2, index) keys in the same order as the one-lane search — lists and float distances bit-identical (test_knn_lists_bit_exact runs both).
// The 2 x 131 072-point frame keeps the 64-query packets: there the inserts bound the walk, not the chains (launch_knn_walk picks by size).
// K <- the twenty smallest of K and the partner's K, sorted (both lanes end up with the same list), as a bitonic merge in place: with a and b ascending,
// c[i] = min(a[i], b[19 - i]) holds exactly the twenty smallest of the forty and rises, then falls; the half-cleaner network of 32 inputs sorts it — the
// sequence padded IN FRONT with twelve -inf that never move, so only the 40 compare-exchanges between real positions are issued. ~140 instructions where
// twenty sorted inserts cost up to 800. (Pairs (j, 19 - j) are fetched before either is overwritten: the partner runs the same instructions on its list.)
template <int ST> ROLO_DEV void merge20(double (&K)[20]) {
#pragma unroll
  for (int j = 0; j < 10; j++) {
    const double o1 = sub_xchg<ST>(K[19 - j]), o2 = sub_xchg<ST>(K[j]);
    K[j] = vmin_f64(K[j], o1);
    K[19 - j] = vmin_f64(K[19 - j], o2);
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {
#pragma unroll
    for (int p = 12; p < 32; p++) {
      if ((p & d) == 0) {
        const int x = p - 12, y = p + d - 12;
        const double lo = vmin_f64(K[x], K[y]);
        K[y] = vmax_f64(K[x], K[y]);
        K[x] = lo;
      }
    }
  }
}

// ROLO_KNN_WALK_MAXOCC (A/B builds): an upper bound on the wavefronts per SIMD the compiler allocates registers for — the way to keep wave slots free for
// other contexts' kernels WITHOUT an LDS pad (which takes the CU's LDS away from them as well)
#ifdef ROLO_KNN_WALK_MAXOCC
#define ROLO_KNN_WALK_OCC_ATTR __attribute__((amdgpu_waves_per_eu(ROLO_KNN_WALK_MAXOCC, ROLO_KNN_WALK_MAXOCC)))
#else
#define ROLO_KNN_WALK_OCC_ATTR
#endif
template <int SUB, bool MOMENTS = false>
__global__ __launch_bounds__(256, MOMENTS ? 6 : ROLO_KNN_WALK_OCC) ROLO_KNN_WALK_OCC_ATTR void knn_walk_sub_kernel(KnnPair A, int split /* first block of cloud 1 */) {
  constexpr int SH = SUB == 2 ? 1 : 2, QPW = 64 / SUB, PPL = KNN_LEAF / SUB, KMAX = 20, E = 4 / SUB /* grandchild boxes per lane */, OWN = QPW / KNN_LEAF;
  static_assert(SUB == 2 || SUB == 4, "lanes per query");
  static_assert(KNN_LEAF == 16, "a wavefront's queries are whole leaves");
  __shared__ int stk_[4][WALK_STACK];
  __shared__ int s_ki[MOMENTS ? KMAX : 1][256];
  const int tid = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, sub = lane & (SUB - 1), ql = lane >> SH;
  const int blk = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x, 4);
  const int which = blk >= split ? 1 : 0;
  const KnnCloud& cl = A.c[which];
  const float4* __restrict__ sorted = cl.sorted;
  const float4* __restrict__ boxes = cl.boxes;
  const int n_sorted = cl.n_sorted, P = cl.P, n_leaves = n_sorted / KNN_LEAF;
  const int j0 = cl.q_begin + ((blk - (which ? split : 0)) * 4 + wv) * QPW;   // wave-uniform
  const int j = j0 + ql;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (j < cl.q_end) { q = sorted[j]; qi = __float_as_int(q.w); }
  const bool active = qi != INT_MAX;   // not padding
  const double sentinel = key_pack(INFINITY, INT_MAX);
  double K[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) K[u] = sentinel;
  double B = active ? sentinel : key_pack(0.f, 0);   // (no key is below (0, 0): an idle query accepts nothing)
  float bd = active ? INFINITY : -1.0f;
  auto score = [&](int g) {
    const float4* __restrict__ leaf = sorted + KNN_LEAF * (size_t)g;
    bool changed = false;
#pragma unroll
    for (int t0 = 0; t0 < PPL; t0 += 4) {   // four candidates in flight at a time
      float4 c[4];
#pragma unroll
      for (int t = 0; t < 4; t++) c[t] = leaf[(t0 + t) * SUB + sub];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const float dx = q.x - c[t].x, dy = q.y - c[t].y, dz = q.z - c[t].z;
        const float cd = ((dx * dx) + (dy * dy)) + (dz * dz);   // (-ffp-contract=off)
        const double ck = key_pack(cd, __float_as_int(c[t].w));
        if (ck < B) { insert_tiered<KMAX>(K, ck); changed = true; }
      }
    }
    if (__any(changed)) {
      B = vmin_f64(vmin_f64(B, sub_min<SUB>(K[KMAX - 1])), sub_max<SUB>(K[KMAX / SUB - 1]));
      bd = key_d2(B);
    }
  };
  // ---- seeds: the wavefront's own leaves, then ROLO_KNN_SEED_EXTRA leaves on either side along the curve ----
  const int g_mine = min(j0 / KNN_LEAF, n_leaves - 1);
  const int g_own0 = max(g_mine - ROLO_KNN_SEED_EXTRA, 0), g_own1 = min(g_mine + OWN + ROLO_KNN_SEED_EXTRA, n_leaves);
  for (int g = g_mine; g < min(g_mine + OWN, n_leaves); g++) score(g);
  for (int g = g_own0; g < g_own1; g++) if (g < g_mine || g >= g_mine + OWN) score(g);
  // ---- the walk ----
  {
    lds_int* stk = (lds_int*)&stk_[wv][0];
    int sp = 0, h = 1;
    while (true) {
      h = __builtin_amdgcn_readfirstlane(h);
      if (2 * h < P) {
        // two levels per step: the four grandchildren of h (nodes 4h .. 4h + 3, their boxes 128 contiguous bytes) are tested by the query's SUB lanes, E each
        float d[E]; bool ok[E];
        float dm = INFINITY;   // the query's nearest live grandchild votes (ties: every one at the minimum)
#pragma unroll
        for (int e = 0; e < E; e++) {
          const int c = sub * E + e;
          const float4 blo = boxes[8 * (size_t)h + 2 * c], bhi = boxes[8 * (size_t)h + 2 * c + 1];
          d[e] = box_d2(blo, bhi, q);
          ok[e] = (d[e] <= bd) && (d[e] < INFINITY);
          dm = fminf(dm, ok[e] ? d[e] : INFINITY);
        }
        dm = fminf(dm, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(dm), 0xB1, 0xF, 0xF, true)));
        if (SUB == 4) dm = fminf(dm, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(dm), 0x4E, 0xF, 0xF, true)));
        unsigned long long m[E], v[E];
#pragma unroll
        for (int e = 0; e < E; e++) { m[e] = __ballot(ok[e]); v[e] = __ballot(ok[e] && d[e] == dm); }
        int key[4];   // wave-uniform: votes * 4 + (3 - c) for a live grandchild, -1 for one no lane reaches
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const unsigned long long sel = (SUB == 4 ? 0x1111111111111111ull : 0x5555555555555555ull) << (c / E);   // the lanes that tested grandchild c
          key[c] = (m[c % E] & sel) != 0ull ? __popcll(v[c % E] & sel) * 4 + (3 - c) : -1;
        }
        auto cx = [](int& a, int& bb) { const int hi = max(a, bb), lo = min(a, bb); a = hi; bb = lo; };
        cx(key[0], key[1]); cx(key[2], key[3]); cx(key[0], key[2]); cx(key[1], key[3]); cx(key[1], key[2]);   // descending
        if (key[0] >= 0) {
#pragma unroll
          for (int r = 3; r >= 1; r--) if (key[r] >= 0 && sp < WALK_STACK) { stk[sp] = 4 * h + (3 - (key[r] & 3)); sp++; }   // worst first: the second best is popped first
          h = 4 * h + (3 - (key[0] & 3));
          continue;
        }
      } else if (h < P) {
        // one level (the children of h are leaves): even sub-lanes test the left child, odd ones the right, the partner's distance comes through DPP
        const int c = sub & 1;
        const float4 blo = boxes[4 * (size_t)h + 2 * c], bhi = boxes[4 * (size_t)h + 2 * c + 1];
        const float d = box_d2(blo, bhi, q);
        const float o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(d), 0xB1, 0xF, 0xF, true));
        const bool ok = (d <= bd) && (d < INFINITY), oko = (o <= bd) && (o < INFINITY);
        const unsigned long long m = __ballot(ok);
        const unsigned long long ml = m & 0x5555555555555555ull, mr = m & 0xaaaaaaaaaaaaaaaaull;
        if (ml != 0ull && mr != 0ull) {
          const unsigned long long pref = __ballot(c == 0 && (ok || oko) && (d <= o));    // even lanes: d = left, o = right
          const unsigned long long either = (ml | (mr >> 1)) & 0x5555555555555555ull;    // even lanes of the queries that reach a child
          const bool left_first = 2 * __popcll(pref) >= __popcll(either);
          if (sp < WALK_STACK) { stk[sp] = left_first ? 2 * h + 1 : 2 * h; sp++; }
          h = left_first ? 2 * h : 2 * h + 1;
          continue;
        }
        if (ml != 0ull) { h = 2 * h; continue; }
        if (mr != 0ull) { h = 2 * h + 1; continue; }
      } else {
        const int g = h - P;
        if (g < g_own0 || g >= g_own1) score(g);   // (the seeds were scored already)
      }
      if (sp == 0) break;
      sp--;
      h = stk[sp];
    }
  }
  merge20<0>(K);
  if (SUB == 4) merge20<1>(K);   // all the query's lanes hold its list now
  if (MOMENTS) {   // the first lane of a query gathers its neighbourhood (one lane: the sums must run in list order, as the oracle's do) — walk_write_moments
    if (active && sub == 0) walk_write_moments<KMAX>(cl, K, KMAX, qi, j, s_ki);
    return;
  }
  // every sub-lane writes 20 / SUB of the twenty slots (slot-major index array for knn_tail_kernel: a store covers SUB slots x QPW queries)
  constexpr int NS = KMAX / SUB;
  int ki[NS];
#pragma unroll
  for (int t = 0; t < NS; t++) {
    if (SUB == 4) {
      const double k01 = sub & 1 ? K[NS + t] : K[t], k23 = sub & 1 ? K[3 * NS + t] : K[2 * NS + t];
      ki[t] = key_idx(sub & 2 ? k23 : k01);
    } else {
      ki[t] = key_idx(sub & 1 ? K[NS + t] : K[t]);
    }
  }
  if (!active) return;
  int32_t* __restrict__ nbr = cl.nbr;
#pragma unroll
  for (int t = 0; t < NS; t++) nbr[(size_t)(sub * NS + t) * n_sorted + j] = ki[t];
  if (cl.knn_idx && sub == 0) {   // the debug lists (rolo_get_knn)
#pragma unroll
    for (int u = 0; u < KMAX; u++) { cl.knn_idx[(size_t)qi * KMAX + u] = key_idx(K[u]); cl.knn_d2[(size_t)qi * KMAX + u] = key_d2(K[u]); }
  }
}

// ---- the cooperative walk: a heavy packet's sub-trees are stolen by the idle wavefronts of its workgroup ------------------------------------------
// The plain walk lasts as long as its heaviest packets: mean wavefront 78 us, p99 134 us, max 214 us of a 168 us kernel — packets whose 64
// curve-adjacent queries lie on scattered fragments pay the SUM of what every lane needs, and no static quantity predicts them (DESIGN.md
// sections 4, 9). Here a workgroup is NW wavefronts = NW packets (NW / 4 runs of four consecutive packets from distant stretches of the curve, the
// mix a CU gets from the dispatcher under the plain kernel), and work moves between them through LDS:
//   * every wavefront walks its own packet. Once it has scored `budget` leaves it is a donor: whenever the small ring it publishes to is empty it
//     moves the bottom entries of its stack (the largest sub-trees) and its lanes' current bounds there, and goes on at the top of its stack;
//   * a wavefront whose packet is done writes its lists and turns thief: it picks a donor of its workgroup, steals entries from that ring and walks
//     them WITH THE DONOR'S 64 queries into a fresh, empty list whose bound is capped by the donor's published bound (CAP) — so it collects exactly
//     the points of those sub-trees that beat that bound — until the donor is done, and leaves the list in an LDS result slot;
//   * after a workgroup barrier every donor merges the result lists addressed to it into its own with the same min / max network and writes.
// Exactness: stolen sub-trees are disjoint from each other and from everything the donor scores itself, every point of the final k beats every
// bound ever published (bounds only tighten), so the k smallest of (donor's list + result lists) are the k smallest of the whole cloud: lists and
// float distances stay bit-identical to the oracle's. LDS traffic is a few words per hand-over; nothing crosses workgroups (device-scope
// atomics would: the XCDs' L2s are not coherent with each other without write-back / invalidate).
// Measured and not kept: (1) TWO launches — budgeted walk + a continuation launch with eight wavefronts per unfinished packet: slower than the
// plain walk at every budget (0.266 against 0.220 ms at 28 leaves; the second launch starts with cold L2s and thousands of empty workgroups);
// (2) publishing the stack ONCE at the budget and popping from it: the donor itself walks the big entries it pops, the ring is empty by the
// time anybody is free to help (0.217 against 0.223 ms).
template <int NW> struct CoopCfg { static constexpr int NSLOT = NW == 16 ? 14 : (NW == 8 ? 7 : 3); };   // result slots of 10 KB: the workgroup's LDS share (160 KB per CU at 4 waves per SIMD)

template <int NW>
__global__ __launch_bounds__(64 * NW, ROLO_KNN_WALK_OCC) void knn_walk_coop_kernel(KnnPair A, int split4 /* first 4-packet block of cloud 1 */, int G4 /* 4-packet blocks */, int budget) {
  constexpr int KMAX = 20, kk = 20, NSLOT = CoopCfg<NW>::NSLOT;
  __shared__ int pstk[NW][WALK_STACK];     // private stacks
  __shared__ int ring[NW][COOP_RING];      // published sub-trees
  __shared__ double d_cap[NW][64];         // published bounds (the donor's k-th key per lane)
  __shared__ double res[NSLOT][KMAX][64];  // result lists of thieves
  __shared__ int r_head[NW], r_tail[NW], r_done[NW], d_j0[NW], d_which[NW];
  __shared__ int res_owner[NSLOT];         // the donor a result list belongs to (-1: none)
  __shared__ int n_res, n_p1;              // result slots handed out; wavefronts done with their own packet
  const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), lane_ = tid & 63;   // (w in a scalar register: everything derived from it — cloud, tree, first query — stays wave-uniform for the compiler)
  // this wavefront's packet: run k = w / 4 of the workgroup is 4-packet block (blockIdx.x + gridDim.x * k) of the plain walk's launch geometry
  const int vb = (int)blockIdx.x + (int)gridDim.x * (w >> 2);
  const int blk4 = vb < G4 ? xcd_contiguous_block(vb, G4, 4) : G4;
  const int which = (blk4 >= split4 && A.n_clouds > 1) ? 1 : 0;
  const KnnCloud& cl = A.c[which];
  const int j0 = blk4 < G4 ? cl.q_begin + (blk4 - (which ? split4 : 0)) * 256 + (w & 3) * 64 : cl.q_end;
  if (lane_ == 0) { r_head[w] = 0; r_tail[w] = 0; r_done[w] = 0; d_j0[w] = j0; d_which[w] = which; }
  if (tid < NSLOT) res_owner[tid] = -1;
  if (tid == 0) { n_res = 0; n_p1 = 0; }
  __syncthreads();
  const int j = j0 + lane_;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (j < cl.q_end) { q = cl.sorted[j]; qi = __float_as_int(q.w); }
  const bool active = qi != INT_MAX;
  unsigned st_nodes = 0, st_leaves = 0, st_ins = 0, st_lane = 0, st_rounds = 0, st_push = 0;
  const double sentinel = key_pack(INFINITY, INT_MAX);
  double K[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) K[u] = sentinel;
  float bd = active ? INFINITY : -1.0f;
  double bkey = active ? sentinel : key_pack(0.f, 0);
#ifdef ROLO_KNN_STATS
  unsigned long long ct0, ct1 = 0, ct2 = 0; unsigned c_sessions = 0, c_p1_leaves = 0, c_full = 0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct0));
#endif

  // ---- the own packet ----
  int n_published = 0;
  {
    const float4* __restrict__ sorted = cl.sorted;
    const float4* __restrict__ boxes = cl.boxes;
    const int P = cl.P, n_leaves = cl.n_sorted / KNN_LEAF;
    const int g_mine0 = __builtin_amdgcn_readfirstlane(j0 / KNN_LEAF);
    const int g_own0 = max(g_mine0 - ROLO_KNN_SEED_EXTRA, 0);
    const int g_own1 = min(g_mine0 + 64 / KNN_LEAF + ROLO_KNN_SEED_EXTRA, n_leaves);
    walk_seeds<KMAX, false, true>(sorted, g_mine0, g_own0, g_own1, n_leaves, q, K, kk, bkey, bd, 0.0, st_leaves, st_ins, st_lane, st_rounds);
    const CoopPub pub{(lds_int*)&ring[w][0], (lds_int*)&r_head[w], (lds_int*)&r_tail[w], (lds_double*)&d_cap[w][0], budget};
    int sp = 0, n_scored = 0, h = 1;
    while (h >= 0) {   // the tree from the root, then whatever of the own published sub-trees nobody took
      packet_walk<KMAX, false, false, true, true>(sorted, boxes, P, g_own0, g_own1, q, K, kk, bkey, bd, 0.0, 0.0, (lds_int*)&pstk[w][0], sp, h, pub, n_scored, n_published,
                                            st_nodes, st_leaves, st_ins, st_lane, st_rounds, st_push);
      h = n_published ? coop_steal(pub.ring, pub.head, pub.tail) : -1;
    }
    if (lane_ == 0) {
      __hip_atomic_store(&r_done[w], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&n_p1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    KNN_STAT(c_p1_leaves = st_leaves; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct1));)
  }
  const bool donor = n_published > 0;
  // a packet nobody helped with is final: its lists go out now, so that no list is live in registers while the wavefront helps
  // (global stores before further walks are harmless: the tree fetches are explicit scalar loads)
  if (!donor && active) walk_write_lists<KMAX>(cl, K, kk, bkey, qi, j);

  // ---- thief: ONE donor per wavefront (its result list needs an LDS slot until the barrier); donors wait at the barrier, their lists stay in registers ----
  while (!donor) {
    int d = -1;
    for (int o = 1; o < NW; o++) {
      const int c = (w + o) % NW;
      if (__hip_atomic_load(&r_head[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > __hip_atomic_load(&r_tail[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { d = c; break; }
    }
    d = __builtin_amdgcn_readfirstlane(d);
    if (d < 0) {
      if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&n_p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= NW) break;   // nobody is left who could publish
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    int slot = 0;
    if (lane_ == 0) slot = __hip_atomic_fetch_add(&n_res, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= NSLOT) { KNN_STAT(c_full = 1;) break; }   // no room for another result list: the donors finish on their own
    KNN_STAT(c_sessions++;)
    // the donor's packet
    const KnnCloud& dc = A.c[__builtin_amdgcn_readfirstlane(d_which[d])];
    const float4* __restrict__ sorted = dc.sorted;
    const float4* __restrict__ boxes = dc.boxes;
    const int P = dc.P, n_leaves = dc.n_sorted / KNN_LEAF;
    const int jd0 = __builtin_amdgcn_readfirstlane(d_j0[d]), jd = jd0 + lane_;
    float4 qd = make_float4(0.f, 0.f, 0.f, 0.f);
    bool act_d = false;
    if (jd < dc.q_end) { qd = sorted[jd]; act_d = __float_as_int(qd.w) != INT_MAX; }
    const int gd0 = __builtin_amdgcn_readfirstlane(jd0 / KNN_LEAF);
    const int gd_own0 = max(gd0 - ROLO_KNN_SEED_EXTRA, 0), gd_own1 = min(gd0 + 64 / KNN_LEAF + ROLO_KNN_SEED_EXTRA, n_leaves);
    double L[KMAX];
#pragma unroll
    for (int u = 0; u < KMAX; u++) L[u] = sentinel;
    double bk = act_d ? sentinel : key_pack(0.f, 0);
    float bdd = -1.0f;
    const CoopPub none{};
    while (true) {
      const int e = coop_steal((lds_int*)&ring[d][0], (lds_int*)&r_head[d], (lds_int*)&r_tail[d]);
      if (e < 0) {
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&r_done[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
        __builtin_amdgcn_s_sleep(4);
        continue;
      }
      const double cap = act_d ? d_cap[d][lane_] : key_pack(0.f, 0);   // the donor's bound as of its last hand-over: it only tightens
      bk = vmin_f64(bk, cap);
      bdd = act_d ? key_d2(bk) : -1.0f;
      int sp = 0, ns = 0, np = 0;
      packet_walk<KMAX, false, true, false, true>(sorted, boxes, P, gd_own0, gd_own1, qd, L, kk, bk, bdd, 0.0, cap, (lds_int*)&pstk[w][0], sp, e, none, ns, np,
                                            st_nodes, st_leaves, st_ins, st_lane, st_rounds, st_push);
    }
#pragma unroll
    for (int u = 0; u < KMAX; u++) res[slot][u][lane_] = L[u];
    if (lane_ == 0) res_owner[slot] = d;
    break;
  }
#ifdef ROLO_KNN_STATS
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct2));
  { const unsigned wid = blockIdx.x * NW + w;   // [0] nodes [1] leaves of the own packet [2] leaves scored as a thief [3] donor | entries published << 8 | sessions << 16 | slots full << 31
    if (lane_ == 0 && wid < 16384) {            // [4] start [5] own packet done [6] stealing done (100 MHz) [7] -
      g_knn_wave_rec[wid][0] = st_nodes; g_knn_wave_rec[wid][1] = c_p1_leaves; g_knn_wave_rec[wid][2] = st_leaves - c_p1_leaves;
      g_knn_wave_rec[wid][3] = (donor ? 1u : 0u) | ((unsigned)min(n_published, 255) << 8) | (c_sessions << 16) | (c_full << 31);
      g_knn_wave_rec[wid][4] = (unsigned)ct0; g_knn_wave_rec[wid][5] = (unsigned)ct1; g_knn_wave_rec[wid][6] = (unsigned)ct2; g_knn_wave_rec[wid][7] = 0;
    } }
#endif
  (void)st_nodes; (void)st_leaves; (void)st_ins; (void)st_lane; (void)st_rounds; (void)st_push;
  __syncthreads();

  // ---- donors merge what the thieves found and write ----
  if (!donor) return;
  const int nr = min(n_res, NSLOT);
  for (int s = 0; s < nr; s++) {
    if (res_owner[s] != w) continue;
    for (int u = 0; u < KMAX; u++) {   // ascending: once no lane's key beats its bound, none of the list's later keys will
      const double ck = res[s][u][lane_];
      if (!__any(ck < bkey)) break;
      if (ck < bkey) {
#pragma unroll
        for (int t = KMAX - 1; t >= 1; t--) insert_slot(K[t], K[t - 1], ck);
        K[0] = vmin_f64(ck, K[0]);
        bkey = K[kk - 1];
      }
    }
  }
  if (!active) return;
  walk_write_lists<KMAX>(cl, K, kk, bkey, qi, j);
}

#ifndef ROLO_KNN_TAIL_OCC
#define ROLO_KNN_TAIL_OCC 2
#endif
template <int KMAX, bool MOMENTS = false>
__global__ __launch_bounds__(256, ROLO_KNN_TAIL_OCC) void knn_tail_kernel(KnnPair A, int split, int k, int reg, VoxelFuse vf) {
  ROLO_TAIL_KERNEL_PRIO();
  // the two clouds' workgroups are dealt ALTERNATELY over the launch: the target's carry the voxel map's accumulation and last longer — with the clouds one behind the
  // other some CUs held four of them (tail 42.8 -> 40.5 us); the tail has no reuse between neighbouring workgroups that an XCD-contiguous order would serve
  int which, blk;
  {
    // (the alternating order is laid over the XCD-contiguous one: neighbouring workgroups of a cloud still share an XCD, whose L2 merges their scattered stores — dealt
    // over the XCDs directly the tail wrote 32 instead of 26 MB)
    const int b = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x), nb0 = split, nb1 = (int)gridDim.x - split, m = nb0 < nb1 ? nb0 : nb1;
    if (b < 2 * m) { which = b & 1; blk = (b >> 1) + (which ? split : 0); }
    else { which = nb0 > nb1 ? 0 : 1; blk = m + (b - 2 * m) + (which ? split : 0); }
  }
  const KnnCloud& cl = A.c[which];
  const int n_sorted = cl.n_sorted;
  const int j = cl.q_begin + (blk - (which ? split : 0)) * 256 + threadIdx.x;
  const bool fuse = vf.enabled && which == vf.which && !cl.stage;   // workgroup-uniform: this workgroup's points also go into the voxel map
                                                                    // (sharded: only a slice is here — the scatter after the exchange accumulates)
  float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (j < cl.q_end) { sp = cl.sorted[j]; qi = __float_as_int(sp.w); }
  const bool act = qi != INT_MAX;
  if (!act && !fuse) return;
  // (the voxel of a target point: two dependent gathers — slot by original index, id by slot — issued HERE, so that they travel while the SVD runs)
  int vox_id = -1;
  if (fuse && act) { const int slot = vf.tgt_slot[qi]; if (slot >= 0) vox_id = slot_id(vf.tab, (unsigned)slot); }
  double c6[6] = {0, 0, 0, 0, 0, 0};
  if (act && MOMENTS) {   // the walk left the neighbourhood's six moments in cov[] (walk_write_moments): regularise them in place
    const size_t pitch = (size_t)n_sorted;
    const double* __restrict__ cv = reinterpret_cast<const double*>(cl.nbr);   // by sorted position: coalesced
    const double m0 = cv[j], m1 = cv[pitch + j], m2 = cv[2 * pitch + j], m3 = cv[3 * pitch + j], m4 = cv[4 * pitch + j], m5 = cv[5 * pitch + j];
    if (cl.stage) {
      double* o = stage_area(cl, 1) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
      knn_covariance_finish(m0, m1, m2, m3, m4, m5, 1, 0, reg, o, c6);
    } else {
      knn_covariance_finish(m0, m1, m2, m3, m4, m5, cl.n, qi, reg, cl.cov, c6, cl.nrm);
    }
  } else if (act) {
    const int32_t* __restrict__ nbr = cl.nbr;
    int ki[KMAX];
#pragma unroll
    for (int u = 0; u < KMAX; u++) ki[u] = nbr[(size_t)u * n_sorted + j];
    if (cl.stage) {   // multi-GPU: into the exchange buffer, sorted order
      double* o = stage_area(cl, 1) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
      knn_covariance_tail<KMAX>(ki, (KMAX == 20) ? 20 : k, cl.xyz, 1, 0, reg, o, c6);   // pitch 1, index 0: six consecutive doubles
    } else {
      knn_covariance_tail<KMAX>(ki, (KMAX == 20) ? 20 : k, cl.xyz, cl.n, qi, reg, cl.cov, c6, cl.nrm);
    }
  }
#ifdef ROLO_VOXEL_ACCUM_WAVE   // (A/B: one set of atomics per RUN of a wavefront, rounds 2-5)
  if (fuse) accumulate_point(vf.tab, vox_id, sp, c6, fix_scales(cl.n, vf.counters), true, const_cast<int*>(vf.counters) + 1);   // every lane of the wavefront takes part in the segmented fold
#else
  if (fuse) accumulate_point_wg<256>(vf.tab, vox_id, sp, c6, fix_scales(cl.n, vf.counters), const_cast<int*>(vf.counters) + 1);   // every thread of the workgroup takes part (fuse is workgroup-uniform)
#endif
}

// k_correspondences > 64: the same kernel with the neighbour slots walked in a loop (knn_covariance_tail_loop) — correct, not tuned
__global__ __launch_bounds__(256) void knn_tail_loop_kernel(KnnPair A, int split, int k, int reg, VoxelFuse vf) {
  const int blk = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
  const int which = blk >= split ? 1 : 0;
  const KnnCloud& cl = A.c[which];
  const int j = cl.q_begin + (blk - (which ? split : 0)) * 256 + threadIdx.x;
  const bool fuse = vf.enabled && which == vf.which && !cl.stage;
  float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (j < cl.q_end) { sp = cl.sorted[j]; qi = __float_as_int(sp.w); }
  const bool act = qi != INT_MAX;
  if (!act && !fuse) return;
  double c6[6] = {0, 0, 0, 0, 0, 0};
  if (act) {
    if (cl.stage) {
      double* o = stage_area(cl, 1) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
      knn_covariance_tail_loop(cl.nbr, (size_t)cl.n_sorted, j, k, cl.xyz, 1, 0, reg, o, c6);
    } else {
      knn_covariance_tail_loop(cl.nbr, (size_t)cl.n_sorted, j, k, cl.xyz, cl.n, qi, reg, cl.cov, c6, cl.nrm);
    }
  }
  if (fuse) {
    int id = -1;
    if (act) { const int slot = vf.tgt_slot[qi]; if (slot >= 0) id = slot_id(vf.tab, (unsigned)slot); }
    accumulate_point_wg<256>(vf.tab, id, sp, c6, fix_scales(cl.n, vf.counters), const_cast<int*>(vf.counters) + 1);
  }
}

// exchange buffer -> cov[] (SoA by original index) for every sorted position (after the all-gather) or the own slice only.
// vf.enabled (sharded registration with all ranks' covariances present): the target's points also go into the voxel map right here —
// the covariance is in registers, the positions come in curve order (long runs of equal voxels per wavefront) — as the tail kernel does on
// one GPU; clear + insert rode on the key / sort launches, which every rank runs on the whole cloud anyway.
__global__ __launch_bounds__(256) void knn_unstage_kernel(KnnPair A, int split, int own_only, VoxelFuse vf) {
  const int which = (int)blockIdx.x >= split ? 1 : 0;
  const KnnCloud& cl = A.c[which];
  const int j = (own_only ? cl.q_begin : 0) + ((int)blockIdx.x - (which ? split : 0)) * 256 + threadIdx.x;
  const bool fuse = vf.enabled && which == vf.which;   // workgroup-uniform
  float4 sp = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (j < (own_only ? cl.q_end : cl.n_sorted)) { sp = cl.sorted[j]; qi = __float_as_int(sp.w); }
  const bool act = qi != INT_MAX;
  if (!act && !fuse) return;
  double c6[6] = {0, 0, 0, 0, 0, 0};
  if (act) {
    const double* __restrict__ o = stage_area(cl, 0) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
    const size_t pitch = (size_t)cl.n;
#pragma unroll
    for (int v = 0; v < 6; v++) { c6[v] = o[v]; cl.cov[v * pitch + qi] = c6[v]; }
  }
  if (fuse) {   // every lane of the wavefront takes part in the segmented fold
    int id = -1;
    if (act) { const int slot = vf.tgt_slot[qi]; if (slot >= 0) id = slot_id(vf.tab, (unsigned)slot); }
    accumulate_point_wg<256>(vf.tab, id, sp, c6, fix_scales(cl.n, vf.counters), const_cast<int*>(vf.counters) + 1);
  }
}

}  // namespace
}  // namespace rolo
