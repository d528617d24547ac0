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

namespace rolo {
namespace {

constexpr int WALK_STACK = 48;

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

// Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch), each XCD with its own 4 MB L2. Neighbouring packets of the
// Hilbert-sorted cloud read the same leaves and boxes, so give every XCD a CONTIGUOUS eighth of the packets: what one wavefront
// pulled in from HBM (1-2 us per cold fetch — the walk's real bound on the ~48k-point feature clouds) the next ones find in L2.
// Small launches (<= 512 blocks: the ~48k-point feature clouds of the odometry pipeline, where every fetch is a cold miss and the walk is
// pure latency) give each XCD ONE contiguous eighth (pipeline frame latency 0.787 -> 0.731 ms, 1552 -> 1769 frames/s); big launches deal
// runs of 64 blocks round-robin instead, because the work per packet varies along the curve and whole eighths balance worse
// (2 x 131 072 points: 0.226 ms contiguous, 0.199 ms in runs, 0.207 ms unmapped). Both are bijections on [0, G).
ROLO_DEV int xcd_contiguous_block(int b, int G) {
#ifdef ROLO_KNN_NO_XCD_REMAP
  return b;
#else
  if (G <= 512) {
    const int x = b & 7, k = b >> 3, q = G >> 3, r = G & 7;   // XCD x owns G / 8 (+1 for x < G % 8) consecutive blocks
    return x * q + min(x, r) + k;
  }
  constexpr int RUN = 64, GROUP = 8 * RUN;
  if (b >= G / GROUP * GROUP) return b;                       // the whole groups are permuted, the remainder stays put
  const int grp = b / GROUP, o = b - grp * GROUP;             // o = k * 8 + x : the k-th block this group sends to XCD x
  return grp * GROUP + (o & 7) * RUN + (o >> 3);
#endif
}

// Instrumented build (-DROLO_KNN_STATS): one record per wavefront of the walk. Everything is counted in scalar registers and stored once at
// the very end; the start time comes from a NON-volatile asm that also produces the root node index, so it cannot move. (A store, an atomic, a
// clock builtin or a volatile asm before the loop is a potential memory clobber to the compiler: after it the wave-uniform box / leaf loads are
// no longer provably unclobbered and turn into vector loads — 64 more VGPRs, spills, a 5-10x slower walk. That is what a first version measured.)
#ifdef ROLO_KNN_STATS
__device__ unsigned g_knn_wave_rec[16384][8];   // nodes, leaves, insert executions, pushes, start, end (100 MHz wall clock), lanes live summed over the
                                                 // insert executions, per-leaf maximum over the lanes of the candidates accepted at leaf entry (summed)
#define KNN_STAT(x) x
#else
#define KNN_STAT(x)
#endif

ROLO_DEV double key_pack(float d2, int idx) {
  return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)idx));
}
ROLO_DEV float key_d2(double k) { return __uint_as_float((unsigned)((unsigned long long)__double_as_longlong(k) >> 32)); }
ROLO_DEV int key_idx(double k) { return (int)(unsigned)((unsigned long long)__double_as_longlong(k) & 0xffffffffull); }

// raw v_min_f64 / v_max_f64: fmin/fmax would add a canonicalising v_max_f64 x,x in front of every operand (sNaN
// quieting) — our operands are never NaN by construction. Pure VALU, no memory: safe as inline asm.
ROLO_DEV double vmin_f64(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
ROLO_DEV double vmax_f64(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// one slot of the sorted insert, K[s] = max(K[s-1], min(c, K[s])), as ONE asm statement: between two statements the compiler puts an s_nop for the
// read-after-write it cannot see into (20 per full insert); inside one statement the hardware interlock does the same job without the slot
#ifndef ROLO_KNN_SPLIT_MINMAX
ROLO_DEV void insert_slot(double& ks, double ksm1, double c) { asm("v_min_f64 %0, %1, %0\n\tv_max_f64 %0, %2, %0" : "+v"(ks) : "v"(c), "v"(ksm1)); }
#else
ROLO_DEV void insert_slot(double& ks, double ksm1, double c) { ks = vmax_f64(ksm1, vmin_f64(c, ks)); }
#endif
// score the KNN_LEAF (16) points of leaf g against this lane's query and insert the ones that beat its current k-th best
template <int KMAX, bool LOWER = false>
ROLO_DEV void knn_score_leaf(const float4* __restrict__ sorted, int g, const float4& q, double (&K)[KMAX], int kk, double& bkey, float& bd,
                             unsigned& n_ins, unsigned& lane_acc, unsigned& rounds, double lo = 0.0) {
  // fetch the whole leaf first: the address is wave-uniform, so these are KNN_LEAF scalar loads in flight behind ONE wait
  // (loading inside the loop serialised the scalar-cache round trips of a leaf behind the insert branch)
  float4 pts[KNN_LEAF];
#pragma unroll
  for (int u = 0; u < KNN_LEAF; u++) pts[u] = sorted[KNN_LEAF * (size_t)g + u];
  KNN_STAT(const double bkey0 = bkey; unsigned my_acc = 0;)   // accepted against the bound at leaf entry: what a per-lane queue would hold
#ifdef ROLO_KNN_STATS2
  unsigned my_cur = 0;
#endif
#pragma unroll
  for (int u = 0; u < KNN_LEAF; u++) {
    const float4 c = pts[u];
    const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
    const float cd = ((dx * dx) + (dy * dy)) + (dz * dz);  // file is compiled with -ffp-contract=off
    const double ck0 = key_pack(cd, __float_as_int(c.w));
    const double ck = (LOWER && !(ck0 > lo)) ? key_pack(INFINITY, INT_MAX) : ck0;   // LOWER: a point of an earlier round's 64 is no candidate
    KNN_STAT(if (__any(ck < bkey)) { n_ins++; lane_acc += (unsigned)__popcll(__ballot(ck < bkey)); })
    KNN_STAT(if (ck < bkey0) my_acc++;)
#ifdef ROLO_KNN_STATS2
    if (ck < bkey) my_cur++;   // accepted against the CURRENT bound: this lane's own inserts
#endif
    if (ck < bkey) {
      // sorted insert, descending slot order so every step reads not-yet-overwritten neighbours — in four tiers: a slot whose lower
      // neighbour is already <= the candidate in EVERY lane keeps its value (min(ck, K[s]) = K[s] and K[s-1] <= K[s]), so the lower tiers
      // run only if some lane's candidate sorts below them. Late in the walk candidates barely beat the k-th best: most executions stop
      // after the first tier (the insert is two thirds of the walk's VALU instructions; v_min_f64 / v_max_f64 issue at the fp32 rate, profiles/tools/valu_rate.hip).
      // tier boundaries (slots [B1, KMAX) always, then [B2, B1), [B3, B2), [0, B3)); tunable for KMAX = 20 (-DROLO_KNN_B1/B2/B3)
#ifndef ROLO_KNN_B1
#define ROLO_KNN_B1 15
#define ROLO_KNN_B2 10
#define ROLO_KNN_B3 5
#endif
      constexpr int B1 = KMAX == 20 ? ROLO_KNN_B1 : 3 * (KMAX / 4), B2 = KMAX == 20 ? ROLO_KNN_B2 : 2 * (KMAX / 4), B3 = KMAX == 20 ? ROLO_KNN_B3 : KMAX / 4;
#pragma unroll
      for (int s = KMAX - 1; s >= B1; s--) insert_slot(K[s], K[s - 1], ck);
      if (__any(ck < K[B1 - 1])) {
#pragma unroll
        for (int s = B1 - 1; s >= B2; s--) insert_slot(K[s], K[s - 1], ck);
        if (__any(ck < K[B2 - 1])) {
#pragma unroll
          for (int s = B2 - 1; s >= B3; s--) insert_slot(K[s], K[s - 1], ck);
          if (__any(ck < K[B3 - 1])) {
#pragma unroll
            for (int s = B3 - 1; s >= 1; s--) insert_slot(K[s], K[s - 1], ck);
            K[0] = vmin_f64(ck, K[0]);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < KMAX; s++) if (s == kk - 1) bkey = K[s];
      bd = key_d2(bkey);
    }
  }
#ifdef ROLO_KNN_STATS
#ifdef ROLO_KNN_STATS2
  rounds += my_cur;   // per LANE (record [7] becomes the wave maximum of a lane's inserts over the whole walk: what a cross-leaf queue would execute at least)
  (void)my_acc;
#else
  { unsigned m = my_acc;   // wave maximum: the drain iterations of a per-lane queue for this leaf
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    rounds += m; }
#endif
#endif
}

// The same leaf scored through a per-lane LDS queue (-DROLO_KNN_LANE_QUEUE, an A/B): every lane first appends the candidates that beat its
// bound AT LEAF ENTRY to its own queue (16 predicated ds_write_b64), then the wavefront drains the queues together — one sorted insert per
// iteration with ALL lanes live, max-over-lanes iterations instead of one partial-exec insert per point ANY lane accepts (the union over
// lanes). A queued key that no longer beats the (tighter) bound when it is popped is a no-op in the min / max network, so the neighbour
// lists are unchanged.
template <int KMAX>
ROLO_DEV void knn_score_leaf_queue(const float4* __restrict__ sorted, int g, const float4& q, double (&K)[KMAX], int kk, double& bkey, float& bd, double sentinel) {
  // declared HERE, not passed in: through a (generic) pointer parameter the compiler no longer knows the stores go to LDS, every later
  // wave-uniform leaf load becomes a potential clobber victim and turns into a vector load (128 VGPRs + 209 spilled: the clobber rule of DESIGN.md section 4)
  __shared__ double qbuf_all[4][KNN_LEAF * 64];
  double* qbuf = qbuf_all[threadIdx.x >> 6];
  float4 pts[KNN_LEAF];
#pragma unroll
  for (int u = 0; u < KNN_LEAF; u++) pts[u] = sorted[KNN_LEAF * (size_t)g + u];
  const int lane = threadIdx.x & 63;
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < KNN_LEAF; u++) {
    const float4 c = pts[u];
    const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
    float cd = ((dx * dx) + (dy * dy)) + (dz * dz);
    asm("" : "+v"(cd), "+v"(cnt));   // one candidate at a time: without this ordering the 16 keys are formed up front (32 more VGPRs, 209 spills)
    const double ck = key_pack(cd, __float_as_int(c.w));
    if (ck < bkey) { qbuf[cnt * 64 + lane] = ck; cnt++; }
  }
  while (__any(cnt > 0)) {
    double ck = sentinel;
    if (cnt > 0) { cnt--; ck = qbuf[cnt * 64 + lane]; }
    constexpr int T = KMAX / 4;
#pragma unroll
    for (int s = KMAX - 1; s >= 3 * T; s--) K[s] = vmax_f64(K[s - 1], vmin_f64(ck, K[s]));
    if (__any(ck < K[3 * T - 1])) {
#pragma unroll
      for (int s = 3 * T - 1; s >= 2 * T; s--) K[s] = vmax_f64(K[s - 1], vmin_f64(ck, K[s]));
      if (__any(ck < K[2 * T - 1])) {
#pragma unroll
        for (int s = 2 * T - 1; s >= T; s--) K[s] = vmax_f64(K[s - 1], vmin_f64(ck, K[s]));
        if (__any(ck < K[T - 1])) {
#pragma unroll
          for (int s = T - 1; s >= 1; s--) K[s] = vmax_f64(K[s - 1], vmin_f64(ck, K[s]));
          K[0] = vmin_f64(ck, K[0]);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < KMAX; s++) if (s == kk - 1) bkey = K[s];
  bd = key_d2(bkey);
}

// The walk keeps only the 20 packed keys and the query live (58 VGPRs): cut for 8 wavefronts per SIMD. Neighbour indices
// go to A.c[].nbr, slot-major so every store is coalesced; knn_tail_kernel turns them into covariances.
#ifndef ROLO_KNN_WALK_OCC
#define ROLO_KNN_WALK_OCC 4   // 128 VGPRs allowed: the loop needs 61, the slack buys the compiler ~3 % (0.196 -> 0.188 ms); a 2 x 131 072-point pair fills 4 waves per SIMD
#endif
template <int KMAX, bool FUSE_TAIL, bool LOWER = false>
__global__ __launch_bounds__(256, KMAX > 32 ? 2 : ROLO_KNN_WALK_OCC) void knn_walk_kernel(KnnPair A, int split, int k, int reg) {   // 64 slots = 128 key registers: 256 VGPRs, 2 waves per SIMD
  __shared__ int stk[4][WALK_STACK];
  const int tid = threadIdx.x;
  const int wv = tid >> 6;
  // which cloud of the pair this workgroup searches (wave-uniform: everything below stays in scalar registers)
  const int blk = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
  const int which = blk >= split ? 1 : 0;
  const float4* __restrict__ sorted = A.c[which].sorted;
  const float4* __restrict__ boxes = A.c[which].boxes;
  int32_t* knn_idx = A.c[which].knn_idx;
  float* knn_d2 = A.c[which].knn_d2;
  const int n_sorted = A.c[which].n_sorted, P = A.c[which].P;
#ifndef ROLO_KNN_PACKET
#define ROLO_KNN_PACKET 64   // queries per wavefront (experiment: 32 / 16 leave the upper lanes idle — shorter dependent chain per wave, more waves)
#endif
  const int lane_ = tid & 63;
  const int j = A.c[which].q_begin + (blk - (which ? split : 0)) * (4 * ROLO_KNN_PACKET) + wv * ROLO_KNN_PACKET + lane_;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  int qi = INT_MAX;
  if (lane_ < ROLO_KNN_PACKET && j < A.c[which].q_end) { q = sorted[j]; qi = __float_as_int(q.w); }
  const bool active = qi != INT_MAX;  // not padding
  const int kk = (KMAX == 20) ? 20 : k;
  const int n_leaves = n_sorted / KNN_LEAF;
  unsigned st_nodes = 0, st_leaves = 0, st_ins = 0, st_lane = 0, st_rounds = 0;

  // K[0..KMAX) ascending; sentinel = (inf, INT_MAX)
  const double sentinel = key_pack(INFINITY, INT_MAX);
  double K[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) K[u] = sentinel;
  // pruning radius of this lane: d2 of its kk-th best so far; inactive lanes never reach anything
  float bd = active ? INFINITY : -1.0f;
  double bkey = active ? sentinel : key_pack(0.f, 0);

  // ---- seed: the wavefront's own 64 / KNN_LEAF leaves ----
#ifndef ROLO_KNN_SEED_EXTRA
#define ROLO_KNN_SEED_EXTRA 1   // leaves on either side of the wavefront's own ones (along the curve) scored before the tree walk starts: curve neighbours are space neighbours,
                                // so every lane enters the walk with a tighter bound (0 / 1 / 2 / 4: walk 0.188 / 0.177 / 0.178 / 0.187 ms at 2 x 131 072 points, 0.147 / 0.136 / 0.138 / 0.138 at 2 x 65 536)
#endif
  const int g_mine0 = __builtin_amdgcn_readfirstlane(j / KNN_LEAF);  // lane 0 of the wave: j is a multiple of 64
  const int g_own0 = max(g_mine0 - ROLO_KNN_SEED_EXTRA, 0);
  const int g_own1 = min(g_mine0 + ROLO_KNN_PACKET / KNN_LEAF + ROLO_KNN_SEED_EXTRA, n_leaves);
#ifdef ROLO_KNN_LANE_QUEUE
#define KNN_SCORE(g) knn_score_leaf_queue<KMAX>(sorted, g, q, K, kk, bkey, bd, sentinel)
#else
#define KNN_SCORE(g) knn_score_leaf<KMAX, LOWER>(sorted, g, q, K, kk, bkey, bd, st_ins, st_lane, st_rounds, lo)
#endif
  // rounds of a search for more than 64 neighbours (LOWER): only keys above the previous round's last one count
  double lo = 0.0;
  if (LOWER && active) lo = A.c[which].lower[j];
  (void)lo;
  {  // the wavefront's own leaves first, then the extra ones: the own points are the nearer ones, so fewer keys are inserted only to be pushed out again
     // (walk 0.1763 -> 0.1725 ms at 2 x 131 072 points, 0.1486 -> 0.1470 at 2 x 43 776)
    const int n_own = min(g_mine0 + ROLO_KNN_PACKET / KNN_LEAF, n_leaves) - g_mine0, n_before = g_mine0 - g_own0;
    for (int i = 0; i < g_own1 - g_own0; i++) {
      const int g = i < n_own ? g_mine0 + i : (i - n_own < n_before ? g_own0 + (i - n_own) : g_mine0 + (i - n_before));
      KNN_SCORE(g); st_leaves++;
    }
  }

  // ---- packet walk ----
  // (a stack in one vector register — slot i in lane i, v_writelane / v_readlane — measured the same as this LDS stack: 0.202 vs 0.200 ms;
  // gfx950's v_writelane takes its lane select from M0 when the value is an SGPR, which inline asm may not clobber safely)
  int sp = 0;
  int h = 1;
#ifdef ROLO_KNN_STATS
  unsigned long long wt0;
  asm("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %1, 1" : "=s"(wt0), "=s"(h));
  unsigned st_push = 0;
#endif
#if defined(ROLO_KNN_PRIO)
  int it = 0;   // s_setprio below is a plain asm tied to this counter: the builtin (and any volatile asm) counts as a memory clobber, after which
                // the wave-uniform leaf loads are no longer provably unclobbered and become vector loads (64 more VGPRs, spills)
#endif
  while (true) {
    h = __builtin_amdgcn_readfirstlane(h);
#if defined(ROLO_KNN_PRIO) && ROLO_KNN_PRIO == 1   // rotate the issue priority among the waves of a SIMD
    it++;
    if ((it & 7) == 0) { switch (((it >> 3) + (int)blockIdx.x) & 3) { case 0: asm("s_setprio 0" : "+s"(it)); break; case 1: asm("s_setprio 1" : "+s"(it)); break; case 2: asm("s_setprio 2" : "+s"(it)); break; default: asm("s_setprio 3" : "+s"(it)); } }
#elif defined(ROLO_KNN_PRIO) && ROLO_KNN_PRIO == 2   // the further a wave has come, the higher its priority
    it++;
    if (it == 48) asm("s_setprio 1" : "+s"(it)); else if (it == 96) asm("s_setprio 2" : "+s"(it)); else if (it == 144) asm("s_setprio 3" : "+s"(it));
#elif defined(ROLO_KNN_PRIO) && ROLO_KNN_PRIO == 3   // the less a wave has done, the higher its priority (fair share)
    it++;
    if (it == 1) asm("s_setprio 3" : "+s"(it)); else if (it == 32) asm("s_setprio 2" : "+s"(it)); else if (it == 64) asm("s_setprio 1" : "+s"(it)); else if (it == 96) asm("s_setprio 0" : "+s"(it));
#endif
#ifdef ROLO_KNN_WIDE
    // Two levels per step where the tree allows it: the four grandchildren of h are nodes 4h .. 4h + 3, their boxes 128 contiguous bytes —
    // ONE dependent fetch instead of two or three, the same box tests or fewer (the children's own boxes are skipped: their union is h's,
    // which is known to be reached). The grandchild most lanes are nearest to goes first, the other live ones are pushed, best on top.
    if (2 * h < P) {
      st_nodes++;
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; u++) b[u] = boxes[8 * (size_t)h + u];
      float d[4]; bool ok[4];
#pragma unroll
      for (int c = 0; c < 4; c++) { d[c] = box_d2(b[2 * c], b[2 * c + 1], q); ok[c] = (d[c] <= bd) && (d[c] < INFINITY); }
      // this lane's nearest live grandchild (ties: the lowest index)
      int best = -1; float bestd = INFINITY;
#pragma unroll
      for (int c = 0; c < 4; c++) if (ok[c] && d[c] < bestd) { bestd = d[c]; best = c; }
      int key[4];   // wave-uniform: votes * 4 + (3 - c) for a live grandchild, -1 for one no lane reaches
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const unsigned long long m = __ballot(ok[c]);
        const int votes = __popcll(__ballot(best == c));
        key[c] = m != 0ull ? votes * 4 + (3 - c) : -1;
      }
      // sort the four keys descending (5 compare-exchanges on scalars)
      auto cx = [](int& a, int& bb) { const int hi = max(a, bb), lo = min(a, bb); a = hi; bb = lo; };
      cx(key[0], key[1]); cx(key[2], key[3]); cx(key[0], key[2]); cx(key[1], key[3]); cx(key[1], key[2]);
      if (key[0] >= 0) {
        // push the runners-up, worst first, so that the second best is popped first
#pragma unroll
        for (int r = 3; r >= 1; r--) if (key[r] >= 0 && sp < WALK_STACK) { stk[wv][sp] = 4 * h + (3 - (key[r] & 3)); sp++; KNN_STAT(st_push++;) }
        h = 4 * h + (3 - (key[0] & 3));
        continue;
      }
    } else
#endif
    if (h < P) {
      st_nodes++;
      const float4 llo = boxes[4 * (size_t)h], lhi = boxes[4 * (size_t)h + 1], rlo = boxes[4 * (size_t)h + 2], rhi = boxes[4 * (size_t)h + 3];
      const float bl = box_d2(llo, lhi, q), br = box_d2(rlo, rhi, q);
      const bool okl = (bl <= bd) && (bl < INFINITY), okr = (br <= bd) && (br < INFINITY);
      const unsigned long long ml = __ballot(okl), mr = __ballot(okr);
      if (ml != 0ull && mr != 0ull) {
        // nearer child first, by majority vote of the lanes that reach either child (a "lane with the largest
        // radius decides" rule needed a 6-step cross-lane max per node and did not reduce the nodes visited)
        const unsigned long long pref = __ballot((okl || okr) && (bl <= br));
        const bool left_first = 2 * __popcll(pref) >= __popcll(ml | mr);
        if (sp < WALK_STACK) { stk[wv][sp] = left_first ? 2 * h + 1 : 2 * h; sp++; KNN_STAT(st_push++;) }
        h = left_first ? 2 * h : 2 * h + 1;
        continue;
      }
      if (ml != 0ull) { h = 2 * h; continue; }
      if (mr != 0ull) { h = 2 * h + 1; continue; }
    } else {
      const int g = h - P;
      if (g < g_own0 || g >= g_own1) { KNN_SCORE(g); st_leaves++; }
    }
    if (sp == 0) break;
    sp--;
    h = stk[wv][sp];
  }
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
  (void)st_nodes; (void)st_leaves; (void)st_ins; (void)st_lane; (void)st_rounds;

  if (!active) return;

  int ki[KMAX];
#pragma unroll
  for (int u = 0; u < KMAX; u++) ki[u] = key_idx(K[u]);
  const int slot0 = (KMAX == 64) ? A.c[which].slot0 : 0;                          // rounds exist for the 64-slot kernel only
  const int ktot = (KMAX == 64 && A.c[which].k_total) ? A.c[which].k_total : kk;
  if (knn_idx) {
#pragma unroll
    for (int u = 0; u < KMAX; u++) if (u < kk) { knn_idx[(size_t)qi * ktot + slot0 + u] = ki[u]; knn_d2[(size_t)qi * ktot + slot0 + u] = key_d2(K[u]); }
  }
  if (KMAX == 64 && A.c[which].lower) A.c[which].lower[j] = bkey;   // the next round starts above this round's last key
  if (!FUSE_TAIL) {   // neighbour indices only (slot-major, coalesced): knn_tail_kernel turns them into covariances
    int32_t* __restrict__ nbr = A.c[which].nbr;
#pragma unroll
    for (int u = 0; u < KMAX; u++) nbr[(size_t)(slot0 + u) * n_sorted + j] = ki[u];
    return;
  }
  // FUSE_TAIL (ROLO_KNN_FUSE_TAIL=1, an A/B): covariance + regularisation right here, so that the light wavefronts do theirs while the heavy
  // ones are still walking and the 42 MB of index traffic disappear. Measured: 0.2335 ms against 0.1884 + 0.0394 ms for walk + tail launch,
  // and the 4-context throughput falls from 2750 to 2490 scans/s — the tails' fp64 work competes with the walking wavefronts for issue slots.
  const KnnCloud& cl = A.c[which];
  if (cl.stage) {   // multi-GPU: into the exchange buffer, sorted order
    double* o = stage_area(cl, 1) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
    double c6[6]; knn_covariance_tail<KMAX>(ki, kk, cl.xyz, 1, 0, reg, o, c6);
  } else {
    double c6[6]; knn_covariance_tail<KMAX>(ki, kk, cl.xyz, cl.n, qi, reg, cl.cov, c6);
  }
}

#ifndef ROLO_KNN_TAIL_OCC
#define ROLO_KNN_TAIL_OCC 2
#endif
template <int KMAX>
__global__ __launch_bounds__(256, ROLO_KNN_TAIL_OCC) void knn_tail_kernel(KnnPair A, int split, int k, int reg, VoxelFuse vf) {
  const int blk = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
  const int which = blk >= split ? 1 : 0;
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
  double c6[6] = {0, 0, 0, 0, 0, 0};
  if (act) {
    const int32_t* __restrict__ nbr = cl.nbr;
    int ki[KMAX];
#pragma unroll
    for (int u = 0; u < KMAX; u++) ki[u] = nbr[(size_t)u * n_sorted + j];
    if (cl.stage) {   // multi-GPU: into the exchange buffer, sorted order
      double* o = stage_area(cl, 1) + (size_t)(j / cl.chunk) * cl.seg + cl.stage_off + (size_t)(j % cl.chunk) * 6;
      knn_covariance_tail<KMAX>(ki, (KMAX == 20) ? 20 : k, cl.xyz, 1, 0, reg, o, c6);   // pitch 1, index 0: six consecutive doubles
    } else {
      knn_covariance_tail<KMAX>(ki, (KMAX == 20) ? 20 : k, cl.xyz, cl.n, qi, reg, cl.cov, c6);
    }
  }
  if (fuse) {   // every lane of the wavefront takes part in the segmented fold
    int id = -1;
    if (act) { const int slot = vf.tgt_slot[qi]; if (slot >= 0) id = vf.tab.ids[slot]; }
    accumulate_point(vf.tab, id, sp, c6, fix_scales(cl.n, vf.counters), true, const_cast<int*>(vf.counters) + 1);
  }
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
      knn_covariance_tail_loop(cl.nbr, (size_t)cl.n_sorted, j, k, cl.xyz, cl.n, qi, reg, cl.cov, c6);
    }
  }
  if (fuse) {
    int id = -1;
    if (act) { const int slot = vf.tgt_slot[qi]; if (slot >= 0) id = vf.tab.ids[slot]; }
    accumulate_point(vf.tab, id, sp, c6, fix_scales(cl.n, vf.counters), true, const_cast<int*>(vf.counters) + 1);
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
    if (act) { const int slot = vf.tgt_slot[qi]; if (slot >= 0) id = vf.tab.ids[slot]; }
    accumulate_point(vf.tab, id, sp, c6, fix_scales(cl.n, vf.counters), true, const_cast<int*>(vf.counters) + 1);
  }
}

}  // namespace
}  // namespace rolo
