// The packet walk of the exact k-nearest-neighbour search, as device functions: packed (d2, index) keys and their min / max sorted insert, wave-uniform
// fetches as explicit scalar loads, leaf scoring, the walk itself and the hooks of the cooperative kernel. Included by knn_walk.hpp (K5: the queries are the
// cloud's own points) and by scan2map.hip (8f.4: foreign queries — the scan's features against the sub-map's tree). Design notes: knn_walk.hpp.
#pragma once
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include <cfloat>
#include <climits>

#ifndef KNN_STAT
#ifdef ROLO_KNN_STATS   // (the instrumented build: every includer sees the same definition — scan2map.hip includes this file without knn_walk.hpp)
#define KNN_STAT(x) x
#else
#define KNN_STAT(x)
#endif
#endif

namespace rolo {
namespace {

constexpr int WALK_STACK = KNN_WALK_STACK;

// squared distance from q to the box [lo, hi] in the operation order of the point distances (a true lower bound of every point distance inside)
ROLO_DEV float box_d2(const float4& lo, const float4& hi, const float4& q) {
  float dx = fmaxf(fmaxf(__fsub_rn(lo.x, q.x), __fsub_rn(q.x, hi.x)), 0.f);
  float dy = fmaxf(fmaxf(__fsub_rn(lo.y, q.y), __fsub_rn(q.y, hi.y)), 0.f);
  float dz = fmaxf(fmaxf(__fsub_rn(lo.z, q.z), __fsub_rn(q.z, hi.z)), 0.f);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

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
// ("+&v": the first instruction writes %0 before the second reads %2 — without the early-clobber mark the compiler may give K[s-1] the register of K[s]
// whenever it knows the two hold the same value, e.g. two sentinels of a fresh list: seen in knn_walk_sub_kernel as lists full of repeated keys)
ROLO_DEV void insert_slot(double& ks, double ksm1, double c) { asm("v_min_f64 %0, %1, %0\n\tv_max_f64 %0, %2, %0" : "+&v"(ks) : "v"(c), "v"(ksm1)); }
#else
ROLO_DEV void insert_slot(double& ks, double ksm1, double c) { ks = vmax_f64(ksm1, vmin_f64(c, ks)); }
#endif
// sorted insert of ck into the ascending K[0 .. KMAX), in four tiers (see knn_score_leaf): the lower tiers run only if some lane's candidate sorts below them
#ifndef ROLO_KNN_B1
#define ROLO_KNN_B1 15
#define ROLO_KNN_B2 10
#define ROLO_KNN_B3 5
#endif
template <int KMAX>
ROLO_DEV void insert_tiered(double (&K)[KMAX], double ck) {
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
}

// ---- wave-uniform fetches as EXPLICIT scalar loads ----------------------------------------------------------------------------------------
// Node boxes and leaf points are fetched through wave-uniform addresses: one s_load per 64 bytes per WAVE instead of a vector load per lane. Rounds
// 1-3 left that to the compiler, which emits scalar loads only while it can prove that nothing in the kernel may have written memory before them
// — a store, an atomic, a fence, a clock builtin or a volatile asm ahead of the loop (or on any path that reaches it again) turned the leaf's 64
// floats into vector loads: 64 more VGPRs, spills, a 5-10 x slower walk ("the clobber rule"). That rule forbade every form of
// work sharing between wavefronts. The loads are inline asm now (s_load_dwordx16 + the wait, outputs in SGPR tuples): scalar by construction,
// whatever else the kernel does. The data they read (sorted points, boxes) is written by EARLIER launches only, so a non-volatile asm is exact.
typedef float sgpr16 __attribute__((ext_vector_type(16)));
#ifndef ROLO_KNN_ASM_LOADS
#define ROLO_KNN_ASM_LOADS 0   // the PLAIN walk kernels (no store, atomic or fence ahead of their loops: the compiler's own scalar loads, as measured in rounds 1-3; 1 = the asm loads there too, an A/B: +2-3 %);
#endif                         // the cooperative kernel, which fences and stores between walks, always takes the asm loads
constexpr bool KNN_PLAIN_ASM = ROLO_KNN_ASM_LOADS != 0;
// (the "s" constraint does not make a pointer uniform by itself: one the compiler believes divergent — e.g. picked through a value read from LDS — would be
// substituted as a VGPR pair, which the instruction does not take)
ROLO_DEV const float4* uniform_ptr(const float4* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (const float4*)(((unsigned long long)hi << 32) | lo);
}
template <bool ASM>
ROLO_DEV void sload_leaf(const float4* __restrict__ p_, float4 (&pts)[KNN_LEAF]) {
  if (!ASM) {
#pragma unroll
    for (int u = 0; u < KNN_LEAF; u++) pts[u] = p_[u];
    return;
  }
  const float4* p = uniform_ptr(p_);
#if 1
  static_assert(KNN_LEAF == 16 || KNN_LEAF == 8, "leaf size");
  sgpr16 a, b;
  if (KNN_LEAF == 16) {
    sgpr16 c, d;
    asm("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx16 %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(p));
#pragma unroll
    for (int u = 0; u < 4; u++) { pts[8 + u] = make_float4(c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3]); pts[12 + u] = make_float4(d[4 * u], d[4 * u + 1], d[4 * u + 2], d[4 * u + 3]); }
  } else {
    asm("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p));
  }
#pragma unroll
  for (int u = 0; u < 4; u++) { pts[u] = make_float4(a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]); pts[4 + u] = make_float4(b[4 * u], b[4 * u + 1], b[4 * u + 2], b[4 * u + 3]); }
#else
#pragma unroll
  for (int u = 0; u < KNN_LEAF; u++) pts[u] = p[u];
#endif
}
// the two child boxes of node h: boxes[4h .. 4h + 3] = left lo, left hi, right lo, right hi (64 bytes)
template <bool ASM>
ROLO_DEV void sload_node(const float4* __restrict__ p_, float4& llo, float4& lhi, float4& rlo, float4& rhi) {
  if (!ASM) { llo = p_[0]; lhi = p_[1]; rlo = p_[2]; rhi = p_[3]; return; }
  const float4* p = uniform_ptr(p_);
#if 1
  sgpr16 a;
  asm("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(p));
  llo = make_float4(a[0], a[1], a[2], a[3]); lhi = make_float4(a[4], a[5], a[6], a[7]); rlo = make_float4(a[8], a[9], a[10], a[11]); rhi = make_float4(a[12], a[13], a[14], a[15]);
#else
  llo = p[0]; lhi = p[1]; rlo = p[2]; rhi = p[3];
#endif
}

// score the KNN_LEAF (16) points of leaf g against this lane's query and insert the ones that beat its current k-th best
// CAP (continuation of a budgeted walk, below): the lane's list starts EMPTY but its bound does not — bkey never rises above bcap, the k-th best
// of the list the first part of the walk left behind
template <int KMAX, bool LOWER = false, bool CAP = false, bool ASM = false>
ROLO_DEV void knn_score_leaf(const float4* __restrict__ sorted, int g, const float4& q, double (&K)[KMAX], int kk, double& bkey, float& bd,
                             unsigned& n_ins, unsigned& lane_acc, unsigned& rounds, double lo = 0.0, double bcap = 0.0) {
  // fetch the whole leaf first: the address is wave-uniform, so these are KNN_LEAF scalar loads in flight behind ONE wait
  // (loading inside the loop serialised the scalar-cache round trips of a leaf behind the insert branch)
  float4 pts[KNN_LEAF];
  sload_leaf<ASM>(sorted + KNN_LEAF * (size_t)g, pts);
  KNN_STAT(const double bkey0 = bkey; unsigned my_acc = 0;)   // accepted against the bound at leaf entry: what a per-lane queue would hold
#ifdef ROLO_KNN_STATS2
  unsigned my_cur = 0;
#endif
#pragma unroll
  for (int u = 0; u < KNN_LEAF; u++) {
    const float4 c = pts[u];
    const float dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
    const float cd = ((dx * dx) + (dy * dy)) + (dz * dz);  // file is compiled with -ffp-contract=off (the compiler already pairs x and y into v_pk_add_f32 / v_pk_mul_f32)
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
      insert_tiered<KMAX>(K, ck);
#pragma unroll
      for (int s = 0; s < KMAX; s++) if (s == kk - 1) bkey = K[s];
      if (CAP) bkey = vmin_f64(bkey, bcap);
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

// ---- SUB lanes per query (round 4): exchanges among 2 / 4 / 8 adjacent lanes by DPP -------------------------------------------------------------------
template <int CTRL>
ROLO_DEV double dpp_f64(double x) {
  const long long v = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)(unsigned long long)v, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)((unsigned long long)v >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// exchange step `st` (0, 1, 2) of a reduction over SUB = 2, 4, 8 adjacent lanes: lane ^ 1, lane ^ 2 (quad_perm), then the mirror of the 8-lane half row
// (which maps each quad onto the other one)
template <int ST> ROLO_DEV double sub_xchg(double x) { return ST == 0 ? dpp_f64<0xB1>(x) : ST == 1 ? dpp_f64<0x4E>(x) : dpp_f64<0x141>(x); }
template <int SUB> ROLO_DEV double sub_min(double x) {
  x = vmin_f64(x, sub_xchg<0>(x));
  if (SUB >= 4) x = vmin_f64(x, sub_xchg<1>(x));
  if (SUB >= 8) x = vmin_f64(x, sub_xchg<2>(x));
  return x;
}
template <int SUB> ROLO_DEV double sub_max(double x) {
  x = vmax_f64(x, sub_xchg<0>(x));
  if (SUB >= 4) x = vmax_f64(x, sub_xchg<1>(x));
  if (SUB >= 8) x = vmax_f64(x, sub_xchg<2>(x));
  return x;
}

// Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch), each XCD with its own 4 MB L2. Neighbouring packets of the
// Hilbert-sorted cloud read the same leaves and boxes, so give every XCD a CONTIGUOUS eighth of the packets: what one wavefront
// pulled in from HBM (1-2 us per cold fetch — the walk's real bound on the ~48k-point feature clouds) the next ones find in L2.
// Small launches (<= 512 blocks: the ~48k-point feature clouds of the odometry pipeline, where every fetch is a cold miss and the walk is
// pure latency) give each XCD ONE contiguous eighth (pipeline frame latency 0.787 -> 0.731 ms, 1552 -> 1769 frames/s); big launches deal
// runs of 64 blocks round-robin instead, because the work per packet varies along the curve and whole eighths balance worse
// (2 x 131 072 points: 0.226 ms contiguous, 0.199 ms in runs, 0.207 ms unmapped). Both are bijections on [0, G).
// wpb = wavefronts (packets) per block: the thresholds are in packets, whatever the workgroup size
ROLO_DEV int xcd_contiguous_block(int b, int G, int wpb = 4) {
#ifdef ROLO_KNN_NO_XCD_REMAP
  return b;
#else
  if (G * wpb <= 2048) {
    const int x = b & 7, k = b >> 3, q = G >> 3, r = G & 7;   // XCD x owns G / 8 (+1 for x < G % 8) consecutive blocks
    return x * q + min(x, r) + k;
  }
  const int RUN = 256 / wpb, GROUP = 8 * RUN;                  // runs of 256 packets
  if (b >= G / GROUP * GROUP) return b;                       // the whole groups are permuted, the remainder stays put
  const int grp = b / GROUP, o = b - grp * GROUP;             // o = k * 8 + x : the k-th block this group sends to XCD x
  return grp * GROUP + (o & 7) * RUN + (o >> 3);
#endif
}

// ---- the packet walk proper --------------------------------------------------------------------------------------------------------------
// One wavefront, 64 queries, from node h down; the far children wait on a small per-wave stack in LDS. Every control decision is wave-uniform.
// PUBLISH (knn_walk_coop_kernel): once the walk has scored `budget` leaves, whenever everything it published before has been taken it moves the
// BOTTOM entries of its stack — the oldest, i.e. the largest sub-trees — into a small ring in LDS where the idle wavefronts of its workgroup
// steal them (work stealing: the owner works at the top of its stack, thieves take from the bottom).
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) double lds_double;
constexpr int COOP_RING = 16;
struct CoopPub { lds_int* ring; lds_int* head; lds_int* tail; lds_double* cap; int budget; };   // head: entries published so far, tail: entries taken so far

template <int KMAX, bool LOWER, bool CAP, bool PUBLISH, bool ASM>
ROLO_DEV void packet_walk(const float4* __restrict__ sorted, const float4* __restrict__ boxes, int P, int g_own0, int g_own1, const float4& q, double (&K)[KMAX], int kk,
                          double& bkey, float& bd, double lo, double bcap, lds_int* stk, int& sp, int h, const CoopPub& pub, int& n_scored, int& n_published,
                          unsigned& st_nodes, unsigned& st_leaves, unsigned& st_ins, unsigned& st_lane, unsigned& st_rounds, unsigned& st_push) {
  (void)st_push; (void)pub; (void)n_scored; (void)n_published;
  while (true) {
    h = __builtin_amdgcn_readfirstlane(h);
    if (h < P) {
      st_nodes++;
      float4 llo, lhi, rlo, rhi;
      sload_node<ASM>(boxes + 4 * (size_t)h, llo, lhi, rlo, rhi);
      const float bl = box_d2(llo, lhi, q), br = box_d2(rlo, rhi, q);
      const bool okl = (bl <= bd) && (bl < INFINITY), okr = (br <= bd) && (br < INFINITY);
      const unsigned long long ml = __ballot(okl), mr = __ballot(okr);
      if (ml != 0ull && mr != 0ull) {
        // nearer child first, by majority vote of the lanes that reach either child (a "lane with the largest
        // radius decides" rule needed a 6-step cross-lane max per node and did not reduce the nodes visited)
        const unsigned long long pref = __ballot((okl || okr) && (bl <= br));
        const bool left_first = 2 * __popcll(pref) >= __popcll(ml | mr);
        if (sp < WALK_STACK) { stk[sp] = left_first ? 2 * h + 1 : 2 * h; sp++; KNN_STAT(st_push++;) }
        h = left_first ? 2 * h : 2 * h + 1;
        continue;
      }
      if (ml != 0ull) { h = 2 * h; continue; }
      if (mr != 0ull) { h = 2 * h + 1; continue; }
    } else {
      const int g = h - P;
      if (g < g_own0 || g >= g_own1) {   // (the wavefront's own leaves and their neighbours along the curve were scored as seeds)
        knn_score_leaf<KMAX, LOWER, CAP, ASM>(sorted, g, q, K, kk, bkey, bd, st_ins, st_lane, st_rounds, lo, bcap);
        st_leaves++;
        if (PUBLISH) n_scored++;
      }
    }
    if (sp == 0) return;
    if (PUBLISH && n_scored >= pub.budget && sp >= 2) {
      const int lane = threadIdx.x & 63;
      const int hd = __builtin_amdgcn_readfirstlane(*pub.head);
      const int tl = __builtin_amdgcn_readfirstlane(__hip_atomic_load((int*)pub.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (hd == tl) {   // the ring is empty: hand out the bottom half of the stack (at most 4 entries), and the bound the thieves may prune with
        const int m = min(sp >> 1, 4);
        if (lane < m) pub.ring[(hd + lane) % COOP_RING] = stk[lane];
        pub.cap[lane] = bkey;
        const int keep = sp - m;
        const int e = lane < keep ? stk[m + lane] : 0;
        if (lane < keep) stk[lane] = e;
        sp = keep;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store((int*)pub.head, hd + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        n_published += m;
      }
    }
    sp--;
    h = stk[sp];
  }
}

// a thief's (or the owner's own) take from the bottom ring of wavefront d: the oldest published sub-tree, or -1. Lane 0 acts for the wavefront.
// The entry is read BEFORE the compare-and-swap on `tail`: slot t is only rewritten by its owner once tail has moved past t, so a successful swap proves the read was of entry t.
ROLO_DEV int coop_steal(lds_int* ring, lds_int* head, lds_int* tail) {
  int e = -1;
  if ((threadIdx.x & 63) == 0) {
    while (true) {
      const int t = __hip_atomic_load((int*)tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int hd = __hip_atomic_load((int*)head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (t >= hd) break;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int v = ring[t % COOP_RING];
      int expect = t;
      if (__hip_atomic_compare_exchange_strong((int*)tail, &expect, t + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { e = v; break; }
    }
  }
  return __builtin_amdgcn_readfirstlane(e);
}

}  // namespace
}  // namespace rolo
