// Device helpers shared by the voxel-map build and the pass kernels: voxel coordinates, packed keys, probing.
#pragma once
#include "rolo_internal.hpp"
#include "dev_math.hpp"
#include "polar_exact.hpp"

namespace rolo {

// vmp_voxel.hpp:199-201 (UNIFORM) and :208-211 (POLAR), fp64, true divisions as written in the reference. The integer key must be the reference's, bit for bit: device atan2 / acos
// differ from a host libm by ulps, which can move a key only for a point whose quotient lies within ~1e-14 of an integer. Such points (within 1e-12 bins of
// an edge: near_edge, counted in rolo_num_edge_points) are RE-KEYED with the correctly rounded atan2 / acos of polar_exact.hpp — the value a correctly
// rounded libm returns by definition — followed by the reference's own fp64 sum, quotient and floor ("flag ... and resolve those (few)";
// round 3 only counted them). exact = false keeps the fast key (ROLO_POLAR_EXACT=0: the A/B that shows which planted points would differ).
// UNIFORM keys are a division and a floor — correctly rounded on both sides, no hazard, nothing counted.
ROLO_DEV void voxel_coord_dev_edge(const VoxelTable& tab, double x, double y, double z, int& kx, int& ky, int& kz, bool& near_edge) {
  near_edge = false;
  const bool exact = tab.polar_exact != 0;
  if (tab.voxel_type == ROLO_VOXEL_POLAR) {
    const double r = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y)), __dmul_rn(z, z)));   // x.head<3>().norm(): no FMA contraction (the pass kernels hand in transformed, non-float coordinates)
    const double a = (atan2(y, x) + 3.14159265358979323846) / tab.polar_res[0], b = acos(z / r) / tab.polar_res[1], c = r / tab.polar_res[2];
    const double fa = floor(a), fb = floor(b), fc = floor(c);
    kx = (int)fa; ky = (int)fb; kz = (int)fc;
    const double lo = 1e-12, hi = 1.0 - 1e-12;
    const bool ea = (a - fa) < lo || (a - fa) > hi, eb = (b - fb) < lo || (b - fb) > hi;
    near_edge = ea || eb || (c - fc) < lo || (c - fc) > hi;
    if (exact && ea) kx = (int)floor((ddx::atan2_cr(y, x) + 3.14159265358979323846) / tab.polar_res[0]);
    if (exact && eb) ky = (int)floor(ddx::acos_cr(z / r) / tab.polar_res[1]);
  } else {
    kx = (int)floor(x / tab.voxel_resolution - 0.5);
    ky = (int)floor(y / tab.voxel_resolution - 0.5);
    kz = (int)floor(z / tab.voxel_resolution - 0.5);
  }
}

// vmp_voxel.hpp:199-201 (UNIFORM) and :208-211 (POLAR) for the pass kernels' correspondence lookup of a transformed source point (rot_vgicp_impl.hpp:184-186): the same
// keys as the map build's, edge points through the correctly rounded functions too (round 4; a rare divergent branch: rot_pass_kernel stays at its register budget)
// UNIFORM (round 5): the three quotients by the wave-uniform leaf are three fp64 divisions = ~40 instructions per lane and pass. x * (1 / leaf) differs from the correctly
// rounded x / leaf by at most two ulps of the quotient — below 1e-9 for |key| < 2^21 — so the product decides the floor unless it lands within 1e-9 of an integer; those
// lanes (two in 1e9 per coordinate) take the division as written in the reference. Keys stay bit-exact; checked against the divisions on the device by the parity tests.
ROLO_DEV void voxel_coord_dev(const VoxelTable& tab, double x, double y, double z, int& kx, int& ky, int& kz) {
  if (tab.voxel_type != ROLO_VOXEL_POLAR) {
    const double inv = tab.inv_voxel_resolution;
    const double tx = x * inv - 0.5, ty = y * inv - 0.5, tz = z * inv - 0.5;
    const double fx = floor(tx), fy = floor(ty), fz = floor(tz);
    const double lo = 1e-9, hi = 1.0 - 1e-9;
    const double rx = tx - fx, ry = ty - fy, rz = tz - fz;
    const bool safe = rx > lo && rx < hi && ry > lo && ry < hi && rz > lo && rz < hi && fabs(fx) < 2097152.0 && fabs(fy) < 2097152.0 && fabs(fz) < 2097152.0;
    if (safe) { kx = (int)fx; ky = (int)fy; kz = (int)fz; return; }
  }
  bool near_edge;
  voxel_coord_dev_edge(tab, x, y, z, kx, ky, kz, near_edge);
}

ROLO_DEV bool pack_key(int kx, int ky, int kz, unsigned long long& key) {
  const unsigned ux = (unsigned)(kx + KEY_BIAS), uy = (unsigned)(ky + KEY_BIAS), uz = (unsigned)(kz + KEY_BIAS);
  if ((ux | uy | uz) >> 21) return false;
  key = (unsigned long long)ux | ((unsigned long long)uy << 21) | ((unsigned long long)uz << 42);
  return true;
}
ROLO_DEV void unpack_key(unsigned long long key, int& kx, int& ky, int& kz) {
  kx = (int)(key & 0x1fffffu) - KEY_BIAS;
  ky = (int)((key >> 21) & 0x1fffffu) - KEY_BIAS;
  kz = (int)((key >> 42) & 0x1fffffu) - KEY_BIAS;
}

ROLO_DEV unsigned hash_key(unsigned long long k) {  // splitmix64 finaliser
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return (unsigned)k;
}

// slot h of the table: 16 bytes = { packed key, compact voxel id }
ROLO_DEV unsigned long long* slot_key(const VoxelTable& tab, unsigned h) { return tab.keys + 2 * (size_t)h; }
ROLO_DEV int slot_id(const VoxelTable& tab, unsigned h) { return (int)(unsigned)tab.keys[2 * (size_t)h + 1]; }
ROLO_DEV void set_slot_id(const VoxelTable& tab, unsigned h, int id) { tab.keys[2 * (size_t)h + 1] = (unsigned long long)(unsigned)id; }

// lookup_voxel (vmp_voxel.hpp:226-233): compact voxel id or -1. Key and id of a slot come back in one 16-byte load.
ROLO_DEV int voxel_lookup(const VoxelTable& tab, int kx, int ky, int kz) {
  unsigned long long key;
  if (!pack_key(kx, ky, kz, key)) return -1;
  unsigned h = hash_key(key) & tab.mask;
  while (true) {
    const ulonglong2 s = *reinterpret_cast<const ulonglong2*>(tab.keys + 2 * (size_t)h);
    if (s.x == key) return (int)(unsigned)s.y;
    if (s.x == KEY_EMPTY) return -1;
    h = (h + 1) & tab.mask;
  }
}

// ---- map build, shared by voxelmap.hip and the search kernels that carry parts of it (knn_cov.hip) ---------------------------------
// counters: [0] voxels, [1] error code, [2] points within 1e-12 of a bin edge, [3] float bits of max |coordinate| of the target
ROLO_DEV void note_point(bool valid, bool near_edge, int* counters) {
  const unsigned long long edge = __ballot(valid && near_edge);
  if (edge && (threadIdx.x & 63) == 0) atomicAdd(&counters[2], __popcll(edge));
}

// max |coordinate| of the target -> counters[3]: from the bounding box the neighbour search left on the device (6 order-preserving
// ints: min xyz, max xyz), or, for a target whose covariances were handed in (no search ran), from the points themselves
// fixed-point scales (powers of two) of the position / covariance sums: n * max|value| * scale < 2^62
struct FixScale { double pos, cov; };
ROLO_DEV FixScale fix_scales(int n, const int* counters) {
  const int bits_n = 32 - __clz(max(n, 1));
  const int e = ((counters[3] >> 23) & 0xff) - 127;      // max|coordinate| < 2^(e+1)
  FixScale s;
  s.pos = ldexp(1.0, 62 - bits_n - (max(e, -1) + 1));
  s.cov = ldexp(1.0, 62 - bits_n - 1);                   // |entries| <= 1 for the spectrally bounded regularisations
  return s;
}
ROLO_DEV long long shfl_up_ll(long long v, int off) {
  return (long long)(((unsigned long long)(unsigned)__shfl_up((int)((unsigned long long)v >> 32), off, 64) << 32) | (unsigned)__shfl_up((int)((unsigned long long)v & 0xffffffffull), off, 64));
}

// one point -> (id, 10 values) -> wave-level fold of runs of equal id -> one atomic per run and value.
// fixed_cov: covariances go through the integer sums too (bounded entries); otherwise they keep fp64 atomics (options the
// reference never selects, or covariances handed in by the caller: unbounded entries).
// err (counters + 1, may be nullptr): a point or covariance that cannot go through the fixed-point sums — non-finite (a degenerate
// neighbourhood, e.g. PLANE_S with a zero singular-value sum; the reference's fp64 sums would carry the NaN into the voxel and on into H), or
// an entry outside the bound the scale assumes — raises ROLO_ENONFINITE there instead of becoming a finite but wrong voxel.
ROLO_DEV void accumulate_point(const VoxelTable& tab, int id, const float4& p, const double (&c)[6], const FixScale& S, bool fixed_cov, int* err = nullptr) {
  const int lane = threadIdx.x & 63;
  if (id >= 0 && err) {
    bool ok = fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY;   // false for NaN too
    if (fixed_cov) {
#pragma unroll
      for (int d = 0; d < 6; d++) ok = ok && (fabs(c[d]) <= 1.0 + 1e-9);
    }
    if (!ok) atomicMin(err, ROLO_ENONFINITE);   // the more negative code wins deterministically when a frame also holds a key out of range (ROLO_EKEYRANGE)
  }
  long long q[10];
  double cv[6];
#pragma unroll
  for (int d = 0; d < 10; d++) q[d] = 0;
#pragma unroll
  for (int d = 0; d < 6; d++) cv[d] = 0.0;
  if (id >= 0) {
    q[0] = __double2ll_rn((double)p.x * S.pos); q[1] = __double2ll_rn((double)p.y * S.pos); q[2] = __double2ll_rn((double)p.z * S.pos);
    if (fixed_cov) {
#pragma unroll
      for (int d = 0; d < 6; d++) q[3 + d] = __double2ll_rn(c[d] * S.cov);
    } else {
#pragma unroll
      for (int d = 0; d < 6; d++) cv[d] = c[d];
    }
    q[9] = 1;
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
      const long long o = shfl_up_ll(q[d], off);
      if (dist >= off) q[d] += o;
    }
    if (!fixed_cov) {
#pragma unroll
      for (int d = 0; d < 6; d++) { const double o = __shfl_up(cv[d], off, 64); if (dist >= off) cv[d] += o; }
    }
  }
  const int next_id = __shfl_down(id, 1, 64);
  const bool tail = (lane == 63) || (next_id != id);
  if (tail && id >= 0) {
    unsigned long long* r = reinterpret_cast<unsigned long long*>(tab.rec + (size_t)id * REC_DOUBLES);
#pragma unroll
    for (int d = 0; d < 3; d++) atomicAdd(&r[d], (unsigned long long)q[d]);
    if (fixed_cov) {
#pragma unroll
      for (int d = 3; d < 9; d++) atomicAdd(&r[d], (unsigned long long)q[d]);
    } else {
      double* rd = tab.rec + (size_t)id * REC_DOUBLES;
#pragma unroll
      for (int d = 0; d < 6; d++) atomicAdd(&rd[3 + d], cv[d]);
    }
    atomicAdd(&r[10], (unsigned long long)q[9]);
  }
}

// The same for a whole workgroup of THREADS consecutive points of the curve (the covariance tail): the runs a wavefront folds are SHORT — along the curve a voxel is
// entered and left again and again: 28 runs per 64 points of the nominal frame, 12.7 distinct voxels — and every run costs ten 64-bit device-scope atomics, which is what
// the tail's voxel part is made of (two atomics per run instead of ten: tail 57 -> 33 us). So the runs' sums meet in a small hash table in LDS first (ds_add_u64, keyed by
// the voxel id) and ONE set of ten atomics per distinct voxel of the workgroup goes out: 39.5 per 256 points instead of 110. Integer sums: the map is the same bits.
// Every thread of the workgroup must call this (barriers); a table that runs full sends the run's sums out directly.
template <int THREADS, int TS = 128>
ROLO_DEV void accumulate_point_wg(const VoxelTable& tab, int id, const float4& p, const double (&c)[6], const FixScale& S, int* err = nullptr) {
  static_assert((TS & (TS - 1)) == 0 && TS <= THREADS, "one thread per table entry");
  __shared__ int t_key[TS];
  __shared__ unsigned long long t_sum[10][TS];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < TS) {
    t_key[tid] = -1;
#pragma unroll
    for (int d = 0; d < 10; d++) t_sum[d][tid] = 0ull;
  }
  if (id >= 0 && err) {
    bool ok = fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY;   // false for NaN too
#pragma unroll
    for (int d = 0; d < 6; d++) ok = ok && (fabs(c[d]) <= 1.0 + 1e-9);
    if (!ok) atomicMin(err, ROLO_ENONFINITE);
  }
  long long q[10];
#pragma unroll
  for (int d = 0; d < 10; d++) q[d] = 0;
  if (id >= 0) {
    q[0] = __double2ll_rn((double)p.x * S.pos); q[1] = __double2ll_rn((double)p.y * S.pos); q[2] = __double2ll_rn((double)p.z * S.pos);
#pragma unroll
    for (int d = 0; d < 6; d++) q[3 + d] = __double2ll_rn(c[d] * S.cov);
    q[9] = 1;
  }
  // runs of equal id inside the wavefront first (as accumulate_point): 28 table updates per wavefront instead of 64
  const int prev_id = __shfl_up(id, 1, 64);
  const bool head = (lane == 0) || (prev_id != id);
  const unsigned long long head_mask = __ballot(head);
  const unsigned long long below = head_mask & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
  const int dist = lane - (63 - __clzll(below));
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
    for (int d = 0; d < 10; d++) {
      const long long o = shfl_up_ll(q[d], off);
      if (dist >= off) q[d] += o;
    }
  }
  const int next_id = __shfl_down(id, 1, 64);
  const bool tail = (lane == 63) || (next_id != id);
  __syncthreads();   // the table is clear
  if (tail && id >= 0) {
    unsigned h = ((unsigned)id * 2654435761u) >> (32 - __builtin_ctz(TS));
    int slot = -1;
    for (int probe = 0; probe < TS; probe++) {
      const int prev = atomicCAS(&t_key[h], -1, id);
      if (prev == -1 || prev == id) { slot = (int)h; break; }
      h = (h + 1) & (unsigned)(TS - 1);
    }
    if (slot >= 0) {
#pragma unroll
      for (int d = 0; d < 10; d++) atomicAdd(&t_sum[d][slot], (unsigned long long)q[d]);
    } else {   // table full: straight to the record
      unsigned long long* r = reinterpret_cast<unsigned long long*>(tab.rec + (size_t)id * REC_DOUBLES);
#pragma unroll
      for (int d = 0; d < 9; d++) atomicAdd(&r[d], (unsigned long long)q[d]);
      atomicAdd(&r[10], (unsigned long long)q[9]);
    }
  }
  __syncthreads();
  if (tid < TS && t_key[tid] >= 0) {
    unsigned long long* r = reinterpret_cast<unsigned long long*>(tab.rec + (size_t)t_key[tid] * REC_DOUBLES);
#pragma unroll
    for (int d = 0; d < 9; d++) atomicAdd(&r[d], t_sum[d][tid]);
    atomicAdd(&r[10], t_sum[9][tid]);
  }
}

// one target point into the hash table: claims / finds the slot of its voxel, leaves slot and key by original index
ROLO_DEV void voxel_insert_point(const VoxelTable& tab, const float4* __restrict__ pts, int n, int i, unsigned long long* tgt_keys, int* tgt_slot, int* counters) {
  const float4 p = i < n ? pts[i] : make_float4(1.f, 1.f, 1.f, 0.f);
  int kx, ky, kz; bool near_edge;
  voxel_coord_dev_edge(tab, (double)p.x, (double)p.y, (double)p.z, kx, ky, kz, near_edge);
  note_point(i < n, near_edge, counters);
  if (i >= n) return;
  unsigned long long key;
  if (!pack_key(kx, ky, kz, key)) { atomicMin(&counters[1], ROLO_EKEYRANGE); tgt_slot[i] = -1; tgt_keys[i] = KEY_EMPTY; return; }
  unsigned h = hash_key(key) & tab.mask;
  bool claimed = false;
  while (true) {
    unsigned long long prev = atomicCAS(slot_key(tab, h), KEY_EMPTY, key);
    if (prev == KEY_EMPTY) { claimed = true; break; }
    if (prev == key) break;
    h = (h + 1) & tab.mask;
  }
  if (claimed) {
    // the compact ids come from ONE counter: the lanes of a wavefront that claimed a slot take theirs with one atomic between them (every claim its own atomic on the one
    // address: ~5 500 serialised device-scope atomics per launch of the frame's three)
    const unsigned long long m = __ballot(true);   // (the lanes inside this branch)
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&counters[0], __popcll(m));
    base = __shfl(base, leader, 64);
    const int id = base + __popcll(m & ((1ull << lane) - 1ull));
    set_slot_id(tab, h, id);
    tab.id_keys[id] = key;
    double* r = tab.rec + (size_t)id * REC_DOUBLES;
#pragma unroll
    for (int d = 0; d < REC_DOUBLES; d++) r[d] = 0.0;
  }
  tgt_slot[i] = (int)h;
  tgt_keys[i] = key;
}

// clears the hash table and the four counters (thread t of nthreads); with the bounding box at hand counters[3] gets max |coordinate| right here
ROLO_DEV void voxel_clear_body(unsigned long long* __restrict__ keys, size_t n_slots, const int* __restrict__ bbox6, int* counters, size_t t, size_t nthreads) {
  for (size_t k = t; k < 2 * n_slots; k += nthreads) keys[k] = KEY_EMPTY;   // both words of every slot (key = free, id = -1)
  if (t < 3) counters[t] = 0;
  if (t == 3) {
    float m = 0.f;
    if (bbox6) for (int k = 0; k < 6; k++) { const int o = bbox6[k]; m = fmaxf(m, fabsf(__int_as_float(o >= 0 ? o : o ^ 0x7fffffff))); }
    counters[3] = __float_as_int(m);
  }
}

}  // namespace rolo
