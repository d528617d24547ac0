// K7-K12 — the Gauss-Newton / Levenberg-Marquardt iteration on gfx950.
// Replaces (reference include/rot_gicp/gicp/impl/rot_vgicp_impl.hpp) update_correspondences :173-222,
// so3_linearize :293-388, linearize :225-290, compute_error :391-417, t3_linearize :499-607, compute_t_error
// :610-658, and (lsq_registration_impl.hpp) the drivers rot_step_lm :273-324, step_lm :225-270, step_gn :208-222,
// step_t_optimize :84-139 with their convergence tests.
//
// MI355X design. One *fused pass* kernel per LM trial replaces the reference's separate linearize and error
// sweeps: for every source point it (A) re-evaluates the cost at the trial pose on the correspondences and
// Mahalanobis matrices of the current linearisation — exactly compute_error — and (B) linearises at the trial pose
// itself (new voxel lookup, new Mahalanobis, residual, Jacobian, H, b, cost) — exactly what the reference's next
// so3_linearize would compute if the trial is accepted. Nothing per-correspondence is cached in HBM except one
// 4-byte voxel id: the 3x3 Mahalanobis is recomputed from the 48-byte source covariance and the 96-byte voxel
// record instead of being stored as a 128-byte Matrix4d. Each point's contributions are summed per thread,
// reduce-scattered across the 64-lane wavefront, combined across the four wavefronts of the workgroup through
// LDS, and written as one row of partials. A one-workgroup controller kernel then sums the rows in a fixed order
// (deterministic) and runs the scalar LM logic on the device: LDLT solve, so3/se3 exponential, gain ratio,
// damping update, convergence — so there is no host round trip inside a solve. All pass / controller launches
// are predicated on the device-side state, so a fixed schedule of launches can be enqueued (or graph-captured)
// without knowing how many trials the data will need.
#include "rolo_internal.hpp"
#include <atomic>
#include "lm_begin.hpp"
#include "dev_math.hpp"
#include "voxel_dev.hpp"
#include "peer_dev.hpp"
#include <cfloat>
#include <cstdlib>
#include <type_traits>

namespace rolo {

namespace {

// neighbor_offsets (vmp_voxel.hpp:13-47): [0] DIRECT1, [1..8) DIRECT7, [8..35) DIRECT27
__constant__ int c_offsets[35][3] = {
    {0, 0, 0},
    {0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1},
    {-1, -1, -1}, {-1, -1, 0}, {-1, -1, 1}, {-1, 0, -1}, {-1, 0, 0}, {-1, 0, 1}, {-1, 1, -1}, {-1, 1, 0}, {-1, 1, 1},
    {0, -1, -1},  {0, -1, 0},  {0, -1, 1},  {0, 0, -1},  {0, 0, 0},  {0, 0, 1},  {0, 1, -1},  {0, 1, 0},  {0, 1, 1},
    {1, -1, -1},  {1, -1, 0},  {1, -1, 1},  {1, 0, -1},  {1, 0, 0},  {1, 0, 1},  {1, 1, -1},  {1, 1, 0},  {1, 1, 1}};

ROLO_DEV int offset_base(int n_off) { return n_off == 1 ? 0 : (n_off == 7 ? 1 : 8); }

struct Rec { Vec3 mean; Sym3 cov; double w; };
ROLO_DEV Rec load_rec(const double* __restrict__ rec, int id) {
  const double* r = rec + (size_t)id * REC_DOUBLES;
  Rec o;
  o.mean = Vec3{r[0], r[1], r[2]};
  o.cov = Sym3{r[3], r[4], r[5], r[6], r[7], r[8]};
  o.w = r[9];
  return o;
}

// DPP lane moves of a double (two 32-bit halves): pure VALU, no LDS crossbar. Lanes a row_mask leaves out receive 0.0.
template <int CTRL, int ROW_MASK = 0xf>
ROLO_DEV double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Sum over the wavefront, result valid in lane 63: pairs, quads (quad_perm), 8 (row_half_mirror), 16 (row_mirror),
// 32 (row_bcast:15 into rows 1 and 3), 64 (row_bcast:31 into rows 2 and 3). ds_bpermute exchanges cost ~40 cycles each
// here (the LDS crossbar is shared by the CU's wavefronts): 144 of them were a third of the SO(3) pass.
ROLO_DEV double wave_sum_dpp63(double v) {
  v += dpp_mov_f64<0xB1>(v);          // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E>(v);          // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141>(v);         // row_half_mirror
  v += dpp_mov_f64<0x140>(v);         // row_mirror
  v += dpp_mov_f64<0x142, 0xa>(v);    // row_bcast:15
  v += dpp_mov_f64<0x143, 0xc>(v);    // row_bcast:31
  return v;
}

// Sum NV per-lane values over the 64 lanes of the wavefront as a reduce-scatter WITHOUT selects (round 5). At every halving step the
// values are taken in pairs (k, k + half): the lanes whose step bit is 0 will own value k, the others value k + half, and each hands the
// value it does not own to its partner and adds what it receives. gfx950's v_permlane32_swap / v_permlane16_swap do the hand-over of a
// whole pair in place — swap(X, Y) leaves [X.lo | Y.lo] in X and [X.hi | Y.hi] in Y, so X + Y IS the step (2 swaps + 1 add per pair of
// doubles, no v_cndmask); inside a row of 16 the same step is three DPP moves under a bank mask (keep = X with Y's banks patched in,
// received = row_ror:8 / row_shl:4 + row_shr:4 of the other value) and the last two distances are plain butterflies on what is left.
// 12 values (SO(3) pass): 63 VALU instructions against 216 for one DPP butterfly per value; 30 values (translation / 6-dof passes): 126
// against ~400 for round 2's reduce-scatter, whose generic lane-xor moves paid four v_cndmask per exchanged double (half of that
// kernel's instructions). Afterwards lane l holds the complete sums of the NP / 16 values j + (NP / 16) * (b2 + 2 b3 + 4 b4 + 8 b5),
// b_i = bit i of l — identical in the four lanes of a quad; the order of the additions is fixed (deterministic).
ROLO_DEV void rs_swap32(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]); y = __hiloint2double((int)hi[1], (int)lo[1]);
}
ROLO_DEV void rs_swap16(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]); y = __hiloint2double((int)hi[1], (int)lo[1]);
}
// one reduce-scatter step inside a row of 16 lanes: OFF = 8 (banks 0,1 own x, banks 2,3 own y) or 4 (banks 0,2 own x, banks 1,3 own y)
template <int OFF>
ROLO_DEV double rs_row_step(double x, double y) {
  constexpr int YB = OFF == 8 ? 0xc : 0xa, XB = OFF == 8 ? 0x3 : 0x5;   // bank masks of the lanes that own y / x
  constexpr int UP = OFF == 8 ? 0x128 : 0x104, DN = OFF == 8 ? 0x128 : 0x114;   // lane <- lane + OFF (row_ror:8 / row_shl:4), lane <- lane - OFF (row_ror:8 / row_shr:4)
  const int xl = __double2loint(x), xh = __double2hiint(x), yl = __double2loint(y), yh = __double2hiint(y);
  const int kl = __builtin_amdgcn_update_dpp(xl, yl, 0xE4, 0xf, YB, false), kh = __builtin_amdgcn_update_dpp(xh, yh, 0xE4, 0xf, YB, false);   // keep: x, y where y is owned
  int rl = __builtin_amdgcn_update_dpp(0, xl, UP, 0xf, XB, false), rh = __builtin_amdgcn_update_dpp(0, xh, UP, 0xf, XB, false);               // received: the partner's x ...
  rl = __builtin_amdgcn_update_dpp(rl, yl, DN, 0xf, YB, false); rh = __builtin_amdgcn_update_dpp(rh, yh, DN, 0xf, YB, false);                 // ... or the partner's y
  return __hiloint2double(kh, kl) + __hiloint2double(rh, rl);
}
template <int NV>
ROLO_DEV void wave_reduce_scatter(const double (&acc)[NV], double* __restrict__ red_row /* LDS, NV_MAX */) {
  constexpr int NP = NV <= 16 ? 16 : 32;
  constexpr int NL = NP / 16;   // values left per lane after the four scatter steps
  static_assert(NV <= 32, "at most 32 values");
  const int lane = threadIdx.x & 63;
  double v[NP];
#pragma unroll
  for (int k = 0; k < NP; k++) v[k] = k < NV ? acc[k] : 0.0;
#pragma unroll
  for (int k = 0; k < NP / 2; k++) { rs_swap32(v[k], v[k + NP / 2]); v[k] += v[k + NP / 2]; }
#pragma unroll
  for (int k = 0; k < NP / 4; k++) { rs_swap16(v[k], v[k + NP / 4]); v[k] += v[k + NP / 4]; }
#pragma unroll
  for (int k = 0; k < NP / 8; k++) v[k] = rs_row_step<8>(v[k], v[k + NP / 8]);
#pragma unroll
  for (int k = 0; k < NL; k++) v[k] = rs_row_step<4>(v[k], v[k + NL]);
#pragma unroll
  for (int k = 0; k < NL; k++) { v[k] += lane_xor_f64<2>(v[k]); v[k] += lane_xor_f64<1>(v[k]); }
  if ((lane & 3) == 0) {
    const int base = NL * (((lane >> 2) & 1) + 2 * ((lane >> 3) & 1) + 4 * ((lane >> 4) & 1) + 8 * ((lane >> 5) & 1));
#pragma unroll
    for (int k = 0; k < NL; k++) if (base + k < NV) red_row[base + k] = v[k];
  }
}

// only_first (wave-uniform): a pass that evaluated the trial cost alone (LmState::lin_skip) has ONE value to sum — acc[0] — and writes zeros to the other slots
template <int NV, int THREADS = PASS_THREADS>
ROLO_DEV void block_reduce_store(double (&acc)[NV], const int (&slot)[NV], double* __restrict__ out_row, bool only_first = false) {
  __shared__ double red[THREADS / 64][NV_MAX];
  const int wv = threadIdx.x >> 6;
  if (only_first) {
    // the additions value 0 goes through in wave_reduce_scatter, in the same order (lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1: lane 0 ends up with the same bits), so a
    // trial's cost does not depend on whether its pass carried the linearisation half
    double t = acc[0];
    t += lane_xor_f64<32>(t); t += lane_xor_f64<16>(t); t += lane_xor_f64<8>(t); t += lane_xor_f64<4>(t); t += lane_xor_f64<2>(t); t += lane_xor_f64<1>(t);
    if ((threadIdx.x & 63) == 0) red[wv][0] = t;
    __syncthreads();
    if (threadIdx.x < NV) {
      double s0 = 0;
      if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < THREADS / 64; w++) s0 += red[w][0];
      }
      int sl = 0;
#pragma unroll
      for (int v = 0; v < NV; v++) if (threadIdx.x == v) sl = slot[v];
      out_row[sl] = s0;
    }
    return;
  }
  wave_reduce_scatter<NV>(acc, red[wv]);
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) s += red[w][threadIdx.x];
    // slot[] is a compile-time table; pick by thread index
    int sl = 0;
#pragma unroll
    for (int v = 0; v < NV; v++) if (threadIdx.x == v) sl = slot[v];
    out_row[sl] = s;
  }
}

// H += wh * J^T M J, b += J^T Mv_b for J = [skew(a) | -I]  (rot_vgicp_impl.hpp:264-266, 353, 576-578). The columns u_i of J are
// u_0 = (0, a.z, -a.y), u_1 = (-a.z, 0, a.x), u_2 = (a.y, -a.x, 0), u_3.. = -e_x, -e_y, -e_z. Round 5: the structure is written out — the
// zero components are dropped (0 * x + y IS y for the finite values a pass sees) and the -I columns are read off M: -M's columns, H's
// translation block wh * M, the mixed block -wh * (M u_j) — instead of the generic dot3(u_i, M u_j) with its multiplications by 0.0 and
// -1.0, which the compiler must keep. Same operation order for what remains, so H and b keep their bits (up to the sign of a zero);
// 42 fp64 instructions against 60 (SO(3)), 60 against 156 (6 dof).
// acc layout inside the kernels: [0]=yi [1]=y [2]=n [3 .. 3+NH) H lower triangle, then b
template <int DOF>
ROLO_DEV void accumulate_hb(const Sym3& M, const Vec3& a, double wh, double wb_unused, const Vec3& Mv_b, double* Hacc, double* bacc) {
  (void)wb_unused;
  // (p2: the generic expression's own order once its zero term is gone — the first product rounded, the second fused into the sum)
  auto p2 = [](double x1, double y1, double x2, double y2) { return fma(x2, y2, x1 * y1); };
  // M u_j for the three skew columns: two products per component
  const Vec3 Mu0{p2(M.xy, a.z, M.xz, -a.y), p2(M.yy, a.z, M.yz, -a.y), p2(M.yz, a.z, M.zz, -a.y)};
  const Vec3 Mu1{p2(M.xx, -a.z, M.xz, a.x), p2(M.xy, -a.z, M.yz, a.x), p2(M.xz, -a.z, M.zz, a.x)};
  const Vec3 Mu2{p2(M.xx, a.y, M.xy, -a.x), p2(M.xy, a.y, M.yy, -a.x), p2(M.xz, a.y, M.yz, -a.x)};
  // rotation block, lower triangle row-major: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2); u_i . v with the zero component left out
  auto d0 = [&](const Vec3& v) { return p2(a.z, v.y, -a.y, v.z); };
  auto d1 = [&](const Vec3& v) { return p2(-a.z, v.x, a.x, v.z); };
  auto d2 = [&](const Vec3& v) { return p2(a.y, v.x, -a.x, v.y); };
  if constexpr (DOF == 3) {
    Hacc[0] += wh * d0(Mu0);
    Hacc[1] += wh * d1(Mu0); Hacc[2] += wh * d1(Mu1);
    Hacc[3] += wh * d2(Mu0); Hacc[4] += wh * d2(Mu1); Hacc[5] += wh * d2(Mu2);
    bacc[0] += d0(Mv_b); bacc[1] += d1(Mv_b); bacc[2] += d2(Mv_b);
  } else {
    static_assert(DOF == 6, "3 or 6 degrees of freedom");
    // row-major lower triangle of the 6 x 6: rows 0..2 as above, row 3+c = [ -(M u_0).c  -(M u_1).c  -(M u_2).c | M(c, 0..c) ]
    Hacc[0] += wh * d0(Mu0);
    Hacc[1] += wh * d1(Mu0); Hacc[2] += wh * d1(Mu1);
    Hacc[3] += wh * d2(Mu0); Hacc[4] += wh * d2(Mu1); Hacc[5] += wh * d2(Mu2);
    Hacc[6] += wh * (-Mu0.x); Hacc[7] += wh * (-Mu1.x); Hacc[8] += wh * (-Mu2.x); Hacc[9] += wh * M.xx;
    Hacc[10] += wh * (-Mu0.y); Hacc[11] += wh * (-Mu1.y); Hacc[12] += wh * (-Mu2.y); Hacc[13] += wh * M.xy; Hacc[14] += wh * M.yy;
    Hacc[15] += wh * (-Mu0.z); Hacc[16] += wh * (-Mu1.z); Hacc[17] += wh * (-Mu2.z); Hacc[18] += wh * M.xz; Hacc[19] += wh * M.yz; Hacc[20] += wh * M.zz;
    bacc[0] += d0(Mv_b); bacc[1] += d1(Mv_b); bacc[2] += d2(Mv_b);
    bacc[3] -= Mv_b.x; bacc[4] -= Mv_b.y; bacc[5] -= Mv_b.z;
  }
}

// wave-uniform values that come out of LDS (the fused launches keep the state there) belong in scalar registers
ROLO_DEV double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
ROLO_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// what a pass needs of one source point: loaded before anything that depends on the LM state
// CA: the six-entry covariance; or, when the PLANE covariances were computed by the library (a.nrm), m with C_A = I - m m^T: 24 bytes per point and pass
// instead of 48, and the rotation R C_A R^T = R R^T - (R m)(R m)^T is 27 multiply-adds instead of 45 (rotated_cov)
struct PtIn { float4 pf; Sym3 CA; Vec3 m; };
ROLO_DEV PtIn load_pt(const PassArgs& a, int i) {
  PtIn o{};
  o.pf = a.src[i];
  const size_t pitch = (size_t)a.n_total;
  if (a.nrm) {
    o.m = Vec3{a.nrm[i], a.nrm[pitch + i], a.nrm[2 * pitch + i]};
    if (o.m.x == o.m.x) return o;   // (NaN: a rank <= 1 neighbourhood whose covariance is not of the plane form — knn_covariance_finish — takes its six entries)
  }
  o.CA = Sym3{a.cov[i], a.cov[pitch + i], a.cov[2 * pitch + i], a.cov[3 * pitch + i], a.cov[4 * pitch + i], a.cov[5 * pitch + i]};
  return o;
}
// R (I - m m^T) R^T = R R^T - (R m)(R m)^T. R R^T is NOT taken as the identity: a caller's guess arrives as a float matrix, orthonormal to 1e-7 only, and the
// reference rotates the covariance with exactly that matrix (rot_vgicp_impl.hpp:204-222) — the first version of this path assumed I and was 2e-6 off in the cost of a
// solve started from a float guess. The product comes from the LM state, where the controller keeps it next to each rotation (lm_set_rrt): 9 + 6 multiply-adds per
// lane are left of the rotation's 45.
ROLO_DEV Sym3 rotated_cov(const PassArgs& a, const double* R, const double* __restrict__ S6 /* R R^T from the LM state (lm_set_rrt) */, const PtIn& in) {
  if (a.nrm && in.m.x == in.m.x) {
    const Sym3 S{uni(S6[0]), uni(S6[1]), uni(S6[2]), uni(S6[3]), uni(S6[4]), uni(S6[5])};
    const Vec3 q = mat3_mulv(R, in.m);
    return Sym3{S.xx - q.x * q.x, S.xy - q.x * q.y, S.xz - q.x * q.z, S.yy - q.y * q.y, S.yz - q.y * q.z, S.zz - q.z * q.z};
  }
  return sym3_rotate(R, in.CA);
}

template <int DOF>
ROLO_DEV void rot_pass_compute(const PassArgs& a, const LmState* __restrict__ st, const int i, const bool valid, const PtIn& in,
                               double (&acc)[3 + DOF * (DOF + 1) / 2 + DOF]) {
  constexpr int NH = DOF * (DOF + 1) / 2;
  const int phase = uni(st->phase);
  const int cur = uni(st->cur);
  const bool skip_lin = phase == 1 && uni(st->lin_skip) != 0;   // (A) only: the trial before this one was rejected (LmState::lin_skip)
  const int* __restrict__ corr_old = a.corr[cur];
  int* __restrict__ corr_new = a.corr[phase == 0 ? cur : (cur ^ 1)];
  double R0[9], R1[9], t1[3];
#pragma unroll
  for (int k = 0; k < 9; k++) { R0[k] = uni(st->x0_R[k]); R1[k] = uni(st->xt_R[k]); }
#pragma unroll
  for (int k = 0; k < 3; k++) t1[k] = uni(st->xt_t[k]);
  if (valid) {
    const float4 pf = in.pf;
    const Vec3 p{(double)pf.x, (double)pf.y, (double)pf.z};
    Vec3 tp = mat3_mulv(R1, p);
    tp.x += t1[0]; tp.y += t1[1]; tp.z += t1[2];
    const int n_off = a.n_off;
    // (A) compute_error(xi): cached correspondences, Mahalanobis of the linearisation pose x0
    // (round 5 measured the reference's per-correspondence Mahalanobis cache here — six fp64 of M(x0) per point, written by the (B) half, read by the next
    // trial's (A) half, rot_vgicp_impl.hpp:204-222 — and it LOST: 96 B per point and pass of extra traffic cost more than the ~100 instructions saved,
    // 2694 against 2875 scans/s; profiles/DEAD_ENDS.md. Neither stage of these PASS kernels caches M: the translation stage's cache, whose M is constant, measured neutral and is not kept either. The resident kernel does — in LDS, where it costs no traffic: lmp_rot_body.)
    if (phase == 1) {
      const Sym3 RCA0 = rotated_cov(a, R0, st->x0_S, in);
      for (int o = 0; o < n_off; o++) {
        const int vid = corr_old[(size_t)i * n_off + o];
        if (vid >= 0) {
          const Rec r = load_rec(a.tab.rec, vid);
          const Sym3 M = sym3_inverse(sym3_add(r.cov, RCA0));
          const Vec3 e{r.mean.x - tp.x, r.mean.y - tp.y, r.mean.z - tp.z};
          acc[0] += r.w * dot3(e, sym3_mulv(M, e));
        }
      }
    }
    // (B) so3_linearize / linearize at xi, with update_correspondences(xi) fused in
    if (skip_lin) return;
    int kx, ky, kz;
    voxel_coord_dev(a.tab, tp.x, tp.y, tp.z, kx, ky, kz);
    const Sym3 RCA1 = rotated_cov(a, R1, st->xt_S, in);
    const int ob = offset_base(n_off);
    for (int o = 0; o < n_off; o++) {
      const int vid = voxel_lookup(a.tab, kx + c_offsets[ob + o][0], ky + c_offsets[ob + o][1], kz + c_offsets[ob + o][2]);
      corr_new[(size_t)i * n_off + o] = vid;
      if (vid >= 0) {
        const Rec r = load_rec(a.tab.rec, vid);
        const Sym3 M = sym3_inverse(sym3_add(r.cov, RCA1));
        const Vec3 e{r.mean.x - tp.x, r.mean.y - tp.y, r.mean.z - tp.z};
        const Vec3 Me = sym3_mulv(M, e);
        acc[1] += r.w * dot3(e, Me);
        acc[2] += 1.0;
        const Vec3 wMe{r.w * Me.x, r.w * Me.y, r.w * Me.z};
        accumulate_hb<DOF>(M, tp, r.w, 0.0, wMe, &acc[3], &acc[3 + NH]);
      }
    }
  }
}

#ifdef ROLO_PASS_STATS
__device__ unsigned long long g_pass_t[8];   // shader-clock ticks per phase of rot_pass_body, summed over thread 0 of every workgroup; [7] = count
extern "C" int rolo_debug_pass_times(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pass_t), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#define PT_STAMP(k) const long long pt##k = clock64()
#else
#define PT_STAMP(k)
#endif
template <int DOF>
ROLO_DEV void rot_pass_body(const PassArgs& a, const LmState* __restrict__ st, const int block) {
  ROLO_SHORT_KERNEL_PRIO();
  PT_STAMP(0);
  if (st->stage != 1) return;
  PT_STAMP(1);
  constexpr int NH = DOF * (DOF + 1) / 2;
  constexpr int NV = 3 + NH + DOF;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) acc[v] = 0.0;
  const int i = a.begin + block * PASS_THREADS + threadIdx.x;
  const bool valid = i < a.end;
  PtIn in{};
  if (valid) in = load_pt(a, i);
  rot_pass_compute<DOF>(a, st, i, valid, in, acc);
#ifdef ROLO_PASS_STATS
  { double sink = 0; for (int v = 0; v < NV; v++) sink += acc[v]; asm volatile("" :: "v"(sink)); }   // the accumulators are final here
#endif
  PT_STAMP(2);
  int slot[NV];
  slot[0] = V_YI; slot[1] = V_Y; slot[2] = V_N;
#pragma unroll
  for (int v = 0; v < NH; v++) slot[3 + v] = V_H + v;
#pragma unroll
  for (int v = 0; v < DOF; v++) slot[3 + NH + v] = V_B + v;
  block_reduce_store<NV>(acc, slot, a.partials + (size_t)block * NV_MAX, st->phase == 1 && st->lin_skip != 0);
#ifdef ROLO_PASS_STATS
  if (threadIdx.x == 0) {   // state flag known | point data + voxel records in, accumulators final | block reduction + row store
    const long long pt3 = clock64();
    atomicAdd(&g_pass_t[0], (unsigned long long)(pt1 - pt0)); atomicAdd(&g_pass_t[1], (unsigned long long)(pt2 - pt1)); atomicAdd(&g_pass_t[2], (unsigned long long)(pt3 - pt2));
    atomicAdd(&g_pass_t[7], 1ull);
  }
#endif
}

// translation stage: t3_linearize (B) + compute_t_error (A) on the correspondences of the last rotation
// linearisation, Mahalanobis from st->tr_R.
ROLO_DEV void trans_pass_compute(const PassArgs& a, const LmState* __restrict__ st, const int i, const bool valid, const PtIn& in, double (&acc)[30]) {
  constexpr int NH = 21;
  const int phase = uni(st->phase);
  const bool skip_lin = phase == 1 && uni(st->lin_skip) != 0;
  const int tr_cur = uni(st->tr_cur);
  const int* __restrict__ corr = a.corr[tr_cur];
  // (the stage's Mahalanobis matrices are those of the LAST rotation linearisation — constant over all its passes. Caching the six fp64 per point
  // after the first pass instead of rotating / inverting again was measured in round 5 and is not kept: 2957 against 2958 scans/s, profiles/DEAD_ENDS.md)
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = uni(st->tr_R[k]);
  const Vec3 tt{uni(st->tt[0]), uni(st->tt[1]), uni(st->tt[2])};
  const Vec3 g{uni(st->g[0]), uni(st->g[1]), uni(st->g[2])};
  const double lam_n = uni(st->lam_over_n);
  // last_transform keeps its initial value — Zero in t3_linearize (:539), (1,0,0,0) in compute_t_error (:637). The quotients
  // last / dt_{n-1} and 1 / dt_n are the same for every point: formed once when the stage starts (trans_consts), not by every lane of every pass
  // (seven fp64 divisions = ~100 of the pass's ~700 instructions).
  const Vec3 lAq{uni(st->lastA_q[0]), uni(st->lastA_q[1]), uni(st->lastA_q[2])}, lBq{uni(st->lastB_q[0]), uni(st->lastB_q[1]), uni(st->lastB_q[2])};
  const double inv_dtn = uni(st->inv_dtn);
  if (valid) {
    const float4 pf = in.pf;
    const Vec3 p{(double)pf.x, (double)pf.y, (double)pf.z};
    const Vec3 tp{p.x + tt.x, p.y + tt.y, p.z + tt.z};
    const Vec3 ba{p.x - g.x, p.y - g.y, p.z - g.z};
    // (times 1 / dt_n instead of the reference's division: one rounding of a uniform factor, far below the rounding noise ba - tp already carries — the
    // difference is -(g + t) for every point in exact arithmetic — and three fp64 divisions = 45 instructions per lane and pass less)
    const Vec3 dv{(ba.x - tp.x) * inv_dtn, (ba.y - tp.y) * inv_dtn, (ba.z - tp.z) * inv_dtn};
    const Vec3 ctA{dv.x - lAq.x, dv.y - lAq.y, dv.z - lAq.z};
    const Vec3 ctB{dv.x - lBq.x, dv.y - lBq.y, dv.z - lBq.z};
    const int n_off = a.n_off;
    const Sym3 RCA = rotated_cov(a, R, st->tr_S, in);
    for (int o = 0; o < n_off; o++) {
      const int vid = corr[(size_t)i * n_off + o];
      if (vid < 0) continue;
      const double* __restrict__ rr = a.tab.rec + (size_t)vid * REC_DOUBLES;
      const Vec3 mean{rr[0], rr[1], rr[2]};
      const double w = rr[9];
      const Sym3 M = sym3_inverse(sym3_add(Sym3{rr[3], rr[4], rr[5], rr[6], rr[7], rr[8]}, RCA));
      const Vec3 e{mean.x - tp.x, mean.y - tp.y, mean.z - tp.z};
      const Vec3 Me = sym3_mulv(M, e);
      const double eMe = dot3(e, Me);
      if (phase == 1) acc[0] += w * (eMe + lam_n * dot3(ctA, sym3_mulv(M, ctA)));
      if (skip_lin) continue;   // the trial before this one was rejected: the cost alone (LmState::lin_skip)
      const Vec3 McB = sym3_mulv(M, ctB);
      acc[1] += w * (eMe + lam_n * dot3(ctB, McB));
      acc[2] += 1.0;
      const double s1 = lam_n * inv_dtn;
      const Vec3 vb{w * (Me.x + s1 * McB.x), w * (Me.y + s1 * McB.y), w * (Me.z + s1 * McB.z)};
      accumulate_hb<6>(M, tp, w * (1.0 + lam_n * inv_dtn * inv_dtn), 0.0, vb, &acc[3], &acc[3 + NH]);
    }
  }
}

ROLO_DEV void trans_pass_body(const PassArgs& a, const LmState* __restrict__ st, const int block) {
  ROLO_SHORT_KERNEL_PRIO();
  if (st->stage != 2) return;
  constexpr int NH = 21, NV = 3 + NH + 6;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) acc[v] = 0.0;
  const int i = a.begin + block * PASS_THREADS + threadIdx.x;
  const bool valid = i < a.end;
  PtIn in{};
  if (valid) in = load_pt(a, i);
  trans_pass_compute(a, st, i, valid, in, acc);
  int slot[NV];
  slot[0] = V_YI; slot[1] = V_Y; slot[2] = V_N;
#pragma unroll
  for (int v = 0; v < NH; v++) slot[3 + v] = V_H + v;
#pragma unroll
  for (int v = 0; v < 6; v++) slot[3 + NH + v] = V_B + v;
  block_reduce_store<NV>(acc, slot, a.partials + (size_t)block * NV_MAX, st->phase == 1 && st->lin_skip != 0);
}

// Workgroup b of a launch runs on XCD b mod 8, and every XCD has its own L2, emptied at every kernel boundary: with the points dealt round-robin each
// XCD fetches (nearly) ALL voxel records and hash slots again in every pass — 8 x the map from the Infinity Cache per launch. A LiDAR cloud arrives in
// firing order (column-major: consecutive points sweep the azimuth), so giving XCD x the x-th EIGHTH of the point blocks makes it a 45-degree sector
// whose voxels no other XCD needs (round 5; the permutation of knn_packet.hpp's small-launch case). Rows stay indexed by the logical block: the
// controller's summation order does not depend on the mapping.
ROLO_DEV int pass_xcd_block(const PassArgs& a, int b, int G) {
  if (!a.xcd_map) return b;
  const int x = b & 7, k = b >> 3, q = G >> 3, r = G & 7;   // XCD x owns G / 8 (+1 for x < G % 8) consecutive blocks
  return x * q + min(x, r) + k;
}
template <int DOF>
__global__ __launch_bounds__(PASS_THREADS) void rot_pass_kernel(PassArgs a, const LmState* __restrict__ st) { rot_pass_body<DOF>(a, st, pass_xcd_block(a, blockIdx.x, gridDim.x)); }
__global__ __launch_bounds__(PASS_THREADS) void trans_pass_kernel(PassArgs a, const LmState* __restrict__ st) { trans_pass_body(a, st, pass_xcd_block(a, blockIdx.x, gridDim.x)); }

// Batched form (rolo_batch_*, BASELINE config 5): one launch evaluates the same LM trial of B independent frame
// pairs. Workgroups [s * bps, (s + 1) * bps) belong to slot s; every slot has its own clouds, voxel table,
// correspondence cache, partial rows and LM state, and is predicated individually. The LM chain is a string of tiny
// latency-bound launches, so batching B frames through it costs almost the same wall time as one.
template <int DOF>
__global__ __launch_bounds__(PASS_THREADS) void rot_pass_batch_kernel(const BatchSlot* __restrict__ slots, int bps) {
  const int s = blockIdx.x / bps, lb = blockIdx.x - s * bps;
  if (lb >= slots[s].grid) return;
  rot_pass_body<DOF>(slots[s].a, slots[s].st, lb);
}
__global__ __launch_bounds__(PASS_THREADS) void trans_pass_batch_kernel(const BatchSlot* __restrict__ slots, int bps) {
  const int s = blockIdx.x / bps, lb = blockIdx.x - s * bps;
  if (lb >= slots[s].grid) return;
  trans_pass_body(slots[s].a, slots[s].st, lb);
}

// fixed-order sum of the per-workgroup rows (deterministic for a given grid)
ROLO_DEV void reduce_rows(const double* __restrict__ partials, int nblocks, double* sums /* shared, NV_MAX */) {
  __shared__ double part[8][NV_MAX];
  const int v = threadIdx.x & 31, q = threadIdx.x >> 5;  // 256 threads = 8 strided groups of 32 values
  // The rows were written by other CUs a moment ago, so every load is a ~1-2 us L2/fabric round trip and this
  // reduction is pure latency: issue 64 independent loads per thread before the first add — one round trip for the
  // 512 rows of a 131k-point cloud (16 in flight: four round trips, half of the controller's time). The combination
  // order is fixed => deterministic for a given grid.
  constexpr int INFLIGHT = 64;
  double s0 = 0;
  for (int b0 = q; b0 < nblocks; b0 += 8 * INFLIGHT) {
    double r[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) {
      const int b = b0 + 8 * u;
      // an always-valid address and a select: `cond ? load : 0` compiles to one exec-masked branch per load (64 of them in front of the data)
      const double x = partials[(size_t)min(b, nblocks - 1) * NV_MAX + v];
      r[u] = (b < nblocks) ? x : 0.0;
    }
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) s0 += r[u];
  }
  part[q][v] = s0;
  __syncthreads();
  if (threadIdx.x < NV_MAX) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) t += part[k][threadIdx.x];
    sums[threadIdx.x] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void reduce_kernel(const double* __restrict__ partials, int nblocks, double* out, const LmState* st, int stage) {
  if (stage >= 0 && st->stage != stage) return;
  __shared__ double sums[NV_MAX];
  reduce_rows(partials, nblocks, sums);
  if (threadIdx.x < NV_MAX) out[threadIdx.x] = sums[threadIdx.x];
}

// ---- scalar LM logic (one thread) -----------------------------------------------------------------------
template <int N>
ROLO_DEV void ldlt_solve(const double* Hfull /* 6x6 storage, row stride 6 */, double lambda, const double* b, double* x) {
  // Solves (H + lambda I) x = -b by an LDL^T factorisation of the lower triangle. The reference uses Eigen::LDLT
  // (diagonal pivoting); H + lambda I is symmetric positive definite here, for which the unpivoted factorisation is
  // backward stable as well, so the two agree to rounding (checked against the oracle's pivoted solve by the parity
  // tests). No pivoting means no dynamic indexing: every loop unrolls and the whole solve lives in registers
  // instead of scratch memory (the pivoted version cost ~800 scratch round trips per controller launch).
  double L[N][N];
  double D[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    double d = Hfull[j * 6 + j] + lambda;
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
    D[j] = d;
    const double inv = (d != 0.0) ? 1.0 / d : 0.0;
#pragma unroll
    for (int i = j + 1; i < N; i++) {
      double v = Hfull[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
      L[i][j] = v * inv;
    }
  }
  double y[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    double v = -b[i];
#pragma unroll
    for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; i++) y[i] = (D[i] != 0.0) ? y[i] / D[i] : 0.0;
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    double v = y[i];
#pragma unroll
    for (int k = i + 1; k < N; k++) v -= L[k][i] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; i++) x[i] = y[i];
}

// sin / cos of a small angle by their Taylor series (|x| <= 0.5: the 10th term is below 2e-23 relative) — the half angle of an LM step is
// a few 1e-2 at most; the library sincos() is ~150 instructions of range reduction on the controller's one serial lane. Agrees with libm
// to the last ulp or two, far inside what the LM decisions can see (the oracle comparison of the traces is the test).
ROLO_DEV void sincos_small(double x, double* s, double* c) {
  if (fabs(x) > 0.5) { sincos(x, s, c); return; }
  const double z = x * x;
  double ps = -1.0 / 121645100408832000.0;            // -1/19!
  ps = fma(ps, z, 1.0 / 355687428096000.0);           //  1/17!
  ps = fma(ps, z, -1.0 / 1307674368000.0);            // -1/15!
  ps = fma(ps, z, 1.0 / 6227020800.0);                //  1/13!
  ps = fma(ps, z, -1.0 / 39916800.0);                 // -1/11!
  ps = fma(ps, z, 1.0 / 362880.0);                    //  1/9!
  ps = fma(ps, z, -1.0 / 5040.0);                     // -1/7!
  ps = fma(ps, z, 1.0 / 120.0);                       //  1/5!
  ps = fma(ps, z, -1.0 / 6.0);                        // -1/3!
  *s = fma(x * z, ps, x);
  double pc = 1.0 / 6402373705728000.0;               //  1/18!
  pc = fma(pc, z, -1.0 / 20922789888000.0);           // -1/16!
  pc = fma(pc, z, 1.0 / 87178291200.0);               //  1/14!
  pc = fma(pc, z, -1.0 / 479001600.0);                // -1/12!
  pc = fma(pc, z, 1.0 / 3628800.0);                   //  1/10!
  pc = fma(pc, z, -1.0 / 40320.0);                    // -1/8!
  pc = fma(pc, z, 1.0 / 720.0);                       //  1/6!
  pc = fma(pc, z, -1.0 / 24.0);                       // -1/4!
  pc = fma(pc, z, 0.5);                               //  1/2!
  *c = fma(-z, pc, 1.0);
}

ROLO_DEV void so3_exp_R(const double* w, double* R) {  // so3.hpp:59-77 + Quaterniond::toRotationMatrix
  const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (theta_sq < 1e-10) {
    const double q4 = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * q4;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * q4;
  } else {
    const double theta = sqrt(theta_sq), half = 0.5 * theta;
    double sh, ch;
    sincos_small(half, &sh, &ch);
    imag = sh / theta;
    real = ch;
  }
  const double qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

ROLO_DEV void se3_exp_Rt(const double* a, double* R, double* t) {  // so3.hpp:80-103
  const double wx = a[0], wy = a[1], wz = a[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  so3_exp_R(a, R);
  double V[9];
  if (theta < 1e-10) {
    for (int i = 0; i < 9; i++) V[i] = R[i];
  } else {
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    const double th2 = theta * theta;
    const double c1 = (1.0 - cos(theta)) / th2, c2 = (theta - sin(theta)) / (th2 * theta);
    for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
  }
  for (int i = 0; i < 3; i++) t[i] = V[i * 3] * a[3] + V[i * 3 + 1] * a[4] + V[i * 3 + 2] * a[5];
}

// (everything the controller runs is templated on the degrees of freedom: with a run-time count the loops below stay loops of dependent
// LDS accesses on ONE lane, and that lane's latency is the controller's whole duration)
template <int DOF>
ROLO_DEV void unpack_hb(LmState* __restrict__ st, const double* __restrict__ S) {
  int t = 0;
#pragma unroll
  for (int i = 0; i < DOF; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) { const double v = S[V_H + t]; st->H[i * 6 + j] = v; st->H[j * 6 + i] = v; t++; }
  }
#pragma unroll
  for (int i = 0; i < DOF; i++) st->b[i] = S[V_B + i];
  st->y0 = S[V_Y];
  st->n_corr = (int)(S[V_N] + 0.5);
}

// ONE: `trace` is a single record (the resident kernel's LDS slot: a wavefront other than the stepping one copies it to the buffer, lm_persist_kernel)
template <int DOF, bool ONE = false>
ROLO_DEV void trace_push(LmState* __restrict__ st, rolo_trace_rec* __restrict__ trace, int stage, int accepted, double yi, double rho) {
  if (!trace || st->trace_count >= TRACE_CAP) { st->trace_count++; return; }
  rolo_trace_rec& r = trace[ONE ? 0 : st->trace_count];
  st->trace_count++;
  r.stage = stage; r.outer = st->outer; r.trial = st->trial; r.accepted = accepted;
  r.y0 = st->y0; r.yi = yi; r.rho = rho; r.lambda = st->lambda;
  double dn = 0;
#pragma unroll
  for (int i = 0; i < DOF; i++) dn += st->d[i] * st->d[i];
  r.dnorm = sqrt(dn);
}

// The damping update lambda * max(1/3, 1 - pow(2 rho - 1, 3)) (lsq_registration_impl.hpp:129, 262, 318): std::pow(double, int) is glibc's pow, whose result is the
// correctly rounded cube but for arguments within ~0.02 ulp of a rounding tie; q * q * q rounds twice and lands one ulp off for about a quarter of all q. The cube here
// carries the rounding errors of both products along (two-product through fma) and rounds once: lambda — and with it every later step of the stage — follows the CPU's bits.
ROLO_DEV double cube_rn(double q) {
  const double p = q * q, ep = fma(q, q, -p);        // q^2 = p + ep exactly
  const double r = p * q, er = fma(p, q, -r);        // p q = r + er exactly
  return r + (er + ep * q);
}
ROLO_DEV double lm_lambda_after_accept(double lambda, double rho) { return lambda * fmax(1.0 / 3.0, 1 - cube_rn(2 * rho - 1)); }

// ---- rotation / 6-dof stage -------------------------------------------------------------------------------
ROLO_DEV bool delta_converged(const LmState* __restrict__ st, bool rot_only) {  // lsq_registration_impl.hpp:182-191 / :328-335
  const double ir = st->inv_rot_eps;   // (1.0 / eps) * |.| as the reference writes it; the quotient is formed once per frame (rot_begin_dev)
  double rmax = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) rmax = fmax(rmax, ir * fabs(st->delta_R[i] - ((i % 4 == 0) ? 1.0 : 0.0)));
  if (rot_only) return rmax < 1;
  const double it = st->inv_trans_eps;
  double tmax = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) tmax = fmax(tmax, it * fabs(st->delta_t[i]));
  return fmax(rmax, tmax) < 1;
}

template <int DOF>
ROLO_DEV void rot_compute_step(LmState* __restrict__ st) {
  if constexpr (DOF == 3) {
    ldlt_solve<3>(st->H, st->lambda, st->b, st->d);
    st->d[3] = st->d[4] = st->d[5] = 0;
    so3_exp_R(st->d, st->delta_R);
    st->delta_t[0] = st->delta_t[1] = st->delta_t[2] = 0;
  } else {
    ldlt_solve<6>(st->H, st->lambda, st->b, st->d);
    se3_exp_Rt(st->d, st->delta_R, st->delta_t);
  }
  // xi = delta * x0
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) st->xt_R[i * 3 + j] = st->delta_R[i * 3] * st->x0_R[j] + st->delta_R[i * 3 + 1] * st->x0_R[3 + j] + st->delta_R[i * 3 + 2] * st->x0_R[6 + j];
    st->xt_t[i] = st->delta_R[i * 3] * st->x0_t[0] + st->delta_R[i * 3 + 1] * st->x0_t[1] + st->delta_R[i * 3 + 2] * st->x0_t[2] + st->delta_t[i];
  }
  lm_set_rrt(st->xt_S, st->xt_R);
}

template <int DOF>
ROLO_DEV void rot_begin_outer(LmState* __restrict__ st) {
  if (st->optimizer == ROLO_OPT_GN) st->lambda = 0.0;
  else if (st->lambda < 0.0) {
    double m = 0;
#pragma unroll
    for (int i = 0; i < DOF; i++) m = fmax(m, fabs(st->H[i * 7]));
    st->lambda = st->lm_init * m;
  }
  st->nu = 2.0; st->trial = 0;
  rot_compute_step<DOF>(st);
}

ROLO_DEV void trans_compute_step(LmState* st) {
  ldlt_solve<6>(st->H, st->lambda, st->b, st->d);
  double Rd[9];
  se3_exp_Rt(st->d, Rd, st->delta_t);
  for (int i = 0; i < 3; i++) st->tt[i] = st->delta_t[i] + st->t0[i];
}
ROLO_DEV void trans_begin_outer(LmState* st) {
  if (st->lambda < 0.0) { double m = 0; for (int i = 0; i < 6; i++) m = fmax(m, fabs(st->H[i * 7])); st->lambda = st->lm_init * m; }
  st->nu = 2.0; st->trial = 0;
  trans_compute_step(st);
}
// the per-stage constants of the translation passes (same IEEE quotients every lane used to form for itself)
ROLO_DEV void trans_consts(LmState* st) {
  double lastA[3] = {1.0, 0.0, 0.0}, lastB[3] = {0.0, 0.0, 0.0};
  if (st->q2_intended) for (int i = 0; i < 3; i++) { lastA[i] = st->l[i]; lastB[i] = st->l[i]; }
  for (int i = 0; i < 3; i++) { st->lastA_q[i] = lastA[i] / st->dtn1; st->lastB_q[i] = lastB[i] / st->dtn1; }
  st->inv_dtn = 1.0 / st->dtn;
}
ROLO_DEV void trans_start(LmState* st) {  // lsq_registration_impl.hpp:55-61
  trans_consts(st);
  st->stage = 2; st->phase = 0; st->lin_skip = 0; st->outer = 0; st->trial = 0; st->lambda = -1.0;
  st->trans_done = 0; st->trans_failed = 0; st->trans_outer = 0; st->trans_passes = 0; st->trans_cost_only = 0;
  for (int i = 0; i < 3; i++) st->tt[i] = st->t0[i];
  // lambda_/pt_size: float / size_t -> float (rot_vgicp.hpp:124, rot_vgicp_impl.hpp:557)
  st->lam_over_n = (double)(st->ct_lambda / (float)st->tr_n_corr);
  if (st->tr_n_corr <= 0) { st->error = ROLO_ENOCORR; st->trans_failed = 1; st->trans_done = 1; st->stage = 0; }
}

ROLO_DEV void rot_finish(LmState* st, bool converged, bool failed) {
  st->rot_done = 1; st->rot_converged = converged ? 1 : 0; st->rot_failed = failed ? 1 : 0;
  // outer iterations STARTED = nr_iterations_ + 1 (lsq_registration_impl.hpp:163): `outer` counts the completed ones, and a step that returns false ("lm not converged!!",
  // :166-169) leaves the loop inside an iteration it never completes
  st->rot_outer = st->outer + (failed && !st->error ? 1 : 0); st->rot_ncorr = st->tr_n_corr;
  if (st->run_trans && !st->error) trans_start(st);
  else st->stage = 0;
}

template <int DOF, bool ONE = false>
ROLO_DEV void rot_step_t(LmState* __restrict__ st, const double* __restrict__ S, rolo_trace_rec* __restrict__ trace) {
  constexpr int dof = DOF;
  st->rot_passes++;
  if (st->phase == 1 && st->lin_skip) st->rot_cost_only++;
  if (st->phase == 0) {
    // max_iterations <= 0: the loop of computeTransformation (:161) never runs — no linearisation, no correspondences, the guess comes back (this first pass was evaluated for nothing)
    if (st->outer == 0 && st->fixed_iterations <= 0 && st->max_iterations <= 0) { st->tr_n_corr = 0; st->n_corr = 0; rot_finish(st, false, false); return; }
    for (int i = 0; i < 9; i++) st->x0_R[i] = st->xt_R[i];
    for (int i = 0; i < 6; i++) st->x0_S[i] = st->xt_S[i];
    for (int i = 0; i < 3; i++) st->x0_t[i] = st->xt_t[i];
    unpack_hb<DOF>(st, S);
    st->tr_cur = st->cur; st->tr_n_corr = st->n_corr;
    for (int i = 0; i < 9; i++) st->tr_R[i] = st->x0_R[i];
    for (int i = 0; i < 6; i++) st->tr_S[i] = st->x0_S[i];
    if (st->n_corr <= 0) { st->error = ROLO_ENOCORR; rot_finish(st, false, true); return; }
    st->phase = 1; st->lin_skip = 0;
    rot_begin_outer<DOF>(st);
    // lm_max_iterations <= 0: the trial loop (:233, :287) has no iteration — the step returns false right after its linearisation (Gauss-Newton has no such loop)
    if (st->optimizer != ROLO_OPT_GN && st->lm_max <= 0) rot_finish(st, false, true);
    return;
  }
  const double yi = S[V_YI];
  double den = 0;
#pragma unroll
  for (int i = 0; i < dof; i++) den += st->d[i] * (st->lambda * st->d[i] - st->b[i]);
  const double rho = (st->y0 - yi) / den;
  const bool gn = st->optimizer == ROLO_OPT_GN;
  if (!gn && rho < 0) {
    if (delta_converged(st, dof == 3)) {  // returns true without moving x0
      trace_push<DOF, ONE>(st, trace, 0, 2, yi, rho);
      st->outer++;
      const bool done = st->fixed_iterations > 0 ? (st->outer >= st->fixed_iterations) : true;
      if (done) { rot_finish(st, true, false); return; }
      // the reference re-linearises at the same x0: identical H, b, y0, correspondences — and, lambda staying what this trial's step was solved with, the identical
      // step: d, delta and xt are already what rot_begin_outer's LDLT + exponential would write again, bit for bit (the scalar step is a third of this launch)
      st->nu = 2.0; st->trial = 0;
      st->lin_skip = st->spec_lin;   // a rejected trial: the next pass evaluates its trial's cost alone (LmState::lin_skip)
      return;
    }
    trace_push<DOF, ONE>(st, trace, 0, 0, yi, rho);
    st->lambda = st->nu * st->lambda; st->nu = 2 * st->nu;
    st->trial++;
    if (st->trial >= st->lm_max) { rot_finish(st, false, true); return; }  // "lm not converged!!"
    rot_compute_step<DOF>(st);
    st->lin_skip = st->spec_lin;
    return;
  }
  trace_push<DOF, ONE>(st, trace, 0, 1, gn ? NAN : yi, gn ? NAN : rho);
  for (int i = 0; i < 9; i++) st->x0_R[i] = st->xt_R[i];
  for (int i = 0; i < 6; i++) st->x0_S[i] = st->xt_S[i];
  for (int i = 0; i < 3; i++) st->x0_t[i] = st->xt_t[i];
  if (!gn) st->lambda = lm_lambda_after_accept(st->lambda, rho);
  if (dof == 6) for (int i = 0; i < 36; i++) st->final_H[i] = st->H[i];  // final_hessian_ = H
  st->outer++;
  const bool conv = delta_converged(st, false);
  const bool done = st->fixed_iterations > 0 ? (st->outer >= st->fixed_iterations) : (conv || st->outer >= st->max_iterations);
  if (done) { rot_finish(st, conv, false); return; }
  if (st->lin_skip) {   // the accepted trial's pass carried no linearisation (the bet after a rejection was "rejected again"): the next pass is so3_linearize(x0_new) alone —
    st->cur ^= 1; st->phase = 0; st->lin_skip = 0;   // phase 0 = "linearise at xt (= x0 now) into corr[cur], then begin the outer iteration": the state after it is the one below
    return;
  }
  // next outer iteration: the (B) half of this pass IS so3_linearize(x0_new)
  unpack_hb<DOF>(st, S);
  st->cur ^= 1;
  st->tr_cur = st->cur; st->tr_n_corr = st->n_corr;
  for (int i = 0; i < 9; i++) st->tr_R[i] = st->x0_R[i];
  for (int i = 0; i < 6; i++) st->tr_S[i] = st->x0_S[i];
  if (st->n_corr <= 0) { st->error = ROLO_ENOCORR; rot_finish(st, false, true); return; }
  rot_begin_outer<DOF>(st);
}

ROLO_DEV void rot_step(LmState* __restrict__ st, const double* __restrict__ S, rolo_trace_rec* __restrict__ trace) {
  if (st->optimizer == ROLO_OPT_SO3_LM) rot_step_t<3>(st, S, trace);
  else rot_step_t<6>(st, S, trace);
}

ROLO_DEV bool t_converged(const LmState* __restrict__ st) {  // :142-148
  const double it = st->inv_trans_eps;
  double m = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) m = fmax(m, it * fabs(st->delta_t[i]));
  return m < 1;
}
ROLO_DEV void trans_finish(LmState* st, bool failed) {
  st->trans_done = 1; st->trans_failed = failed ? 1 : 0; st->trans_outer = st->outer + (failed ? 1 : 0) /* iterations started: rot_finish */; st->stage = 0;
}

template <bool ONE = false>
ROLO_DEV void trans_step(LmState* __restrict__ st, const double* __restrict__ S, rolo_trace_rec* __restrict__ trace) {
  st->trans_passes++;
  if (st->phase == 1 && st->lin_skip) st->trans_cost_only++;
  if (st->phase == 0) {
    if (st->outer == 0 && st->max_iterations <= 0) { trans_finish(st, false); return; }   // the loop of computeTranslation (:63) never runs: the start value comes back
    const int keep = st->n_corr;
    unpack_hb<6>(st, S);
    st->n_corr = keep;
    st->phase = 1; st->lin_skip = 0;
    trans_begin_outer(st);
    if (st->lm_max <= 0) trans_finish(st, true);   // no trial at all (:98): "lm not converged!!" right after the linearisation
    return;
  }
  const double yi = S[V_YI];
  double den = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) den += st->d[i] * (st->lambda * st->d[i] - st->b[i]);
  const double rho = (st->y0 - yi) / den;
  if (rho < 0) {
    if (t_converged(st)) { trace_push<6, ONE>(st, trace, 1, 2, yi, rho); st->outer++; trans_finish(st, false); return; }
    trace_push<6, ONE>(st, trace, 1, 0, yi, rho);
    st->lambda = st->nu * st->lambda; st->nu = 2 * st->nu;
    st->trial++;
    if (st->trial >= st->lm_max) { trans_finish(st, true); return; }
    trans_compute_step(st);
    st->lin_skip = st->spec_lin;   // a rejected trial: the next pass evaluates its trial's cost alone
    return;
  }
  trace_push<6, ONE>(st, trace, 1, 1, yi, rho);
  for (int i = 0; i < 3; i++) st->t0[i] = st->tt[i];
  st->lambda = lm_lambda_after_accept(st->lambda, rho);
  st->outer++;
  const bool conv = t_converged(st);
  if (conv || st->outer >= st->max_iterations) { trans_finish(st, false); return; }
  if (st->lin_skip) { st->phase = 0; st->lin_skip = 0; return; }   // accepted after a cost-only pass: the next pass is t3_linearize at tt (= t0 now) alone
  const int keep = st->n_corr;
  unpack_hb<6>(st, S);
  st->n_corr = keep;
  trans_begin_outer(st);
}

// One workgroup: sums the partial rows in a fixed order (or takes the all-reduced sums on the multi-GPU path) and
// runs the scalar LM step on the device. Measured alternatives that did NOT pay on MI355X: running
// this in the last workgroup of the pass (arrival ticket + agent-scope release: 26.2 us per trial vs 11.7 + 9.9), and
// running it redundantly in the prologue of the next pass (every workgroup re-reduces the rows: faster alone,
// slower when four contexts share the GPU).
#ifdef ROLO_CTRL_STATS
__device__ unsigned long long g_ctrl_t[8];   // accumulated shader-clock ticks per phase of ctrl_body + [7] = launches counted
#define CT_STAMP(k) const long long ct##k = clock64()
#else
#define CT_STAMP(k)
#endif
// pub: pinned host copy of the state, written by the LAST controller launch of a frame's schedule whether or not it has a step to take
// (replaces a device-to-host copy launch per frame)
// MODE: -1 = decided at run time from `stage` and the state's optimizer (batched launches); 0 = rotation stage, SO(3) optimiser; 1 = rotation
// stage, 6-dof optimisers; 2 = translation stage. The specialised instances carry ONE step function instead of four: the generic kernel
// is ~11 k instructions of which a launch executes ~700 along a branchy path — on a cold instruction cache every taken branch is a fetch.
template <bool PEER = false, int MODE = -1>
ROLO_DEV void ctrl_body(LmState* st, const double* __restrict__ partials, int nblocks, const double* __restrict__ sums_in,
                        rolo_trace_rec* trace, int stage, LmState* pub = nullptr, const PeerArgs* peer = nullptr) {
  __shared__ double sums[NV_MAX];
  __shared__ double part[8][NV_MAX];
  // the scalar LM step touches ~150 fields: stage the whole state through LDS (one coalesced read, one write)
  // instead of paying a global-memory round trip per field from a single lane
  __shared__ LmState sst;
  static_assert(sizeof(LmState) % sizeof(int) == 0, "LmState must be int-copyable");
  constexpr int NW = sizeof(LmState) / sizeof(int);
  constexpr int NWT = (NW + 255) / 256;
  constexpr int INFLIGHT = 64;
  ROLO_SHORT_KERNEL_PRIO();
  CT_STAMP(0);
  const int v = threadIdx.x & 31, q = threadIdx.x >> 5;  // 256 threads = 8 strided groups of 32 values
  // Everything this controller reads was written by other CUs a moment ago: every load is a ~1-2 us L2 / fabric round
  // trip. Issue ALL of them before the first use — the state copy and 64 partial rows per thread (the 512 rows of a
  // 131k-point cloud) — so the launch pays one round trip, not "stage flag, then state, then rows" (three).
  int sreg[NWT];
  {
    const int* g = reinterpret_cast<const int*>(st);
#pragma unroll
    for (int k = 0; k < NWT; k++) { const int i = threadIdx.x + 256 * k; sreg[k] = i < NW ? g[i] : 0; }
  }
  double r[INFLIGHT];
  if (partials) {
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) {   // always-valid address + select: `cond ? load : 0` is an exec-masked branch per load, 64 of them in front of the data
      const int b = q + 8 * u; const double x = partials[(size_t)min(b, nblocks - 1) * NV_MAX + v]; r[u] = (b < nblocks) ? x : 0.0;
    }
  }
  {
    int* l = reinterpret_cast<int*>(&sst);
#pragma unroll
    for (int k = 0; k < NWT; k++) { const int i = threadIdx.x + 256 * k; if (i < NW) l[i] = sreg[k]; }
  }
  __syncthreads();
  if (sst.stage != stage) {  // predicated launch: nothing to do (uniform)
    if (pub) { int* h = reinterpret_cast<int*>(pub); const int* l = reinterpret_cast<const int*>(&sst); for (int i = threadIdx.x; i < NW; i += 256) h[i] = l[i]; }
    return;
  }
  CT_STAMP(1);
  if (partials) {
    // fixed combination order => deterministic for a given grid
    double s0 = 0;
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) s0 += r[u];
    for (int b0 = q + 8 * INFLIGHT; b0 < nblocks; b0 += 8 * INFLIGHT) {   // clouds above 131k points
#pragma unroll
      for (int u = 0; u < INFLIGHT; u++) { const int b = b0 + 8 * u; const double x = partials[(size_t)min(b, nblocks - 1) * NV_MAX + v]; r[u] = (b < nblocks) ? x : 0.0; }
#pragma unroll
      for (int u = 0; u < INFLIGHT; u++) s0 += r[u];
    }
    part[q][v] = s0;
    __syncthreads();
    if (threadIdx.x < NV_MAX) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) t += part[k][threadIdx.x];
      sums[threadIdx.x] = t;
    }
  } else if (threadIdx.x < NV_MAX) {
    sums[threadIdx.x] = sums_in[threadIdx.x];
  }
  __syncthreads();
  bool peer_ok = true;
  if constexpr (PEER) {
    // multi-GPU: this rank's sums cover its shard of the source points — exchange them through the peers' mailboxes and add all ranks'
    // values in rank order (peer_dev.hpp); every rank then runs the identical scalar step below
    __shared__ unsigned xw[PEER_MAX * PEER_SLOT_WORDS];
    __shared__ int bad;
    peer_ok = peer_allreduce_block<256>(sums, xw, &bad, *peer);
  }
  CT_STAMP(2);
  if (threadIdx.x == 0) {
    // (the step on a private copy of the whole state instead of LDS: 256 VGPRs + 396 B of scratch, 2.2 -> 7.1 us)
    if (!peer_ok) {   // a peer did not answer within the timeout: end the registration with ROLO_ECOMM instead of waiting forever
      sst.error = ROLO_ECOMM; sst.stage = 0; sst.rot_done = 1; sst.rot_failed = 1; sst.trans_done = 1; sst.trans_failed = 1;
    } else if constexpr (MODE == 0) rot_step_t<3>(&sst, sums, trace);
    else if constexpr (MODE == 1) rot_step_t<6>(&sst, sums, trace);
    else if constexpr (MODE == 2) trans_step(&sst, sums, trace);
    else if (stage == 1) rot_step(&sst, sums, trace);
    else trans_step(&sst, sums, trace);
  }
  __syncthreads();
  CT_STAMP(3);
  {
    int* g = reinterpret_cast<int*>(st);
    const int* l = reinterpret_cast<const int*>(&sst);
    for (int i = threadIdx.x; i < NW; i += 256) g[i] = l[i];
    if (pub) { int* h = reinterpret_cast<int*>(pub); for (int i = threadIdx.x; i < NW; i += 256) h[i] = l[i]; }
  }
#ifdef ROLO_CTRL_STATS
  if (threadIdx.x == 0) {   // phases of a launch that had a step to take: loads issued -> first use | row sums | scalar step | write-back
    const long long ct4 = clock64();
    atomicAdd(&g_ctrl_t[0], (unsigned long long)(ct1 - ct0)); atomicAdd(&g_ctrl_t[1], (unsigned long long)(ct2 - ct1));
    atomicAdd(&g_ctrl_t[2], (unsigned long long)(ct3 - ct2)); atomicAdd(&g_ctrl_t[3], (unsigned long long)(ct4 - ct3)); atomicAdd(&g_ctrl_t[7], 1ull);
  }
#endif
}
#ifdef ROLO_CTRL_STATS
extern "C" int rolo_debug_ctrl_times(unsigned long long* out8) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ctrl_t), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif

template <int MODE>
__global__ __launch_bounds__(256) void ctrl_kernel(LmState* st, const double* __restrict__ partials, int nblocks,
                                                  const double* __restrict__ sums_in, rolo_trace_rec* trace, int stage, LmState* pub) {
  ctrl_body<false, MODE>(st, partials, nblocks, sums_in, trace, stage, pub);
}
template <int MODE>
__global__ __launch_bounds__(256) void ctrl_peer_kernel(LmState* st, const double* __restrict__ partials, int nblocks, rolo_trace_rec* trace, int stage,
                                                       LmState* pub, PeerArgs peer) {
  ctrl_body<true, MODE>(st, partials, nblocks, nullptr, trace, stage, pub, &peer);
}
__global__ __launch_bounds__(256) void ctrl_batch_kernel(const BatchSlot* __restrict__ slots, int stage) {
  const BatchSlot& S = slots[blockIdx.x];
  ctrl_body(S.st, S.a.partials, S.grid, nullptr, S.trace, stage);
}

// ---- one launch per LM trial: controller in the prologue of the pass ---------------------------------------------------------------
// Every workgroup of launch j first finishes trial j-1 — sums the partial rows launch j-1 wrote (fixed order: deterministic, and the
// same bits in every workgroup) and runs the scalar LM step on a copy of the state in LDS — then evaluates the pass the new state asks
// for (rotation / 6-dof or translation stage) over its points and writes ITS row. Workgroup 0 also writes the new state back (and the
// LM trace). State and rows are double-buffered between consecutive launches (a workgroup of launch j may still be reading what a
// faster one would overwrite). Versus pass + ctrl_kernel launches this halves the launches of a solve; versus the first attempt at
// this fusion (every one of 512 workgroups re-reading 512 rows of 256 B) the rows are compact (12 or 30 doubles), there
// are n / THREADS of them with fat workgroups, and the point / covariance loads of the pass are issued before the prologue so that
// they overlap the row reads and the scalar step. do_body = 0: the closing launch of a schedule (one workgroup, no pass).
template <int COLS, int THREADS>
ROLO_DEV void reduce_rows_compact(const double* __restrict__ rows, int nrows, int nv, double* __restrict__ part /* THREADS */, double* __restrict__ sums /* NV_MAX */) {
  constexpr int GROUPS = THREADS / COLS;
  constexpr int INFLIGHT = 16;
  const int v = threadIdx.x % COLS, q = threadIdx.x / COLS;
  double s0 = 0;
  for (int b0 = q; b0 < nrows; b0 += GROUPS * INFLIGHT) {
    double r[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) { const int b = b0 + GROUPS * u; const double x = rows[(size_t)min(b, nrows - 1) * NV_MAX + v]; r[u] = (b < nrows && v < nv) ? x : 0.0; }
#pragma unroll
    for (int u = 0; u < INFLIGHT; u++) s0 += r[u];
  }
  part[q * COLS + v] = s0;
  __syncthreads();
  if (threadIdx.x < COLS) {
    double t = 0;
#pragma unroll 8
    for (int k = 0; k < GROUPS; k++) t += part[k * COLS + threadIdx.x];
    sums[threadIdx.x] = t;
  }
  __syncthreads();
}

template <int DOF, int THREADS>
__global__ __launch_bounds__(THREADS) void lm_kernel(PassArgs a, const LmState* st_in, LmState* st_out /* the closing launch of an even chunk passes st_in == st_out: no __restrict__ */,
                                                    const double* __restrict__ rows_in, double* __restrict__ rows_out, int nrows,
                                                    rolo_trace_rec* trace, int do_body, int ppt, LmState* pub) {
  __shared__ LmState sst;
  __shared__ double part[THREADS];
  __shared__ double csum[NV_MAX];   // compact: yi, y, n, H (lower triangle), b
  __shared__ double sums[NV_MAX];   // V_* layout of the scalar step
  static_assert(sizeof(LmState) % sizeof(int) == 0, "LmState must be int-copyable");
  constexpr int NW = sizeof(LmState) / sizeof(int);
  constexpr int NWT = (NW + THREADS - 1) / THREADS;
  constexpr int NHR = DOF * (DOF + 1) / 2, NVR = 3 + NHR + DOF;
  int sreg[NWT];
  {
    const int* g = reinterpret_cast<const int*>(st_in);
#pragma unroll
    for (int k = 0; k < NWT; k++) { const int w = threadIdx.x + THREADS * k; sreg[k] = w < NW ? g[w] : 0; }
  }
  // this thread's first point: independent of the state, in flight while the prologue runs. A workgroup owns ppt consecutive
  // slabs of THREADS points (fewer, fatter workgroups leave the rest of the chip to the other contexts' kernels).
  const int i0 = a.begin + (int)blockIdx.x * ppt * THREADS + (int)threadIdx.x;
  const bool valid0 = do_body && i0 < a.end;
  PtIn in{};
  if (valid0) in = load_pt(a, i0);
  {
    int* l = reinterpret_cast<int*>(&sst);
#pragma unroll
    for (int k = 0; k < NWT; k++) { const int w = threadIdx.x + THREADS * k; if (w < NW) l[w] = sreg[k]; }
  }
  __syncthreads();
  if (sst.stage != 0 && sst.pending) {
    const int stage = sst.stage;
    const int nv = stage == 1 ? NVR : 30;
    if (nv <= 16) reduce_rows_compact<16, THREADS>(rows_in, nrows, nv, part, csum);
    else reduce_rows_compact<32, THREADS>(rows_in, nrows, nv, part, csum);
    if (threadIdx.x < NV_MAX) {
      // compact -> V_* slots
      const int nh = stage == 1 ? NHR : 21;
      const int t = threadIdx.x;
      if (t < nv) { const int sl = t < 3 ? t : (t < 3 + nh ? V_H + (t - 3) : V_B + (t - 3 - nh)); sums[sl] = csum[t]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      rolo_trace_rec* tr = blockIdx.x == 0 ? trace : nullptr;
      if (stage == 1) rot_step(&sst, sums, tr); else trans_step(&sst, sums, tr);   // without a buffer trace_count still advances
    }
    __syncthreads();
  }
  const int stage = sst.stage;
  const bool body = do_body && stage != 0;
  if (threadIdx.x == 0) sst.pending = body ? 1 : 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    int* g = reinterpret_cast<int*>(st_out);
    const int* l = reinterpret_cast<const int*>(&sst);
    for (int w = threadIdx.x; w < NW; w += THREADS) g[w] = l[w];
    if (pub) { int* h = reinterpret_cast<int*>(pub); for (int w = threadIdx.x; w < NW; w += THREADS) h[w] = l[w]; }   // pinned host copy (closing launch)
  }
  if (!body) return;
  double* out_row = rows_out + (size_t)blockIdx.x * NV_MAX;
  if (stage == 1) {
    double acc[NVR];
    int slot[NVR];
#pragma unroll
    for (int v = 0; v < NVR; v++) { acc[v] = 0.0; slot[v] = v; }
    rot_pass_compute<DOF>(a, &sst, i0, valid0, in, acc);
    for (int p = 1; p < ppt; p++) {
      const int i = i0 + p * THREADS;
      const bool valid = i < a.end;
      if (valid) in = load_pt(a, i);
      rot_pass_compute<DOF>(a, &sst, i, valid, in, acc);
    }
    block_reduce_store<NVR, THREADS>(acc, slot, out_row, sst.phase == 1 && sst.lin_skip != 0);
  } else {
    double acc[30];
    int slot[30];
#pragma unroll
    for (int v = 0; v < 30; v++) { acc[v] = 0.0; slot[v] = v; }
    trans_pass_compute(a, &sst, i0, valid0, in, acc);
    for (int p = 1; p < ppt; p++) {
      const int i = i0 + p * THREADS;
      const bool valid = i < a.end;
      if (valid) in = load_pt(a, i);
      trans_pass_compute(a, &sst, i, valid, in, acc);
    }
    block_reduce_store<30, THREADS>(acc, slot, out_row, sst.phase == 1 && sst.lin_skip != 0);
  }
}

// ---- ONE launch per frame: the resident LM kernel (round 6; rolo_params::fused_lm = 2) --------------------------------------------------------------------
// A frame's LM chain is ~30 trials = ~60 launches of pass + controller: 60 kernel boundaries, each a dispatch, a cold first fetch and an L2 write-back / invalidate
// that every OTHER kernel on the chip feels (profiles/r06/concurrency.md). Here the whole chain — both stages — is one launch of G <= 128 workgroups that stay
// resident: per trial every workgroup evaluates the pass over ITS points (the same thread owns the same points in every trial, so the correspondence cache never
// crosses a workgroup), leaves its row of 1 / 12 / 30 sums in an exchange buffer as self-validating words {epoch : 32 | half a double : 32} — agent-scope relaxed
// atomic stores, the protocol of the multi-GPU peer exchange (peer_dev.hpp) between workgroups instead of ranks — polls everybody's rows (agent-scope loads: they
// miss the XCD's L2), adds them in row order (the same bits in every workgroup) and runs the scalar LM step on its own copy of the state in LDS. No fence, no
// barrier object, no atomic read-modify-write: nothing but the words crosses. Rows are double-buffered by the epoch's parity (a workgroup can be at most one
// exchange ahead of another); the epoch continues from launch to launch through the buffer's header, so a stale word never matches.
// A poll that lasts longer than timeout_ticks (wall clock, 100 MHz) gives up: the stage ends with ROLO_ECOMM in the state — a workgroup that never became
// resident costs a bounded wait, never a hung GPU. Co-residency: G x THREADS threads with ~30 KB of LDS per workgroup fit the chip several times over (256 CUs),
// and every kernel that can occupy the slots in the meantime terminates by itself.
// ADMISSION. Spinning workgroups never give their slots back, so two resident kernels that each got only part of their workgroups onto the chip — two processes sharing
// the GPU, more contexts in flight than the sizing assumed — would wait for each other forever. Before the first trial the workgroups therefore exchange one EMPTY row under
// a short timeout (admit_ticks, default 1 ms: longer than any kernel of this library that may hold the slots in the meantime): if every row arrives, all G workgroups are
// resident and will stay so; if not, the workgroup that ran out of time raises the bail word of this launch, everybody — whoever is polling now, whoever becomes resident
// later — sees it and leaves, the LM state is untouched (LmState::lmp_bailed = 1 apart), and the host finishes the frame with pass + controller launches. A lost race costs
// the admission time, never a hung GPU and never an error.
constexpr int LMP_HDR = 8;   // words in front of the rows: [0] = last epoch used, [2] = bail word (the admission epoch of the launch that gave up)
#ifdef ROLO_LMP_STATS
// the phases of a trial as workgroup 0 sees them (wall clock, 100 MHz ticks, summed): [0] pass body + block reduction, [1] exchange (publish, poll, row sums), [2] the scalar step, [7] trials
__device__ unsigned long long g_lmp_t[8];
__device__ unsigned long long g_lmp_adm[4];   // admission as workgroup 0 sees it: [0] ticks from its start to everybody admitted, [1] launches, [2] the longest one, [3] ticks from its start to the kernel's end
extern "C" int rolo_debug_lmp_admission(unsigned long long* out4, int reset) {
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_lmp_adm), sizeof(unsigned long long) * 4) != hipSuccess) return -1;
  if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lmp_adm), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int rolo_debug_lmp_times(unsigned long long* out8, int reset) {
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_lmp_t), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lmp_t), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define LMP_STAMP(k) const long long lt##k = wall_clock64()
#define LMP_ACC() do { if (wg == 0 && t == 0) { const long long lt3 = wall_clock64(); atomicAdd(&g_lmp_t[0], (unsigned long long)(lt1 - lt0)); atomicAdd(&g_lmp_t[1], (unsigned long long)(lt2 - lt1)); \
                       atomicAdd(&g_lmp_t[2], (unsigned long long)(lt3 - lt2)); atomicAdd(&g_lmp_t[7], 1ull); \
                       if (only_first) { atomicAdd(&g_lmp_t[3], (unsigned long long)(lt1 - lt0)); atomicAdd(&g_lmp_t[4], (unsigned long long)(lt2 - lt1)); \
                                         atomicAdd(&g_lmp_t[5], (unsigned long long)(lt3 - lt2)); atomicAdd(&g_lmp_t[6], 1ull); } } } while (0)   /* [3..6]: the cost-only trials among them */
#else
#define LMP_STAMP(k)
#define LMP_ACC()
#endif
// the scalar step as a CALL: one lane runs ~700 dependent instructions with its own register needs (the LDLT's factors) — inlined into the resident kernel's loop they
// were live across the pass bodies and spilled (620 bytes of scratch per lane)
// (state and sums live in LDS: the round trip through an address_space(3) pointer tells the compiler so — a call boundary hides it, and generic pointers would make every
// one of the step's ~150 state accesses a flat instruction)
// (round 6, second form: state and sums are FILE-scope LDS variables, so that the step — a function of its own — names them directly: every access is a ds_ instruction
// with a constant address. As pointer arguments they arrived generic, and the cast back to LDS cost a null check of three instructions per access — ~1 300 of the
// step's 4 300 instructions, all issued by one lane at four cycles each)
__shared__ LmState lmp_sst;
__shared__ double lmp_sums[NV_MAX];
// (a trial's trace record goes to an LDS slot: written to the buffer by the stepping lane, the store's acknowledgement — the function returns behind s_waitcnt vmcnt(0) — was
// ~0.4 us of workgroup 0's step, for which all the others then wait in the next exchange)
__shared__ rolo_trace_rec lmp_trace_slot;
template <int DOF> __device__ __noinline__ void lmp_rot_step(bool want_trace) { rot_step_t<DOF, true>(&lmp_sst, lmp_sums, want_trace ? &lmp_trace_slot : nullptr); }
__device__ __noinline__ void lmp_trans_step(bool want_trace) { trans_step<true>(&lmp_sst, lmp_sums, want_trace ? &lmp_trace_slot : nullptr); }
// after the barrier behind a step: the last wavefront of workgroup 0 copies the record the step left (if it left one) — plain stores nobody waits for before the kernel ends
template <int THREADS>
ROLO_DEV void lmp_trace_flush(rolo_trace_rec* __restrict__ trace, int count_before) {
  static_assert(sizeof(rolo_trace_rec) % sizeof(int) == 0, "record copied in dwords");
  constexpr int NWORD = sizeof(rolo_trace_rec) / sizeof(int);
  const int t = (int)threadIdx.x - (THREADS - 64);
  if (t < 0 || t >= NWORD) return;
  const int count = lmp_sst.trace_count;
  if (count == count_before || count_before >= TRACE_CAP) return;
  reinterpret_cast<int*>(trace + count_before)[t] = reinterpret_cast<const int*>(&lmp_trace_slot)[t];
}

// publish row[0 .. nv) as words of epoch e, collect all G rows, add them in a fixed order into sums[] (V_* slots). Returns false if a row did not arrive in time.
#ifndef ROLO_LMP_SPIN_PRIO
#define ROLO_LMP_SPIN_PRIO 1   // issue priority while a workgroup polls the others' rows (A/B: 0 = the spinning wavefronts step back behind everything else on their SIMD)
#endif
#ifndef ROLO_LMP_SPIN_SLEEP
#define ROLO_LMP_SPIN_SLEEP 1  // s_sleep argument between two polls of a missing word (x 64 clocks)
#endif
template <int THREADS, bool ADMIT = false>
ROLO_DEV bool lmp_exchange(const double* __restrict__ row, int nv, int nh, unsigned e, unsigned long long* __restrict__ xbuf, int G, int wg, unsigned* __restrict__ xw,
                           double (*part)[NV_MAX], double* __restrict__ sums, int* __restrict__ bad, unsigned long long timeout_ticks) {
  const int t = (int)threadIdx.x;
  unsigned long long* rows = xbuf + LMP_HDR + (size_t)(e & 1u) * G * PEER_SLOT_WORDS;
  const int nw = 2 * nv;
  if (t < nw) {
    const double v = row[t >> 1];
    const unsigned half = (t & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
    __hip_atomic_store(rows + (size_t)wg * PEER_SLOT_WORDS + t, ((unsigned long long)e << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int total = G * nw;
  const long long t0 = wall_clock64();
  if (ROLO_LMP_SPIN_PRIO != ROLO_SHORT_PRIO) __builtin_amdgcn_s_setprio(ROLO_LMP_SPIN_PRIO);
  for (int base = t; base < total; base += THREADS * 4) {
    unsigned long long x[4]; const unsigned long long* q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = min(base + u * THREADS, total - 1);
      const int r = i / nw, w = i - r * nw;
      q[u] = rows + (size_t)r * PEER_SLOT_WORDS + w;
      x[u] = __hip_atomic_load(q[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int i = base + u * THREADS;
      if (i >= total) continue;
      while ((unsigned)(x[u] >> 32) != e) {
        if ((unsigned long long)(wall_clock64() - t0) > timeout_ticks) {
          *bad = 1;
          if (ADMIT) __hip_atomic_store(xbuf + 2, (unsigned long long)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // this launch gives up: tell everybody
          break;
        }
        if (ADMIT && (unsigned)__hip_atomic_load(xbuf + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e) { *bad = 1; break; }   // somebody gave up
        __builtin_amdgcn_s_sleep(ADMIT ? 8 : ROLO_LMP_SPIN_SLEEP);
        x[u] = __hip_atomic_load(q[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      xw[i] = (unsigned)x[u];
    }
  }
  if (ROLO_LMP_SPIN_PRIO != ROLO_SHORT_PRIO) __builtin_amdgcn_s_setprio(ROLO_SHORT_PRIO);
  if (ADMIT && t == 0 && (unsigned)__hip_atomic_load(xbuf + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == e) *bad = 1;   // (a workgroup that found every row may still be the last to hear)
  __syncthreads();
  if (*bad) return false;
  if (ADMIT) return true;
  {   // row sums in a fixed order (16 groups of rows, then the groups): the same bits in every workgroup
    const int v = t & 31;
    for (int q = t >> 5; q < 16; q += THREADS / 32) {
      double s0 = 0.0;
      if (v < nv) for (int r = q; r < G; r += 16) s0 += __hiloint2double((int)xw[r * nw + 2 * v + 1], (int)xw[r * nw + 2 * v]);
      part[q][v] = s0;
    }
  }
  __syncthreads();
  if (t < nv) {
    double s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) s1 += part[k][t];
    sums[t < 3 ? t : (t < 3 + nh ? V_H + (t - 3) : V_B + (t - 3 - nh))] = s1;
  }
  __syncthreads();
  return true;
}

// The pass bodies of the resident kernel: PPT points per thread, INTERLEAVED — all the record fetches of half (A) first, the first hash probes of half (B) behind them,
// then (A)'s arithmetic while the probes are in flight, (B)'s record fetches, (B)'s arithmetic: a trial costs two dependent round trips whatever PPT is, where a loop over
// the points costs two per point (one launch per trial with 2 / 4 / 8 points per thread: +70 us of frame latency per doubling, DEAD_ENDS rounds 2-3). A thread owns the same
// points in every trial, so the float coordinates, the covariance (as m) and BOTH correspondence ids of a point stay in registers for the whole chain; the ids are still
// written to corr[] — the getters and a later rolo_compute_translation read them there.
struct PtReg { float x, y, z; Vec3 m; };   // m.x is NaN for a point whose covariance is not of the plane form (or when the pass has no m at all): six entries from memory
template <int PPT>
ROLO_DEV void lmp_load_points(const PassArgs& a, int i0, int threads, PtReg (&pt)[PPT], int (&cid)[PPT][2], bool (&valid)[PPT]) {
  const size_t pitch = (size_t)a.n_total;
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const int i = i0 + p * threads;
    valid[p] = i < a.end;
    pt[p] = PtReg{0.f, 0.f, 0.f, Vec3{__builtin_nan(""), 0.0, 0.0}};
    cid[p][0] = cid[p][1] = -1;
    if (valid[p]) {
      const float4 pf = a.src[i];
      pt[p].x = pf.x; pt[p].y = pf.y; pt[p].z = pf.z;
      if (a.nrm) pt[p].m = Vec3{a.nrm[i], a.nrm[pitch + i], a.nrm[2 * pitch + i]};
      if (a.n_off == 1) { cid[p][0] = a.corr[0][i]; cid[p][1] = a.corr[1][i]; }
    }
  }
}
ROLO_DEV Sym3 lmp_rotated_cov(const PassArgs& a, const double* R, const double* __restrict__ S6, const PtReg& q, int i) {
  if (q.m.x == q.m.x) {
    const Sym3 S{uni(S6[0]), uni(S6[1]), uni(S6[2]), uni(S6[3]), uni(S6[4]), uni(S6[5])};
    const Vec3 r = mat3_mulv(R, q.m);
    return Sym3{S.xx - r.x * r.x, S.xy - r.x * r.y, S.xz - r.x * r.z, S.yy - r.y * r.y, S.yz - r.y * r.z, S.zz - r.z * r.z};
  }
  const size_t pitch = (size_t)a.n_total;
  return sym3_rotate(R, Sym3{a.cov[i], a.cov[pitch + i], a.cov[2 * pitch + i], a.cov[3 * pitch + i], a.cov[4 * pitch + i], a.cov[5 * pitch + i]});
}
// The reference's per-correspondence Mahalanobis cache (rot_vgicp_impl.hpp:204-222: update_correspondences leaves M = (C_B + R C_A R^T)^-1 of the linearisation point with
// every correspondence, compute_error reads it) — in LDS: mc[k * ms + slot], k = the six entries, slot = this thread's p-th point; a thread only ever touches its own slots.
// The (B) half writes it, the (A) half of the trials that follow reads it (m_hit: wave-uniform, lm_persist_kernel) instead of fetching the voxel covariance, rotating C_A and
// inverting again: a cost-only trial is 26 instead of ~100 fp64 instructions per point. As global memory this cache LOST (48 B written + read per point and pass: DEAD_ENDS
// round 5, entry 1); the resident kernel keeps a thread's points for the whole chain, so it costs no traffic here.
typedef __attribute__((address_space(3))) double lds_f64;
ROLO_DEV void mc_store(lds_f64* mc, int ms, int slot, const Sym3& M) {
  mc[slot] = M.xx; mc[ms + slot] = M.xy; mc[2 * ms + slot] = M.xz; mc[3 * ms + slot] = M.yy; mc[4 * ms + slot] = M.yz; mc[5 * ms + slot] = M.zz;
}
ROLO_DEV Sym3 mc_load(const lds_f64* mc, int ms, int slot) { return Sym3{mc[slot], mc[ms + slot], mc[2 * ms + slot], mc[3 * ms + slot], mc[4 * ms + slot], mc[5 * ms + slot]}; }
struct RecMW { Vec3 mean; double w; };
ROLO_DEV RecMW load_rec_mw(const double* __restrict__ rec, int id) {
  const double* r = rec + (size_t)id * REC_DOUBLES;
  return RecMW{Vec3{r[0], r[1], r[2]}, r[9]};
}
// M of every point of this thread for the pose (R, S6 = R R^T) and the correspondences of buffer `buf`, one point after the other: what a (B) half would have left, had
// it not been overwritten by a rejected trial's speculation (once per rejection), and the translation stage's constant matrices (once per stage)
template <int NP>
ROLO_DEV void lmp_fill_cache(const PassArgs& a, const double* R9, const double* S6, int buf, int i0, int threads, const PtReg* pt, const int (*cid)[2], const bool* valid,
                             lds_f64* mc, int ms, int slot0) {
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = uni(R9[k]);
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int v = buf ? cid[p][1] : cid[p][0];
    if (valid[p] && v >= 0) {
      const double* r = a.tab.rec + (size_t)v * REC_DOUBLES;
      const Sym3 cb{r[3], r[4], r[5], r[6], r[7], r[8]};
      mc_store(mc, ms, slot0 + p * threads, sym3_inverse(sym3_add(cb, lmp_rotated_cov(a, R, S6, pt[p], i0 + p * threads))));
    }
    asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
  }
}
template <int DOF, int PPT, bool MC>
ROLO_DEV void lmp_rot_body(const PassArgs& a, const LmState* __restrict__ st, int i0, int threads, const PtReg* pt, int (*cid)[2], const bool* valid,
                           double (&acc)[3 + DOF * (DOF + 1) / 2 + DOF], lds_f64* mc, int ms, int slot0) {
  constexpr int NH = DOF * (DOF + 1) / 2;
  const int phase = uni(st->phase), cur = uni(st->cur);
  const bool skip_lin = phase == 1 && uni(st->lin_skip) != 0;
  const int newb = phase == 0 ? cur : (cur ^ 1);
  double R0[9], R1[9], t1[3];
#pragma unroll
  for (int k = 0; k < 9; k++) { R0[k] = uni(st->x0_R[k]); R1[k] = uni(st->xt_R[k]); }
#pragma unroll
  for (int k = 0; k < 3; k++) t1[k] = uni(st->xt_t[k]);
  Vec3 tp[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    tp[p] = mat3_mulv(R1, Vec3{(double)pt[p].x, (double)pt[p].y, (double)pt[p].z});
    tp[p].x += t1[0]; tp[p].y += t1[1]; tp[p].z += t1[2];
  }
  // (A) compute_error(xi): the records of the cached correspondences — their ids are in registers, so these fetches depend on nothing this trial computed
  // (with the cache — mc, filled for x0 by whoever ran before: lm_persist_kernel — only the mean and the weight of a record are fetched here)
  Rec ra[PPT]; bool ha[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const int v = cur ? cid[p][1] : cid[p][0];
    ha[p] = phase == 1 && valid[p] && v >= 0;
    if (ha[p]) {
      if (MC) { const RecMW l = load_rec_mw(a.tab.rec, v); ra[p].mean = l.mean; ra[p].w = l.w; }
      else ra[p] = load_rec(a.tab.rec, v);
    }
  }
  // (B) first probe of every point's voxel
  unsigned long long key[PPT]; unsigned h[PPT]; ulonglong2 sl[PPT]; bool ok[PPT];
  if (!skip_lin) {
#pragma unroll
    for (int p = 0; p < PPT; p++) {
      int kx, ky, kz;
      voxel_coord_dev(a.tab, tp[p].x, tp[p].y, tp[p].z, kx, ky, kz);
      ok[p] = valid[p] && pack_key(kx, ky, kz, key[p]);
      h[p] = hash_key(key[p]) & a.tab.mask;
      if (ok[p]) sl[p] = *reinterpret_cast<const ulonglong2*>(a.tab.keys + 2 * (size_t)h[p]);
    }
  }
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    if (ha[p]) {
      Sym3 M;
      if constexpr (MC) M = mc_load(mc, ms, slot0 + p * threads);
      else M = sym3_inverse(sym3_add(ra[p].cov, lmp_rotated_cov(a, R0, st->x0_S, pt[p], i0 + p * threads)));
      const Vec3 e{ra[p].mean.x - tp[p].x, ra[p].mean.y - tp[p].y, ra[p].mean.z - tp[p].z};
      acc[0] += ra[p].w * dot3(e, sym3_mulv(M, e));
    }
    if (PPT > 1 && !MC) __builtin_amdgcn_sched_barrier(0);   // the FETCHES of the points are batched, their arithmetic is not: interleaved, four inverses' temporaries are 250 registers
  }
  if (skip_lin) return;
  int vid[PPT];
  int* __restrict__ corr_new = a.corr[newb];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    vid[p] = -1;
    if (ok[p]) {
      while (true) {   // (the table is at most half full: the first probe answers almost always)
        if (sl[p].x == key[p]) { vid[p] = (int)(unsigned)sl[p].y; break; }
        if (sl[p].x == KEY_EMPTY) break;
        h[p] = (h[p] + 1) & a.tab.mask;
        sl[p] = *reinterpret_cast<const ulonglong2*>(a.tab.keys + 2 * (size_t)h[p]);
      }
    }
    if (newb) cid[p][1] = vid[p]; else cid[p][0] = vid[p];
    if (valid[p]) corr_new[i0 + p * threads] = vid[p];
  }
  Rec rb[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) if (vid[p] >= 0) rb[p] = load_rec(a.tab.rec, vid[p]);
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    if (vid[p] >= 0) {
      const Sym3 M = sym3_inverse(sym3_add(rb[p].cov, lmp_rotated_cov(a, R1, st->xt_S, pt[p], i0 + p * threads)));
      if (MC) mc_store(mc, ms, slot0 + p * threads, M);
      const Vec3 e{rb[p].mean.x - tp[p].x, rb[p].mean.y - tp[p].y, rb[p].mean.z - tp[p].z};
      const Vec3 Me = sym3_mulv(M, e);
      acc[1] += rb[p].w * dot3(e, Me);
      acc[2] += 1.0;
      const Vec3 wMe{rb[p].w * Me.x, rb[p].w * Me.y, rb[p].w * Me.z};
      accumulate_hb<DOF>(M, tp[p], rb[p].w, 0.0, wMe, &acc[3], &acc[3 + NH]);
    }
    if (PPT > 1) __builtin_amdgcn_sched_barrier(0);
  }
}
template <int PPT, bool MC>
ROLO_DEV void lmp_trans_body(const PassArgs& a, const LmState* __restrict__ st, int i0, int threads, const PtReg* pt, const int (*cid)[2], const bool* valid, double (&acc)[30],
                             lds_f64* mc, int ms, int slot0) {
  constexpr int NH = 21;
  const int phase = uni(st->phase);
  const bool skip_lin = phase == 1 && uni(st->lin_skip) != 0;
  const int tr_cur = uni(st->tr_cur);
  double R[9];
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = uni(st->tr_R[k]);
  const Vec3 tt{uni(st->tt[0]), uni(st->tt[1]), uni(st->tt[2])};
  const Vec3 g{uni(st->g[0]), uni(st->g[1]), uni(st->g[2])};
  const double lam_n = uni(st->lam_over_n);
  const Vec3 lAq{uni(st->lastA_q[0]), uni(st->lastA_q[1]), uni(st->lastA_q[2])}, lBq{uni(st->lastB_q[0]), uni(st->lastB_q[1]), uni(st->lastB_q[2])};
  const double inv_dtn = uni(st->inv_dtn);
  Rec r[PPT]; bool has[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const int v = tr_cur ? cid[p][1] : cid[p][0];
    has[p] = valid[p] && v >= 0;
    if (has[p]) {
      if (MC) { const RecMW l = load_rec_mw(a.tab.rec, v); r[p].mean = l.mean; r[p].w = l.w; }
      else r[p] = load_rec(a.tab.rec, v);
    }
  }
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    if (!has[p]) continue;
    const Vec3 q{(double)pt[p].x, (double)pt[p].y, (double)pt[p].z};
    const Vec3 tp{q.x + tt.x, q.y + tt.y, q.z + tt.z};
    const Vec3 ba{q.x - g.x, q.y - g.y, q.z - g.z};
    const Vec3 dv{(ba.x - tp.x) * inv_dtn, (ba.y - tp.y) * inv_dtn, (ba.z - tp.z) * inv_dtn};
    const Vec3 ctA{dv.x - lAq.x, dv.y - lAq.y, dv.z - lAq.z};
    const Vec3 ctB{dv.x - lBq.x, dv.y - lBq.y, dv.z - lBq.z};
    // (the stage's Mahalanobis matrices are those of the LAST rotation linearisation: constant over all its passes — lm_persist_kernel fills the cache once)
    Sym3 M;
    if constexpr (MC) M = mc_load(mc, ms, slot0 + p * threads);
    else M = sym3_inverse(sym3_add(r[p].cov, lmp_rotated_cov(a, R, st->tr_S, pt[p], i0 + p * threads)));
    const Vec3 e{r[p].mean.x - tp.x, r[p].mean.y - tp.y, r[p].mean.z - tp.z};
    const Vec3 Me = sym3_mulv(M, e);
    const double eMe = dot3(e, Me), w = r[p].w;
    if (phase == 1) acc[0] += w * (eMe + lam_n * dot3(ctA, sym3_mulv(M, ctA)));
    if (skip_lin) continue;
    const Vec3 McB = sym3_mulv(M, ctB);
    acc[1] += w * (eMe + lam_n * dot3(ctB, McB));
    acc[2] += 1.0;
    const double s1 = lam_n * inv_dtn;
    const Vec3 vb{w * (Me.x + s1 * McB.x), w * (Me.y + s1 * McB.y), w * (Me.z + s1 * McB.z)};
    accumulate_hb<6>(M, tp, w * (1.0 + lam_n * inv_dtn * inv_dtn), 0.0, vb, &acc[3], &acc[3 + NH]);
    if (PPT > 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// PPT > 0: the interleaved bodies above (DIRECT1 only: one correspondence per point); PPT = 0: any number of points per thread and any neighbour search, one point after the other
// BATCH = points of a thread that go through a body together (interleaved); OCC = wavefronts per SIMD the register budget must allow (2: 256 registers, 4: 128 — then a
// walk's wavefronts fit on the same SIMDs and use the issue slots the resident kernel leaves idle while it exchanges and steps)
// MC: the Mahalanobis cache of lmp_rot_body in LDS (the interleaved bodies only)
template <int DOF, int THREADS, int PPT, int BATCH = (PPT > 0 ? PPT : 1), int OCC = 2, bool MC = (PPT > 0)>
__global__ __launch_bounds__(THREADS, OCC) void lm_persist_kernel(PassArgs a, LmState* st_io, unsigned long long* __restrict__ xbuf, rolo_trace_rec* trace, int ppt, LmState* pub,
                                                                             unsigned long long timeout_ticks, unsigned long long admit_ticks, int max_trials) {
  LmState& sst = lmp_sst;
  double* const sums = lmp_sums;
  __shared__ double row[NV_MAX];
  __shared__ double part[16][NV_MAX];
  __shared__ int bad;
  extern __shared__ unsigned xw[];   // G x 60 halves
  static_assert(sizeof(LmState) % sizeof(int) == 0, "LmState must be int-copyable");
  constexpr int NW = sizeof(LmState) / sizeof(int);
  constexpr int NHR = DOF * (DOF + 1) / 2, NVR = 3 + NHR + DOF;
  constexpr int NP = PPT > 0 ? PPT : 1;
  static_assert(THREADS == 512 || THREADS == 256, "16 summation groups of 32 values, dealt over the workgroup's 32-lane halves");
  const int G = (int)gridDim.x, wg = (int)blockIdx.x, t = (int)threadIdx.x;
  if (PPT > 0) ppt = PPT;
  ROLO_SHORT_KERNEL_PRIO();
#ifdef ROLO_LMP_STATS
  const long long lt_entry = wall_clock64();
#endif
  {
    const int* g = reinterpret_cast<const int*>(st_io);
    int* l = reinterpret_cast<int*>(&sst);
    for (int w = t; w < NW; w += THREADS) l[w] = g[w];
  }
  unsigned e = (unsigned)xbuf[0];   // written by the previous launch on this context (a kernel boundary ago)
  if (t == 0) bad = 0;
  // XCD x (= wg mod 8: the dispatcher deals workgroups round-robin) owns a contiguous eighth of the point blocks — a sector of the cloud whose voxels no other XCD's
  // L2 has to hold (pass_xcd_block); and with no kernel boundary between the trials that L2 stays warm from the second trial on
  const int i0 = a.begin + pass_xcd_block(a, wg, G) * ppt * THREADS + t;
  PtReg pt[NP]; int cid[NP][2]; bool valid[NP];
  if (PPT > 0) lmp_load_points<NP>(a, i0, THREADS, pt, cid, valid);
  // the Mahalanobis cache (lmp_rot_body) behind the G rows of xw: 6 x THREADS x PPT doubles
  constexpr int MS = THREADS * NP;
  static_assert(!MC || PPT > 0, "the cache belongs to the interleaved bodies");
  lds_f64* mc = MC ? (lds_f64*)(__attribute__((address_space(3))) void*)(xw + (size_t)G * 60) : nullptr;
  bool m_valid = false;   // the cache holds M(x0) for the correspondences of buffer `cur` (rotation stage) / M(tr_R) for those of `tr_cur` (translation stage)
  __syncthreads();
  // admission (above): one empty row per workgroup under the short timeout
  if (t == 0) row[0] = 0.0;
  __syncthreads();
  if (admit_ticks == 0 /* test switch: nobody is admitted */ || !lmp_exchange<THREADS, true>(row, 1, 0, ++e, xbuf, G, wg, xw, part, sums, &bad, admit_ticks)) {
    // not everybody is resident: leave the stage to the host, exactly as it was (every leaving workgroup writes the same words)
    if (t == 0) {
      xbuf[0] = admit_ticks == 0 ? e + 1 : e;
      st_io->lmp_bailed = 1;
      if (pub) { int* h = reinterpret_cast<int*>(pub); const int* l = reinterpret_cast<const int*>(&sst); for (int w = 0; w < NW; w++) h[w] = l[w]; pub->lmp_bailed = 1; }
    }
    return;
  }
  if (t == 0) sst.lmp_bailed = 0;
#ifdef ROLO_LMP_STATS
  if (wg == 0 && t == 0) { const unsigned long long dta = (unsigned long long)(wall_clock64() - lt_entry); atomicAdd(&g_lmp_adm[0], dta); atomicAdd(&g_lmp_adm[1], 1ull); atomicMax(&g_lmp_adm[2], dta); }
#endif
  rolo_trace_rec* tr = wg == 0 ? trace : nullptr;   // without a buffer trace_count still advances
  int trial = 0;
  bool ok = true;
  // ---- rotation / 6-dof stage ----
  while (ok && uni(sst.stage) == 1) {
    const bool only_first = uni(sst.phase) == 1 && uni(sst.lin_skip) != 0;
    const int phase_in = uni(sst.phase), cur_in = uni(sst.cur), tc_in = uni(sst.trace_count);
    LMP_STAMP(0);
    if (MC && !m_valid && phase_in == 1) lmp_fill_cache<NP>(a, sst.x0_R, sst.x0_S, cur_in, i0, THREADS, pt, cid, valid, mc, MS, t);
    {
      double acc[NVR]; int slot[NVR];
#pragma unroll
      for (int v = 0; v < NVR; v++) { acc[v] = 0.0; slot[v] = v; }
      if (PPT > 0) {
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += BATCH) {
          lmp_rot_body<DOF, BATCH, MC>(a, &sst, i0 + p0 * THREADS, THREADS, pt + p0, cid + p0, valid + p0, acc, mc, MS, p0 * THREADS + t);
          if (BATCH < NP) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }   // one batch after the other: left alone, the compiler interleaves them all (and spills)
        }
      }
      else for (int p = 0; p < ppt; p++) {
        const int i = i0 + p * THREADS;
        const bool vl = i < a.end;
        PtIn in{};
        if (vl) in = load_pt(a, i);
        rot_pass_compute<DOF>(a, &sst, i, vl, in, acc);
      }
      block_reduce_store<NVR, THREADS>(acc, slot, row, only_first);   // compact: yi, y, n, H lower triangle, b
    }
    __syncthreads();
    LMP_STAMP(1);
    ok = lmp_exchange<THREADS>(row, only_first ? 1 : NVR, NHR, ++e, xbuf, G, wg, xw, part, sums, &bad, timeout_ticks) && ++trial <= max_trials;
    LMP_STAMP(2);
    if (ok && t == 0) lmp_rot_step<DOF>(tr != nullptr);
    __syncthreads();
    if (ok && tr) lmp_trace_flush<THREADS>(tr, tc_in);
    // what the cache holds now (rot_step_t): a linearisation pass (phase 0) left M(x0); a trial WITH a (B) half overwrote it with M(xt) — M(x0) of the next trial exactly if
    // the step accepted and took the speculated linearisation (cur flipped); a cost-only trial only read it (if it was accepted, a phase-0 pass follows and writes)
    m_valid = phase_in == 0 || only_first || uni(sst.cur) != cur_in;
    LMP_ACC();
  }
  m_valid = false;   // (tr_R is the LAST linearisation point, not always x0: the first translation pass fills the cache)
  // ---- translation stage ----
  while (ok && uni(sst.stage) == 2) {
    const bool only_first = uni(sst.phase) == 1 && uni(sst.lin_skip) != 0;
    const int tc_in = uni(sst.trace_count);
    LMP_STAMP(0);
    if (MC && !m_valid) lmp_fill_cache<NP>(a, sst.tr_R, sst.tr_S, uni(sst.tr_cur), i0, THREADS, pt, cid, valid, mc, MS, t);
    {
      double acc[30]; int slot[30];
#pragma unroll
      for (int v = 0; v < 30; v++) { acc[v] = 0.0; slot[v] = v; }
      if (PPT > 0) {
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += BATCH) {
          lmp_trans_body<BATCH, MC>(a, &sst, i0 + p0 * THREADS, THREADS, pt + p0, cid + p0, valid + p0, acc, mc, MS, p0 * THREADS + t);
          if (BATCH < NP) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        }
      }
      else for (int p = 0; p < ppt; p++) {
        const int i = i0 + p * THREADS;
        const bool vl = i < a.end;
        PtIn in{};
        if (vl) in = load_pt(a, i);
        trans_pass_compute(a, &sst, i, vl, in, acc);
      }
      block_reduce_store<30, THREADS>(acc, slot, row, only_first);
    }
    __syncthreads();
    LMP_STAMP(1);
    ok = lmp_exchange<THREADS>(row, only_first ? 1 : 30, 21, ++e, xbuf, G, wg, xw, part, sums, &bad, timeout_ticks) && ++trial <= max_trials;
    LMP_STAMP(2);
    if (ok && t == 0) lmp_trans_step(tr != nullptr);
    __syncthreads();
    if (ok && tr) lmp_trace_flush<THREADS>(tr, tc_in);
    m_valid = true;
    LMP_ACC();
  }
  if (!ok) {   // a row never arrived (or the stages do not end): an error code instead of waiting forever
    if (t == 0) { sst.error = ROLO_ECOMM; sst.stage = 0; sst.rot_done = 1; sst.rot_failed = 1; sst.trans_done = 1; sst.trans_failed = 1; }
    __syncthreads();
  }
  if (wg == 0) {
    int* g = reinterpret_cast<int*>(st_io);
    const int* l = reinterpret_cast<const int*>(&sst);
    for (int w = t; w < NW; w += THREADS) g[w] = l[w];
    if (pub) { int* h = reinterpret_cast<int*>(pub); for (int w = t; w < NW; w += THREADS) h[w] = l[w]; }
    if (t == 0) xbuf[0] = e;   // the next launch's epochs continue here
#ifdef ROLO_LMP_STATS
    if (t == 0) atomicAdd(&g_lmp_adm[3], (unsigned long long)(wall_clock64() - lt_entry));
#endif
  }
}

__global__ void rot_begin_kernel(LmState* st, RotBegin a) {
  if (threadIdx.x != 0) return;
  rot_begin_dev(st, a);
}

ROLO_DEV void trans_knobs(LmState* st, const TransBegin& a) {   // TransBegin: the stage's knobs as they are when computeTranslation is called
  st->max_iterations = a.max_iterations; st->lm_max = a.lm_max; st->q2_intended = a.q2_intended;
  st->trans_eps = a.trans_eps; st->inv_trans_eps = 1.0 / a.trans_eps; st->lm_init = a.lm_init;
}
__global__ void trans_begin_kernel(LmState* st, TransBegin a) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 3; i++) { st->t0[i] = a.t0[i]; st->g[i] = a.g[i]; st->l[i] = a.l[i]; }
  st->dtn = a.dtn; st->dtn1 = a.dtn1; st->ct_lambda = a.ct_lambda; st->pending = 0; st->lmp_bailed = 0;
  trans_knobs(st, a);
  if (a.direct) trans_start(st);
}

__global__ void frame_begin_kernel(LmState* st, const FrameArgs* a) {
  if (threadIdx.x != 0) return;
  frame_begin_dev(st, a);
}

__global__ void frame_begin_batch_kernel(const BatchSlot* __restrict__ slots, const FrameArgs* __restrict__ args, int n) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  LmState* st = slots[s].st;
  const RotBegin r = args[s].rot;
  const TransBegin t = args[s].trans;
  rot_begin_dev(st, r);
  for (int i = 0; i < 3; i++) { st->t0[i] = t.t0[i]; st->g[i] = t.g[i]; st->l[i] = t.l[i]; }
  st->dtn = t.dtn; st->dtn1 = t.dtn1; st->ct_lambda = t.ct_lambda;
}

// stage-level evaluation (rolo_so3_linearize & co): mode 0 = linearise at (R,t): phase 0, correspondences into
// buffer 0; mode 1 = error at (R,t) on the cached correspondences: phase 1.
__global__ void eval_begin_kernel(LmState* st, RotBegin a, int mode) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 9; i++) st->xt_R[i] = a.R[i];
  lm_set_rrt(st->xt_S, st->xt_R);
  for (int i = 0; i < 3; i++) st->xt_t[i] = a.t[i];
  st->optimizer = a.optimizer; st->q2_intended = a.q2_intended;
  st->stage = 1; st->error = 0; st->lin_skip = 0;
  if (mode == 0) {
    st->phase = 0; st->cur = 0; st->tr_cur = 0;
    for (int i = 0; i < 9; i++) { st->x0_R[i] = a.R[i]; st->tr_R[i] = a.R[i]; }
    for (int i = 0; i < 6; i++) { st->x0_S[i] = st->xt_S[i]; st->tr_S[i] = st->xt_S[i]; }
    for (int i = 0; i < 3; i++) st->x0_t[i] = a.t[i];
  } else {
    st->phase = 1;
  }
}
// after a mode-0 evaluation: remember the correspondence count for later t3 evaluations
__global__ void eval_end_kernel(LmState* st, const double* sums, int mode) {
  if (threadIdx.x != 0) return;
  if (mode == 0) { st->n_corr = (int)(sums[V_N] + 0.5); st->tr_n_corr = st->n_corr; }
  st->stage = 0;
}
__global__ void t3_eval_begin_kernel(LmState* st, TransBegin a, int phase) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 3; i++) { st->tt[i] = a.t0[i]; st->t0[i] = a.t0[i]; st->g[i] = a.g[i]; st->l[i] = a.l[i]; }
  st->dtn = a.dtn; st->dtn1 = a.dtn1; st->ct_lambda = a.ct_lambda;
  trans_knobs(st, a);
  st->lam_over_n = (double)(a.ct_lambda / (float)st->tr_n_corr);
  trans_consts(st);
  st->stage = 2; st->phase = phase; st->lin_skip = 0;
}

}  // namespace

hipError_t launch_rot_pass(int dof, const PassArgs& a, const LmState* st, int grid, hipStream_t s) {
  if (dof == 3) rot_pass_kernel<3><<<grid, PASS_THREADS, 0, s>>>(a, st);
  else rot_pass_kernel<6><<<grid, PASS_THREADS, 0, s>>>(a, st);
  return hipGetLastError();
}
hipError_t launch_trans_pass(const PassArgs& a, const LmState* st, int grid, hipStream_t s) {
  trans_pass_kernel<<<grid, PASS_THREADS, 0, s>>>(a, st);
  return hipGetLastError();
}
hipError_t launch_batch_pass(int stage, int dof, const BatchSlot* slots, int n_slots, int bps, hipStream_t s) {
  const int grid = n_slots * bps;
  if (stage == 1) {
    if (dof == 3) rot_pass_batch_kernel<3><<<grid, PASS_THREADS, 0, s>>>(slots, bps);
    else rot_pass_batch_kernel<6><<<grid, PASS_THREADS, 0, s>>>(slots, bps);
  } else {
    trans_pass_batch_kernel<<<grid, PASS_THREADS, 0, s>>>(slots, bps);
  }
  return hipGetLastError();
}
hipError_t launch_batch_ctrl(int stage, const BatchSlot* slots, int n_slots, hipStream_t s) {
  ctrl_batch_kernel<<<n_slots, 256, 0, s>>>(slots, stage);
  return hipGetLastError();
}
hipError_t launch_batch_begin(const BatchSlot* slots, const FrameArgs* args, int n_slots, hipStream_t s) {
  frame_begin_batch_kernel<<<(n_slots + 63) / 64, 64, 0, s>>>(slots, args, n_slots);
  return hipGetLastError();
}
hipError_t launch_lm(int dof, int threads, int ppt, const PassArgs& a, const LmState* st_in, LmState* st_out, const double* rows_in, double* rows_out, int nrows,
                     rolo_trace_rec* trace, int do_body, hipStream_t s, LmState* pub) {
  const int grid = do_body ? nrows : 1;
  if (threads == 1024) {
    if (dof == 3) lm_kernel<3, 1024><<<grid, 1024, 0, s>>>(a, st_in, st_out, rows_in, rows_out, nrows, trace, do_body, ppt, pub);
    else lm_kernel<6, 1024><<<grid, 1024, 0, s>>>(a, st_in, st_out, rows_in, rows_out, nrows, trace, do_body, ppt, pub);
  } else {
    if (dof == 3) lm_kernel<3, 512><<<grid, 512, 0, s>>>(a, st_in, st_out, rows_in, rows_out, nrows, trace, do_body, ppt, pub);
    else lm_kernel<6, 512><<<grid, 512, 0, s>>>(a, st_in, st_out, rows_in, rows_out, nrows, trace, do_body, ppt, pub);
  }
  return hipGetLastError();
}
// one launch of an instantiation: dynamic LDS = the G rows of the exchange (+ the Mahalanobis cache); above 64 KB the function's limit is raised first — once per
// instantiation (KERN is a template argument: a static per kernel, not per signature) and per device of the process
template <auto KERN, typename... Args>
hipError_t lmp_launch(int nrows, int threads, size_t lds, hipStream_t s, Args... args) {
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> raised{0};   // bit d: done on device d
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      if (e != hipSuccess) return e;
      raised.fetch_or(bit, std::memory_order_relaxed);
    }
  }
  KERN<<<nrows, threads, lds, s>>>(args...);
  return hipGetLastError();
}
hipError_t launch_lm_persist(int dof, int threads, int ppt, const PassArgs& a, LmState* st, unsigned long long* xbuf, int nrows, rolo_trace_rec* trace, LmState* pub, unsigned long long timeout_ticks,
                             unsigned long long admit_ticks, int max_trials, hipStream_t s) {
  // the interleaved bodies (1, 2 or 4 points per thread in registers) for the reference's own configuration — SO(3) optimiser, DIRECT1; everything else one point after the other
  static const bool interleave = [] { const char* e = getenv("ROLO_LM_PERSIST_INTERLEAVE"); return !(e && atoi(e) == 0); }();
  const int sp = (interleave && dof == 3 && a.n_off == 1 && (ppt == 1 || ppt == 2 || ppt == 4)) ? ppt : 0;
  // ROLO_LM_PERSIST_MCACHE=0 (A/B): no Mahalanobis cache in LDS — every trial inverts again, as the pass kernels do
  static const int mcache_on = [] { const char* e = getenv("ROLO_LM_PERSIST_MCACHE"); return (e && atoi(e) == 0) ? 0 : 1; }();
  // (the A/B form exists for the headline's case; it is also the form of a launch whose rows + cache would not fit a CU's LDS: four points per thread AND more than ~220
  // workgroups — a cloud of more than 450 000 points on an idle device — is 60 KB of rows + 96 KB of cache + 10 KB of state)
  const bool fits = sizeof(unsigned) * (size_t)nrows * 60 + sizeof(double) * 6 * (size_t)threads * sp <= (size_t)(160 - 12) * 1024;
  const bool mcache = sp > 0 && !(sp == 4 && threads == 512 && (!mcache_on || !fits));
  const size_t lds = sizeof(unsigned) * (size_t)nrows * 60 + (mcache ? sizeof(double) * 6 * (size_t)threads * sp : 0);
  // (builds for four wavefronts per SIMD — 128 registers, so that a walk's wavefronts could share the SIMDs — spill 85 / 159 / 270 registers at 1 / 2 / 4 points per thread and are
  // not instantiated: lm_persist_kernel<3, 512, PPT, 1, 4>, profiles/DEAD_ENDS.md round 6)
  static const int batch4 = [] { const char* e = getenv("ROLO_LM_PERSIST_BATCH"); const int v = e ? atoi(e) : 2; return (v == 1 || v == 4) ? v : 2; }();   // four points per thread go through the bodies in batches of 2 (default: 4 087 scans/s with four contexts, final kernels) / 1 (4 046: twice the dependent round trips and the same 6.6 us per linearising body — the bodies are bound by their fp64 issue at two wavefronts per SIMD, not by their fetches) / 4 (3 761: 144 spilled registers)
#define LMP_GO(...) lmp_launch<&lm_persist_kernel<__VA_ARGS__>>(nrows, threads, lds, s, a, st, xbuf, trace, ppt, pub, timeout_ticks, admit_ticks, max_trials)
  // (A/B, ROLO_LM_PERSIST_BUSY_THREADS=256: 128 workgroups of 256 threads — one wavefront per SIMD at 256 registers, so that other kernels' wavefronts share the SIMDs
  // instead of finding 64 CUs closed: 3 353 / 3 339 against 3 639 / 3 635 scans/s, profiles/DEAD_ENDS.md round 6)
  if (threads == 256) return (dof == 3 && sp == 4) ? LMP_GO(3, 256, 4, 2) : hipErrorInvalidValue;
  if (dof != 3) return LMP_GO(6, 512, 0);
  if (sp == 1) return LMP_GO(3, 512, 1);
  if (sp == 2) return LMP_GO(3, 512, 2);
  if (sp == 4 && batch4 == 1) return LMP_GO(3, 512, 4, 1);
  if (sp == 4 && !mcache) return LMP_GO(3, 512, 4, 2, 2, false);
  if (sp == 4 && batch4 == 2) return LMP_GO(3, 512, 4, 2);
  if (sp == 4) return LMP_GO(3, 512, 4);
  return LMP_GO(3, 512, 0);
#undef LMP_GO
}
size_t lm_persist_words(int nrows) { return (size_t)LMP_HDR + 2 * (size_t)nrows * PEER_SLOT_WORDS; }
hipError_t launch_reduce(const double* partials, int nblocks, double* sums, const LmState* st, int stage, hipStream_t s) {
  reduce_kernel<<<1, 256, 0, s>>>(partials, nblocks, sums, st, stage);
  return hipGetLastError();
}
hipError_t launch_ctrl(LmState* st, const double* partials, int nblocks, const double* sums, rolo_trace_rec* trace, int stage, hipStream_t s, LmState* pub,
                       const PeerArgs* peer, int dof) {
  static const bool generic = [] { const char* e = getenv("ROLO_CTRL_GENERIC"); return e && atoi(e) != 0; }();   // A/B: the one-size-fits-all kernel of round 2
  const int mode = generic ? -1 : (stage == 2 ? 2 : (dof == 3 ? 0 : (dof == 6 ? 1 : -1)));
  const bool p = peer && peer->world > 1 && partials;
  switch (mode) {
    case 0: if (p) ctrl_peer_kernel<0><<<1, 256, 0, s>>>(st, partials, nblocks, trace, stage, pub, *peer); else ctrl_kernel<0><<<1, 256, 0, s>>>(st, partials, nblocks, sums, trace, stage, pub); break;
    case 1: if (p) ctrl_peer_kernel<1><<<1, 256, 0, s>>>(st, partials, nblocks, trace, stage, pub, *peer); else ctrl_kernel<1><<<1, 256, 0, s>>>(st, partials, nblocks, sums, trace, stage, pub); break;
    case 2: if (p) ctrl_peer_kernel<2><<<1, 256, 0, s>>>(st, partials, nblocks, trace, stage, pub, *peer); else ctrl_kernel<2><<<1, 256, 0, s>>>(st, partials, nblocks, sums, trace, stage, pub); break;
    default: if (p) ctrl_peer_kernel<-1><<<1, 256, 0, s>>>(st, partials, nblocks, trace, stage, pub, *peer); else ctrl_kernel<-1><<<1, 256, 0, s>>>(st, partials, nblocks, sums, trace, stage, pub);
  }
  return hipGetLastError();
}
hipError_t launch_rot_begin(LmState* st, const RotBegin& a, hipStream_t s) {
  rot_begin_kernel<<<1, 64, 0, s>>>(st, a);
  return hipGetLastError();
}
// graph-replayable form: both argument blocks come from a device buffer refreshed by a captured H2D copy
hipError_t launch_frame_begin(LmState* st, const FrameArgs* a, hipStream_t s) {
  frame_begin_kernel<<<1, 64, 0, s>>>(st, a);
  return hipGetLastError();
}
hipError_t launch_trans_begin(LmState* st, const TransBegin& a, hipStream_t s) {
  trans_begin_kernel<<<1, 64, 0, s>>>(st, a);
  return hipGetLastError();
}
hipError_t launch_eval_begin(LmState* st, const RotBegin& a, int mode, hipStream_t s) {
  eval_begin_kernel<<<1, 64, 0, s>>>(st, a, mode);
  return hipGetLastError();
}
hipError_t launch_eval_end(LmState* st, const double* sums, int mode, hipStream_t s) {
  eval_end_kernel<<<1, 64, 0, s>>>(st, sums, mode);
  return hipGetLastError();
}
hipError_t launch_t3_eval_begin(LmState* st, const TransBegin& a, int phase, hipStream_t s) {
  t3_eval_begin_kernel<<<1, 64, 0, s>>>(st, a, phase);
  return hipGetLastError();
}

}  // namespace rolo
