// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product (see orc_linalg.hpp header).
// PARITY UNPINNED by the reference itself: pinned by oracle/twin.py golden vectors + analytic tests only.
//
// CPU restatement (C++17 + OpenMP, fp64 where the reference is fp64, fp32 where it is fp32) of the
// registration half of sdwyc/ROLO's scan-matching hot path:
//   include/rot_gicp/gicp/impl/rot_vgicp_impl.hpp      (RotVGICP: K5 covariances, K7 correspondences,
//                                                       K8 so3_linearize, K9 linearize, K10 compute_error,
//                                                       K11 t3_linearize / compute_t_error)
//   include/rot_gicp/gicp/impl/lsq_registration_impl.hpp (K12 LM / GN drivers)
//   include/rot_gicp/gicp/vmp_voxel.hpp                (K6 voxel map, polar / uniform voxel coordinates)
//   include/rot_gicp/so3/so3.hpp                       (skewd, so3_exp, se3_exp)
// Third-party semantics restated from documented behaviour (sources not under /root/reference, no pins):
//   Eigen >=3.3.7 (JacobiSVD, LDLT, inverse, Quaternion), PCL >=1.10 + FLANN (exact sorted kNN, float L2;
//   transformPointCloud), see SURVEY.md Appendix A.
#include "rolo_oracle.h"
#include "orc_linalg.hpp"
#include "orc_kdtree.hpp"

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <unordered_map>
#include <queue>
#include <numeric>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

inline int resolve_threads(int n) {
#ifdef _OPENMP
  if (n <= 0) return omp_get_max_threads();
  return n;
#else
  (void)n; return 1;
#endif
}

struct Pose { M3 R; V3 t; };  // Eigen::Isometry3d
inline Pose pose_from_rowmajor(const double* T) {
  Pose p;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) p.R.a[i][j] = T[i * 4 + j]; p.t.v[i] = T[i * 4 + 3]; }
  return p;
}
inline void pose_to_rowmajor(const Pose& p, double* T) {
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[i * 4 + j] = p.R.a[i][j]; T[i * 4 + 3] = p.t.v[i]; }
  T[12] = T[13] = T[14] = 0; T[15] = 1;
}
// Isometry3d * Vector4d with w = 1 : sum_k T(i,k) p_k, k = 0..3 left to right
inline V3 pose_apply(const Pose& T, const V3& p) {
  V3 r;
  for (int i = 0; i < 3; i++) r.v[i] = ((T.R.a[i][0] * p[0] + T.R.a[i][1] * p[1]) + T.R.a[i][2] * p[2]) + T.t.v[i] * 1.0;
  return r;
}
inline Pose pose_mul(const Pose& A, const Pose& B) {  // delta * x0
  Pose C;
  C.R = m3_mul(A.R, B.R);
  V3 rt = m3_mulv(A.R, B.t);
  for (int i = 0; i < 3; i++) C.t.v[i] = rt[i] + A.t.v[i];
  return C;
}

struct Key { int32_t k[3]; bool operator==(const Key& o) const { return k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2]; } };
struct KeyHash {  // vmp_voxel.hpp:49-58 Vector3iHash with boost::hash_combine (pre-1.81). Only bucket placement; unobservable.
  size_t operator()(const Key& x) const {
    size_t seed = 0;
    for (int i = 0; i < 3; i++) seed ^= (size_t)(int64_t)x.k[i] + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};

struct Voxel {  // vmp_voxel.hpp:60-109 VmfVoxel / AdditiveVmfVoxel (kappa, r_bar, dir_reg are never read by a solver)
  int num_points = 0;
  double mean[4] = {0, 0, 0, 0};
  double cov[4][4] = {{0}};
  Key key;
};

// vmp_voxel.hpp:199-201
inline Key voxel_coord(const V3& x, double res) {
  Key k;
  for (int i = 0; i < 3; i++) k.k[i] = (int32_t)std::floor(x[i] / res - 0.5);
  return k;
}
// vmp_voxel.hpp:208-211
inline Key polar_coord(const V3& x, const double* pres) {
  double r = std::sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
  Key k;
  k.k[0] = (int32_t)std::floor((std::atan2(x[1], x[0]) + M_PI) / pres[0]);
  k.k[1] = (int32_t)std::floor(std::acos(x[2] / r) / pres[1]);
  k.k[2] = (int32_t)std::floor(r / pres[2]);
  return k;
}

// vmp_voxel.hpp:13-47
std::vector<Key> neighbor_offsets(int method) {
  std::vector<Key> o;
  if (method == ORC_DIRECT1) { o.push_back(Key{{0, 0, 0}}); return o; }
  if (method == ORC_DIRECT7) {
    const int d[7][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    for (auto& e : d) o.push_back(Key{{e[0], e[1], e[2]}});
    return o;
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) o.push_back(Key{{i - 1, j - 1, k - 1}});
  return o;
}

}  // namespace

struct orc_reg {
  orc_params P;
  std::vector<P4> target, source;
  KdTree kd_target, kd_source;
  std::vector<M3> target_covs, source_covs;  // 3x3 block of the Matrix4d; last row/col are zero
  bool have_tcov = false, have_scov = false;
  // voxel map
  bool have_map = false;
  std::vector<Voxel> voxels;  // order of first appearance
  std::unordered_map<Key, int, KeyHash> vmap;
  // correspondences
  std::vector<int> corr_src, corr_vox;
  std::vector<M3> mahal;
  float lambda_ = 1.0f;  // rot_vgicp.hpp:124 (float)
  double lm_lambda = -1.0;
  std::vector<orc_trace_rec> trace;
  // Scripted evaluations (tests of the LM drivers' branches, orc_reg_set_script): while set, so3_linearize / linearize6 / compute_error / t3_eval return the
  // script's numbers — indexed by the (outer iteration, trial) the drivers are in — instead of evaluating clouds; the drivers themselves (the restatement of
  // lsq_registration_impl.hpp:55-179, 225-324) run unchanged, so every exit of theirs can be reached with inputs that are the same bits on both sides of a test.
  const orc_lm_script* script = nullptr;
  int cur_outer = 0, cur_trial = 0;
};

namespace {

// rot_vgicp_impl.hpp:421-496 calculate_covariances
int calc_covs(orc_reg* r, const std::vector<P4>& cloud, const KdTree& kd, std::vector<M3>& covs) {
  const int n = (int)cloud.size();
  const int k = r->P.k_correspondences;
  if (n < k) return -2;  // FLANN would clamp k and leave uninitialised columns; refuse instead
  covs.resize(n);
  const int nt = resolve_threads(r->P.num_threads);
  int bad = 0;
#pragma omp parallel for num_threads(nt) schedule(guided, 8) reduction(+ : bad)
  for (int i = 0; i < n; i++) {
    std::vector<int> idx(k);
    int found = kd.knn(cloud[i], k, idx.data(), nullptr);
    if (found != k) { bad++; continue; }
    std::vector<double> nb(4 * k);
    for (int j = 0; j < k; j++) {
      const P4& q = cloud[idx[j]];
      nb[0 * k + j] = (double)q.x; nb[1 * k + j] = (double)q.y; nb[2 * k + j] = (double)q.z; nb[3 * k + j] = 1.0;
    }
    double mean[4];
    for (int d = 0; d < 4; d++) { double s = 0; for (int j = 0; j < k; j++) s += nb[d * k + j]; mean[d] = s / k; }
    for (int d = 0; d < 4; d++) for (int j = 0; j < k; j++) nb[d * k + j] -= mean[d];
    M3 cov;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
      double s = 0; for (int j = 0; j < k; j++) s += nb[a * k + j] * nb[b * k + j];
      cov.a[a][b] = s / k;
    }
    const int reg = r->P.regularization;
    if (reg == ORC_REG_NONE) { covs[i] = cov; continue; }
    if (reg == ORC_REG_FROBENIUS) {
      M3 C = cov; for (int d = 0; d < 3; d++) C.a[d][d] += 1e-3;
      M3 Ci = m3_inverse(C);
      double nrm = 0; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) nrm += Ci.a[a][b] * Ci.a[a][b];
      nrm = std::sqrt(nrm);
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Ci.a[a][b] /= nrm;
      covs[i] = m3_inverse(Ci);
      continue;
    }
    M3 U, V; V3 sv;
    jacobi_svd3(cov, U, sv, V);
    V3 values;
    switch (reg) {
      case ORC_REG_PLANE: values = {{1, 1, 1e-3}}; break;
      case ORC_REG_MIN_EIG: for (int d = 0; d < 3; d++) values.v[d] = std::max(sv[d], 1e-3); break;
      case ORC_REG_NORMALIZED_MIN_EIG: {
        double mx = std::max(sv[0], std::max(sv[1], sv[2]));
        for (int d = 0; d < 3; d++) values.v[d] = std::max(sv[d] / mx, 1e-3);
        break;
      }
      case ORC_REG_PLANE_S: {
        double s = sv[0] + sv[1] + sv[2];
        for (int d = 0; d < 3; d++) values.v[d] = sv[d] / s;
        values.v[2] = 1e-3;
        break;
      }
      default: values = {{1, 1, 1e-3}}; break;
    }
    M3 out;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
      out.a[a][b] = (U.a[a][0] * values[0]) * V.a[b][0] + (U.a[a][1] * values[1]) * V.a[b][1] + (U.a[a][2] * values[2]) * V.a[b][2];
    covs[i] = out;
  }
  return bad ? -3 : 0;
}

inline Key coord_of(const orc_reg* r, const V3& x) {
  return r->P.voxel_type == ORC_VOXEL_POLAR ? polar_coord(x, r->P.polar_resolution) : voxel_coord(x, r->P.voxel_resolution);
}

// vmp_voxel.hpp:167-197 create_voxelmap (+ append :93-99, finalize :101-108)
int build_map(orc_reg* r) {
  if (!r->have_tcov) return -1;
  r->voxels.clear(); r->vmap.clear();
  for (size_t i = 0; i < r->target.size(); i++) {
    const P4& p = r->target[i];
    V3 x = {{(double)p.x, (double)p.y, (double)p.z}};
    Key key = coord_of(r, x);
    auto it = r->vmap.find(key);
    int vid;
    if (it == r->vmap.end()) { vid = (int)r->voxels.size(); r->vmap.emplace(key, vid); Voxel v; v.key = key; r->voxels.push_back(v); }
    else vid = it->second;
    Voxel& v = r->voxels[vid];
    v.num_points++;
    v.mean[0] += x[0]; v.mean[1] += x[1]; v.mean[2] += x[2]; v.mean[3] += 1.0;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) v.cov[a][b] += r->target_covs[i].a[a][b];
  }
  for (auto& v : r->voxels) {
    for (int d = 0; d < 4; d++) v.mean[d] /= v.num_points;
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) v.cov[a][b] /= v.num_points;
  }
  r->have_map = true;
  return 0;
}

// rot_vgicp_impl.hpp:173-222 update_correspondences
void update_correspondences(orc_reg* r, const Pose& T) {
  const int n = (int)r->source.size();
  const int nt = resolve_threads(r->P.num_threads);
  auto offsets = neighbor_offsets(r->P.neighbor_search);
  std::vector<std::vector<std::pair<int, int>>> corrs(nt);
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) {
    const P4& p = r->source[i];
    V3 a = {{(double)p.x, (double)p.y, (double)p.z}};
    V3 ta = pose_apply(T, a);
    Key c = coord_of(r, ta);
    for (const auto& o : offsets) {
      Key q{{c.k[0] + o.k[0], c.k[1] + o.k[1], c.k[2] + o.k[2]}};
      auto it = r->vmap.find(q);
      if (it != r->vmap.end()) {
#ifdef _OPENMP
        corrs[omp_get_thread_num()].push_back({i, it->second});
#else
        corrs[0].push_back({i, it->second});
#endif
      }
    }
  }
  r->corr_src.clear(); r->corr_vox.clear();
  for (auto& c : corrs) for (auto& pr : c) { r->corr_src.push_back(pr.first); r->corr_vox.push_back(pr.second); }
  const int nc = (int)r->corr_src.size();
  r->mahal.resize(nc);
  M3 Rt = m3_transpose(T.R);
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < nc; i++) {
    const M3& cov_A = r->source_covs[r->corr_src[i]];
    const Voxel& vx = r->voxels[r->corr_vox[i]];
    // RCR = cov_B + T * cov_A * T^T (4x4; the translation column only touches row/col 3 because cov_A's
    // last row/col are zero) ; RCR(3,3) = 1 ; inverse ; (3,3) = 0   => 3x3 inverse of the upper block
    M3 RA = m3_mul(T.R, cov_A);
    M3 RAR = m3_mul(RA, Rt);
    M3 RCR;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) RCR.a[a][b] = vx.cov[a][b] + RAR.a[a][b];
    r->mahal[i] = m3_inverse(RCR);
  }
}

inline int ensure_map(orc_reg* r) {
  if (!r->have_map) return build_map(r);
  return 0;
}

inline double quad3(const M3& M, const V3& e) {  // e^T M e, Eigen evaluates (e^T M) e
  double s = 0;
  for (int j = 0; j < 3; j++) { double c = e[0] * M.a[0][j] + e[1] * M.a[1][j] + e[2] * M.a[2][j]; s += c * e[j]; }
  return s;
}

// scripted evaluations (orc_reg::script): the linearisation opened by outer iteration o, the trial cost of (o, t)
inline int script_o(const orc_reg* r) { return std::min(std::max(r->cur_outer, 0), r->script->n_outer - 1); }
double script_lin(orc_reg* r, int dof, double* H, double* b) {
  const orc_lm_script* S = r->script;
  const int o = script_o(r);
  r->corr_src.assign((size_t)std::max(S->lin_n[o], 0), 0); r->corr_vox.assign(r->corr_src.size(), 0);
  if (H && b) {
    for (int a = 0; a < dof; a++) { for (int c = 0; c < dof; c++) H[a * dof + c] = S->lin_H[(size_t)o * 36 + a * 6 + c]; b[a] = S->lin_b[(size_t)o * 6 + a]; }
  }
  return S->lin_y[o];
}
double script_err(const orc_reg* r) {
  const orc_lm_script* S = r->script;
  const int t = std::min(std::max(r->cur_trial, 0), S->n_trial - 1);
  return S->err_y[(size_t)script_o(r) * S->n_trial + t];
}

// rot_vgicp_impl.hpp:293-388
double so3_linearize(orc_reg* r, const Pose& T, double* H9, double* b3) {
  if (r->script) return script_lin(r, 3, H9, b3);
  if (ensure_map(r) != 0) return NAN;
  update_correspondences(r, T);
  const int nc = (int)r->corr_src.size();
  const int nt = resolve_threads(r->P.num_threads);
  std::vector<M3> Hs(nt, m3_zero());
  std::vector<V3> bs(nt, V3{{0, 0, 0}});
  double sum_errors = 0;
#pragma omp parallel for num_threads(nt) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < nc; i++) {
    const P4& p = r->source[r->corr_src[i]];
    const Voxel& vx = r->voxels[r->corr_vox[i]];
    V3 a = {{(double)p.x, (double)p.y, (double)p.z}};
    V3 ta = pose_apply(T, a);
    V3 e = {{vx.mean[0] - ta[0], vx.mean[1] - ta[1], vx.mean[2] - ta[2]}};
    double w = std::sqrt((double)vx.num_points);
    const M3& M = r->mahal[i];
    sum_errors += w * quad3(M, e);
    if (!H9 || !b3) continue;
    M3 J = skewd(ta);
    M3 Jt = m3_transpose(J);
    M3 JtM = m3_mul(Jt, M);
    for (int a2 = 0; a2 < 3; a2++) for (int b2 = 0; b2 < 3; b2++) JtM.a[a2][b2] *= w;  // w * J^T (left-assoc.)
    M3 Hi = m3_mul(JtM, J);
    V3 bi = m3_mulv(JtM, e);
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    Hs[tid] = m3_add(Hs[tid], Hi);
    for (int d = 0; d < 3; d++) bs[tid].v[d] += bi[d];
  }
  if (H9 && b3) {
    M3 H = m3_zero(); V3 b = {{0, 0, 0}};
    for (int t = 0; t < nt; t++) { H = m3_add(H, Hs[t]); for (int d = 0; d < 3; d++) b.v[d] += bs[t].v[d]; }
    for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) H9[a * 3 + c] = H.a[a][c]; b3[a] = b[a]; }
  }
  return sum_errors;
}

// J (3x6 upper part of the 4x6) = [skew(ta) | -I] ; accumulates w * J^T M J (6x6) and w * J^T M v (6)
inline void accum6(const M3& M, const V3& ta, double w, const V3& e, double scale2, const V3* e2,
                   double (&H)[6][6], double (&b)[6]) {
  double J[3][6];
  M3 S = skewd(ta);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) { J[i][j] = S.a[i][j]; J[i][3 + j] = (i == j) ? -1.0 : 0.0; } }
  double JtM[6][3];
  for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) JtM[a][c] = J[0][a] * M.a[0][c] + J[1][a] * M.a[1][c] + J[2][a] * M.a[2][c];
  for (int a = 0; a < 6; a++) {
    for (int c = 0; c < 6; c++) {
      double h = JtM[a][0] * J[0][c] + JtM[a][1] * J[1][c] + JtM[a][2] * J[2][c];
      H[a][c] += w * (h + scale2 * h);  // scale2 = lambda/N * (1/dt)^2 for the CT term, 0 otherwise
    }
    double g = JtM[a][0] * e[0] + JtM[a][1] * e[1] + JtM[a][2] * e[2];
    double g2 = 0;
    if (e2) g2 = JtM[a][0] * (*e2)[0] + JtM[a][1] * (*e2)[1] + JtM[a][2] * (*e2)[2];
    b[a] += w * (g + g2);
  }
}

// rot_vgicp_impl.hpp:225-290 (6-dof variant; reachable through setOptimizerType)
double linearize6(orc_reg* r, const Pose& T, double* H36, double* b6) {
  if (r->script) return script_lin(r, 6, H36, b6);
  if (ensure_map(r) != 0) return NAN;
  update_correspondences(r, T);
  const int nc = (int)r->corr_src.size();
  const int nt = resolve_threads(r->P.num_threads);
  struct Acc { double H[6][6]; double b[6]; };
  std::vector<Acc> acc(nt);
  for (auto& a : acc) memset(&a, 0, sizeof(Acc));
  double sum_errors = 0;
#pragma omp parallel for num_threads(nt) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < nc; i++) {
    const P4& p = r->source[r->corr_src[i]];
    const Voxel& vx = r->voxels[r->corr_vox[i]];
    V3 a = {{(double)p.x, (double)p.y, (double)p.z}};
    V3 ta = pose_apply(T, a);
    V3 e = {{vx.mean[0] - ta[0], vx.mean[1] - ta[1], vx.mean[2] - ta[2]}};
    double w = std::sqrt((double)vx.num_points);
    sum_errors += w * quad3(r->mahal[i], e);
    if (!H36 || !b6) continue;
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    accum6(r->mahal[i], ta, w, e, 0.0, nullptr, acc[tid].H, acc[tid].b);
  }
  if (H36 && b6) {
    for (int a = 0; a < 6; a++) { b6[a] = 0; for (int c = 0; c < 6; c++) H36[a * 6 + c] = 0; }
    for (int t = 0; t < nt; t++) for (int a = 0; a < 6; a++) { b6[a] += acc[t].b[a]; for (int c = 0; c < 6; c++) H36[a * 6 + c] += acc[t].H[a][c]; }
  }
  return sum_errors;
}

// rot_vgicp_impl.hpp:391-417 : cached correspondences and Mahalanobis, trial pose
double compute_error(orc_reg* r, const Pose& T) {
  if (r->script) return script_err(r);
  const int nc = (int)r->corr_src.size();
  const int nt = resolve_threads(r->P.num_threads);
  double sum_errors = 0;
#pragma omp parallel for num_threads(nt) reduction(+ : sum_errors)
  for (int i = 0; i < nc; i++) {
    const P4& p = r->source[r->corr_src[i]];
    const Voxel& vx = r->voxels[r->corr_vox[i]];
    V3 a = {{(double)p.x, (double)p.y, (double)p.z}};
    V3 ta = pose_apply(T, a);
    V3 e = {{vx.mean[0] - ta[0], vx.mean[1] - ta[1], vx.mean[2] - ta[2]}};
    double w = std::sqrt((double)vx.num_points);
    sum_errors += w * quad3(r->mahal[i], e);
  }
  return sum_errors;
}

// rot_vgicp_impl.hpp:499-607 (t3_linearize, last_transform init Zero) and :610-658 (compute_t_error, init
// Vector4d::Identity() = (1,0,0,0)).  SURVEY Q2: the `.col(3).head<3>() = last_t0` store on a 4x1 vector is
// out of bounds (UB, asserts off); restated as "initial value retained, stray store dropped".
// q2_intended = 1 uses last_t0 in both.
double t3_eval(orc_reg* r, const V3& trans, const V3& init_guess, const V3& last_t0, double dtn, double dtn1,
               bool is_error_variant, double* H36, double* b6) {
  if (r->script) { if (is_error_variant) return script_err(r); const size_t keep = r->corr_src.size(); const double y = script_lin(r, 6, H36, b6); r->corr_src.resize(keep); r->corr_vox.resize(keep); return y; }
  const int nc = (int)r->corr_src.size();
  const int nt = resolve_threads(r->P.num_threads);
  struct Acc { double H[6][6]; double b[6]; };
  std::vector<Acc> acc(nt);
  for (auto& a : acc) memset(&a, 0, sizeof(Acc));
  const size_t pt_size = (size_t)nc;
  const float lam_over_n_f = r->lambda_ / (float)pt_size;  // float / size_t -> float division (lambda_ is float)
  const double lam_over_n = (double)lam_over_n_f;
  V3 last_transform;
  if (r->P.q2_intended) last_transform = last_t0;
  else if (is_error_variant) last_transform = {{1, 0, 0}};
  else last_transform = {{0, 0, 0}};
  double sum_errors = 0;
  const bool want = (H36 && b6);
#pragma omp parallel for num_threads(nt) reduction(+ : sum_errors) schedule(guided, 8)
  for (int i = 0; i < nc; i++) {
    const P4& p = r->source[r->corr_src[i]];
    const Voxel& vx = r->voxels[r->corr_vox[i]];
    V3 a = {{(double)p.x, (double)p.y, (double)p.z}};
    V3 ta = {{a[0] + trans[0], a[1] + trans[1], a[2] + trans[2]}};                 // transform * mean_A
    V3 ba = {{a[0] - init_guess[0], a[1] - init_guess[1], a[2] - init_guess[2]}};  // propagation_transform.inverse() * mean_A
    V3 e = {{vx.mean[0] - ta[0], vx.mean[1] - ta[1], vx.mean[2] - ta[2]}};
    V3 ct;
    for (int d = 0; d < 3; d++) ct.v[d] = (ba[d] - ta[d]) / dtn - last_transform[d] / dtn1;
    double w = std::sqrt((double)vx.num_points);
    const M3& M = r->mahal[i];
    sum_errors += w * (quad3(M, e) + lam_over_n * quad3(M, ct));
    if (!want) continue;
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    // Hi = w (J1^T M J1 + lam/N J2^T M J2), J2 = J1/dtn ; bi = w (J1^T M e + lam/N J2^T M ct)
    double inv = 1.0 / dtn;
    V3 ct_s = {{lam_over_n * inv * ct[0], lam_over_n * inv * ct[1], lam_over_n * inv * ct[2]}};
    accum6(M, ta, w, e, lam_over_n * inv * inv, &ct_s, acc[tid].H, acc[tid].b);
  }
  if (want) {
    for (int a = 0; a < 6; a++) { b6[a] = 0; for (int c = 0; c < 6; c++) H36[a * 6 + c] = 0; }
    for (int t = 0; t < nt; t++) for (int a = 0; a < 6; a++) { b6[a] += acc[t].b[a]; for (int c = 0; c < 6; c++) H36[a * 6 + c] += acc[t].H[a][c]; }
  }
  return sum_errors;
}

// lsq_registration_impl.hpp:182-191 / :328-335
inline bool is_converged(const orc_reg* r, const Pose& delta, bool rot_only) {
  double rmax = 0, tmax = 0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rmax = std::max(rmax, 1.0 / r->P.rotation_epsilon * std::fabs(delta.R.a[i][j] - (i == j ? 1.0 : 0.0)));
  if (rot_only) return rmax < 1;
  for (int i = 0; i < 3; i++) tmax = std::max(tmax, 1.0 / r->P.transformation_epsilon * std::fabs(delta.t[i]));
  return std::max(rmax, tmax) < 1;
}
inline bool is_t_converged(const orc_reg* r, const V3& d) {  // :142-148
  double m = 0;
  for (int i = 0; i < 3; i++) m = std::max(m, 1.0 / r->P.transformation_epsilon * std::fabs(d[i]));
  return m < 1;
}

inline void push_trace(orc_reg* r, int stage, int outer, int trial, int accepted, double y0, double yi, double rho, double lambda, double dn) {
  orc_trace_rec t; t.stage = stage; t.outer = outer; t.trial = trial; t.accepted = accepted; t.y0 = y0; t.yi = yi; t.rho = rho; t.lambda = lambda; t.dnorm = dn;
  r->trace.push_back(t);
}

// lsq_registration_impl.hpp:273-324
bool rot_step_lm(orc_reg* r, Pose& x0, Pose& delta, int outer) {
  double H9[9], b3[3];
  double y0 = so3_linearize(r, x0, H9, b3);
  if (r->lm_lambda < 0.0) r->lm_lambda = r->P.lm_init_lambda_factor * std::max(std::fabs(H9[0]), std::max(std::fabs(H9[4]), std::fabs(H9[8])));
  double nu = 2.0;
  for (int i = 0; i < r->P.lm_max_iterations; i++) {
    double A[3][3], rhs[3], d[3];
    for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) A[a][c] = H9[a * 3 + c] + (a == c ? r->lm_lambda : 0.0); rhs[a] = -b3[a]; }
    ldlt_solve<3>(A, rhs, d);
    V3 dv = {{d[0], d[1], d[2]}};
    double q[4]; so3_exp_quat(dv, q);
    delta.R = quat_to_rot(q); delta.t = {{0, 0, 0}};
    Pose xi = pose_mul(delta, x0);
    r->cur_trial = i;
    double yi = compute_error(r, xi);
    double den = 0; for (int a = 0; a < 3; a++) den += d[a] * (r->lm_lambda * d[a] - b3[a]);
    double rho = (y0 - yi) / den;
    double dn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (rho < 0) {
      if (is_converged(r, delta, true)) { push_trace(r, 0, outer, i, 2, y0, yi, rho, r->lm_lambda, dn); return true; }
      push_trace(r, 0, outer, i, 0, y0, yi, rho, r->lm_lambda, dn);
      r->lm_lambda = nu * r->lm_lambda; nu = 2 * nu;
      continue;
    }
    push_trace(r, 0, outer, i, 1, y0, yi, rho, r->lm_lambda, dn);
    x0 = xi;
    r->lm_lambda = r->lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    return true;
  }
  return false;
}

// :225-270 step_lm and :208-222 step_gn
bool step_lm6(orc_reg* r, Pose& x0, Pose& delta, int outer) {
  double H36[36], b6[6];
  double y0 = linearize6(r, x0, H36, b6);
  if (r->lm_lambda < 0.0) { double m = 0; for (int a = 0; a < 6; a++) m = std::max(m, std::fabs(H36[a * 7])); r->lm_lambda = r->P.lm_init_lambda_factor * m; }
  double nu = 2.0;
  for (int i = 0; i < r->P.lm_max_iterations; i++) {
    double A[6][6], rhs[6], d[6];
    for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) A[a][c] = H36[a * 6 + c] + (a == c ? r->lm_lambda : 0.0); rhs[a] = -b6[a]; }
    ldlt_solve<6>(A, rhs, d);
    se3_exp(d, delta.R, delta.t);
    Pose xi = pose_mul(delta, x0);
    r->cur_trial = i;
    double yi = compute_error(r, xi);
    double den = 0, dn = 0; for (int a = 0; a < 6; a++) { den += d[a] * (r->lm_lambda * d[a] - b6[a]); dn += d[a] * d[a]; }
    double rho = (y0 - yi) / den;
    dn = std::sqrt(dn);
    if (rho < 0) {
      if (is_converged(r, delta, false)) { push_trace(r, 0, outer, i, 2, y0, yi, rho, r->lm_lambda, dn); return true; }
      push_trace(r, 0, outer, i, 0, y0, yi, rho, r->lm_lambda, dn);
      r->lm_lambda = nu * r->lm_lambda; nu = 2 * nu;
      continue;
    }
    push_trace(r, 0, outer, i, 1, y0, yi, rho, r->lm_lambda, dn);
    x0 = xi;
    r->lm_lambda = r->lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    return true;
  }
  return false;
}
bool step_gn6(orc_reg* r, Pose& x0, Pose& delta, int outer) {
  double H36[36], b6[6];
  double y0 = linearize6(r, x0, H36, b6);
  double A[6][6], rhs[6], d[6];
  for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) A[a][c] = H36[a * 6 + c]; rhs[a] = -b6[a]; }
  ldlt_solve<6>(A, rhs, d);
  se3_exp(d, delta.R, delta.t);
  x0 = pose_mul(delta, x0);
  double dn = 0; for (int a = 0; a < 6; a++) dn += d[a] * d[a];
  push_trace(r, 0, outer, 0, 1, y0, NAN, NAN, 0.0, std::sqrt(dn));
  return true;
}

// :84-139
bool step_t_optimize(orc_reg* r, V3& x0, V3& delta, const V3& g, const V3& l, double dtn, double dtn1, int outer) {
  double H36[36], b6[6];
  double y0 = t3_eval(r, x0, g, l, dtn, dtn1, false, H36, b6);
  if (r->lm_lambda < 0.0) { double m = 0; for (int a = 0; a < 6; a++) m = std::max(m, std::fabs(H36[a * 7])); r->lm_lambda = r->P.lm_init_lambda_factor * m; }
  double nu = 2.0;
  for (int i = 0; i < r->P.lm_max_iterations; i++) {
    double A[6][6], rhs[6], d[6];
    for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) A[a][c] = H36[a * 6 + c] + (a == c ? r->lm_lambda : 0.0); rhs[a] = -b6[a]; }
    ldlt_solve<6>(A, rhs, d);
    M3 Rd; se3_exp(d, Rd, delta);
    V3 xi = {{delta[0] + x0[0], delta[1] + x0[1], delta[2] + x0[2]}};
    r->cur_trial = i;
    double yi = t3_eval(r, xi, g, l, dtn, dtn1, true, nullptr, nullptr);
    double den = 0, dn = 0; for (int a = 0; a < 6; a++) { den += d[a] * (r->lm_lambda * d[a] - b6[a]); dn += d[a] * d[a]; }
    double rho = (y0 - yi) / den;
    dn = std::sqrt(dn);
    if (rho < 0) {
      if (is_t_converged(r, delta)) { push_trace(r, 1, outer, i, 2, y0, yi, rho, r->lm_lambda, dn); return true; }
      push_trace(r, 1, outer, i, 0, y0, yi, rho, r->lm_lambda, dn);
      r->lm_lambda = nu * r->lm_lambda; nu = 2 * nu;
      continue;
    }
    push_trace(r, 1, outer, i, 1, y0, yi, rho, r->lm_lambda, dn);
    x0 = xi;
    r->lm_lambda = r->lm_lambda * std::max(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
    return true;
  }
  return false;
}

int set_cloud(std::vector<P4>& dst, const float* pts, int n, int stride) {
  if (n < 0 || stride < 3 || (!pts && n > 0)) return -1;
  dst.resize(n);
  for (int i = 0; i < n; i++) dst[i] = P4{pts[(size_t)i * stride], pts[(size_t)i * stride + 1], pts[(size_t)i * stride + 2], 1.0f};
  return 0;
}

void cov_out(const std::vector<M3>& c, double* out) {
  for (size_t i = 0; i < c.size(); i++) {
    double* o = out + i * 16;
    for (int k = 0; k < 16; k++) o[k] = 0;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) o[a * 4 + b] = c[i].a[a][b];
  }
}

}  // namespace

extern "C" {

void orc_default_params(orc_params* p) {
  p->k_correspondences = 20;
  p->regularization = ORC_REG_PLANE;
  p->neighbor_search = ORC_DIRECT1;
  p->voxel_type = ORC_VOXEL_POLAR;
  p->voxel_resolution = 1.0;
  p->polar_resolution[0] = 1; p->polar_resolution[1] = 0; p->polar_resolution[2] = 0;  // Vector3d::Identity(), SURVEY Q9
  p->optimizer = ORC_OPT_SO3_LM;
  p->max_iterations = 64;
  p->rotation_epsilon = 2e-3;
  p->transformation_epsilon = 5e-4;
  p->lm_max_iterations = 10;
  p->lm_init_lambda_factor = 1e-9;
  p->num_threads = 0;
  p->fixed_iterations = 0;
  p->q2_intended = 0;
}

orc_reg* orc_reg_create(const orc_params* p) {
  orc_reg* r = new orc_reg();
  if (p) r->P = *p; else orc_default_params(&r->P);
  return r;
}
void orc_reg_destroy(orc_reg* r) { delete r; }

int orc_reg_set_target(orc_reg* r, const float* pts, int n, int stride) {
  int rc = set_cloud(r->target, pts, n, stride);
  if (rc) return rc;
  r->kd_target.build(r->target);
  r->have_tcov = false; r->have_map = false; r->target_covs.clear();
  r->corr_src.clear(); r->corr_vox.clear(); r->mahal.clear();
  return 0;
}
int orc_reg_set_source(orc_reg* r, const float* pts, int n, int stride) {
  int rc = set_cloud(r->source, pts, n, stride);
  if (rc) return rc;
  r->kd_source.build(r->source);
  r->have_scov = false; r->source_covs.clear();
  r->corr_src.clear(); r->corr_vox.clear(); r->mahal.clear();
  return 0;
}
int orc_reg_compute_covariances(orc_reg* r) {
  if (!r->have_scov) { int rc = calc_covs(r, r->source, r->kd_source, r->source_covs); if (rc) return rc; r->have_scov = true; }
  if (!r->have_tcov) { int rc = calc_covs(r, r->target, r->kd_target, r->target_covs); if (rc) return rc; r->have_tcov = true; }
  return 0;
}
int orc_reg_get_source_covs(orc_reg* r, double* covs) { if (!r->have_scov) return -1; cov_out(r->source_covs, covs); return 0; }
int orc_reg_get_target_covs(orc_reg* r, double* covs) { if (!r->have_tcov) return -1; cov_out(r->target_covs, covs); return 0; }
int orc_reg_set_source_covs(orc_reg* r, const double* covs) {
  r->source_covs.resize(r->source.size());
  for (size_t i = 0; i < r->source.size(); i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) r->source_covs[i].a[a][b] = covs[i * 16 + a * 4 + b];
  r->have_scov = true;
  return 0;
}
int orc_reg_build_voxelmap(orc_reg* r) { int rc = orc_reg_compute_covariances(r); if (rc) return rc; return build_map(r); }
int orc_reg_num_voxels(orc_reg* r) { return r->have_map ? (int)r->voxels.size() : -1; }
int orc_reg_get_voxels(orc_reg* r, int32_t* keys, int32_t* counts, double* means, double* covs) {
  if (!r->have_map) return -1;
  for (size_t i = 0; i < r->voxels.size(); i++) {
    const Voxel& v = r->voxels[i];
    if (keys) for (int d = 0; d < 3; d++) keys[i * 3 + d] = v.key.k[d];
    if (counts) counts[i] = v.num_points;
    if (means) for (int d = 0; d < 4; d++) means[i * 4 + d] = v.mean[d];
    if (covs) for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) covs[i * 16 + a * 4 + b] = v.cov[a][b];
  }
  return 0;
}

double orc_reg_so3_linearize(orc_reg* r, const double* T, double* H9, double* b3) {
  if (orc_reg_compute_covariances(r)) return NAN;
  return so3_linearize(r, pose_from_rowmajor(T), H9, b3);
}
double orc_reg_linearize(orc_reg* r, const double* T, double* H36, double* b6) {
  if (orc_reg_compute_covariances(r)) return NAN;
  return linearize6(r, pose_from_rowmajor(T), H36, b6);
}
double orc_reg_compute_error(orc_reg* r, const double* T) { return compute_error(r, pose_from_rowmajor(T)); }
int orc_reg_num_correspondences(orc_reg* r) { return (int)r->corr_src.size(); }
int orc_reg_get_correspondences(orc_reg* r, int32_t* src_idx, int32_t* voxel_idx, double* m) {
  for (size_t i = 0; i < r->corr_src.size(); i++) {
    if (src_idx) src_idx[i] = r->corr_src[i];
    if (voxel_idx) voxel_idx[i] = r->corr_vox[i];
    if (m) { double* o = m + i * 16; for (int k = 0; k < 16; k++) o[k] = 0; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) o[a * 4 + b] = r->mahal[i].a[a][b]; }
  }
  return (int)r->corr_src.size();
}
double orc_reg_t3_linearize(orc_reg* r, const double* t3, const double* g, const double* l, double dtn, double dtn1, float ct_lambda, double* H36, double* b6) {
  r->lambda_ = ct_lambda;
  return t3_eval(r, V3{{t3[0], t3[1], t3[2]}}, V3{{g[0], g[1], g[2]}}, V3{{l[0], l[1], l[2]}}, dtn, dtn1, false, H36, b6);
}
double orc_reg_compute_t_error(orc_reg* r, const double* t3, const double* g, const double* l, double dtn, double dtn1, float ct_lambda) {
  r->lambda_ = ct_lambda;
  return t3_eval(r, V3{{t3[0], t3[1], t3[2]}}, V3{{g[0], g[1], g[2]}}, V3{{l[0], l[1], l[2]}}, dtn, dtn1, true, nullptr, nullptr);
}

// rot_vgicp_impl.hpp:146-160 computeTransformation + lsq_registration_impl.hpp:152-179
int orc_reg_align(orc_reg* r, const float* guess16, float* Tf, double* Td, int* n_outer, int* converged) {
  if (!r->script) {
    if (r->source.empty() || r->target.empty()) return -1;
    r->have_map = false;  // voxelmap_.reset()
    int rc = orc_reg_compute_covariances(r);
    if (rc) return rc;
  }
  Pose x0;
  if (guess16) { double g[16]; for (int i = 0; i < 16; i++) g[i] = (double)guess16[i]; x0 = pose_from_rowmajor(g); }
  else { x0.R = m3_identity(); x0.t = {{0, 0, 0}}; }
  r->lm_lambda = -1.0;
  bool conv = false;
  int iters = 0;
  const int maxit = r->P.fixed_iterations > 0 ? r->P.fixed_iterations : r->P.max_iterations;
  int status = 0;
  for (int i = 0; i < maxit && (r->P.fixed_iterations > 0 || !conv); i++) {
    iters = i + 1;
    r->cur_outer = i;
    Pose delta; delta.R = m3_identity(); delta.t = {{0, 0, 0}};
    bool ok;
    switch (r->P.optimizer) {
      case ORC_OPT_LM: ok = step_lm6(r, x0, delta, i); break;
      case ORC_OPT_GN: ok = step_gn6(r, x0, delta, i); break;
      default: ok = rot_step_lm(r, x0, delta, i); break;
    }
    if (!ok) { status = 1; break; }  // "lm not converged!!"
    conv = is_converged(r, delta, false);
  }
  double T[16]; pose_to_rowmajor(x0, T);
  if (Td) memcpy(Td, T, sizeof(T));
  if (Tf) for (int i = 0; i < 16; i++) Tf[i] = (float)T[i];
  if (n_outer) *n_outer = iters;
  if (converged) *converged = conv ? 1 : 0;
  return status;
}

// rot_vgicp_impl.hpp:163-169 + lsq_registration_impl.hpp:55-80
int orc_reg_compute_translation(orc_reg* r, double* trans, const double* g3, const double* l3, double dtn, double dtn1, float ct_lambda, int* n_outer) {
  r->lambda_ = ct_lambda;
  if (r->corr_src.empty()) return -4;  // empty correspondence set => NaN in the reference
  V3 t0 = {{trans[0], trans[1], trans[2]}};
  V3 g = {{g3[0], g3[1], g3[2]}}, l = {{l3[0], l3[1], l3[2]}};
  r->lm_lambda = -1.0;
  bool conv = false;
  int iters = 0, status = 0;
  for (int i = 0; i < r->P.max_iterations && !conv; i++) {
    iters = i + 1;
    r->cur_outer = i;
    V3 delta = {{0, 0, 0}};
    if (!step_t_optimize(r, t0, delta, g, l, dtn, dtn1, i)) { status = 1; break; }
    conv = is_t_converged(r, delta);
  }
  trans[0] = t0[0]; trans[1] = t0[1]; trans[2] = t0[2];
  if (n_outer) *n_outer = iters;
  return status;
}

void orc_reg_set_driver_params(orc_reg* r, const orc_params* p) {
  r->P.optimizer = p->optimizer; r->P.max_iterations = p->max_iterations; r->P.rotation_epsilon = p->rotation_epsilon; r->P.transformation_epsilon = p->transformation_epsilon;
  r->P.lm_max_iterations = p->lm_max_iterations; r->P.lm_init_lambda_factor = p->lm_init_lambda_factor; r->P.fixed_iterations = p->fixed_iterations;
}
void orc_reg_set_script(orc_reg* r, const orc_lm_script* s) { r->script = s; r->cur_outer = 0; r->cur_trial = 0; }

int orc_reg_trace(orc_reg* r, orc_trace_rec* out, int cap) {
  int n = (int)r->trace.size();
  for (int i = 0; i < n && i < cap; i++) out[i] = r->trace[i];
  return n;
}
void orc_reg_clear_trace(orc_reg* r) { r->trace.clear(); }

int orc_knn(const float* pts, int n, int stride, int k, int threads, int32_t* idx, float* d2) {
  std::vector<P4> c;
  if (set_cloud(c, pts, n, stride)) return -1;
  if (n < k) return -2;
  KdTree kd; kd.build(c);
  int nt = resolve_threads(threads);
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
  for (int i = 0; i < n; i++) kd.knn(c[i], k, idx + (size_t)i * k, d2 ? d2 + (size_t)i * k : nullptr);
  return 0;
}

int orc_voxel_keys(const float* pts, int n, int stride, int voxel_type, double vres, const double* pres, const double* T, int32_t* keys) {
  Pose P; if (T) P = pose_from_rowmajor(T);
  for (int i = 0; i < n; i++) {
    V3 x = {{(double)pts[(size_t)i * stride], (double)pts[(size_t)i * stride + 1], (double)pts[(size_t)i * stride + 2]}};
    if (T) x = pose_apply(P, x);
    Key k = voxel_type == ORC_VOXEL_POLAR ? polar_coord(x, pres) : voxel_coord(x, vres);
    keys[i * 3] = k.k[0]; keys[i * 3 + 1] = k.k[1]; keys[i * 3 + 2] = k.k[2];
  }
  return 0;
}

void orc_so3_exp(const double* w, double* R9) {
  double q[4]; so3_exp_quat(V3{{w[0], w[1], w[2]}}, q);
  M3 R = quat_to_rot(q);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R9[i * 3 + j] = R.a[i][j];
}
void orc_se3_exp(const double* a6, double* R9, double* t3) {
  M3 R; V3 t; se3_exp(a6, R, t);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R9[i * 3 + j] = R.a[i][j]; t3[i] = t[i]; }
}
void orc_svd3(const double* A9, double* U9, double* s3, double* V9) {
  M3 A, U, V; V3 s;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A.a[i][j] = A9[i * 3 + j];
  jacobi_svd3(A, U, s, V);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) { U9[i * 3 + j] = U.a[i][j]; V9[i * 3 + j] = V.a[i][j]; } s3[i] = s[i]; }
}
int orc_ldlt_solve(int n, const double* A, const double* rhs, double* x) {
  if (n == 3) {
    double a[3][3], r3[3], x3[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) a[i][j] = A[i * 3 + j]; r3[i] = rhs[i]; }
    bool ok = ldlt_solve<3>(a, r3, x3);
    for (int i = 0; i < 3; i++) x[i] = x3[i];
    return ok ? 0 : 1;
  }
  if (n == 6) {
    double a[6][6], r6[6], x6[6];
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) a[i][j] = A[i * 6 + j]; r6[i] = rhs[i]; }
    bool ok = ldlt_solve<6>(a, r6, x6);
    for (int i = 0; i < 6; i++) x[i] = x6[i];
    return ok ? 0 : 1;
  }
  return -1;
}

// pcl::transformPointCloud float path (PCL >=1.10 SSE2 Transformer::se3): p0 + (p1 + (p2 + c3)), w <- 1, other fields copied
void orc_transform_cloud_f(const float* in, float* out, int n, int stride, const float* T) {
  for (int i = 0; i < n; i++) {
    const float* s = in + (size_t)i * stride;
    float* d = out + (size_t)i * stride;
    float x = s[0], y = s[1], z = s[2];
    float o[3];
    for (int rI = 0; rI < 3; rI++) o[rI] = T[rI * 4 + 0] * x + (T[rI * 4 + 1] * y + (T[rI * 4 + 2] * z + T[rI * 4 + 3]));
    for (int k = 3; k < stride; k++) d[k] = s[k];
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
    if (stride > 3) d[3] = 1.0f;
  }
}

}  // extern "C"
