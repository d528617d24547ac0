// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product (see orc_linalg.hpp header). PARITY UNPINNED by the reference itself.
//
// CPU restatement (C++17 + OpenMP, float where the reference is float) of the back end's scan-to-submap optimisation, reference src/backMapping.cpp:
//   scan2MapOptimization :681-711, updatePointAssociateToMap :714-717 + trans2Affine3f :339-342 + pointAssociateToMap :293-299,
//   cornerOptimization :720-824, surfOptimization :827-901, combineOptimizationCoeffs :904-925, LMOptimization :929-1058
// in the reference's loop structure (one OpenMP loop per feature kind, the flags combined in index order afterwards).
// Third-party numerics restated from their published algorithms (sources not under /root/reference, no pins):
//   * pcl::KdTreeFLANN::nearestKSearch(p, 5, ...)  -> exact 5-NN, float squared L2, (d2, index) order (orc_kdtree.hpp);
//   * cv::eigen of a symmetric float matrix (3 x 3 :771, 6 x 6 :1009) -> OpenCV's Jacobi scheme: the off-diagonal element of largest magnitude is
//     annihilated each step (row / column maxima kept in index arrays), rotation from  y = (w_l - w_k) / 2, t = |y| + hypot(p, y), s = hypot(p, t),
//     c = t / s, s = p / s, t = (p / t) p, at most 30 n^2 steps, stop when |p| <= FLT_EPSILON; eigenvalues sorted descending, eigenvectors as ROWS;
//   * Eigen::Matrix<float, 5, 3>::colPivHouseholderQr().solve (:861) -> column-pivoted Householder QR (largest remaining column norm first, LAPACK
//     working note 176 norm down-dating, rank threshold eps * max column norm / rows scaled by the remaining rows), Q^T applied to the right-hand
//     side, back substitution on the non-zero pivots, permutation undone;
//   * cv::solve(AtA, AtB, X, DECOMP_QR) (:999) -> Householder QR of the 6 x 6 float system; matAt * matA / matAt * matB (:997-998) -> products
//     accumulated in double and narrowed to float (cv::gemm's work type for CV_32F); matV.inv() (:1023) -> LU with partial pivoting.
#include "rolo_oracle_backend.h"
#include "orc_kdtree.hpp"

#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;

namespace {

inline float cv_hypot(float a, float b) {   // OpenCV's own hypot (lapack.cpp): scaled, only + * / sqrt
  a = std::fabs(a); b = std::fabs(b);
  if (a > b) { b /= a; return a * std::sqrt(1 + b * b); }
  if (b > 0) { a /= b; return b * std::sqrt(1 + a * a); }
  return 0;
}

// cv::eigen(src, eigenvalues, eigenvectors) for a symmetric n x n float matrix (n <= 6); A is destroyed
void cv_eigen_sym(float* A, int n, float* W, float* V) {
  const float eps = FLT_EPSILON;
  int indR[6], indC[6];
  for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) V[i * n + j] = 0.f; V[i * n + i] = 1.f; }
  auto row_max = [&](int k) { int m = k + 1; float mv = std::fabs(A[k * n + m]); for (int i = k + 2; i < n; i++) { const float v = std::fabs(A[k * n + i]); if (mv < v) { mv = v; m = i; } } return m; };
  auto col_max = [&](int k) { int m = 0; float mv = std::fabs(A[k]); for (int i = 1; i < k; i++) { const float v = std::fabs(A[i * n + k]); if (mv < v) { mv = v; m = i; } } return m; };
  for (int k = 0; k < n; k++) {
    W[k] = A[k * n + k];
    if (k < n - 1) indR[k] = row_max(k);
    if (k > 0) indC[k] = col_max(k);
  }
  if (n > 1) for (int it = 0; it < n * n * 30; it++) {
    int k = 0; float mv = std::fabs(A[indR[0]]);
    for (int i = 1; i < n - 1; i++) { const float v = std::fabs(A[i * n + indR[i]]); if (mv < v) { mv = v; k = i; } }
    int l = indR[k];
    for (int i = 1; i < n; i++) { const float v = std::fabs(A[indC[i] * n + i]); if (mv < v) { mv = v; k = indC[i]; l = i; } }
    const float p = A[k * n + l];
    if (std::fabs(p) <= eps) break;
    const float y = (W[l] - W[k]) * 0.5f;
    float t = std::fabs(y) + cv_hypot(p, y);
    float s = cv_hypot(p, t);
    const float c = t / s;
    s = p / s; t = (p / t) * p;
    if (y < 0) { s = -s; t = -t; }
    A[k * n + l] = 0;
    W[k] -= t; W[l] += t;
    auto rot = [&](float& v0, float& v1) { const float a0 = v0, b0 = v1; v0 = a0 * c - b0 * s; v1 = a0 * s + b0 * c; };
    for (int i = 0; i < k; i++) rot(A[i * n + k], A[i * n + l]);
    for (int i = k + 1; i < l; i++) rot(A[k * n + i], A[i * n + l]);
    for (int i = l + 1; i < n; i++) rot(A[k * n + i], A[l * n + i]);
    for (int i = 0; i < n; i++) rot(V[k * n + i], V[l * n + i]);
    for (int j = 0; j < 2; j++) {
      const int idx = j == 0 ? k : l;
      if (idx < n - 1) indR[idx] = row_max(idx);
      if (idx > 0) indC[idx] = col_max(idx);
    }
  }
  for (int k = 0; k < n - 1; k++) {   // descending
    int m = k;
    for (int i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
    if (k != m) { std::swap(W[m], W[k]); for (int i = 0; i < n; i++) std::swap(V[m * n + i], V[k * n + i]); }
  }
}

// Eigen::Matrix<float, 5, 3> A; x = A.colPivHouseholderQr().solve(b)  (b of 5 rows)
void colpiv_qr_solve_5x3(const float (&Ain)[5][3], const float (&bin)[5], float (&x)[3]) {
  constexpr int R = 5, C = 3;
  float qr[R][C]; std::memcpy(qr, Ain, sizeof(qr));
  float hco[C] = {0, 0, 0}, normU[C], normD[C];
  int trans[C], nonzero = C;
  for (int k = 0; k < C; k++) { float s = 0; for (int i = 0; i < R; i++) s += qr[i][k] * qr[i][k]; normU[k] = normD[k] = std::sqrt(s); }
  float mxn = std::fmax(normU[0], std::fmax(normU[1], normU[2]));
  const float th0 = mxn * FLT_EPSILON / (float)R, threshold_helper = th0 * th0, downdate = std::sqrt(FLT_EPSILON);
  float maxpivot = 0;
  for (int k = 0; k < C; k++) {
    int big = k; float bn = normU[k];
    for (int j = k + 1; j < C; j++) if (normU[j] > bn) { bn = normU[j]; big = j; }
    const float big_sq = bn * bn;
    if (nonzero == C && big_sq < threshold_helper * (float)(R - k)) nonzero = k;
    trans[k] = big;
    if (k != big) { for (int i = 0; i < R; i++) std::swap(qr[i][k], qr[i][big]); std::swap(normU[k], normU[big]); std::swap(normD[k], normD[big]); }
    // makeHouseholderInPlace on qr[k..R-1][k]
    float tail = 0; for (int i = k + 1; i < R; i++) tail += qr[i][k] * qr[i][k];
    const float c0 = qr[k][k];
    float beta, tau;
    if (tail <= FLT_MIN) { tau = 0; beta = c0; for (int i = k + 1; i < R; i++) qr[i][k] = 0; }
    else {
      beta = std::sqrt(c0 * c0 + tail);
      if (c0 >= 0) beta = -beta;
      for (int i = k + 1; i < R; i++) qr[i][k] = qr[i][k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hco[k] = tau; qr[k][k] = beta;
    if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
    // applyHouseholderOnTheLeft to the trailing columns: tmp = essential^T bottom + row k; row k -= tau tmp; bottom -= tau essential tmp
    for (int j = k + 1; j < C; j++) {
      float tmp = 0; for (int i = k + 1; i < R; i++) tmp += qr[i][k] * qr[i][j];
      tmp += qr[k][j];
      qr[k][j] -= tau * tmp;
      for (int i = k + 1; i < R; i++) qr[i][j] -= tau * qr[i][k] * tmp;
    }
    for (int j = k + 1; j < C; j++) {   // LAPACK working note 176
      if (normU[j] != 0.f) {
        float temp = std::fabs(qr[k][j]) / normU[j];
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        const float q = normU[j] / normD[j], temp2 = temp * (q * q);
        if (temp2 <= downdate) { float s = 0; for (int i = k + 1; i < R; i++) s += qr[i][j] * qr[i][j]; normD[j] = std::sqrt(s); normU[j] = normD[j]; }
        else normU[j] *= std::sqrt(temp);
      }
    }
  }
  (void)maxpivot;
  int perm[C] = {0, 1, 2};
  for (int k = 0; k < C; k++) std::swap(perm[k], perm[trans[k]]);
  x[0] = x[1] = x[2] = 0.f;
  if (nonzero == 0) return;
  float c[R]; std::memcpy(c, bin, sizeof(c));
  for (int k = 0; k < nonzero; k++) {   // Q^T b: H_0, then H_1, ...
    float tmp = 0; for (int i = k + 1; i < R; i++) tmp += qr[i][k] * c[i];
    tmp += c[k];
    c[k] -= hco[k] * tmp;
    for (int i = k + 1; i < R; i++) c[i] -= hco[k] * qr[i][k] * tmp;
  }
  for (int i = nonzero - 1; i >= 0; i--) {   // upper-triangular solve in place
    c[i] /= qr[i][i];
    for (int r = 0; r < i; r++) c[r] -= qr[r][i] * c[i];
  }
  for (int i = 0; i < nonzero; i++) x[perm[i]] = c[i];
}

// cv::solve(A, b, x, DECOMP_QR), 6 x 6 float: Householder QR
bool qr_solve6(const float* Ain, const float* bin, float* x) {
  float A[36], b[6];
  std::memcpy(A, Ain, sizeof(A)); std::memcpy(b, bin, sizeof(b));
  for (int k = 0; k < 6; k++) {
    float nrm = 0; for (int i = k; i < 6; i++) nrm += A[i * 6 + k] * A[i * 6 + k];
    nrm = std::sqrt(nrm);
    if (nrm == 0.f) return false;
    const float alpha = A[k * 6 + k] > 0 ? -nrm : nrm;
    float v[6] = {0, 0, 0, 0, 0, 0};
    for (int i = k; i < 6; i++) v[i] = A[i * 6 + k];
    v[k] -= alpha;
    float vv = 0; for (int i = k; i < 6; i++) vv += v[i] * v[i];
    if (vv == 0.f) continue;
    for (int j = k; j < 6; j++) { float d = 0; for (int i = k; i < 6; i++) d += v[i] * A[i * 6 + j]; d = 2 * d / vv; for (int i = k; i < 6; i++) A[i * 6 + j] -= d * v[i]; }
    { float d = 0; for (int i = k; i < 6; i++) d += v[i] * b[i]; d = 2 * d / vv; for (int i = k; i < 6; i++) b[i] -= d * v[i]; }
  }
  for (int i = 5; i >= 0; i--) {
    float s = b[i];
    for (int j = i + 1; j < 6; j++) s -= A[i * 6 + j] * x[j];
    if (A[i * 6 + i] == 0.f) return false;
    x[i] = s / A[i * 6 + i];
  }
  return true;
}

bool lu_invert6(const float* Ain, float* inv) {   // cv::Mat::inv() (DECOMP_LU): Gauss-Jordan with partial pivoting
  float a[36]; std::memcpy(a, Ain, sizeof(a));
  for (int i = 0; i < 36; i++) inv[i] = (i % 7 == 0) ? 1.f : 0.f;
  for (int c = 0; c < 6; c++) {
    int p = c; float best = std::fabs(a[c * 6 + c]);
    for (int r = c + 1; r < 6; r++) if (std::fabs(a[r * 6 + c]) > best) { best = std::fabs(a[r * 6 + c]); p = r; }
    if (best == 0.f) return false;
    if (p != c) for (int j = 0; j < 6; j++) { std::swap(a[p * 6 + j], a[c * 6 + j]); std::swap(inv[p * 6 + j], inv[c * 6 + j]); }
    const float d = 1.f / a[c * 6 + c];
    for (int j = 0; j < 6; j++) { a[c * 6 + j] *= d; inv[c * 6 + j] *= d; }
    for (int r = 0; r < 6; r++) if (r != c) { const float f = a[r * 6 + c]; if (f != 0.f) for (int j = 0; j < 6; j++) { a[r * 6 + j] -= f * a[c * 6 + j]; inv[r * 6 + j] -= f * inv[c * 6 + j]; } }
  }
  return true;
}

struct Coeff { float x, y, z, intensity; };

}  // namespace

extern "C" int orc_scan2map(const float* corner, int n_corner, const float* surf, int n_surf, const float* map_corner, int m_corner, const float* map_surf, int m_surf,
                            float* tf, int edge_min, int surf_min, int threads, int* stats5, unsigned char* selected_out, float* coeff_out) {
  int st[5] = {0, 0, 0, 0, 0};   // skipped, iterations, converged, degenerate, n_selected
  auto done = [&](int rc) { if (stats5) std::memcpy(stats5, st, sizeof(st)); return rc; };
  if (!tf || n_corner < 0 || n_surf < 0 || m_corner < 0 || m_surf < 0) return done(-1);
  if (!(n_corner > edge_min && n_surf > surf_min)) { st[0] = 1; return done(0); }   // :689
  if (m_corner < 5 || m_surf < 5) { st[0] = 2; return done(0); }                   // a 5-NN query needs five points (the reference would read past a shorter result)
#ifdef _OPENMP
  const int nth = threads > 0 ? threads : omp_get_max_threads();
#else
  const int nth = 1; (void)threads;
#endif
  std::vector<P4> mc((size_t)m_corner), ms((size_t)m_surf);
  for (int i = 0; i < m_corner; i++) mc[(size_t)i] = P4{map_corner[4 * (size_t)i], map_corner[4 * (size_t)i + 1], map_corner[4 * (size_t)i + 2], 1.f};
  for (int i = 0; i < m_surf; i++) ms[(size_t)i] = P4{map_surf[4 * (size_t)i], map_surf[4 * (size_t)i + 1], map_surf[4 * (size_t)i + 2], 1.f};
  KdTree kc, ks;   // kdtreeCornerFromMap->setInputCloud / kdtreeSurfFromMap->setInputCloud :690-691
  kc.build(mc); ks.build(ms);
  std::vector<unsigned char> flagC((size_t)n_corner), flagS((size_t)n_surf);
  std::vector<Coeff> coefC((size_t)n_corner), coefS((size_t)n_surf);
  bool isDegenerate = false;
  float matP[36]; for (int i = 0; i < 36; i++) matP[i] = (i % 7 == 0) ? 1.f : 0.f;
  for (int iterCount = 0; iterCount < 30; iterCount++) {
    // trans2Affine3f :339-342 = pcl::getTransformation(x, y, z, roll, pitch, yaw), float
    float T[12];
    {
      const float A = std::cos(tf[2]), B = std::sin(tf[2]), C = std::cos(tf[1]), D = std::sin(tf[1]), E = std::cos(tf[0]), F = std::sin(tf[0]);
      const float DE = D * E, DF = D * F;
      T[0] = A * C; T[1] = A * DF - B * E; T[2] = B * F + A * DE; T[3] = tf[3];
      T[4] = B * C; T[5] = A * E + B * DF; T[6] = B * DE - A * F; T[7] = tf[4];
      T[8] = -D; T[9] = C * F; T[10] = C * E; T[11] = tf[5];
    }
    std::fill(flagC.begin(), flagC.end(), 0); std::fill(flagS.begin(), flagS.end(), 0);
    // ---- cornerOptimization :720-824 ----
#pragma omp parallel for num_threads(nth)
    for (int i = 0; i < n_corner; i++) {
      const float ox = corner[4 * (size_t)i], oy = corner[4 * (size_t)i + 1], oz = corner[4 * (size_t)i + 2];
      const float sx = T[0] * ox + T[1] * oy + T[2] * oz + T[3], sy = T[4] * ox + T[5] * oy + T[6] * oz + T[7], sz = T[8] * ox + T[9] * oy + T[10] * oz + T[11];
      int ind[5]; float sq[5];
      if (kc.knn(P4{sx, sy, sz, 1.f}, 5, ind, sq) < 5) continue;
      if (!(sq[4] < 1.0)) continue;
      float cx = 0, cy = 0, cz = 0;
      for (int j = 0; j < 5; j++) { cx += mc[(size_t)ind[j]].x; cy += mc[(size_t)ind[j]].y; cz += mc[(size_t)ind[j]].z; }
      cx /= 5; cy /= 5; cz /= 5;
      float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
      for (int j = 0; j < 5; j++) {
        const float ax = mc[(size_t)ind[j]].x - cx, ay = mc[(size_t)ind[j]].y - cy, az = mc[(size_t)ind[j]].z - cz;
        a11 += ax * ax; a12 += ax * ay; a13 += ax * az; a22 += ay * ay; a23 += ay * az; a33 += az * az;
      }
      a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
      float A1[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33}, D1[3], V1[9];
      cv_eigen_sym(A1, 3, D1, V1);
      if (D1[0] > 3 * D1[1]) {
        const float x0 = sx, y0 = sy, z0 = sz;
        const float x1 = cx + 0.1 * V1[0], y1 = cy + 0.1 * V1[1], z1 = cz + 0.1 * V1[2];   // (double 0.1, narrowed on assignment: as written)
        const float x2 = cx - 0.1 * V1[0], y2 = cy - 0.1 * V1[1], z2 = cz - 0.1 * V1[2];
        const float a012 = std::sqrt(((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) +
                                     ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) +
                                     ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1)));
        const float l12 = std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
        const float la = ((y1 - y2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) + (z1 - z2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1))) / a012 / l12;
        const float lb = -((x1 - x2) * ((x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1)) - (z1 - z2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
        const float lc = -((x1 - x2) * ((x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1)) + (y1 - y2) * ((y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1))) / a012 / l12;
        const float ld2 = a012 / l12;
        const float s = 1 - 0.9 * std::fabs(ld2);
        if (s > 0.1) { coefC[(size_t)i] = Coeff{s * la, s * lb, s * lc, s * ld2}; flagC[(size_t)i] = 1; }
      }
    }
    // ---- surfOptimization :827-901 ----
#pragma omp parallel for num_threads(nth)
    for (int i = 0; i < n_surf; i++) {
      const float ox = surf[4 * (size_t)i], oy = surf[4 * (size_t)i + 1], oz = surf[4 * (size_t)i + 2];
      const float sx = T[0] * ox + T[1] * oy + T[2] * oz + T[3], sy = T[4] * ox + T[5] * oy + T[6] * oz + T[7], sz = T[8] * ox + T[9] * oy + T[10] * oz + T[11];
      int ind[5]; float sq[5];
      if (ks.knn(P4{sx, sy, sz, 1.f}, 5, ind, sq) < 5) continue;
      if (!(sq[4] < 1.0)) continue;
      float A0[5][3], B0[5] = {-1, -1, -1, -1, -1}, X0[3];
      for (int j = 0; j < 5; j++) { A0[j][0] = ms[(size_t)ind[j]].x; A0[j][1] = ms[(size_t)ind[j]].y; A0[j][2] = ms[(size_t)ind[j]].z; }
      colpiv_qr_solve_5x3(A0, B0, X0);
      float pa = X0[0], pb = X0[1], pc = X0[2], pd = 1;
      const float ps = std::sqrt(pa * pa + pb * pb + pc * pc);
      pa /= ps; pb /= ps; pc /= ps; pd /= ps;
      bool planeValid = true;
      for (int j = 0; j < 5; j++)
        if (std::fabs(pa * A0[j][0] + pb * A0[j][1] + pc * A0[j][2] + pd) > 0.2) { planeValid = false; break; }
      if (planeValid) {
        const float pd2 = pa * sx + pb * sy + pc * sz + pd;
        const float s = 1 - 0.9 * std::fabs(pd2) / std::sqrt(std::sqrt(ox * ox + oy * oy + oz * oz));
        if (s > 0.1) { coefS[(size_t)i] = Coeff{s * pa, s * pb, s * pc, s * pd2}; flagS[(size_t)i] = 1; }
      }
    }
    if (selected_out) { std::memcpy(selected_out, flagC.data(), (size_t)n_corner); std::memcpy(selected_out + n_corner, flagS.data(), (size_t)n_surf); }
    if (coeff_out) {
      for (int i = 0; i < n_corner; i++) { const Coeff c = flagC[(size_t)i] ? coefC[(size_t)i] : Coeff{0, 0, 0, 0}; std::memcpy(coeff_out + 4 * (size_t)i, &c, sizeof(c)); }
      for (int i = 0; i < n_surf; i++) { const Coeff c = flagS[(size_t)i] ? coefS[(size_t)i] : Coeff{0, 0, 0, 0}; std::memcpy(coeff_out + 4 * ((size_t)n_corner + i), &c, sizeof(c)); }
    }
    // ---- combineOptimizationCoeffs :904-925 + LMOptimization :929-1058 ----
    int nsel = 0;
    for (int i = 0; i < n_corner; i++) nsel += flagC[(size_t)i];
    for (int i = 0; i < n_surf; i++) nsel += flagS[(size_t)i];
    st[1] = iterCount + 1; st[4] = nsel;
    if (nsel < 50) break;   // LMOptimization returns false without moving the pose: the remaining iterations would repeat this one
    const float srx = std::sin(tf[1]), crx = std::cos(tf[1]), sry = std::sin(tf[2]), cry = std::cos(tf[2]), srz = std::sin(tf[0]), crz = std::cos(tf[0]);
    double AtAd[36], AtBd[6];
    for (int i = 0; i < 36; i++) AtAd[i] = 0;
    for (int i = 0; i < 6; i++) AtBd[i] = 0;
    auto add_row = [&](const float* po, const Coeff& cf) {
      const float px = po[1], py = po[2], pz = po[0];            // lidar -> camera
      const float kx = cf.y, ky = cf.z, kz = cf.x;
      const float arx = (crx * sry * srz * px + crx * crz * sry * py - srx * sry * pz) * kx + (-srx * srz * px - crz * srx * py - crx * pz) * ky +
                        (crx * cry * srz * px + crx * cry * crz * py - cry * srx * pz) * kz;
      const float ary = ((cry * srx * srz - crz * sry) * px + (sry * srz + cry * crz * srx) * py + crx * cry * pz) * kx +
                        ((-cry * crz - srx * sry * srz) * px + (cry * srz - crz * srx * sry) * py - crx * sry * pz) * kz;
      const float arz = ((crz * srx * sry - cry * srz) * px + (-cry * crz - srx * sry * srz) * py) * kx + (crx * crz * px - crx * srz * py) * ky +
                        ((sry * srz + cry * crz * srx) * px + (crz * sry - cry * srx * srz) * py) * kz;
      const double row[6] = {arz, arx, ary, kz, kx, ky};
      const double b = -(double)cf.intensity;
      for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) AtAd[r * 6 + c] += row[r] * row[c]; AtBd[r] += row[r] * b; }
    };
    for (int i = 0; i < n_corner; i++) if (flagC[(size_t)i]) add_row(corner + 4 * (size_t)i, coefC[(size_t)i]);
    for (int i = 0; i < n_surf; i++) if (flagS[(size_t)i]) add_row(surf + 4 * (size_t)i, coefS[(size_t)i]);
    float AtA[36], AtB[6], X[6];
    for (int i = 0; i < 36; i++) AtA[i] = (float)AtAd[i];
    for (int i = 0; i < 6; i++) AtB[i] = (float)AtBd[i];
    if (!qr_solve6(AtA, AtB, X)) for (int r = 0; r < 6; r++) X[r] = 0.f;
    if (iterCount == 0) {   // :1004-1026
      float Acp[36], E[6], V[36], V2[36], Vi[36];
      std::memcpy(Acp, AtA, sizeof(Acp));
      cv_eigen_sym(Acp, 6, E, V);
      std::memcpy(V2, V, sizeof(V2));
      isDegenerate = false;
      for (int i = 5; i >= 0; i--) {
        if (E[i] < 100.f) { for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0; isDegenerate = true; } else break;
      }
      if (lu_invert6(V, Vi)) for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { float a = 0; for (int k = 0; k < 6; k++) a += Vi[r * 6 + k] * V2[k * 6 + c]; matP[r * 6 + c] = a; }
    }
    if (isDegenerate) { float X2[6]; std::memcpy(X2, X, sizeof(X2)); for (int r = 0; r < 6; r++) { float a = 0; for (int k = 0; k < 6; k++) a += matP[r * 6 + k] * X2[k]; X[r] = a; } }
    for (int r = 0; r < 6; r++) tf[r] += X[r];
    const float r2d = 57.29578f;   // pcl::rad2deg(float); pow(float, int) promotes to double (:1041-1048), the result is narrowed on assignment
    const float deltaR = (float)std::sqrt(std::pow((double)(X[0] * r2d), 2) + std::pow((double)(X[1] * r2d), 2) + std::pow((double)(X[2] * r2d), 2));
    const float deltaT = (float)std::sqrt(std::pow((double)(X[3] * 100), 2) + std::pow((double)(X[4] * 100), 2) + std::pow((double)(X[5] * 100), 2));
    if (deltaR < 0.05f && deltaT < 0.05f) { st[2] = 1; break; }
  }
  st[3] = isDegenerate ? 1 : 0;
  return done(0);
}
