// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product. Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may build, link or call anything under oracle/.
//
// PARITY UNPINNED: the reference (sdwyc/ROLO) cannot be compiled here (needs Eigen, PCL, FLANN, Boost,
// ROS) and its tests hold no golden vectors for this path, so this restatement is pinned only by our own
// independent numpy/scipy twin (oracle/twin.py -> tests/golden/*.npz) and analytic known-answer tests.
//
// Small fixed-size fp64 linear algebra that restates the Eigen operations the reference hot path uses.
// Each routine names the Eigen call site (reference file:line) it stands in for. Eigen itself is a
// third-party dependency absent from /root/reference (README.md:21-31 asks for >=3.3.7, no pin); the
// algorithms below are restated from its published behaviour.
#pragma once
#include <cmath>
#include <cfloat>
#include <algorithm>
#include <limits>

namespace orc {

struct M3 {
  double a[3][3];
  double& operator()(int i, int j) { return a[i][j]; }
  double operator()(int i, int j) const { return a[i][j]; }
};
struct V3 {
  double v[3];
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};

inline M3 m3_zero() { M3 m; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m.a[i][j] = 0; return m; }
inline M3 m3_identity() { M3 m = m3_zero(); m.a[0][0] = m.a[1][1] = m.a[2][2] = 1; return m; }
inline M3 m3_mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.a[i][j] = A.a[i][0] * B.a[0][j] + A.a[i][1] * B.a[1][j] + A.a[i][2] * B.a[2][j];
  return C;
}
inline M3 m3_transpose(const M3& A) { M3 T; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.a[i][j] = A.a[j][i]; return T; }
inline V3 m3_mulv(const M3& A, const V3& x) {
  V3 y;
  for (int i = 0; i < 3; i++) y.v[i] = A.a[i][0] * x.v[0] + A.a[i][1] * x.v[1] + A.a[i][2] * x.v[2];
  return y;
}
inline M3 m3_add(const M3& A, const M3& B) { M3 C; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C.a[i][j] = A.a[i][j] + B.a[i][j]; return C; }

// so3.hpp:21-32 skewd
inline M3 skewd(const V3& x) {
  M3 s = m3_zero();
  s.a[0][1] = -x[2]; s.a[0][2] = x[1];
  s.a[1][0] = x[2];  s.a[1][2] = -x[0];
  s.a[2][0] = -x[1]; s.a[2][1] = x[0];
  return s;
}

// Eigen Matrix4d::inverse() on the block-diagonal [A 0; 0 1] used at rot_vgicp_impl.hpp:215-219
// reduces to the 3x3 cofactor inverse of A.
inline M3 m3_inverse(const M3& A) {
  M3 c;
  c.a[0][0] = A.a[1][1] * A.a[2][2] - A.a[1][2] * A.a[2][1];
  c.a[0][1] = A.a[0][2] * A.a[2][1] - A.a[0][1] * A.a[2][2];
  c.a[0][2] = A.a[0][1] * A.a[1][2] - A.a[0][2] * A.a[1][1];
  c.a[1][0] = A.a[1][2] * A.a[2][0] - A.a[1][0] * A.a[2][2];
  c.a[1][1] = A.a[0][0] * A.a[2][2] - A.a[0][2] * A.a[2][0];
  c.a[1][2] = A.a[0][2] * A.a[1][0] - A.a[0][0] * A.a[1][2];
  c.a[2][0] = A.a[1][0] * A.a[2][1] - A.a[1][1] * A.a[2][0];
  c.a[2][1] = A.a[0][1] * A.a[2][0] - A.a[0][0] * A.a[2][1];
  c.a[2][2] = A.a[0][0] * A.a[1][1] - A.a[0][1] * A.a[1][0];
  double det = A.a[0][0] * c.a[0][0] + A.a[0][1] * c.a[1][0] + A.a[0][2] * c.a[2][0];
  double inv = 1.0 / det;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.a[i][j] *= inv;
  return c;
}

// ---- Eigen::JacobiSVD<Matrix3d>(FullU|FullV), used at rot_vgicp_impl.hpp:468 -------------------------
// Two-sided Jacobi on the square matrix (no QR preconditioner for square inputs), sweep order
// p=1..n-1, q=0..p-1, threshold max(min_positive, 2*eps*maxDiag), singular values made non-negative by
// flipping U's column, then sorted descending with matching column swaps.
struct Rot { double c, s; };

inline bool make_jacobi(double x, double y, double z, Rot& r) {  // Eigen Jacobi.h makeJacobi(x,y,z)
  double deno = 2.0 * std::fabs(y);
  if (deno < std::numeric_limits<double>::min()) { r.c = 1; r.s = 0; return false; }
  double tau = (x - z) / deno;
  double w = std::sqrt(tau * tau + 1.0);
  double t = (tau > 0) ? 1.0 / (tau + w) : 1.0 / (tau - w);
  double sign_t = t > 0 ? 1.0 : -1.0;
  double n = 1.0 / std::sqrt(t * t + 1.0);
  r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
  r.c = n;
  return true;
}

// rows p,q of W:  x <- c x + s y ; y <- -s x + c y
inline void apply_left(M3& W, int p, int q, const Rot& j) {
  for (int k = 0; k < 3; k++) {
    double x = W.a[p][k], y = W.a[q][k];
    W.a[p][k] = j.c * x + j.s * y;
    W.a[q][k] = -j.s * x + j.c * y;
  }
}
// cols p,q of W with j.transpose() = (c,-s):  x <- c x - s y ; y <- s x + c y
inline void apply_right(M3& W, int p, int q, const Rot& j) {
  for (int k = 0; k < 3; k++) {
    double x = W.a[k][p], y = W.a[k][q];
    W.a[k][p] = j.c * x - j.s * y;
    W.a[k][q] = j.s * x + j.c * y;
  }
}

inline void real_2x2_jacobi_svd(const M3& W, int p, int q, Rot& j_left, Rot& j_right) {
  double m00 = W.a[p][p], m01 = W.a[p][q], m10 = W.a[q][p], m11 = W.a[q][q];
  Rot rot1;
  double t = m00 + m11, d = m10 - m01;
  if (std::fabs(d) < std::numeric_limits<double>::min()) { rot1.s = 0; rot1.c = 1; }
  else { double u = t / d; double tmp = std::sqrt(1.0 + u * u); rot1.s = 1.0 / tmp; rot1.c = u / tmp; }
  // m.applyOnTheLeft(0,1,rot1)
  double n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
  double n11 = -rot1.s * m01 + rot1.c * m11;
  make_jacobi(n00, n01, n11, j_right);
  // j_left = rot1 * j_right.transpose();  (c,s)*(c2,s2) = (c c2 - s s2, c s2 + s c2), transpose -> s2=-s
  double c2 = j_right.c, s2 = -j_right.s;
  j_left.c = rot1.c * c2 - rot1.s * s2;
  j_left.s = rot1.c * s2 + rot1.s * c2;
}

inline void jacobi_svd3(const M3& A, M3& U, V3& sv, M3& V) {
  double scale = 0;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) scale = std::max(scale, std::fabs(A.a[i][j]));
  if (!(scale > 0) || !std::isfinite(scale)) scale = 1.0;  // Eigen: if(scale==0) scale=1
  M3 W;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W.a[i][j] = A.a[i][j] / scale;
  U = m3_identity(); V = m3_identity();
  const double precision = 2.0 * DBL_EPSILON;
  const double consider_zero = std::numeric_limits<double>::min();
  double max_diag = std::max(std::fabs(W.a[0][0]), std::max(std::fabs(W.a[1][1]), std::fabs(W.a[2][2])));
  bool finished = false;
  int guard = 0;
  while (!finished && guard++ < 100) {
    finished = true;
    for (int p = 1; p < 3; p++) {
      for (int q = 0; q < p; q++) {
        double threshold = std::max(consider_zero, precision * max_diag);
        if (std::fabs(W.a[p][q]) > threshold || std::fabs(W.a[q][p]) > threshold) {
          finished = false;
          Rot jl, jr;
          real_2x2_jacobi_svd(W, p, q, jl, jr);
          apply_left(W, p, q, jl);
          Rot jlt = {jl.c, -jl.s};
          apply_right(U, p, q, jlt);  // U.applyOnTheRight(p,q,j_left.transpose())
          apply_right(W, p, q, jr);
          apply_right(V, p, q, jr);
          max_diag = std::max(max_diag, std::max(std::fabs(W.a[p][p]), std::fabs(W.a[q][q])));
        }
      }
    }
  }
  for (int i = 0; i < 3; i++) {
    double a = std::fabs(W.a[i][i]);
    sv[i] = a;
    if (a != 0) { double sgn = W.a[i][i] / a; for (int k = 0; k < 3; k++) U.a[k][i] *= sgn; }
  }
  for (int i = 0; i < 3; i++) sv[i] *= scale;
  for (int i = 0; i < 3; i++) {
    int pos = i; double mx = sv[i];
    for (int k = i + 1; k < 3; k++) if (sv[k] > mx) { mx = sv[k]; pos = k; }
    if (mx == 0) break;
    if (pos != i) {
      std::swap(sv[i], sv[pos]);
      for (int k = 0; k < 3; k++) { std::swap(U.a[k][i], U.a[k][pos]); std::swap(V.a[k][i], V.a[k][pos]); }
    }
  }
}

// ---- Eigen::Transform<float,3,Affine>::rotation(), src/lidarOdometry.cpp:130,474,548 ---------------------------------------------
// For an Affine (not Isometry) transform rotation() is computeRotationScaling(): JacobiSVD<Matrix3f>(linear(), FullU | FullV) in FLOAT,
// x = det(U V^T) < 0 ? -1 : 1, U.col(2) *= x, rotation = U V^T — the polar factor of the linear part, not the linear part.
// The same two-sided Jacobi as above, on float, flat row-major arrays.
inline void rotation_of_affine3f(const float* L /* 3x3 row-major linear part */, float* R) {
  typedef float T;
  T W[3][3], U[3][3], V[3][3];
  T scale = 0;
  for (int i = 0; i < 9; i++) scale = std::max(scale, std::fabs(L[i]));
  if (!(scale > 0) || !std::isfinite(scale)) scale = 1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { W[i][j] = L[i * 3 + j] / scale; U[i][j] = V[i][j] = (i == j) ? T(1) : T(0); }
  const T precision = T(2) * std::numeric_limits<T>::epsilon(), tiny = std::numeric_limits<T>::min();
  T max_diag = std::max(std::fabs(W[0][0]), std::max(std::fabs(W[1][1]), std::fabs(W[2][2])));
  bool finished = false;
  int guard = 0;
  while (!finished && guard++ < 100) {
    finished = true;
    for (int p = 1; p < 3; p++) for (int q = 0; q < p; q++) {
      const T threshold = std::max(tiny, precision * max_diag);
      if (!(std::fabs(W[p][q]) > threshold || std::fabs(W[q][p]) > threshold)) continue;
      finished = false;
      // real_2x2_jacobi_svd
      const T m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
      T c1, s1;
      const T t = m00 + m11, d = m10 - m01;
      if (std::fabs(d) < tiny) { s1 = 0; c1 = 1; } else { const T u = t / d, tmp = std::sqrt(T(1) + u * u); s1 = T(1) / tmp; c1 = u / tmp; }
      const T n00 = c1 * m00 + s1 * m10, n01 = c1 * m01 + s1 * m11, n11 = -s1 * m01 + c1 * m11;
      T cr, sr;   // makeJacobi(n00, n01, n11)
      const T deno = T(2) * std::fabs(n01);
      if (deno < tiny) { cr = 1; sr = 0; }
      else {
        const T tau = (n00 - n11) / deno, w = std::sqrt(tau * tau + T(1));
        const T tt = tau > 0 ? T(1) / (tau + w) : T(1) / (tau - w);
        const T sign_t = tt > 0 ? T(1) : T(-1), n = T(1) / std::sqrt(tt * tt + T(1));
        sr = -sign_t * (n01 / std::fabs(n01)) * std::fabs(tt) * n; cr = n;
      }
      const T cl = c1 * cr + s1 * sr, sl = -c1 * sr + s1 * cr;   // j_left = rot1 * j_right^T
      for (int k = 0; k < 3; k++) { const T x = W[p][k], y = W[q][k]; W[p][k] = cl * x + sl * y; W[q][k] = -sl * x + cl * y; }      // W.applyOnTheLeft(p, q, j_left)
      for (int k = 0; k < 3; k++) { const T x = U[k][p], y = U[k][q]; U[k][p] = cl * x + sl * y; U[k][q] = -sl * x + cl * y; }      // U.applyOnTheRight(p, q, j_left^T)
      for (int k = 0; k < 3; k++) { const T x = W[k][p], y = W[k][q]; W[k][p] = cr * x - sr * y; W[k][q] = sr * x + cr * y; }      // W.applyOnTheRight(p, q, j_right)
      for (int k = 0; k < 3; k++) { const T x = V[k][p], y = V[k][q]; V[k][p] = cr * x - sr * y; V[k][q] = sr * x + cr * y; }      // V.applyOnTheRight(p, q, j_right)
      max_diag = std::max(max_diag, std::max(std::fabs(W[p][p]), std::fabs(W[q][q])));
    }
  }
  T sv[3];
  for (int i = 0; i < 3; i++) { const T a = std::fabs(W[i][i]); sv[i] = a; if (a != 0) { const T sgn = W[i][i] / a; for (int k = 0; k < 3; k++) U[k][i] *= sgn; } }
  for (int i = 0; i < 3; i++) {
    int pos = i; T mx = sv[i];
    for (int k = i + 1; k < 3; k++) if (sv[k] > mx) { mx = sv[k]; pos = k; }
    if (mx == 0) break;
    if (pos != i) { std::swap(sv[i], sv[pos]); for (int k = 0; k < 3; k++) { std::swap(U[k][i], U[k][pos]); std::swap(V[k][i], V[k][pos]); } }
  }
  T M[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + U[i][2] * V[j][2];
  const T det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) + M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  const T x = det < 0 ? T(-1) : T(1);
  for (int k = 0; k < 3; k++) U[k][2] *= x;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = U[i][0] * V[j][0] + U[i][1] * V[j][1] + U[i][2] * V[j][2];
}

// ---- Eigen::LDLT<Matrix<double,N,N>> (lower, diagonal pivoting), lsq_registration_impl.hpp:102,213,236,288 ----
// solve A x = rhs for symmetric A (lower triangle read). Returns false if a zero pivot is met.
template <int N>
inline bool ldlt_solve(const double (&Ain)[N][N], const double (&rhs)[N], double (&x)[N]) {
  double A[N][N];
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) A[i][j] = (j <= i) ? Ain[i][j] : Ain[j][i];
  int perm[N];
  for (int i = 0; i < N; i++) perm[i] = i;
  bool ok = true;
  for (int k = 0; k < N; k++) {
    int piv = k; double big = std::fabs(A[k][k]);
    for (int i = k + 1; i < N; i++) if (std::fabs(A[i][i]) > big) { big = std::fabs(A[i][i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < N; j++) std::swap(A[k][j], A[piv][j]);
      for (int i = 0; i < N; i++) std::swap(A[i][k], A[i][piv]);
      std::swap(perm[k], perm[piv]);
    }
    double d = A[k][k];
    if (d == 0.0) { ok = false; continue; }
    for (int i = k + 1; i < N; i++) A[i][k] /= d;
    for (int j = k + 1; j < N; j++)
      for (int i = j; i < N; i++) { A[i][j] -= A[i][k] * d * A[j][k]; A[j][i] = A[i][j]; }
  }
  double y[N];
  for (int i = 0; i < N; i++) y[i] = rhs[perm[i]];
  for (int i = 0; i < N; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < N; i++) y[i] = (A[i][i] != 0.0) ? y[i] / A[i][i] : 0.0;
  for (int i = N - 1; i >= 0; i--) for (int j = i + 1; j < N; j++) y[i] -= A[j][i] * y[j];
  for (int i = 0; i < N; i++) x[perm[i]] = y[i];
  return ok;
}

// so3.hpp:59-77 so3_exp (quaternion w,x,y,z) and Eigen Quaterniond::toRotationMatrix
inline void so3_exp_quat(const V3& omega, double q[4]) {
  double theta_sq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2];
  double imag_factor, real_factor;
  if (theta_sq < 1e-10) {
    double theta_quad = theta_sq * theta_sq;
    imag_factor = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad;
    real_factor = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad;
  } else {
    double theta = std::sqrt(theta_sq);
    double half_theta = 0.5 * theta;
    imag_factor = std::sin(half_theta) / theta;
    real_factor = std::cos(half_theta);
  }
  q[0] = real_factor; q[1] = imag_factor * omega[0]; q[2] = imag_factor * omega[1]; q[3] = imag_factor * omega[2];
}
inline M3 quat_to_rot(const double q[4]) {  // Eigen QuaternionBase::toRotationMatrix (no normalisation)
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  M3 R;
  R.a[0][0] = 1 - (tyy + tzz); R.a[0][1] = txy - twz;       R.a[0][2] = txz + twy;
  R.a[1][0] = txy + twz;       R.a[1][1] = 1 - (txx + tzz); R.a[1][2] = tyz - twx;
  R.a[2][0] = txz - twy;       R.a[2][1] = tyz + twx;       R.a[2][2] = 1 - (txx + tyy);
  return R;
}
// so3.hpp:80-103 se3_exp: returns R and t = V(omega) * v
inline void se3_exp(const double a[6], M3& R, V3& t) {
  V3 omega = {{a[0], a[1], a[2]}};
  double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  double q[4];
  so3_exp_quat(omega, q);
  M3 Omega = skewd(omega);
  M3 Omega_sq = m3_mul(Omega, Omega);
  M3 V;
  R = quat_to_rot(q);
  if (theta < 1e-10) {
    V = R;  // so3.matrix()
  } else {
    double theta_sq = theta * theta;
    double c1 = (1.0 - std::cos(theta)) / theta_sq;
    double c2 = (theta - std::sin(theta)) / (theta_sq * theta);
    V = m3_identity();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V.a[i][j] += c1 * Omega.a[i][j] + c2 * Omega_sq.a[i][j];
  }
  V3 v = {{a[3], a[4], a[5]}};
  t = m3_mulv(V, v);
}

}  // namespace orc
