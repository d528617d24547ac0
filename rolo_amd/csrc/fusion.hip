// Second half of the rolo_lidarOdometry node: rolo::eskf::PoseESEKF and TransformFusion — host code (the state is 18 numbers and runs at
// 20 / 30 Hz: there is nothing to put on the GPU). C ABI: include/rolo_fusion.h.
// Restates, as written: include/rolo/eskf/eskf.hpp:39-358 on the iterated ESKF of the IKFoM toolkit
// (include/rolo/eskf/IKFoM_toolkit/esekfom/esekfom.hpp: predict :275-403, update_iterated :406-703; mtk/types/SOn.hpp:186-350 SO3,
// mtk/src/mtkmath.hpp cos_sinc_sqrt / exp / log / A_matrix) specialised to this state — pos, rot (SO(3)), vel, omega, acc, alpha: 18 dof, no
// S2 / SEN blocks — and src/lidarOdometry.cpp:34-323 (odom2affine, TransformFusion). Eigen is not available: small fixed-size loops.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/rolo_fusion.h"
#include "polar_f32.hpp"
#include "../../include/rolo_hip.h"

namespace {

constexpr int N = 18, L = 6;   // state dof, measurement dof
constexpr int I_POS = 0, I_ROT = 3, I_VEL = 6, I_OMEGA = 9, I_ACC = 12, I_ALPHA = 15;
constexpr double TOL = 1e-11;  // MTK::tolerance<double>()

struct Quat { double w = 1, x = 0, y = 0, z = 0; };
Quat qmul(const Quat& a, const Quat& b) {   // Eigen quaternion product
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
Quat qconj(const Quat& a) { return Quat{a.w, -a.x, -a.y, -a.z}; }
double qnorm(const Quat& a) { return std::sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z); }
Quat qnormalized(Quat a) { const double n = qnorm(a); a.w /= n; a.x /= n; a.y /= n; a.z /= n; return a; }
void q_to_R(const Quat& q, double* R) {   // Eigen::Quaternion::toRotationMatrix
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
Quat R_to_q(const double* m) {   // Eigen::Quaternion(Matrix3): quaternionbase_assign_impl<3,3>
  Quat q; double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t;
    q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0; if (m[4] > m[0]) i = 1; if (m[8] > m[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
    double v[3]; v[i] = 0.5 * t; t = 0.5 / t;
    q.w = (m[k * 3 + j] - m[j * 3 + k]) * t; v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

// mtkmath.hpp:142-173
void cos_sinc_sqrt(double x2, double& cosi, double& sinc) {
  const double taylor_0 = DBL_EPSILON, taylor_2 = std::sqrt(taylor_0), taylor_n = std::sqrt(taylor_2);
  if (x2 >= taylor_n) { const double x = std::sqrt(x2); cosi = std::cos(x); sinc = std::sin(x) / x; return; }
  static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
  cosi = 1.; sinc = 1.;
  double term = -1 / 2. * x2;
  for (int i = 0; i < 3; ++i) { cosi += term; term *= inv[2 * i]; sinc += term; term *= -inv[2 * i + 1] * x2; }
}
// SO3::exp(dvec, scale) (SOn.hpp:333-337): w = MTK::exp(vec_out, dvec, scale / 2)
Quat so3_exp(const double* v, double scale) {
  const double s = scale / 2, n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double c, sc; cos_sinc_sqrt(s * s * n2, c, sc);
  const double mult = sc * s;
  return Quat{c, mult * v[0], mult * v[1], mult * v[2]};
}
// SO3::log(orient) = MTK::log(w, vec, 2, plus_minus_periodicity = true) (SOn.hpp:342-346, mtkmath.hpp:268-289)
void so3_log(const Quat& q, double* out) {
  double nv = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (nv < TOL) nv = TOL;
  const double s = 2.0 / nv * std::atan(nv / q.w);
  out[0] = s * q.x; out[1] = s * q.y; out[2] = s * q.z;
}
void A_matrix(const double* v, double* A) {   // mtkmath.hpp:235-247
  const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], n = std::sqrt(sq);
  for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
  if (n < TOL) return;
  const double H[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
  double H2[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) H2[i * 3 + j] = H[i * 3] * H[j] + H[i * 3 + 1] * H[3 + j] + H[i * 3 + 2] * H[6 + j];
  const double a = (1 - std::cos(n)) / sq, b = (1 - std::sin(n) / n) / sq;
  for (int i = 0; i < 9; i++) A[i] += a * H[i] + b * H2[i];
}

// general inverse with partial pivoting (Eigen's fixed-size 6x6 inverse goes through PartialPivLU)
bool invert(const double* Ain, double* inv, int n) {
  std::vector<double> a(Ain, Ain + n * n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) inv[i * n + j] = i == j ? 1.0 : 0.0;
  for (int c = 0; c < n; c++) {
    int p = c; double best = std::fabs(a[c * n + c]);
    for (int r = c + 1; r < n; r++) if (std::fabs(a[r * n + c]) > best) { best = std::fabs(a[r * n + c]); p = r; }
    if (best == 0.0) return false;
    if (p != c) for (int j = 0; j < n; j++) { std::swap(a[p * n + j], a[c * n + j]); std::swap(inv[p * n + j], inv[c * n + j]); }
    const double d = 1.0 / a[c * n + c];
    for (int j = 0; j < n; j++) { a[c * n + j] *= d; inv[c * n + j] *= d; }
    for (int r = 0; r < n; r++) if (r != c) {
      const double f = a[r * n + c];
      if (f != 0.0) for (int j = 0; j < n; j++) { a[r * n + j] -= f * a[c * n + j]; inv[r * n + j] -= f * inv[c * n + j]; }
    }
  }
  return true;
}

struct State {
  double pos[3] = {0, 0, 0}; Quat rot; double vel[3] = {0, 0, 0}, omega[3] = {0, 0, 0}, acc[3] = {0, 0, 0}, alpha[3] = {0, 0, 0};
  // vect blocks: += scale * d ; SO3: rot = rot * exp(d, scale)   (oplus and boxplus are the same functions, SOn.hpp:241-253)
  void plus(const double* d, double scale) {
    for (int i = 0; i < 3; i++) { pos[i] += scale * d[I_POS + i]; vel[i] += scale * d[I_VEL + i]; omega[i] += scale * d[I_OMEGA + i]; acc[i] += scale * d[I_ACC + i]; alpha[i] += scale * d[I_ALPHA + i]; }
    rot = qmul(rot, so3_exp(d + I_ROT, scale));
  }
  void minus(const State& o, double* d) const {   // boxminus
    for (int i = 0; i < 3; i++) { d[I_POS + i] = pos[i] - o.pos[i]; d[I_VEL + i] = vel[i] - o.vel[i]; d[I_OMEGA + i] = omega[i] - o.omega[i]; d[I_ACC + i] = acc[i] - o.acc[i]; d[I_ALPHA + i] = alpha[i] - o.alpha[i]; }
    so3_log(qmul(qconj(o.rot), rot), d + I_ROT);
  }
};

// processModel eskf.hpp:299-306
void process_model(const State& s, double dt, double* f) {
  for (int i = 0; i < N; i++) f[i] = 0;
  for (int i = 0; i < 3; i++) { f[I_POS + i] = s.vel[i] + 0.5 * dt * s.acc[i]; f[I_ROT + i] = s.omega[i] + 0.5 * dt * s.alpha[i]; f[I_VEL + i] = s.acc[i]; f[I_OMEGA + i] = s.alpha[i]; }
}

}  // namespace

struct rolo_eskf {
  rolo_eskf_options opt;
  State x;
  double P[N * N];
  double Q[L * L];          // process noise (6 x 6: linear jerk, angular jerk)
  bool initialized = false;
  double last_time = 0.0;

  void configure() {   // eskf.hpp:249-255
    for (int i = 0; i < 36; i++) Q[i] = 0;
    for (int i = 0; i < 3; i++) { Q[i * 7] = opt.q_linear_jerk_std * opt.q_linear_jerk_std; Q[(3 + i) * 7] = opt.q_angular_jerk_std * opt.q_angular_jerk_std; }
    x = State();
    reset();
  }
  void initial_covariance() {   // :257-266
    for (int i = 0; i < N * N; i++) P[i] = 0;
    const double sd[6] = {opt.init_position_std, opt.init_rotation_std, opt.init_velocity_std, opt.init_angular_velocity_std, opt.init_acceleration_std, opt.init_angular_acceleration_std};
    for (int b = 0; b < 6; b++) for (int i = 0; i < 3; i++) P[(3 * b + i) * (N + 1)] = sd[b] * sd[b];
  }
  void reset() { initialized = false; last_time = 0.0; initial_covariance(); }
  static Quat normalized_quaternion(Quat q) {   // :283-290
    if (!std::isfinite(q.w) || !std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z) || qnorm(q) < 1e-12) return Quat();
    return qnormalized(q);
  }
  void initialize(double stamp, const double* p, const Quat& q) {   // :97-112
    x = State();
    for (int i = 0; i < 3; i++) x.pos[i] = p[i];
    x.rot = normalized_quaternion(q);
    initial_covariance();
    initialized = true; last_time = stamp;
  }

  // esekfom.hpp:275-403 for this state
  void predict(double dt) {
    double f[N]; process_model(x, dt, f);
    // processJacobian :308-318, processNoiseJacobian :320-327
    double fx[N * N] = {0}, fw[N * L] = {0};
    for (int i = 0; i < 3; i++) {
      fx[(I_POS + i) * N + I_VEL + i] = 1.0; fx[(I_POS + i) * N + I_ACC + i] = 0.5 * dt;
      fx[(I_ROT + i) * N + I_OMEGA + i] = 1.0; fx[(I_ROT + i) * N + I_ALPHA + i] = 0.5 * dt;
      fx[(I_VEL + i) * N + I_ACC + i] = 1.0; fx[(I_OMEGA + i) * N + I_ALPHA + i] = 1.0;
      fw[(I_ACC + i) * L + i] = 1.0; fw[(I_ALPHA + i) * L + 3 + i] = 1.0;
    }
    x.plus(f, dt);   // x_.oplus(f_, dt)
    // SO3 block: seg = -f * dt; F_x1 block = exp(seg, scalar_type(1/2) == 0).toRotationMatrix() = Identity (:359, as written);
    // rows of f_x / f_w multiplied by A_matrix(seg)
    double seg[3], A[9];
    for (int i = 0; i < 3; i++) seg[i] = -1 * f[I_ROT + i] * dt;
    A_matrix(seg, A);
    double fxf[N * N], fwf[N * L];
    std::memcpy(fxf, fx, sizeof(fx)); std::memcpy(fwf, fw, sizeof(fw));
    for (int c = 0; c < N; c++) for (int r = 0; r < 3; r++) fxf[(I_ROT + r) * N + c] = A[r * 3] * fx[(I_ROT)*N + c] + A[r * 3 + 1] * fx[(I_ROT + 1) * N + c] + A[r * 3 + 2] * fx[(I_ROT + 2) * N + c];
    for (int c = 0; c < L; c++) for (int r = 0; r < 3; r++) fwf[(I_ROT + r) * L + c] = A[r * 3] * fw[(I_ROT)*L + c] + A[r * 3 + 1] * fw[(I_ROT + 1) * L + c] + A[r * 3 + 2] * fw[(I_ROT + 2) * L + c];
    double F[N * N];
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) F[i * N + j] = (i == j ? 1.0 : 0.0) + fxf[i * N + j] * dt;
    // P = F P F^T + (dt f_w) Q (dt f_w)^T
    double FP[N * N], Pn[N * N];
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < N; k++) s += F[i * N + k] * P[k * N + j]; FP[i * N + j] = s; }
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < N; k++) s += FP[i * N + k] * F[j * N + k]; Pn[i * N + j] = s; }
    double GQ[N * L];
    for (int i = 0; i < N; i++) for (int j = 0; j < L; j++) { double s = 0; for (int k = 0; k < L; k++) s += (dt * fwf[i * L + k]) * Q[k * L + j]; GQ[i * L + j] = s; }
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < L; k++) s += GQ[i * L + k] * (dt * fwf[j * L + k]); Pn[i * N + j] += s; }
    std::memcpy(P, Pn, sizeof(Pn));
  }

  // left-multiply rows [I_ROT, I_ROT+3) of an (N x cols) matrix by M (3x3)
  static void rot_rows(double* X, int cols, const double* M, const double* src) {
    for (int c = 0; c < cols; c++) {
      const double a = src[(I_ROT)*cols + c], b = src[(I_ROT + 1) * cols + c], d = src[(I_ROT + 2) * cols + c];
      for (int r = 0; r < 3; r++) X[(I_ROT + r) * cols + c] = M[r * 3] * a + M[r * 3 + 1] * b + M[r * 3 + 2] * d;
    }
  }
  // right-multiply columns [I_ROT, I_ROT+3) of an (N x N) matrix by M^T
  static void rot_cols(double* X, const double* M) {
    for (int i = 0; i < N; i++) {
      const double a = X[i * N + I_ROT], b = X[i * N + I_ROT + 1], d = X[i * N + I_ROT + 2];
      for (int r = 0; r < 3; r++) X[i * N + I_ROT + r] = a * M[r * 3] + b * M[r * 3 + 1] + d * M[r * 3 + 2];
    }
  }

  // esekfom.hpp:406-703 for this state and measurement (pos, rot): h_x = [I 0 ...; 0 I 0 ...], h_v = I
  void update_iterated(const double* zp, const Quat& zq, const double* R) {
    int t = 0;
    const State x_prop = x;
    double P_prop[N * N]; std::memcpy(P_prop, P, sizeof(P_prop));
    for (int it = 0; it < opt.maximum_iteration; it++) {
      double dx[N], dx_new[N];
      x.minus(x_prop, dx);
      std::memcpy(dx_new, dx, sizeof(dx));
      std::memcpy(P, P_prop, sizeof(P_prop));
      double At[9];
      { double A[9]; A_matrix(dx + I_ROT, A); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i * 3 + j] = A[j * 3 + i]; }
      for (int r = 0; r < 3; r++) dx_new[I_ROT + r] = At[r * 3] * dx[I_ROT] + At[r * 3 + 1] * dx[I_ROT + 1] + At[r * 3 + 2] * dx[I_ROT + 2];
      { double tmp[N * N]; std::memcpy(tmp, P, sizeof(tmp)); rot_rows(P, N, At, tmp); }
      rot_cols(P, At);
      // K = P H^T (H P H^T + R)^-1 : H picks the first six rows / columns
      double S[L * L], Si[L * L], K[N * L];
      for (int i = 0; i < L; i++) for (int j = 0; j < L; j++) S[i * L + j] = P[i * N + j] + R[i * L + j];
      invert(S, Si, L);
      for (int i = 0; i < N; i++) for (int j = 0; j < L; j++) { double s = 0; for (int k = 0; k < L; k++) s += P[i * N + k] * Si[k * L + j]; K[i * L + j] = s; }
      double innov[L];
      for (int i = 0; i < 3; i++) innov[i] = zp[i] - x.pos[i];
      so3_log(qmul(qconj(x.rot), zq), innov + 3);
      // dx_ = K innov + (K H - I) dx_new
      double dxu[N];
      for (int i = 0; i < N; i++) {
        double s = 0;
        for (int k = 0; k < L; k++) s += K[i * L + k] * innov[k];
        double s2 = 0;
        for (int j = 0; j < N; j++) s2 += ((j < L ? K[i * L + j] : 0.0) - (i == j ? 1.0 : 0.0)) * dx_new[j];
        dxu[i] = s + s2;
      }
      x.plus(dxu, 1.0);
      bool converg = true;
      for (int i = 0; i < N; i++) if (std::fabs(dxu[i]) > opt.convergence_limit) { converg = false; break; }
      if (converg) t++;
      if (t > 1 || it == opt.maximum_iteration - 1) {
        double Lm[N * N]; std::memcpy(Lm, P, sizeof(Lm));
        double A[9], At2[9]; A_matrix(dxu + I_ROT, A);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At2[i * 3 + j] = A[j * 3 + i];
        rot_rows(Lm, N, At2, P);                                       // L rows = At * P rows
        { double tmp[N * L]; std::memcpy(tmp, K, sizeof(tmp)); rot_rows(K, L, At2, tmp); }   // K rows = At * K rows
        rot_cols(Lm, At2); rot_cols(P, At2);
        // P = L - K H P
        double Pn[N * N];
        for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { double s = 0; for (int k = 0; k < L; k++) s += K[i * L + k] * P[k * N + j]; Pn[i * N + j] = Lm[i * N + j] - s; }
        std::memcpy(P, Pn, sizeof(Pn));
        return;
      }
    }
  }

  int process_measurement(double stamp, const double* p, const Quat& q, const double* Rin) {   // eskf.hpp:108-147
    double R[L * L];
    if (Rin) std::memcpy(R, Rin, sizeof(R));
    else { for (int i = 0; i < L * L; i++) R[i] = 0; for (int i = 0; i < 3; i++) { R[i * 7] = opt.r_position_std * opt.r_position_std; R[(3 + i) * 7] = opt.r_rotation_std * opt.r_rotation_std; } }
    if (!initialized) { initialize(stamp, p, q); return 1; }
    const double dt = stamp - last_time;
    if (dt <= 0.0 || !std::isfinite(dt)) return 0;
    if (dt > opt.max_dt) { initialize(stamp, p, q); return 1; }
    predict(dt);
    for (int i = 0; i < L; i++) if (!std::isfinite(R[i * 7]) || R[i * 7] < 1e-12) R[i * 7] = 1e-12;   // sanitizeMeasurementNoise :292-299
    update_iterated(p, normalized_quaternion(q), R);
    last_time = stamp;
    return 1;
  }
  int state_predict(double stamp) {   // :149-171
    if (!initialized) return 0;
    const double dt = stamp - last_time;
    if (dt <= 0.0 || !std::isfinite(dt)) return 0;
    if (dt > opt.max_dt) return 0;
    predict(dt);
    last_time = stamp;
    return 1;
  }
};

namespace {

struct Aff { float m[16]; };
Aff aff_identity() { Aff a; for (int i = 0; i < 16; i++) a.m[i] = (i % 5 == 0) ? 1.f : 0.f; return a; }
Aff get_transformation(float x, float y, float z, float roll, float pitch, float yaw) {   // pcl::getTransformation
  const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll);
  const float DE = D * E, DF = D * F;
  Aff t;
  t.m[0] = A * C; t.m[1] = A * DF - B * E; t.m[2] = B * F + A * DE; t.m[3] = x;
  t.m[4] = B * C; t.m[5] = A * E + B * DF; t.m[6] = B * DE - A * F; t.m[7] = y;
  t.m[8] = -D; t.m[9] = C * F; t.m[10] = C * E; t.m[11] = z;
  t.m[12] = 0; t.m[13] = 0; t.m[14] = 0; t.m[15] = 1;
  return t;
}
Aff aff_mul(const Aff& a, const Aff& b) {
  Aff c;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0; for (int k = 0; k < 4; k++) s += a.m[i * 4 + k] * b.m[k * 4 + j]; c.m[i * 4 + j] = s; }
  return c;
}
Aff aff_inverse(const Aff& T) {   // Eigen::Transform<float,3,Affine>::inverse()
  const float* a = T.m; float c[9];
  c[0] = a[5] * a[10] - a[6] * a[9]; c[1] = a[2] * a[9] - a[1] * a[10]; c[2] = a[1] * a[6] - a[2] * a[5];
  c[3] = a[6] * a[8] - a[4] * a[10]; c[4] = a[0] * a[10] - a[2] * a[8]; c[5] = a[2] * a[4] - a[0] * a[6];
  c[6] = a[4] * a[9] - a[5] * a[8]; c[7] = a[1] * a[8] - a[0] * a[9]; c[8] = a[0] * a[5] - a[1] * a[4];
  const float det = a[0] * c[0] + a[1] * c[3] + a[2] * c[6], inv = 1.0f / det;
  Aff o;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[i * 4 + j] = c[i * 3 + j] * inv;
  for (int i = 0; i < 3; i++) o.m[i * 4 + 3] = -(o.m[i * 4] * a[3] + o.m[i * 4 + 1] * a[7] + o.m[i * 4 + 2] * a[11]);
  o.m[12] = o.m[13] = o.m[14] = 0; o.m[15] = 1;
  return o;
}
// tf::Matrix3x3(q).getRPY (setRotation + getEulerYPR, solution 1)
void get_rpy(const double q[4], double& roll, double& pitch, double& yaw) {
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs, xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs, yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
  if (std::fabs(m20) >= 1.0) { yaw = 0.0; const double delta = std::atan2(m21, m22); pitch = m20 < 0 ? M_PI / 2.0 : -M_PI / 2.0; roll = delta; return; }
  pitch = -std::asin(m20);
  const double c = std::cos(pitch);
  roll = std::atan2(m21 / c, m22 / c);
  yaw = std::atan2(m10 / c, m00 / c);
}
// odom2affine (lidarOdometry.cpp:34-45): doubles narrowed to the float arguments of pcl::getTransformation
Aff odom2affine(const double* p, const double* q) {
  double r, pi, y; get_rpy(q, r, pi, y);
  return get_transformation((float)p[0], (float)p[1], (float)p[2], (float)r, (float)pi, (float)y);
}
// affineToPose :127-136. affine.rotation() of an Affine3f is Eigen's float polar factor of the linear part (polar_f32.hpp), cast to double,
// then Quaterniond(rotation).normalize()
void affine_to_pose(const Aff& a, double* pos, Quat& q) {
  pos[0] = a.m[3]; pos[1] = a.m[7]; pos[2] = a.m[11];
  float L[9], Rf[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) L[i * 3 + j] = a.m[i * 4 + j];
  rolo::polar::rotation_f32(L, Rf);
  double R[9];
  for (int i = 0; i < 9; i++) R[i] = (double)Rf[i];
  q = qnormalized(R_to_q(R));
}

struct OdomMsg { double stamp; double p[3]; double q[4]; };

}  // namespace

struct rolo_fusion {
  rolo_eskf pose_regulator;
  Aff mappingOdomAffine = aff_identity(), lidarOdomAffineFront = aff_identity();
  double mappingOdomTime = -1, lastProcessedLidarTime = -1, lastPathTime = -1;
  std::deque<OdomMsg> lidarOdomQueue;
  std::deque<double> path_stamps;   // lidarPath.poses (their stamps; the poses themselves go out through rolo_fusion_odometry)
};

extern "C" {

void rolo_eskf_default_options(rolo_eskf_options* o) {
  if (!o) return;
  o->max_dt = 1.0; o->q_linear_jerk_std = 0.5; o->q_angular_jerk_std = 0.5; o->r_position_std = 0.20; o->r_rotation_std = 0.10;
  o->init_position_std = 0.05; o->init_rotation_std = 0.05; o->init_velocity_std = 5.0; o->init_angular_velocity_std = 2.0;
  o->init_acceleration_std = 5.0; o->init_angular_acceleration_std = 2.0; o->maximum_iteration = 3; o->convergence_limit = 1e-4;
}
int rolo_eskf_create(const rolo_eskf_options* o, rolo_eskf** out) {
  if (!out) return ROLO_EINVAL;
  rolo_eskf* f = new rolo_eskf();
  if (o) f->opt = *o; else rolo_eskf_default_options(&f->opt);
  f->configure();
  *out = f;
  return ROLO_OK;
}
void rolo_eskf_destroy(rolo_eskf* f) { delete f; }
int rolo_eskf_copy(const rolo_eskf* src, rolo_eskf* dst) { if (!src || !dst) return ROLO_EINVAL; *dst = *src; return ROLO_OK; }
void rolo_eskf_reset(rolo_eskf* f) { if (f) f->reset(); }
int rolo_eskf_initialized(const rolo_eskf* f) { return f && f->initialized ? 1 : 0; }
double rolo_eskf_last_time(const rolo_eskf* f) { return f ? f->last_time : 0.0; }
int rolo_eskf_process_measurement(rolo_eskf* f, double stamp, const double* p, const double* q, const double* R36) {
  if (!f || !p || !q) return ROLO_EINVAL;
  return f->process_measurement(stamp, p, Quat{q[3], q[0], q[1], q[2]}, R36);
}
int rolo_eskf_state_predict(rolo_eskf* f, double stamp) { return f ? f->state_predict(stamp) : ROLO_EINVAL; }
void rolo_eskf_get_state(const rolo_eskf* f, double* p, double* q, double* v, double* w, double* a, double* al) {
  if (!f) return;
  const Quat n = qnormalized(f->x.rot);
  for (int i = 0; i < 3; i++) { if (p) p[i] = f->x.pos[i]; if (v) v[i] = f->x.vel[i]; if (w) w[i] = f->x.omega[i]; if (a) a[i] = f->x.acc[i]; if (al) al[i] = f->x.alpha[i]; }
  if (q) { q[0] = n.x; q[1] = n.y; q[2] = n.z; q[3] = n.w; }
}
void rolo_eskf_get_covariance(const rolo_eskf* f, double* P) { if (f && P) std::memcpy(P, f->P, sizeof(f->P)); }
int rolo_eskf_state_propagate(const rolo_eskf* f, double dt, double dis, double* poses7, int cap) {   // eskf.hpp:213-246
  if (!f) return ROLO_EINVAL;
  if (!f->initialized || dt <= 0.0 || dis <= 0.0 || !std::isfinite(dt) || !std::isfinite(dis)) return 0;
  State s = f->x;
  double last[3] = {s.pos[0], s.pos[1], s.pos[2]}, propagated = 0.0;
  int n = 0;
  while (propagated < dis) {
    double dx[N]; process_model(s, dt, dx);
    s.plus(dx, dt);
    const double d0 = s.pos[0] - last[0], d1 = s.pos[1] - last[1], d2 = s.pos[2] - last[2];
    const double step = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (!std::isfinite(step) || step < 1e-12) break;
    propagated += step;
    for (int i = 0; i < 3; i++) last[i] = s.pos[i];
    const Quat q = qnormalized(s.rot);
    if (poses7 && n < cap) { double* o = poses7 + 7 * (size_t)n; o[0] = s.pos[0]; o[1] = s.pos[1]; o[2] = s.pos[2]; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w; }
    n++;
    if (n > (1 << 20)) break;   // guard: the reference loop is unbounded when the state barely moves
  }
  return n;
}

int rolo_fusion_create(const rolo_eskf_options* o, rolo_fusion** out) {
  if (!out) return ROLO_EINVAL;
  rolo_fusion* f = new rolo_fusion();
  if (o) f->pose_regulator.opt = *o; else rolo_eskf_default_options(&f->pose_regulator.opt);
  f->pose_regulator.configure();
  *out = f;
  return ROLO_OK;
}
void rolo_fusion_destroy(rolo_fusion* f) { delete f; }
rolo_eskf* rolo_fusion_filter(rolo_fusion* f) { return f ? &f->pose_regulator : nullptr; }
int rolo_fusion_mapping_odometry(rolo_fusion* f, double stamp, const double* p, const double* q) {   // :109-117
  if (!f || !p || !q) return ROLO_EINVAL;
  f->mappingOdomAffine = odom2affine(p, q);
  f->mappingOdomTime = stamp;
  return ROLO_OK;
}
int rolo_fusion_lidar_odometry(rolo_fusion* f, double stamp, const double* p, const double* q) {   // :119-125
  if (!f || !p || !q) return ROLO_EINVAL;
  OdomMsg m; m.stamp = stamp; std::memcpy(m.p, p, sizeof(m.p)); std::memcpy(m.q, q, sizeof(m.q));
  f->lidarOdomQueue.push_back(m);
  return ROLO_OK;
}
int rolo_fusion_timer(rolo_fusion* f, double now, rolo_fusion_odometry* out) {   // fusionTimerHandler :138-241
  if (!f || !out) return ROLO_EINVAL;
  if (f->mappingOdomTime == -1) return 0;
  while (!f->lidarOdomQueue.empty() && f->lidarOdomQueue.front().stamp <= f->mappingOdomTime) f->lidarOdomQueue.pop_front();
  if (f->lidarOdomQueue.empty()) return 0;
  f->lidarOdomAffineFront = odom2affine(f->lidarOdomQueue.front().p, f->lidarOdomQueue.front().q);
  const bool has_new = f->lidarOdomQueue.back().stamp > f->lastProcessedLidarTime;
  if (has_new) {
    const OdomMsg& lo = f->lidarOdomQueue.back();
    double mp[3]; Quat mq;
    affine_to_pose(odom2affine(lo.p, lo.q), mp, mq);
    if (f->pose_regulator.process_measurement(lo.stamp, mp, mq, nullptr)) f->lastProcessedLidarTime = lo.stamp;
  }
  if (!f->pose_regulator.initialized) return 0;
  rolo_eskf preview = f->pose_regulator;
  preview.state_predict(now);
  const Quat pq = qnormalized(preview.x.rot);
  Aff back = aff_identity();
  double R[9]; q_to_R(pq, R);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) back.m[i * 4 + j] = (float)R[i * 3 + j]; back.m[i * 4 + 3] = (float)preview.x.pos[i]; }
  const Aff incre = aff_mul(aff_inverse(f->lidarOdomAffineFront), back);
  const Aff last = aff_mul(f->mappingOdomAffine, incre);
  Quat oq; affine_to_pose(last, out->position, oq);
  out->orientation[0] = oq.x; out->orientation[1] = oq.y; out->orientation[2] = oq.z; out->orientation[3] = oq.w;
  for (int i = 0; i < 3; i++) out->velocity[i] = preview.x.vel[i];
  out->speed = std::sqrt(out->velocity[0] * out->velocity[0] + out->velocity[1] * out->velocity[1] + out->velocity[2] * out->velocity[2]);
  out->path_appended = 0;
  if (now - f->lastPathTime > 0.05) {
    f->lastPathTime = now;
    f->path_stamps.push_back(now);
    while (!f->path_stamps.empty() && f->path_stamps.front() < now - 1.0) f->path_stamps.pop_front();
    out->path_appended = 1;
  }
  out->path_length = (int)f->path_stamps.size();
  return 1;
}
int rolo_fusion_predict_timer(rolo_fusion* f, rolo_future_point* pts, int cap) {   // predictTimerHandler :243-322
  if (!f) return ROLO_EINVAL;
  const rolo_eskf& kf = f->pose_regulator;
  if (!kf.initialized) return 0;
  const int n = rolo_eskf_state_propagate(&kf, 0.2, 8.0, nullptr, 0);
  if (n <= 0) return 0;
  std::vector<double> poses((size_t)n * 7);
  rolo_eskf_state_propagate(&kf, 0.2, 8.0, poses.data(), n);
  const Quat cq = qnormalized(kf.x.rot);
  double Rc[9]; q_to_R(cq, Rc);
  double local_v[3];
  for (int i = 0; i < 3; i++) local_v[i] = Rc[i] * kf.x.vel[0] + Rc[3 + i] * kf.x.vel[1] + Rc[6 + i] * kf.x.vel[2];   // R^T v
  const double heading_rate = kf.x.omega[2];
  for (int i = 0; i < n && i < cap; i++) {
    const double* p = &poses[7 * (size_t)i];
    const Quat fq = qnormalized(Quat{p[6], p[3], p[4], p[5]});
    double Rf[9]; q_to_R(fq, Rf);
    // current_pose.inverse() * future_pose (Affine3d: rotation part R_c^T R_f, translation R_c^T (t_f - t_c))
    double Rl[9], tl[3];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) Rl[r * 3 + c] = Rc[r] * Rf[c] + Rc[3 + r] * Rf[3 + c] + Rc[6 + r] * Rf[6 + c];
      tl[r] = Rc[r] * (p[0] - kf.x.pos[0]) + Rc[3 + r] * (p[1] - kf.x.pos[1]) + Rc[6 + r] * (p[2] - kf.x.pos[2]);
    }
    const Quat lq = qnormalized(R_to_q(Rl));
    rolo_future_point& o = pts[i];
    o.position[0] = tl[0]; o.position[1] = tl[1]; o.position[2] = 0.0;
    o.orientation[0] = lq.x; o.orientation[1] = lq.y; o.orientation[2] = lq.z; o.orientation[3] = lq.w;
    o.longitudinal_velocity_mps = local_v[0]; o.lateral_velocity_mps = local_v[1]; o.heading_rate_rps = heading_rate;
    o.is_final = (i + 1 == n) ? 1 : 0;
  }
  return n;
}

}  // extern "C"
