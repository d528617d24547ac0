// Eigen::Transform<float, 3, Affine>::rotation() for the host-side pose chains (odometry.hip, fusion.hip).
//
// The reference reads rotations off its float Affine3f poses with .rotation() (src/lidarOdometry.cpp:130, :474, :548), which for an Affine
// (not Isometry) transform is NOT the linear part: Eigen runs computeRotationScaling() — JacobiSVD<Matrix3f>(linear(), ComputeFullU |
// ComputeFullV), x = det(U V^T) < 0 ? -1 : +1, U.col(2) *= x, rotation = U V^T — i.e. the polar factor, computed in float. For the
// rotation-like matrices of the pose chain the two differ in the last float ulps; this header restates Eigen's algorithm (two-sided Jacobi
// on the 3x3 in float: sweep order (1,0), (2,0), (2,1); threshold max(FLT_MIN, 2 eps max|diag|); 2x2 step = symmetrising rotation + Jacobi
// rotation; singular values made positive, sorted descending with the column swaps) so that the HIP path takes the same value as the
// reference would, not a neighbour of it. Eigen's sources are not in /root/reference; restated from its published algorithm
// (Eigen/src/SVD/JacobiSVD.h, Eigen/src/Jacobi/Jacobi.h, Eigen/src/Geometry/Transform.h).
// EIGEN VERSION ASSUMED: 3.4 (computeRotationScaling as above: x = sign of det, U.col(2) *= x). Eigen 3.3.x — the stock version of ROS1 Noetic /
// Ubuntu 20.04, which the reference's README floor ("Eigen >= 3.3.7") admits — uses x = det(U V^T) itself (about +-1, not exactly) and
// m.col(0) /= x: with 3.3 the Rotation at lidarOdometry.cpp:474 / :130 differs from this restatement in the last float ulps (far inside the
// 1e-5 rad bar; the oracle shares this restatement, so only a dump of the real reference built with the target Eigen —
// tools/dump_reference_golden.cpp — can tell the two apart).
#pragma once
#include <cfloat>
#include <cmath>
#include <utility>

namespace rolo {
namespace polar {

struct Rot2 { float c, s; };

// JacobiRotation::makeJacobi(x, y, z) for the symmetric 2x2 [[x, y], [y, z]]
inline Rot2 make_jacobi(float x, float y, float z) {
  const float deno = 2.0f * std::fabs(y);
  if (deno < FLT_MIN) return Rot2{1.0f, 0.0f};
  const float tau = (x - z) / deno;
  const float w = std::sqrt(tau * tau + 1.0f);
  const float t = tau > 0.0f ? 1.0f / (tau + w) : 1.0f / (tau - w);
  const float sign_t = t > 0.0f ? 1.0f : -1.0f;
  const float n = 1.0f / std::sqrt(t * t + 1.0f);
  return Rot2{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}

// rows p, q of M (row-major 3x3):  (x, y) <- (c x + s y, -s x + c y)       [applyOnTheLeft(p, q, j)]
inline void rot_rows(float* M, int p, int q, Rot2 j) {
  for (int k = 0; k < 3; k++) { const float x = M[p * 3 + k], y = M[q * 3 + k]; M[p * 3 + k] = j.c * x + j.s * y; M[q * 3 + k] = -j.s * x + j.c * y; }
}
// columns p, q of M with the transposed rotation: (x, y) <- (c x - s y, s x + c y)     [applyOnTheRight(p, q, j)]
inline void rot_cols(float* M, int p, int q, Rot2 j) {
  for (int k = 0; k < 3; k++) { const float x = M[k * 3 + p], y = M[k * 3 + q]; M[k * 3 + p] = j.c * x - j.s * y; M[k * 3 + q] = j.s * x + j.c * y; }
}

// JacobiSVD<Matrix3f, FullU | FullV> of the row-major 3x3 A: A = U diag(sv) V^T, sv descending
inline void jacobi_svd3_f32(const float* A, float* U, float* sv, float* V) {
  float scale = 0.0f;
  for (int i = 0; i < 9; i++) scale = std::fmax(scale, std::fabs(A[i]));
  if (!(scale > 0.0f) || !std::isfinite(scale)) scale = 1.0f;
  float W[9];
  for (int i = 0; i < 9; i++) { W[i] = A[i] / scale; U[i] = V[i] = (i % 4 == 0) ? 1.0f : 0.0f; }
  const float precision = 2.0f * FLT_EPSILON;
  float max_diag = std::fmax(std::fabs(W[0]), std::fmax(std::fabs(W[4]), std::fabs(W[8])));
  bool finished = false;
  for (int sweep = 0; !finished && sweep < 100; sweep++) {
    finished = true;
    for (int p = 1; p < 3; p++) {
      for (int q = 0; q < p; q++) {
        const float threshold = std::fmax(FLT_MIN, precision * max_diag);
        if (std::fabs(W[p * 3 + q]) > threshold || std::fabs(W[q * 3 + p]) > threshold) {
          finished = false;
          // real_2x2_jacobi_svd: rot1 makes the 2x2 block symmetric, j_right diagonalises it, j_left = rot1 * j_right^T
          const float m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
          Rot2 rot1;
          const float t = m00 + m11, d = m10 - m01;
          if (std::fabs(d) < FLT_MIN) rot1 = Rot2{1.0f, 0.0f};
          else { const float u = t / d, tmp = std::sqrt(1.0f + u * u); rot1 = Rot2{u / tmp, 1.0f / tmp}; }
          const float n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11, n11 = -rot1.s * m01 + rot1.c * m11;
          const Rot2 jr = make_jacobi(n00, n01, n11);
          const Rot2 jl{rot1.c * jr.c + rot1.s * jr.s, -rot1.c * jr.s + rot1.s * jr.c};
          rot_rows(W, p, q, jl);
          rot_cols(U, p, q, Rot2{jl.c, -jl.s});
          rot_cols(W, p, q, jr);
          rot_cols(V, p, q, jr);
          max_diag = std::fmax(max_diag, std::fmax(std::fabs(W[p * 3 + p]), std::fabs(W[q * 3 + q])));
        }
      }
    }
  }
  for (int i = 0; i < 3; i++) {
    const float a = std::fabs(W[i * 3 + i]);
    sv[i] = a;
    if (a != 0.0f) { const float sgn = W[i * 3 + i] / a; for (int k = 0; k < 3; k++) U[k * 3 + i] *= sgn; }
  }
  for (int i = 0; i < 3; i++) sv[i] *= scale;
  for (int i = 0; i < 3; i++) {
    int pos = i; float mx = sv[i];
    for (int k = i + 1; k < 3; k++) if (sv[k] > mx) { mx = sv[k]; pos = k; }
    if (mx == 0.0f) break;
    if (pos != i) { std::swap(sv[i], sv[pos]); for (int k = 0; k < 3; k++) { std::swap(U[k * 3 + i], U[k * 3 + pos]); std::swap(V[k * 3 + i], V[k * 3 + pos]); } }
  }
}

// Transform::rotation() of an Affine3f whose linear part is the row-major 3x3 L
inline void rotation_f32(const float* L, float* R) {
  float U[9], V[9], sv[3];
  jacobi_svd3_f32(L, U, sv, V);
  float UVt[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) UVt[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
  const float det = UVt[0] * (UVt[4] * UVt[8] - UVt[5] * UVt[7]) - UVt[1] * (UVt[3] * UVt[8] - UVt[5] * UVt[6]) + UVt[2] * (UVt[3] * UVt[7] - UVt[4] * UVt[6]);
  const float x = det < 0.0f ? -1.0f : 1.0f;
  for (int k = 0; k < 3; k++) U[k * 3 + 2] *= x;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
}

}  // namespace polar
}  // namespace rolo
