// Device-side fp64 small-matrix helpers for the registration kernels (gfx950). Symmetric 3x3 matrices are
// kept as 6 values in the order xx, xy, xz, yy, yz, zz.
#pragma once
#include <hip/hip_runtime.h>

#define ROLO_DEV __device__ __forceinline__

namespace rolo {

struct Sym3 { double xx, xy, xz, yy, yz, zz; };
struct Vec3 { double x, y, z; };
struct Mat3 { double m[9]; };  // row-major

ROLO_DEV Vec3 mat3_mulv(const double* R, const Vec3& p) {
  return Vec3{R[0] * p.x + R[1] * p.y + R[2] * p.z, R[3] * p.x + R[4] * p.y + R[5] * p.z, R[6] * p.x + R[7] * p.y + R[8] * p.z};
}

// R * C * R^T for symmetric C
ROLO_DEV Sym3 sym3_rotate(const double* R, const Sym3& C) {
  // A = R * C
  double a00 = R[0] * C.xx + R[1] * C.xy + R[2] * C.xz, a01 = R[0] * C.xy + R[1] * C.yy + R[2] * C.yz, a02 = R[0] * C.xz + R[1] * C.yz + R[2] * C.zz;
  double a10 = R[3] * C.xx + R[4] * C.xy + R[5] * C.xz, a11 = R[3] * C.xy + R[4] * C.yy + R[5] * C.yz, a12 = R[3] * C.xz + R[4] * C.yz + R[5] * C.zz;
  double a20 = R[6] * C.xx + R[7] * C.xy + R[8] * C.xz, a21 = R[6] * C.xy + R[7] * C.yy + R[8] * C.yz, a22 = R[6] * C.xz + R[7] * C.yz + R[8] * C.zz;
  Sym3 o;
  o.xx = a00 * R[0] + a01 * R[1] + a02 * R[2];
  o.xy = a00 * R[3] + a01 * R[4] + a02 * R[5];
  o.xz = a00 * R[6] + a01 * R[7] + a02 * R[8];
  o.yy = a10 * R[3] + a11 * R[4] + a12 * R[5];
  o.yz = a10 * R[6] + a11 * R[7] + a12 * R[8];
  o.zz = a20 * R[6] + a21 * R[7] + a22 * R[8];
  return o;
}

ROLO_DEV Sym3 sym3_add(const Sym3& a, const Sym3& b) { return Sym3{a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz}; }

// cofactor inverse (the 4x4 [A 0; 0 1] inverse of rot_vgicp_impl.hpp:215-219 restricted to its 3x3 block)
ROLO_DEV Sym3 sym3_inverse(const Sym3& A) {
  double c00 = A.yy * A.zz - A.yz * A.yz;
  double c01 = A.xz * A.yz - A.xy * A.zz;
  double c02 = A.xy * A.yz - A.xz * A.yy;
  double c11 = A.xx * A.zz - A.xz * A.xz;
  double c12 = A.xz * A.xy - A.xx * A.yz;
  double c22 = A.xx * A.yy - A.xy * A.xy;
  double det = A.xx * c00 + A.xy * c01 + A.xz * c02;
  double inv = 1.0 / det;
  return Sym3{c00 * inv, c01 * inv, c02 * inv, c11 * inv, c12 * inv, c22 * inv};
}

ROLO_DEV Vec3 sym3_mulv(const Sym3& M, const Vec3& e) {
  return Vec3{M.xx * e.x + M.xy * e.y + M.xz * e.z, M.xy * e.x + M.yy * e.y + M.yz * e.z, M.xz * e.x + M.yz * e.y + M.zz * e.z};
}
ROLO_DEV double dot3(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// wave64 butterfly sum
ROLO_DEV double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace rolo
