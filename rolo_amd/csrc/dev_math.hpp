// Device-side fp64 small-matrix helpers for the registration kernels (gfx950). Symmetric 3x3 matrices are
// kept as 6 values in the order xx, xy, xz, yy, yz, zz.
#pragma once
#include <hip/hip_runtime.h>

#define ROLO_DEV __device__ __forceinline__

namespace rolo {

// floats as order-preserving ints (min / max by integer compare, atomics)
ROLO_DEV int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
ROLO_DEV float ord2f(int k) { int i = k >= 0 ? k : k ^ 0x7fffffff; return __int_as_float(i); }


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

// The value lane (l ^ OFF) holds, without the LDS crossbar: gfx950's v_permlane32_swap / v_permlane16_swap for the
// cross-row distances, DPP row rotate / shifts / quad_perm inside a row of 16.
template <int OFF>
ROLO_DEV int lane_xor_b32(int x) {
  if constexpr (OFF == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);  // {vdst', src'}: vdst'[32..63] = x[0..31], src'[0..31] = x[32..63]
    return (int)((threadIdx.x & 32) ? r[0] : r[1]);
  } else if constexpr (OFF == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);  // odd rows of vdst' <- even rows of x, even rows of src' <- odd rows of x
    return (int)((threadIdx.x & 16) ? r[0] : r[1]);
  } else if constexpr (OFF == 8) {
    return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);          // row_ror:8
  } else if constexpr (OFF == 4) {
    const int a = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);   // row_shl:4 into banks 0 and 2 (lane <- lane + 4)
    return __builtin_amdgcn_update_dpp(a, x, 0x114, 0xf, 0xa, false);          // row_shr:4 into banks 1 and 3 (lane <- lane - 4)
  } else if constexpr (OFF == 2) {
    return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);           // quad_perm [2,3,0,1]
  } else {
    return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);           // quad_perm [1,0,3,2]
  }
}
template <int OFF>
ROLO_DEV double lane_xor_f64(double v) {
  return __hiloint2double(lane_xor_b32<OFF>(__double2hiint(v)), lane_xor_b32<OFF>(__double2loint(v)));
}

template <int OFF>
ROLO_DEV unsigned long long lane_xor_u64(unsigned long long v) {
  const unsigned lo = (unsigned)lane_xor_b32<OFF>((int)(unsigned)(v & 0xffffffffull)), hi = (unsigned)lane_xor_b32<OFF>((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

}  // namespace rolo
