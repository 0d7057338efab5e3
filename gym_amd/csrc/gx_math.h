// Scalar math pieces shared by the kernels.  Plain C++ with explicit fma(), so
// the same text compiles for the device (hipcc) and for the host, where
// tests/host_math_check.cpp checks it against the platform libm.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include "gx_sincos_table.h"

#if defined(__HIPCC__)
#define GX_HD __host__ __device__ __forceinline__
#else
#define GX_HD inline
#endif

namespace gx {
namespace math {

GX_HD uint64_t f64_bits(double x) {
  uint64_t b;
  memcpy(&b, &x, 8);
  return b;
}
GX_HD double f64_from_bits(uint64_t b) {
  double x;
  memcpy(&x, &b, 8);
  return x;
}

// From sin and cos of the reduced argument to those of x = r + q pi/2: swap when q is odd,
// then the signs.  Written on the bit patterns (mask selects and a sign-bit xor) so that no
// comparison results have to be kept in scalar registers across the polynomial evaluations.
// (x & mask) | (y & ~mask)
GX_HD uint32_t bit_select(uint32_t mask, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(x), "v"(y));
  return d;
#else
  return (x & mask) | (y & ~mask);
#endif
}
GX_HD void quadrant(int q, double sn, double cs, double& s, double& c) {
  const uint32_t odd = 0u - ((uint32_t)q & 1u);
  const uint64_t bs = f64_bits(sn), bc = f64_bits(cs);
  const uint32_t sl = (uint32_t)bs, sh = (uint32_t)(bs >> 32), cl = (uint32_t)bc, ch = (uint32_t)(bc >> 32);
  const uint32_t al = bit_select(odd, cl, sl), ah = bit_select(odd, ch, sh);
  const uint32_t bl = bit_select(odd, sl, cl), bh = bit_select(odd, sh, ch);
  const uint32_t sign_s = ((uint32_t)q & 2u) << 30, sign_c = (((uint32_t)q + 1u) & 2u) << 30;
  s = f64_from_bits(((uint64_t)(ah ^ sign_s) << 32) | al);
  c = f64_from_bits(((uint64_t)(bh ^ sign_c) << 32) | bl);
}

// a * b + k for a compile-time constant k.  Written as one v_fma_f64 with the constant in a scalar
// register pair: left to itself the compiler puts such a constant into a vector register pair
// (two moves) and accumulates into it, three vector instructions per polynomial term.
GX_HD double fma_k(double a, double b, double k) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
#else
  return fma(a, b, k);
#endif
}

// sin(x) and cos(x) in binary64 for |x| <= pi/4, no argument reduction.
// Minimax kernels of the classic fdlibm form (sin: x + x^3 S(x^2), cos:
// 1 - x^2/2 + x^4 C(x^2)) evaluated by Horner's rule with fused multiply-adds;
// the error stays under one unit in the last place of the double, so rounding
// the result to float gives the same value as rounding a correctly rounded
// double except when the true value lies within ~2^-29 (relative) of a float
// rounding boundary.
GX_HD void sincos_small(double x, double& s, double& c) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  double rs = fma_k(z, S6, S5);
  rs = fma_k(z, rs, S4);
  rs = fma_k(z, rs, S3);
  rs = fma_k(z, rs, S2);
  rs = fma_k(z, rs, S1);
  s = fma(z * x, rs, x);
  double rc = fma_k(z, C6, C5);
  rc = fma_k(z, rc, C4);
  rc = fma_k(z, rc, C3);
  rc = fma_k(z, rc, C2);
  rc = fma_k(z, rc, C1);
  // 1 + z (-1/2 + z rc): for |x| <= pi/4 the inner term is below 0.31, so its rounding error
  // reaches the result scaled by z <= 0.62 and the sum rounds once
  c = fma(z, fma_k(z, rc, -0.5), 1.0);
}

// sin(x) and cos(x) in binary64 for |x| up to a few thousand (the angles the
// dynamics can produce: everything derived from atan2 plus one integration
// step stays below 16).  Cody-Waite reduction against pi/2 split in three
// parts -- the first has 33 significant bits, so n * P1 is exact for |n| < 2^20
// and x - n P1 is exact by cancellation -- carried as a double-double
// (r, tail), then the same minimax kernels as sincos_small with the first
// order tail terms, then the quadrant.  No table, no branch; inf/NaN give NaN.
GX_HD void sincos_bounded(double x, double& s, double& c) {
  const double INV_PIO2 = 0x1.45f306dc9c883p-1;
  const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a626331p-34, P3 = 0x1.1701b839a2520p-88;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  // round(x * 2/pi) by adding and subtracting 1.5 * 2^52; the quadrant number is in the sum's low bits
  const double MAGIC = 0x1.8p+52;
  const double tq = fma(x, INV_PIO2, MAGIC);
  const double fn = tq - MAGIC;
  const double r0 = fma(-fn, P1, x);           // exact
  const double r = fma(-fn, P2, r0);           // head of the reduced argument
  double t = fma(-fn, P2, r0 - r);             // what the rounding of r dropped
  t = fma(-fn, P3, t);                         // tail
  const double z = r * r;
  double rs = fma_k(z, S6, S5);
  rs = fma_k(z, rs, S4);
  rs = fma_k(z, rs, S3);
  rs = fma_k(z, rs, S2);
  rs = fma_k(z, rs, S1);
  // sin(r + t) = sin r + t cos r: r + r^3 S(z) rounds once, t (1 - z/2) is below 2^-53 of it
  const double sn = fma(t, fma(-0.5, z, 1.0), fma(z * r, rs, r));
  double rc = fma_k(z, C6, C5);
  rc = fma_k(z, rc, C4);
  rc = fma_k(z, rc, C3);
  rc = fma_k(z, rc, C2);
  rc = fma_k(z, rc, C1);
  // cos(r + t) = cos r - t sin r: 1 + z (-1/2 + z rc) rounds once, r t is below 2^-53
  const double cs = fma(-r, t, fma(z, fma_k(z, rc, -0.5), 1.0));
  quadrant((int)(uint32_t)f64_bits(tq), sn, cs, s, c);
}

// sin(x) and cos(x) in binary64 to a few units in the last place, for |x| up to
// a few thousand: two-part Cody-Waite reduction (no tail), the kernels of
// sincos_small, the quadrant.  For the binary64 dynamics, whose results are not
// rounded to float afterwards and are held to a relative tolerance, not to the
// last bit; about a quarter cheaper than sincos_bounded.
GX_HD void sincos_reduced(double x, double& s, double& c) {
  const double INV_PIO2 = 0x1.45f306dc9c883p-1;
  const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a626331p-34;
  // round(x * 2/pi) by adding and subtracting 1.5 * 2^52: the sum's low mantissa bits are the
  // quadrant number, so neither a rounding instruction nor a float-to-int conversion is needed
  const double MAGIC = 0x1.8p+52;
  const double t = fma(x, INV_PIO2, MAGIC);
  const double fn = t - MAGIC;
  const double r = fma(-fn, P2, fma(-fn, P1, x));
  double sn, cs;
  sincos_small(r, sn, cs);
  quadrant((int)(uint32_t)f64_bits(t), sn, cs, s, c);
}

// ---- table-driven sin and cos ----
// sin(x) and cos(x) in binary64 to the last place (about 0.9 units, like the routines above) for |x| up to a
// few thousand, at two thirds of their instruction count and with no quadrant logic: x = j pi/32 + r with
// |r| <= pi/64 (Cody-Waite against pi/32 in three parts, the first of 33 bits so that j times it is exact),
// sin and cos of the table angle a = j pi/32 read as head + tail from a 64-entry table (2 KiB: the kernels keep
// it in LDS, the host checker in ordinary memory), short kernels for sin r and cos r - 1, and
//   sin x = sin a + (cos a sin r + sin a (cos r - 1)),  cos x = cos a - (sin a sin r - cos a (cos r - 1)),
// where the bracket is below 0.05: it is rounded at 2^-58 and the one rounding that shows is the final sum's.
// Table entries at the multiples of pi/2 hold exact zeros, so the results keep their relative accuracy next to the
// zeros of sin and cos (there the bracket alone is the result).  TAIL carries what the rounding of r dropped
// (about 2^-58) into the result as well.
struct alignas(32) SinCosEntry {
  double sh, sl, ch, cl;  // sin a = sh + sl, cos a = ch + cl
};

template <bool TAIL>
GX_HD void sincos_table(double x, const SinCosEntry* table, double& s, double& c) {
  const double MAGIC = 0x1.8p+52;
  // round(x * 32/pi) by adding and subtracting 1.5 * 2^52; the table index is in the sum's low bits
  const double tq = fma(x, GX_SINCOS_INV_H, MAGIC);
  const double fj = tq - MAGIC;
  const double r0 = fma(-fj, GX_SINCOS_H1, x);  // exact
  const double r = fma(-fj, GX_SINCOS_H2, r0);
  const SinCosEntry e = table[(uint32_t)f64_bits(tq) & (GX_SINCOS_TABLE_ENTRIES - 1)];
  const double z = r * r;
  double p = fma_k(z, GX_SINCOS_S3, GX_SINCOS_S2);
  p = fma_k(z, p, GX_SINCOS_S1);
  double sr = fma(z * r, p, r);  // sin r
  double q = fma_k(z, GX_SINCOS_C2, GX_SINCOS_C1);
  q = fma_k(z, q, GX_SINCOS_C0);
  double cm = z * q;             // cos r - 1
  if (TAIL) {
    double t = fma(-fj, GX_SINCOS_H2, r0 - r);
    t = fma(-fj, GX_SINCOS_H3, t);
    sr += t;              // sin(r + t) = sin r + t cos r, and t (1 - cos r) is below 2^-68
    cm = fma(-r, t, cm);  // cos(r + t) = cos r - t sin r
  }
  s = e.sh + fma(e.ch, sr, fma(e.sh, cm, e.sl));
  c = e.ch + fma(-e.sh, sr, fma(e.ch, cm, e.cl));
}

// 1 / x in binary64 to the last place or two (hardware estimate plus two Newton steps), for the
// binary64 dynamics, which are held to a relative tolerance and need no correctly rounded divide.
GX_HD double rcp64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

// atan2(y, x) in binary64 to about one unit in the last place, for any arguments (the same
// special values as the C library: signed zeros, infinities, NaN).  The smaller magnitude over
// the larger is a quotient q in [0, 1] refined to the last place (Newton reciprocal, one residual
// step); atan q = q + q t P(t) with t = q^2 and P the degree-20 near-minimax fit (Chebyshev
// nodes, 60-digit arithmetic; 0.4 units in the last place after rounding the coefficients) of
// (atan(sqrt t) - sqrt t) / (t sqrt t) on [0, 1]; then the octant, the half-plane and the sign of y.
// One division and no table, like the library routine, at half its instruction count: every
// polynomial term is a single v_fma_f64 (see fma_k).
GX_HD double atan2_f64(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  // fmax / fmin drop a NaN operand; NaN arguments are put back at the end
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const double r = rcp64(mx);
  double q = mn * r;
  q = fma(fma(-q, mx, mn), r, q);
  // the quotient comes out NaN exactly when the larger magnitude is 0 (0 times inf) or infinite
  // (a product of 0 and inf in the residual): 0 / 0 -> 0 (the angle then depends on the signs alone),
  // inf / inf -> 1, finite / inf -> 0
  if (q != q) q = (mn == mx && mx != 0.0) ? 1.0 : 0.0;
  const double t = q * q;
  double p = fma_k(t, -0x1.8d087295ab9f4p-17, 0x1.1f0adb240a1fbp-13);
  p = fma_k(t, p, -0x1.8a2f3ae537bb1p-11);
  p = fma_k(t, p, 0x1.57cc888719aa0p-9);
  p = fma_k(t, p, -0x1.aef1a8e977bf8p-8);
  p = fma_k(t, p, 0x1.a1fe88baf72d4p-7);
  p = fma_k(t, p, -0x1.4b974a27ac39fp-6);
  p = fma_k(t, p, 0x1.c3ac54a572735p-6);
  p = fma_k(t, p, -0x1.147b2825d82f6p-5);
  p = fma_k(t, p, 0x1.3d3f26d5ea7ffp-5);
  p = fma_k(t, p, -0x1.60ebd1489b946p-5);
  p = fma_k(t, p, 0x1.854a7b99af61dp-5);
  p = fma_k(t, p, -0x1.af00bd568a3d2p-5);
  p = fma_k(t, p, 0x1.e1dc0d7679adcp-5);
  p = fma_k(t, p, -0x1.1110bfea8ed5ep-4);
  p = fma_k(t, p, 0x1.3b13aaf4aba4ap-4);
  p = fma_k(t, p, -0x1.745d16f26d0a3p-4);
  p = fma_k(t, p, 0x1.c71c71c472843p-4);
  p = fma_k(t, p, -0x1.249249248c7dap-3);
  p = fma_k(t, p, 0x1.99999999998f9p-3);
  p = fma_k(t, p, -0x1.5555555555555p-2);
  double a = fma(q * t, p, q);
  // pi/2 and pi carried as head + tail, so that the subtractions round once
  const double PIO2_HI = 0x1.921fb54442d18p+0, PIO2_LO = 0x1.1a62633145c07p-54;
  a = ay > ax ? (PIO2_HI - (a - PIO2_LO)) : a;
  a = __builtin_signbit(x) ? (2.0 * PIO2_HI - (a - 2.0 * PIO2_LO)) : a;
  a = copysign(a, y);
  return (x != x || y != y) ? x + y : a;
}

// (float)atan2((double)s, (double)c) for s = (float)S, c = (float)C with S = sin(theta), C = cos(theta) as
// sincos_bounded returns them, |theta| < 1000: the angle a step that keeps (cos, sin) of an angle as floats
// reads back from them.  The angle of (c, s) is theta brought into [-pi, pi] plus the small rotation the two
// roundings cause: with es = s - S and ec = c - C (exact in binary64),
//   atan2(s, c) = w + atan2(s C - c S, c C + s S) = w + (es C - ec S) / (1 + ec C + es S)
// up to terms below 2^-70, where the quotient is below 2^-23, so 1 / (1 + h) = 1 - h is enough and the arc
// tangent is its argument.  The errors of S and C (a unit in their last place) enter scaled by the other
// one, i.e. by at most the size of the result: the sum is within about two units in the last place of the
// binary64 angle, like atan2_f64, at a third of its cost.  The distance of a float to an odd multiple of pi
// is above 1e-8 for |theta| < 16, far more than the rotation, so the sum cannot leave [-pi, pi].
// The sign is that of s, as atan2 gives it (also for zeros).  NaN or infinite theta gives NaN.
GX_HD float atan2_of_rounded_sincos(float theta, double S, double C, float s, float c) {
  const double INV_2PI = 0x1.45f306dc9c883p-3;
  const double TWO_PI_HI = 0x1.921fb54442d18p+2, TWO_PI_LO = 0x1.1a62633145c07p-52;
  const double MAGIC = 0x1.8p+52;
  const double x = (double)theta;
  const double k = fma(x, INV_2PI, MAGIC) - MAGIC;       // whole turns, rounded to nearest
  const double w = fma(-k, TWO_PI_LO, fma(-k, TWO_PI_HI, x));
  const double es = (double)s - S, ec = (double)c - C;
  const double num = fma(es, C, -(ec * S));
  const double h = fma(es, S, ec * C);
  const double a = w + fma(-num, h, num);
  return copysignf((float)a, s);
}

// ---- binary32 routines of the FAST32 mode (held to 1e-5 relative, not to the last bit) ----

// sin(x) and cos(x) in binary32 for |x| up to a few thousand, absolute error about 1e-7:
// Cody-Waite reduction against pi/2 in three parts (the first has 8 significant bits, so
// n * P1 is exact), degree-7 / degree-8 minimax kernels on [-pi/4, pi/4], the quadrant.
GX_HD void sincosf_bounded(float x, float& s, float& c) {
  const float P1 = 1.5703125f, P2 = 4.837512969970703125e-4f, P3 = 7.54978995489188216e-8f;
  const float fn = rintf(x * 0.636619772367581343f);
  float r = fmaf(-fn, P1, x);
  r = fmaf(-fn, P2, r);
  r = fmaf(-fn, P3, r);
  const float z = r * r;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  const float sn = fmaf(z * r, ps, r);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  const float cs = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const int q = (int)fn;
  const float a = (q & 1) ? cs : sn;
  const float b = (q & 1) ? sn : cs;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

GX_HD float fast_rcpf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}

// atan2(y, x) in binary32 for finite arguments, absolute error about 2e-7: the smaller
// magnitude over the larger, one more reduction at tan(pi/8), a four-term odd kernel, then
// the octant, the half-plane and the sign of y.  atan2(0, 0) = 0 with the sign of y.
GX_HD float atan2f_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
  float q = mx > 0.0f ? mn * fast_rcpf(mx) : 0.0f;
  const bool big = q > 0.4142135623730950f;
  q = big ? (q - 1.0f) * fast_rcpf(q + 1.0f) : q;
  const float z = q * q;
  float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = fmaf(z, p, 1.99777106478e-1f);
  p = fmaf(z, p, -3.33329491539e-1f);
  float a = fmaf(p * z, q, q);
  a = big ? a + 0.785398163397448310f : a;
  a = ay > ax ? 1.570796326794896619f - a : a;
  a = __builtin_signbit(x) ? 3.141592653589793238f - a : a;  // also for x = -0, like atan2
  return copysignf(a, y);
}

// x / 1.1f in binary32, correctly rounded, without a hardware divide: with
// y = RN(1 / c), q = RN(x y) is within an ulp of x / c, r = x - q c is exact in
// one fma, and RN(q + r y) is the correctly rounded quotient (Markstein) as
// long as nothing over/underflows.  Valid for 2^-100 <= |x| <= 2^100; callers
// fall back to a true divide outside.  Checked against `x / 1.1f` for every
// float in that range by tests/host_math_check.cpp.
GX_HD float div_by_total_mass(float x) {
  const float c = 1.0f + 0.1f;
  const float y = 1.0f / c;
  const float q = x * y;
  const float r = fmaf(-q, c, x);
  return fmaf(r, y, q);
}
// a / b in binary32, correctly rounded, without the hardware divide sequence, for dividends that share a
// divisor: y = 1 / (double)b to within two units in the last place (rcp64), then (float)((double)a * y).
// Why the second rounding is harmless: the double product is within 2^-51 (relative) of a / b, while the
// quotient of two floats that is not itself representable stays at least 2^-48 away from every midpoint
// between neighbouring floats (a - m b is a non-zero multiple of 2^-47 of its scale for a 25-bit midpoint
// m); where the quotient is subnormal the same count gives 2^-174 against an error below 2^-177 for
// divisors in [0.5, 8), and an exact midpoint needs a divisor that is a power of two, whose reciprocal is
// exact.  b must be finite and non-zero (the Newton steps of rcp64 turn 1/0 into NaN): the callers'
// divisors are bounded away from zero by construction.
GX_HD double rcp_of_f32(float b) { return rcp64((double)b); }
GX_HD float div_by_rcp(float a, double y) { return (float)((double)a * y); }
GX_HD bool div_fast_range(float x) {
  const float a = fabsf(x);
  return a >= 0x1.0p-100f && a <= 0x1.0p100f;
}

}  // namespace math
}  // namespace gx
