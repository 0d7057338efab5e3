// Host-side check of gym_amd/csrc/gx_math.h against the platform libm (the
// same header compiles for the device).  Build: g++ -O2 -ffp-contract=off
// -mfma tests/host_math_check.cpp -o ... ; run: ./host_math_check [stride]
//   sincos_small: (float)sin / (float)cos of every `stride`-th float in
//                 [-pi/4, pi/4] against (float)sin((double)x), (float)cos(...)
//   sincos_bounded: floats in [-16, 16] and their shifts by -pi/2 and +pi, the same way
//   sincos_table:   the same arguments (with and without the tail of the reduced argument), and its
//                   binary64 error off the float grid
//   atan2_f64: random float pairs against (float)atan2((double)y, (double)x), and in binary64
//   atan2_of_rounded_sincos: floats in [-16, 16] against (float)atan2 of the rounded sine and cosine
//   div_by_rcp: floats over every divisor of [2.5, 4.5] and [0.5, 1.25] against a / b
//   div_by_total_mass: every `stride`-th float in [2^-100, 2^100], both signs,
//                 against x / 1.1f
// Prints the mismatch counts; exit status 0 when both are within bounds.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../gym_amd/csrc/gx_math.h"

static const gx::math::SinCosEntry kSinCosTable[GX_SINCOS_TABLE_ENTRIES] = GX_SINCOS_TABLE_VALUES;
static float from_bits(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
static uint32_t to_bits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }

int main(int argc, char** argv) {
  const uint32_t stride = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 64;
  // ---- sincos_small
  const uint32_t hi = to_bits(0.78539816f);
  unsigned long long n = 0, bad_s = 0, bad_c = 0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= hi; b += stride) {
      const float x = from_bits(b | (sign << 31));
      double s, c;
      gx::math::sincos_small((double)x, s, c);
      const float fs = (float)s, fc = (float)c;
      const float rs = (float)std::sin((double)x), rc = (float)std::cos((double)x);
      n++;
      bad_s += to_bits(fs) != to_bits(rs);
      bad_c += to_bits(fc) != to_bits(rc);
    }
  std::printf("sincos_small: %llu inputs, sin mismatches %llu, cos mismatches %llu\n", n, bad_s, bad_c);
  // ---- sincos_bounded: floats |x| <= 16, and the shifted arguments the dynamics form
  // ((double)x - pi/2 and (double)x + pi)
  const uint32_t hi16 = to_bits(16.0f);
  unsigned long long nb = 0, bad_b = 0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= hi16; b += stride * 3 + 1) {
      const float x = from_bits(b | (sign << 31));
      const double args[3] = {(double)x, (double)x - M_PI / 2.0, (double)x + M_PI};
      for (int k = 0; k < 3; k++) {
        double s, c;
        gx::math::sincos_bounded(args[k], s, c);
        nb += 2;
        bad_b += to_bits((float)s) != to_bits((float)std::sin(args[k]));
        bad_b += to_bits((float)c) != to_bits((float)std::cos(args[k]));
      }
    }
  std::printf("sincos_bounded: %llu results, mismatches %llu\n", nb, bad_b);
  // ---- sincos_table: the same arguments, against the same libm results
  unsigned long long nt = 0, bad_t = 0, bad_tt = 0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= hi16; b += stride * 3 + 1) {
      const float x = from_bits(b | (sign << 31));
      const double args[3] = {(double)x, (double)x - M_PI / 2.0, (double)x + M_PI};
      for (int k = 0; k < 3; k++) {
        double s, c;
        const float rs = (float)std::sin(args[k]), rc = (float)std::cos(args[k]);
        gx::math::sincos_table<false>(args[k], kSinCosTable, s, c);
        nt += 2;
        bad_t += (to_bits((float)s) != to_bits(rs)) + (to_bits((float)c) != to_bits(rc));
        gx::math::sincos_table<true>(args[k], kSinCosTable, s, c);
        bad_tt += (to_bits((float)s) != to_bits(rs)) + (to_bits((float)c) != to_bits(rc));
      }
    }
  std::printf("sincos_table: %llu results, mismatches %llu (%llu with the tail)\n", nt, bad_t, bad_tt);
  // its accuracy in binary64 itself, off the float grid, |x| <= 16
  double worst_tab = 0.0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= hi16; b += stride * 5 + 3) {
      const double x = (double)from_bits(b | (sign << 31)) * 1.0000000123;
      double s, c;
      gx::math::sincos_table<false>(x, kSinCosTable, s, c);
      const double rs = std::sin(x), rc = std::cos(x);
      const double es = std::fabs(s - rs) / std::fmax(std::fabs(rs), 0x1.0p-20);
      const double ec = std::fabs(c - rc) / std::fmax(std::fabs(rc), 0x1.0p-20);
      worst_tab = std::fmax(worst_tab, std::fmax(es, ec));
    }
  std::printf("sincos_table: worst relative error in binary64 %.3g\n", worst_tab);
  // ---- sincos_reduced: binary64 accuracy itself, |x| <= 16, against libm
  double worst = 0.0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= hi16; b += stride * 5 + 3) {
      const double x = (double)from_bits(b | (sign << 31)) * 1.0000000123;  // off the float grid
      double s, c;
      gx::math::sincos_reduced(x, s, c);
      const double rs = std::sin(x), rc = std::cos(x);
      // relative to the larger of |value| and 2^-20: next to a zero of sin or cos the reduction's
      // absolute error (~1e-17) is what counts
      const double es = std::fabs(s - rs) / std::fmax(std::fabs(rs), 0x1.0p-20);
      const double ec = std::fabs(c - rc) / std::fmax(std::fabs(rc), 0x1.0p-20);
      worst = std::fmax(worst, std::fmax(es, ec));
    }
  std::printf("sincos_reduced: worst relative error %.3g\n", worst);
  // ---- binary32 routines of the FAST32 mode: absolute error against libm in double
  double worst_f = 0.0, worst_at = 0.0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= to_bits(64.0f); b += stride * 7 + 5) {
      const float x = from_bits(b | (sign << 31));
      float s, c;
      gx::math::sincosf_bounded(x, s, c);
      worst_f = std::fmax(worst_f, std::fmax(std::fabs((double)s - std::sin((double)x)), std::fabs((double)c - std::cos((double)x))));
    }
  {
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (double)(state >> 11) * 0x1.0p-53; };
    for (int i = 0; i < 4000000; i++) {
      const float scale = i & 1 ? 1.0f : 0.05f;  // unit-circle pairs and the small reset values
      const float y = (float)(next() * 2 - 1) * scale, x = (float)(next() * 2 - 1) * scale;
      worst_at = std::fmax(worst_at, std::fabs((double)gx::math::atan2f_fast(y, x) - std::atan2((double)y, (double)x)));
    }
    const float ax[] = {0.0f, -0.0f, 1.0f, -1.0f};
    for (float y : ax)
      for (float x : ax) worst_at = std::fmax(worst_at, std::fabs((double)gx::math::atan2f_fast(y, x) - std::atan2((double)y, (double)x)));
  }
  std::printf("sincosf_bounded: worst absolute error %.3g; atan2f_fast: %.3g\n", worst_f, worst_at);
  // ---- atan2_f64: rounded to float against (float)atan2 for float arguments (unit-circle pairs, the
  // small reset values, wide states), the error in binary64 units in the last place, special values
  unsigned long long na = 0, bad_a = 0;
  double worst_ulp = 0.0;
  {
    uint64_t state = 0xD1B54A32D192ED03ull;
    auto next = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (double)(state >> 11) * 0x1.0p-53; };
    const long long count = 40000000ll / (stride > 64 ? 8 : 1);
    for (long long i = 0; i < count; i++) {
      const float scale = (i % 3) == 0 ? 1.0f : ((i % 3) == 1 ? 0.05f : 40.0f);
      const float y = (float)(next() * 2 - 1) * scale, x = (float)(next() * 2 - 1) * scale;
      const double got = gx::math::atan2_f64((double)y, (double)x), ref = std::atan2((double)y, (double)x);
      na++;
      bad_a += to_bits((float)got) != to_bits((float)ref);
      const double ulp = std::ldexp(1.0, std::ilogb(std::fabs(ref) > 0 ? ref : 1.0) - 52);
      if (ref != 0.0) worst_ulp = std::fmax(worst_ulp, std::fabs(got - ref) / ulp);
    }
  }
  bool special_ok = true;
  {
    const double inf = INFINITY, vals[] = {0.0, -0.0, 1.0, -1.0, 3.5, -2.25, 1e-300, -1e-300, 1e300, inf, -inf};
    for (double y : vals)
      for (double x : vals) {
        const double got = gx::math::atan2_f64(y, x), ref = std::atan2(y, x);
        const bool same = std::fabs(got - ref) <= 2.3e-16 * std::fabs(ref) && std::signbit(got) == std::signbit(ref);
        if (!same) { special_ok = false; std::printf("atan2_f64(%g, %g) = %.17g, libm %.17g\n", y, x, got, ref); }
      }
    special_ok = special_ok && std::isnan(gx::math::atan2_f64(NAN, 1.0)) && std::isnan(gx::math::atan2_f64(1.0, NAN));
  }
  std::printf("atan2_f64: %llu float pairs, mismatches after rounding to float %llu, worst error %.3f ulp, special values %s\n", na, bad_a,
              worst_ulp, special_ok ? "ok" : "WRONG");
  // ---- atan2_of_rounded_sincos: the angle read back from the float roundings of sin and cos, for the
  // angles one Acrobot step can produce (|theta| <= pi + 0.2 * 9 pi < 9) and beyond, against atan2 of the pair
  unsigned long long nr = 0, bad_r = 0;
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = 0; b <= hi16; b += stride / 4 + 1) {
      const float th = from_bits(b | (sign << 31));
      // with the sine and cosine of either routine the kernels take them from
      for (int which = 0; which < 2; which++) {
        double S, C;
        if (which == 0) gx::math::sincos_bounded((double)th, S, C);
        else gx::math::sincos_table<false>((double)th, kSinCosTable, S, C);
        const float s = (float)S, c = (float)C;
        const float got = gx::math::atan2_of_rounded_sincos(th, S, C, s, c);
        const float ref = (float)std::atan2((double)s, (double)c);
        nr++;
        if (to_bits(got) != to_bits(ref)) {
          bad_r++;
          if (bad_r <= 5) std::printf("atan2_of_rounded_sincos(%.9g): %.9g, libm %.9g\n", th, got, ref);
        }
      }
    }
  std::printf("atan2_of_rounded_sincos: %llu angles, mismatches %llu\n", nr, bad_r);
  // ---- division through a binary64 reciprocal: every divisor of the two ranges the Acrobot step has
  // (stepping through them), dividends of every magnitude including those with subnormal quotients, and
  // the reciprocal pushed two units in the last place either way (the device's Newton reciprocal is not
  // correctly rounded)
  unsigned long long nq = 0, bad_q = 0;
  {
    uint64_t state = 0xA0761D6478BD642Full;
    auto next32 = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (uint32_t)(state >> 32); };
    const float ranges[2][2] = {{2.5f, 4.5f}, {0.5f, 1.25f}};
    for (int rg = 0; rg < 2; rg++)
      for (uint32_t bb = to_bits(ranges[rg][0]); bb <= to_bits(ranges[rg][1]); bb += stride / 8 + 1) {
        const float b = from_bits(bb);
        const double y = 1.0 / (double)b;
        const double ys[3] = {y, std::nextafter(std::nextafter(y, 0.0), 0.0), std::nextafter(std::nextafter(y, 9.0), 9.0)};
        for (int k = 0; k < 6; k++) {
          uint32_t ab = next32();
          if (k == 4) ab = (ab & 0x807FFFFFu) | ((next32() % 4) << 23);  // tiny dividends: subnormal quotients
          if ((ab & 0x7F800000u) == 0x7F800000u) continue;               // inf / NaN
          const float a = k == 5 ? b * from_bits((ab & 0x007FFFFFu) | 0x3F800000u) : from_bits(ab);  // and exact quotients' neighbours
          volatile float ref = a / b;
          for (double yy : ys) {
            nq++;
            bad_q += to_bits(gx::math::div_by_rcp(a, yy)) != to_bits(ref);
          }
        }
      }
  }
  std::printf("div_by_rcp: %llu quotients, mismatches %llu\n", nq, bad_q);
  // ---- division by 1.1f
  const float c11 = 1.0f + 0.1f;
  unsigned long long nd = 0, bad_d = 0;
  const uint32_t lo_b = to_bits(0x1.0p-100f), hi_b = to_bits(0x1.0p100f);
  for (uint32_t sign = 0; sign < 2; sign++)
    for (uint32_t b = lo_b; b <= hi_b; b += stride) {
      const float x = from_bits(b | (sign << 31));
      volatile float q_ref = x / c11;
      const float q = gx::math::div_by_total_mass(x);
      nd++;
      bad_d += to_bits(q) != to_bits(q_ref);
    }
  std::printf("div_by_total_mass: %llu inputs, mismatches %llu\n", nd, bad_d);
  // one rounding-boundary event per ~2^28 trig calls is expected; the division must be exact
  const bool ok = special_ok && bad_r * (1ull << 23) <= nr + (1ull << 23) && worst_ulp <= 2.0 && bad_a * (1ull << 23) <= na + (1ull << 23) && worst < 1e-15 && worst_f < 3e-7 && worst_at < 6e-7 && bad_d == 0 && bad_q == 0 && bad_b * (1ull << 24) <= nb + (1ull << 24) && bad_t * (1ull << 24) <= nt + (1ull << 24) && bad_tt * (1ull << 24) <= nt + (1ull << 24) && worst_tab < 4e-16 && bad_s * (1ull << 24) <= n + (1ull << 24) && bad_c * (1ull << 24) <= n + (1ull << 24);
  return ok ? 0 : 1;
}
