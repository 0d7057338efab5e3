// Trigonometry policies of the continuous-control kernels.
//
// MODE  EXACT : Math.Sin/Cos/Atan2 of the C# source take and return double and
//               the source casts the result to float, which rounds once; the
//               kernels do the same in binary64.
//       FAST32: binary32 device trig with identities applied.
// BOUNDED     : the caller guarantees the angle ranges stated at each function
//               (the host selects it only when the env's invariants hold), so
//               the reduction-light routines of gx_math.h replace the
//               full-range library ones.  Both are accurate to the last place
//               of the double; the float results differ from each other (and
//               from glibc's) only when the true value sits within ~2^-29 of a
//               float rounding boundary.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gymx.h"
#include "gx_math.h"

namespace gx {

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

template <int MODE, bool BOUNDED>
struct Trig;

// The 2 KiB table of math::sincos_table: a constant array in device memory, and the copy in LDS that the
// kernels of the envs which use it fill once per workgroup (sincos_lds_fill, every thread of the workgroup
// calls it before the first use) and read with two 16-byte LDS loads per evaluation.
__attribute__((unused)) static __device__ const math::SinCosEntry g_sincos_table[GX_SINCOS_TABLE_ENTRIES] = GX_SINCOS_TABLE_VALUES;
__device__ __forceinline__ math::SinCosEntry* sincos_lds() {
  __shared__ math::SinCosEntry table[GX_SINCOS_TABLE_ENTRIES];
  return table;
}
__device__ __forceinline__ void sincos_lds_fill() {
  math::SinCosEntry* table = sincos_lds();
  for (int i = threadIdx.x; i < GX_SINCOS_TABLE_ENTRIES; i += blockDim.x) table[i] = g_sincos_table[i];
  __syncthreads();
}

// EXACT mode with bounded arguments through the table-driven routine (the same last-place accuracy as
// Trig<GX_MODE_EXACT, true>, two thirds of its instructions, no quadrant selection).  For kernels that
// are bound by instruction issue and have called sincos_lds_fill().
struct TrigTable {
  static __device__ __forceinline__ void sincos64(double x, double& s, double& c) { math::sincos_table<false>(x, sincos_lds(), s, c); }
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) {
    double sd, cd;
    sincos64((double)x, sd, cd);
    s = (float)sd;
    c = (float)cd;
  }
  static __device__ __forceinline__ float cos(float x) {
    double sd, cd;
    sincos64((double)x, sd, cd);
    return (float)cd;
  }
  // (float)Math.Cos(x - Math.PI / 2.0) with x a float: the subtraction is in double
  static __device__ __forceinline__ float cos_minus_half_pi(float x) {
    double sd, cd;
    sincos64((double)x - M_PI / 2.0, sd, cd);
    return (float)cd;
  }
  // (float)Math.Sin(x + Math.PI)
  static __device__ __forceinline__ float sin_plus_pi(float x) {
    double sd, cd;
    sincos64((double)x + M_PI, sd, cd);
    return (float)sd;
  }
  static __device__ __forceinline__ float atan2(float y, float x) { return (float)math::atan2_f64((double)y, (double)x); }
};

template <>
struct Trig<GX_MODE_EXACT, false> {
  // |x| <= pi/4 promised only by the BOUNDED variant; here any float
  static __device__ __forceinline__ void sincos_quarter(float x, float& s, float& c) { sincos(x, s, c); }
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) {
    double sd, cd;
    ::sincos((double)x, &sd, &cd);
    s = (float)sd;
    c = (float)cd;
  }
  static __device__ __forceinline__ float cos(float x) { return (float)::cos((double)x); }
  // (float)Math.Cos(x - Math.PI / 2.0) with x a float: the subtraction is in double
  static __device__ __forceinline__ float cos_minus_half_pi(float x) { return (float)::cos((double)x - M_PI / 2.0); }
  // (float)Math.Sin(x + Math.PI)
  static __device__ __forceinline__ float sin_plus_pi(float x) { return (float)::sin((double)x + M_PI); }
  static __device__ __forceinline__ float atan2(float y, float x) { return (float)math::atan2_f64((double)y, (double)x); }
};

template <>
struct Trig<GX_MODE_EXACT, true> {
  // requires |x| <= pi/4
  static __device__ __forceinline__ void sincos_quarter(float x, float& s, float& c) {
    double sd, cd;
    math::sincos_small((double)x, sd, cd);
    s = (float)sd;
    c = (float)cd;
  }
  // the functions below require |argument| below a few thousand
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) {
    double sd, cd;
    math::sincos_bounded((double)x, sd, cd);
    s = (float)sd;
    c = (float)cd;
  }
  static __device__ __forceinline__ float cos(float x) {
    double sd, cd;
    math::sincos_bounded((double)x, sd, cd);
    return (float)cd;
  }
  static __device__ __forceinline__ float cos_minus_half_pi(float x) {
    double sd, cd;
    math::sincos_bounded((double)x - M_PI / 2.0, sd, cd);
    return (float)cd;
  }
  static __device__ __forceinline__ float sin_plus_pi(float x) {
    double sd, cd;
    math::sincos_bounded((double)x + M_PI, sd, cd);
    return (float)sd;
  }
  static __device__ __forceinline__ float atan2(float y, float x) { return (float)math::atan2_f64((double)y, (double)x); }
};

template <>
struct Trig<GX_MODE_FAST32, false> {
  static __device__ __forceinline__ void sincos_quarter(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
  static __device__ __forceinline__ float cos(float x) { return ::cosf(x); }
  static __device__ __forceinline__ float cos_minus_half_pi(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float sin_plus_pi(float x) { return -::sinf(x); }
  static __device__ __forceinline__ float atan2(float y, float x) { return ::atan2f(y, x); }
};

// bounded arguments: the library's full-range reduction (its large-argument path alone is
// several hundred instructions, evaluated branch-free) gives way to the routines of gx_math.h
template <>
struct Trig<GX_MODE_FAST32, true> {
  static __device__ __forceinline__ void sincos_quarter(float x, float& s, float& c) { math::sincosf_bounded(x, s, c); }
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) { math::sincosf_bounded(x, s, c); }
  static __device__ __forceinline__ float cos(float x) {
    float s, c;
    math::sincosf_bounded(x, s, c);
    return c;
  }
  static __device__ __forceinline__ float cos_minus_half_pi(float x) {
    float s, c;
    math::sincosf_bounded(x, s, c);
    return s;
  }
  static __device__ __forceinline__ float sin_plus_pi(float x) {
    float s, c;
    math::sincosf_bounded(x, s, c);
    return -s;
  }
  static __device__ __forceinline__ float atan2(float y, float x) { return math::atan2f_fast(y, x); }
};

// a / b: the IEEE quotient in EXACT mode, a times the hardware reciprocal (about 1.5 ulp) in FAST32
template <int MODE>
__device__ __forceinline__ float divide(float a, float b) {
  if (MODE == GX_MODE_EXACT) return a / b;
  return a * __builtin_amdgcn_rcpf(b);
}

// Math.Clamp(float, float, float)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace gx
