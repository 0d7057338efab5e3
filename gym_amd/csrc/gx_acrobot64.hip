// Acrobot with the upstream dynamics: classical RK4 in binary64.
//
// EXTENSION -- the reference integrates Acrobot with semi-implicit Euler in
// binary32 (Envs.Acrobot.cs:69-74); this variant follows the published
// Gymnasium "book" equations with the reference's constants
// (Envs.Acrobot.cs:33-44): state (theta1, theta2, dtheta1, dtheta2) in double,
// one RK4 step of dt = 0.2, both angles wrapped to [-pi, pi], velocity bounds
// 4pi / 9pi, terminal when -cos(t1) - cos(t1 + t2) > 1 or after 500 steps.
// It is the FP64 vector-ALU workload: in a K-step rollout the state never
// leaves registers.  Two envs per lane (16 bytes of each double row).  While an
// env is in registers the sines and cosines of its two angles ride along with
// the state: the termination test of one step, the first RK4 stage of the next
// and the observation all need exactly those four values.
//
// Every a*b+c is an explicit fma() (and nothing else is contracted) so that a
// CPU restatement written with the same fma() calls can match wherever the
// sin/cos results agree; sin and cos come from the table-driven binary64
// routine of gx_math.h (within a unit in the last place, so results differ from
// a libm build by parts in 10^16).
#pragma clang fp contract(off)

#include "gx_classic_kernel.h"

namespace gx {

struct Acrobot64 {
  static constexpr bool kPrefetchNextTrip = true;
  static constexpr int NS = 4;
  static constexpr int NR = 8;  // + sin(theta1), cos(theta1), sin(theta2), cos(theta2) of the current state
  using Real = double;
  static constexpr int W = 2;
  static constexpr bool kDerivedObs = true;
  static constexpr int NOBS = 6;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 3;
  static constexpr bool kCheckAction = true;
  static __device__ __forceinline__ bool action_valid(Act a) { return (uint32_t)a <= 2u; }
  // the angles are wrapped to [-pi, pi] and the velocities bounded after every step, so
  // every sin/cos argument of an RK4 stage stays below 16 for any finite state
  static constexpr bool kAlwaysBounded = true;
  static constexpr int kStepReturn = -1;  // reward -1 on every step but the final one
#ifdef GX_ACROBOT64_WAVES  // A/B builds only: resident waves per SIMD the fused kernels are compiled for
  static constexpr int kMinWaves = GX_ACROBOT64_WAVES;
#endif
#ifdef GX_ACROBOT_NO_TABLE  // A/B builds only: the polynomial routine with quadrant selection
  static constexpr bool kSinCosTable = false;
  static __device__ __forceinline__ void sincos(double x, double& s, double& c) { math::sincos_reduced(x, s, c); }
#else
  // eight sin/cos pairs per step are two thirds of its instructions: the table-driven routine (gx_math.h)
  static constexpr bool kSinCosTable = true;
  static __device__ __forceinline__ void sincos(double x, double& s, double& c) { TrigTable::sincos64(x, s, c); }
#endif

  static __device__ __forceinline__ void derive(double (&s)[NR]) {
    sincos(s[0], s[4], s[5]);
    sincos(s[1], s[6], s[7]);
  }

  static __device__ __forceinline__ void reset(double (&s)[NR], const RngKey& key, uint64_t env) {
    Draws d = philox_block(key, env, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = fma(d.uniform(i), 0.2, -0.1);
    derive(s);
  }

  // this extension is held to 1e-9 relative, so a correctly rounded divide is not needed
  static __device__ __forceinline__ double recip(double x) { return math::rcp64(x); }

  // Derivative of (theta1, theta2, dtheta1, dtheta2).  Two sincos per evaluation: the book
  // equations' cos(theta1 - pi/2) is sin(theta1) and cos(theta1 + theta2 - pi/2) is
  // sin(theta1 + theta2) = s1 c2 + c1 s2.
  static __device__ __forceinline__ void dsdt(const double (&y)[4], double u, double (&dy)[4]) {
    double s1, c1, s2, c2;
    sincos(y[0], s1, c1);
    sincos(y[1], s2, c2);
    dsdt(y[2], y[3], s1, c1, s2, c2, u, dy);
  }
  static __device__ __forceinline__ void dsdt(double w1, double w2, double s1, double c1, double s2, double c2, double u,
                                              double (&dy)[4]) {
    const double g = 9.8;
    const double s12 = fma(s1, c2, c1 * s2);
    const double d1 = c2 + 3.5;
    const double d2 = fma(0.5, c2, 1.25);
    const double phi2 = (0.5 * g) * s12;
    double phi1 = fma(1.5 * g, s1, phi2);
    phi1 = fma(-(w2 * s2), w1, phi1);
    phi1 = fma(-0.5 * (w2 * s2), w2, phi1);
    const double inv_d1 = recip(d1);
    const double r = d2 * inv_d1;
    double num = fma(r, phi1, u);
    num = fma(-0.5 * (w1 * s2), w1, num) - phi2;
    const double den = fma(-d2, r, 1.25);
    const double a2 = num * recip(den);
    const double a1 = -fma(d2, a2, phi1) * inv_d1;
    dy[0] = w1; dy[1] = w2; dy[2] = a1; dy[3] = a2;
  }

  // Into [-pi, pi] in bounded time.  A step moves an angle by less than one turn (the stored velocities
  // are bounded by 9 pi and dt = 0.2), so one single-turn correction -- the exact `x -= two_pi` (or
  // `+=`) a loop would make, as five instructions: the turn takes the angle's sign -- covers what the env
  // itself produces; anything still outside (a stage slope beyond the velocity bounds, a huge angle
  // written by a caller) is reduced in one multiply-add, and a non-finite value ends as NaN instead of
  // keeping a loop alive.
  static __device__ __forceinline__ double wrap_turn(double x) {
    const double two_pi = 2.0 * M_PI;
    if (fabs(x) > M_PI) x -= copysign(two_pi, x);
    return x;
  }
  static __device__ __forceinline__ double wrap_far(double x) {
    const double two_pi = 2.0 * M_PI;
    if (!(fabs(x) <= M_PI)) x = fma(-two_pi, rint(x * (1.0 / two_pi)), x);
    return x;
  }

  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(double (&s)[NR], int32_t& steps, Act action, Rew& reward, bool& done) {
    const double dt = 0.2;
    const double u = action == 0 ? -1.0 : (action == 2 ? 1.0 : 0.0);
    double k1[4], k2[4], k3[4], k4[4], y[4];
    dsdt(s[2], s[3], s[4], s[5], s[6], s[7], u, k1);
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = fma(0.5 * dt, k1[i], s[i]);
    dsdt(y, u, k2);
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = fma(0.5 * dt, k2[i], s[i]);
    dsdt(y, u, k3);
#pragma unroll
    for (int i = 0; i < 4; i++) y[i] = fma(dt, k3[i], s[i]);
    dsdt(y, u, k4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double sum = fma(2.0, k2[i], k1[i]);
      sum = fma(2.0, k3[i], sum) + k4[i];
      y[i] = fma(dt / 6.0, sum, s[i]);
    }
    y[0] = wrap_turn(y[0]);
    y[1] = wrap_turn(y[1]);
#ifdef GX_ACROBOT64_WRAP_ALWAYS  // A/B builds only
    y[0] = wrap_far(y[0]);
    y[1] = wrap_far(y[1]);
#else
    // what one turn did not bring home is rare enough (never, for states the env itself produced at ordinary
    // accelerations) that the whole wave skips the general reduction unless one of its lanes needs it
    if (__any(!(fabs(y[0]) <= M_PI) || !(fabs(y[1]) <= M_PI))) {
      y[0] = wrap_far(y[0]);
      y[1] = wrap_far(y[1]);
    }
#endif
    const double mv1 = 4.0 * M_PI, mv2 = 9.0 * M_PI;
    y[2] = y[2] < -mv1 ? -mv1 : (y[2] > mv1 ? mv1 : y[2]);
    y[3] = y[3] < -mv2 ? -mv2 : (y[3] > mv2 ? mv2 : y[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = y[i];
    steps++;
    derive(s);
    const double sa = s[4], ca = s[5], sb = s[6], cb = s[7];
    // -cos(t1) - cos(t1 + t2), the second by the angle-addition formula
    done = (-ca - fma(ca, cb, -(sa * sb)) > 1.0) || steps >= 500;
    reward = done ? 0.0f : -1.0f;
  }

  // the reference's observation layout (Envs.Acrobot.cs:77-82)
  static __device__ __forceinline__ void observe(const double (&s)[NR], float (&o)[NOBS]) {
    o[0] = (float)s[5]; o[1] = (float)s[4];
    o[2] = (float)s[7]; o[3] = (float)s[6];
    o[4] = (float)s[2];
    o[5] = (float)s[3];
  }

  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

hipError_t launch_acrobot64_step(const StepArgs& a, int policy, bool fused, dim3 grid, hipStream_t stream) {
  return launch_step_env_exact<Acrobot64>(a, policy, fused, true, grid, stream);
}
hipError_t launch_acrobot64_reset(const StepArgs& a, int only_observe, dim3 grid, hipStream_t stream) {
  return launch_reset_env<Acrobot64>(a, only_observe, grid, stream);
}
hipError_t launch_acrobot64_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  return launch_sample_env<Acrobot64>(a, out, grid, stream);
}

}  // namespace gx
