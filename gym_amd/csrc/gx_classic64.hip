// The five classic-control envs in binary64 (GX_MODE_FP64).
//
// EXTENSION -- the reference computes these in binary32 (Envs.CartPole.cs,
// Envs.MountainCar.cs, Envs.MountainCarContinuous.cs, Envs.Pendulum.cs,
// Envs.Acrobot.cs); here the same formulas run with every float read as
// double and the (float) casts dropped.  State rows are double (two envs per
// lane: 16 bytes of each row), the observation is the state image itself,
// actions keep the reference's types (int32 / float), rewards keep the types
// of the binary32 kernels.
//
// Results are held to a relative tolerance against a CPU build of the same
// formulas, not to the last bit, so the kernels may use identities that are
// exact in real arithmetic: a quotient by a constant is a product with its
// reciprocal, other quotients use a Newton reciprocal, and (BOUNDED only,
// where every stored (cos, sin) pair was written by these kernels or by Reset)
// the sine and cosine of an angle that was just taken from atan2(y, x) are
// y / r and x / r.  Nothing is contracted implicitly; fused multiply-adds are
// written out where they are wanted.
#pragma clang fp contract(off)

#include "gx_classic_kernel.h"

namespace gx {

// 1 / sqrt(x) for finite x > 0: hardware estimate plus two Newton steps
static __device__ __forceinline__ double rsqrt64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x, y * y, 1.5);
  y = y * fma(-0.5 * x, y * y, 1.5);
  return y;
}

static __device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// (x, y) / |(x, y)|, and (1, 0) for the origin, where atan2 gives angle 0
static __device__ __forceinline__ void unit(double x, double y, double& c, double& s) {
  const double r2 = fma(x, x, y * y);
  const double inv = rsqrt64(r2);
  c = r2 > 0.0 ? x * inv : 1.0;
  s = r2 > 0.0 ? y * inv : 0.0;
}

struct CartPole64 {
  static constexpr int NS = 4;
  using Real = double;
  static constexpr int W = 2;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 4;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 2;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  // with auto-reset every stored theta is within the 12-degree threshold or a fresh draw
  static constexpr bool kAlwaysBounded = false;
  static constexpr int kStepReturn = 1;

  // Envs.CartPole.cs:22-32
  static __device__ __forceinline__ void reset(double (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d = philox_block(key, env, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = d.uniform(i) * 0.1 - 0.05;
  }

  // Envs.CartPole.cs:34-75
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(double (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    const double gravity = 9.8, masspole = 0.1, length = 0.5, tau = 0.02;
    const double inv_total_mass = 1.0 / (1.0 + 0.1);
    const double polemass_length = masspole * length;
    const double theta_threshold_radians = 12 * M_PI / 180;
    const double x_threshold = 2.4;

    double x = s[0], xDot = s[1], theta = s[2], thetaDot = s[3];
    const double force = action == 1 ? 10.0 : -10.0;
    double sintheta, costheta;
    if (BOUNDED) math::sincos_small(theta, sintheta, costheta);
    else ::sincos(theta, &sintheta, &costheta);

    const double temp = fma(polemass_length * thetaDot * thetaDot, sintheta, force) * inv_total_mass;
    const double den = length * fma(-masspole * inv_total_mass * costheta, costheta, 4.0 / 3.0);
    const double thetaacc = fma(gravity, sintheta, -(costheta * temp)) * math::rcp64(den);
    const double xacc = fma(-polemass_length * inv_total_mass * thetaacc, costheta, temp);

    x = fma(tau, xDot, x);
    xDot = fma(tau, xacc, xDot);
    theta = fma(tau, thetaDot, theta);
    thetaDot = fma(tau, thetaacc, thetaDot);
    s[0] = x; s[1] = xDot; s[2] = theta; s[3] = thetaDot;
    steps++;

    done = x < -x_threshold || x > x_threshold || theta < -theta_threshold_radians ||
           theta > theta_threshold_radians || steps >= 500;
    reward = !done ? 1.0f : 0.0f;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

// cos(3 * position): with auto-reset the position stays inside [-1.2, 0.57]
template <bool BOUNDED>
static __device__ __forceinline__ double hill_slope(double position) {
  if (BOUNDED) {
    double sn, cs;
    math::sincos_reduced(3 * position, sn, cs);
    return cs;
  }
  return ::cos(3 * position);
}

struct MountainCar64 {
  static constexpr int NS = 2;
  using Real = double;
  static constexpr int W = 2;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 2;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 3;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  static constexpr bool kAlwaysBounded = false;
  static constexpr int kStepReturn = -1;

  // Envs.MountainCar.cs:22-30
  static __device__ __forceinline__ void reset(double (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d0 = philox_block(key, env, 0, 0);
    s[0] = d0.uniform(0) * 0.2 - 0.6;
    s[1] = 0.0;
  }

  // Envs.MountainCar.cs:32-50
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(double (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    double position = s[0], velocity = s[1];
    const double force = (double)(action - 1);
    velocity += fma(0.001, force, -(0.0025 * hill_slope<BOUNDED>(position)));
    velocity = clampd(velocity, -0.07, 0.07);
    position += velocity;
    if (position < -1.2) {
      position = -1.2;
      velocity = 0.0;
    }
    done = position >= 0.5;
    reward = done ? 0.0f : -1.0f;
    s[0] = position; s[1] = velocity;
    steps++;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

struct MountainCarContinuous64 {
  static constexpr bool kPrefetchNextTrip = true;
  static constexpr int NS = 2;
  using Real = double;
  static constexpr int W = 2;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 2;
  using Act = float;
  using Rew = double;
  using EpRet = double;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  static constexpr bool kAlwaysBounded = false;
  static constexpr int kStepReturn = kNoStepReturn;

  // Envs.MountainCarContinuous.cs:19-27
  static __device__ __forceinline__ void reset(double (&s)[NS], const RngKey& key, uint64_t env) {
    MountainCar64::reset(s, key, env);
  }

  // Envs.MountainCarContinuous.cs:29-49
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(double (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    double position = s[0], velocity = s[1];
    const double force = clampd((double)action, -1.0, 1.0);
    velocity += fma(force, 0.0015, -(0.0025 * hill_slope<BOUNDED>(position)));
    velocity = clampd(velocity, -0.07, 0.07);
    position += velocity;
    if (position < -1.2) {
      position = -1.2;
      velocity = 0.0;
    }
    done = position >= 0.45;
    double r = velocity * 0.1;
    if (done) r += 100.0;
    reward = r;
    s[0] = position; s[1] = velocity;
    steps++;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return box_sample(word_uniform(w), -1.0f, 1.0f); }
};

struct Pendulum64 {
  static constexpr int NS = 3;
  using Real = double;
  static constexpr int W = 2;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 3;
  using Act = float;
  using Rew = double;
  using EpRet = double;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  // BOUNDED here also means: the stored (cos, sin) pair was written by Reset or by a step
  static constexpr bool kAlwaysBounded = false;
  static constexpr int kStepReturn = kNoStepReturn;

  // Envs.Pendulum.cs:19-28
  static __device__ __forceinline__ void reset(double (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d0 = philox_block(key, env, 0, 0);
    const double theta = d0.uniform(0) * 2 * M_PI - M_PI;
    const double thetaDot = d0.uniform(1) * 1 - 0.5;
    math::sincos_reduced(theta, s[1], s[0]);
    s[2] = thetaDot;
  }

  // Envs.Pendulum.cs:30-60
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(double (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    const double max_speed = 8.0, max_torque = 2.0, dt = 0.05, g = 10.0, m = 1.0, l = 1.0;

    const double theta = math::atan2_f64(s[1], s[0]);
    const double thetaDot = s[2];
    const double u = clampd((double)action, -max_torque, max_torque);

    // sin(theta + pi) = -sin(theta), and sin(atan2(y, x)) = y / r
    double sin_theta;
    if (BOUNDED) {
      double c;
      unit(s[0], s[1], c, sin_theta);
    } else {
      sin_theta = -::sin(theta + M_PI);
    }
    double new_thetaDot = fma(fma(3 * g / (2 * l), sin_theta, 3.0 / (m * l * l) * u), dt, thetaDot);
    new_thetaDot = clampd(new_thetaDot, -max_speed, max_speed);
    const double new_theta = fma(new_thetaDot, dt, theta);

    math::sincos_reduced(new_theta, s[1], s[0]);  // |new_theta| <= pi + 0.4
    s[2] = new_thetaDot;
    steps++;

    reward = -fma(0.001 * u, u, fma(0.1 * thetaDot, thetaDot, theta * theta));
    done = false;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return box_sample(word_uniform(w), -2.0f, 2.0f); }
};

// Acrobot as the reference integrates it (semi-implicit Euler), in binary64.
struct AcrobotEuler64 {
  static constexpr int NS = 6;
  using Real = double;
  static constexpr int W = 2;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 6;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 3;
  static constexpr bool kCheckAction = true;
  static __device__ __forceinline__ bool action_valid(Act a) { return (uint32_t)a <= 2u; }
  static constexpr bool kAlwaysBounded = false;
  static constexpr int kStepReturn = -1;

  // Envs.Acrobot.cs:22-28
  static __device__ __forceinline__ void reset(double (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d0 = philox_block(key, env, 0, 0), d1 = philox_block(key, env, 0, 1);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = d0.uniform(i) * 0.1 - 0.05;
    s[4] = d1.uniform(0) * 0.1 - 0.05;
    s[5] = d1.uniform(1) * 0.1 - 0.05;
  }

  // Envs.Acrobot.cs:30-91 with the unit masses, lengths and inertias and the 0.5 centre-of-mass
  // positions of Envs.Acrobot.cs:33-40 folded in:
  //   d1 = 3.5 + cos2, d2 = 1.25 + cos2 / 2, phi2 = g/2 sin(theta1 + theta2),
  //   phi1 = -dtheta2^2 sin2 / 2 - dtheta2 dtheta1 sin2 + 3g/2 sin(theta1) + phi2
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(double (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    const double MAX_VEL_1 = 4 * M_PI, MAX_VEL_2 = 9 * M_PI;
    const double dt = 0.2, g = 9.8;

    double theta1 = math::atan2_f64(s[1], s[0]);
    double theta2 = math::atan2_f64(s[3], s[2]);
    double dtheta1 = s[4];
    double dtheta2 = s[5];
    const double u = action == 0 ? -1.0 : (action == 2 ? 1.0 : 0.0);

    double sin1, cos1, sin2, cos2;
    if (BOUNDED) {
      unit(s[0], s[1], cos1, sin1);
      unit(s[2], s[3], cos2, sin2);
    } else {
      ::sincos(theta1, &sin1, &cos1);
      ::sincos(theta2, &sin2, &cos2);
    }
    const double sin12 = fma(sin1, cos2, cos1 * sin2);  // cos(theta1 + theta2 - pi/2)

    const double d1 = cos2 + 3.5;
    const double d2 = fma(0.5, cos2, 1.25);
    const double phi2 = (0.5 * g) * sin12;
    double phi1 = fma(1.5 * g, sin1, phi2);
    phi1 = fma(-(dtheta2 * sin2), dtheta1, phi1);
    phi1 = fma(-0.5 * (dtheta2 * sin2), dtheta2, phi1);
    const double inv_d1 = math::rcp64(d1);
    const double ratio = d2 * inv_d1;
    double num = fma(ratio, phi1, u);
    num = fma(-0.5 * (dtheta1 * sin2), dtheta1, num) - phi2;
    const double ddtheta2 = num * math::rcp64(fma(-d2, ratio, 1.25));
    const double ddtheta1 = -fma(d2, ddtheta2, phi1) * inv_d1;

    dtheta1 = clampd(fma(dt, ddtheta1, dtheta1), -MAX_VEL_1, MAX_VEL_1);
    dtheta2 = clampd(fma(dt, ddtheta2, dtheta2), -MAX_VEL_2, MAX_VEL_2);
    theta1 = fma(dt, dtheta1, theta1);
    theta2 = fma(dt, dtheta2, theta2);

    double s1n, c1n, s2n, c2n;
    math::sincos_reduced(theta1, s1n, c1n);  // |theta| <= pi + 0.2 * 9 pi
    math::sincos_reduced(theta2, s2n, c2n);
    s[0] = c1n; s[1] = s1n; s[2] = c2n; s[3] = s2n; s[4] = dtheta1; s[5] = dtheta2;
    steps++;

    // -cos(theta1) - cos(theta1 + theta2) > 1 (Envs.Acrobot.cs:80-81), the second by the addition formula
    done = (-c1n - fma(c1n, c2n, -(s1n * s2n)) > 1.0) || steps >= 500;
    reward = done ? 0.0f : -1.0f;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

hipError_t launch_classic64_step(int env_id, const StepArgs& a, int policy, bool fused, bool bounded, dim3 grid,
                                 hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_step_env_exact<CartPole64>(a, policy, fused, bounded, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_step_env_exact<MountainCar64>(a, policy, fused, bounded, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_step_env_exact<MountainCarContinuous64>(a, policy, fused, bounded, grid, stream);
    case GX_ENV_PENDULUM: return launch_step_env_exact<Pendulum64>(a, policy, fused, bounded, grid, stream);
    case GX_ENV_ACROBOT: return launch_step_env_exact<AcrobotEuler64>(a, policy, fused, bounded, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic64_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_reset_env<CartPole64>(a, 0, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_reset_env<MountainCar64>(a, 0, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_reset_env<MountainCarContinuous64>(a, 0, grid, stream);
    case GX_ENV_PENDULUM: return launch_reset_env<Pendulum64>(a, 0, grid, stream);
    case GX_ENV_ACROBOT: return launch_reset_env<AcrobotEuler64>(a, 0, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic64_sample(int env_id, const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_sample_env<CartPole64>(a, out, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_sample_env<MountainCar64>(a, out, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_sample_env<MountainCarContinuous64>(a, out, grid, stream);
    case GX_ENV_PENDULUM: return launch_sample_env<Pendulum64>(a, out, grid, stream);
    case GX_ENV_ACROBOT: return launch_sample_env<AcrobotEuler64>(a, out, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace gx
