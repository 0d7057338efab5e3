// Classic-control dynamics as gfx950 kernels: CartPole, MountainCar,
// MountainCarContinuous, Pendulum (Acrobot, the reference's semi-implicit
// Euler form, is in gx_acrobot.hip).  Four consecutive envs per lane: every SoA row is read and
// written as one 16-byte vector per lane.
//
// EXACT mode reproduces the C# source operation for operation: every
// `float op float` is one binary32 operation (this file is compiled with
// -ffp-contract=off so nothing is fused), Math.Sin/Cos/Atan2 are evaluated in
// binary64 and rounded once by the (float) cast.
#pragma clang fp contract(off)

#include "gx_classic_kernel.h"

namespace gx {

// `x / total_mass` with total_mass = 1.0f + 0.1f: a correctly rounded quotient from three
// multiply-adds (bit-identical to the divide for every float of moderate size), with the
// true divide kept for zero, infinities, NaN and the far ends of the exponent range.
__device__ __forceinline__ float div_total_mass(float x) {
  float q = math::div_by_total_mass(x);
  if (__builtin_expect(!math::div_fast_range(x), 0)) q = x / (1.0f + 0.1f);
  return q;
}

// Each env: NS real components, action type, reward type, Reset and Step that
// follow the cited reference lines.  `steps` is the env's own `_steps` field.

struct CartPole {
  // the lane's four env states wait in LDS while one is stepped, and the stores form their addresses again: 89 -> 55
  // registers, eight waves per SIMD instead of five.  The single step does not care (0.146 ms either way, it is
  // not short of waves); the K = 64 rollout, which is arithmetic from end to end, gains 10 % (3.37 -> 3.03 ms).
  // MountainCar, MountainCarContinuous and Pendulum lose 1-2 % on the single step with the same change and keep
  // their states in registers.
  static constexpr bool kStateInLds = true;
  static constexpr bool kRecomputeAddresses = true;
  static constexpr int NS = 4;
  using Real = float;
  static constexpr int W = 4;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 4;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 2;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  // theta is unbounded when the pole keeps falling without a reset; with auto-reset every
  // stored theta is within the 12-degree threshold (Envs.CartPole.cs:45,70) or a fresh draw
  static constexpr bool kAlwaysBounded = false;
  // reward 1 on every step but the final one (Envs.CartPole.cs:72)
  static constexpr int kStepReturn = 1;

  // Envs.CartPole.cs:22-32
  static __device__ __forceinline__ void reset(float (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d = philox_block(key, env, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = (float)(d.uniform(i) * 0.1 - 0.05);
  }

  // Envs.CartPole.cs:34-75
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    const float gravity = 9.8f;
    const float masspole = 0.1f;
    const float length = 0.5f;
    const float polemass_length = masspole * length;
    const float force_mag = 10.0f;
    const float tau = 0.02f;
    const float theta_threshold_radians = 12 * (float)M_PI / 180;
    const float x_threshold = 2.4f;

    float x = s[0], xDot = s[1], theta = s[2], thetaDot = s[3];
    float force = action == 1 ? force_mag : -force_mag;
    float sintheta, costheta;
    Trig<MODE, BOUNDED>::sincos_quarter(theta, sintheta, costheta);

    float temp = div_total_mass(force + polemass_length * thetaDot * thetaDot * sintheta);
    float thetaacc = divide<MODE>(gravity * sintheta - costheta * temp,
                                  length * (4.0f / 3.0f - div_total_mass(masspole * costheta * costheta)));
    float xacc = temp - div_total_mass(polemass_length * thetaacc * costheta);

    x += tau * xDot;
    xDot += tau * xacc;
    theta += tau * thetaDot;
    thetaDot += tau * thetaacc;
    s[0] = x; s[1] = xDot; s[2] = theta; s[3] = thetaDot;
    steps++;

    done = x < -x_threshold || x > x_threshold || theta < -theta_threshold_radians ||
           theta > theta_threshold_radians || steps >= 500;
    reward = !done ? 1.0f : 0.0f;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

// TABLE: EXACT sin and cos through the LDS table (gx_trig.h).  Over every float argument of these envs (1.3 * 10^10
// results, tests/host_math_check.cpp with stride 0) the table routine differs from the C library's float in two
// results and the polynomial routine in three, so which one a kernel uses is a matter of speed only: the table routine is a third shorter, and its table costs every
// workgroup a 2 KiB fill.  The K = 64 rollouts gain with it (Pendulum 4.90 -> 4.22 ms at 16 Mi envs), the single
// steps, which wait on memory, lose 0.5-2 % and keep the polynomial routine.
template <int MODE, bool TABLE>
struct ClassicTrigOf { using type = Trig<MODE, true>; };
template <>
struct ClassicTrigOf<GX_MODE_EXACT, true> { using type = TrigTable; };

// MountainCar's position is unbounded without auto-reset, so the table routine (bounded arguments) serves its
// BOUNDED EXACT form only.
template <int MODE, bool BOUNDED, bool TABLE>
struct CarTrigOf { using type = Trig<MODE, BOUNDED>; };
template <>
struct CarTrigOf<GX_MODE_EXACT, true, true> { using type = TrigTable; };
template <int MODE, bool BOUNDED, bool TABLE>
using CarTrig = typename CarTrigOf<MODE, BOUNDED, TABLE>::type;

template <bool TABLE>
struct MountainCarT {
  static constexpr bool kSinCosTable = TABLE;
  static constexpr int NS = 2;
  using Real = float;
  static constexpr int W = 4;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 2;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 3;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  // the position has no upper clamp (Envs.MountainCar.cs:38-44): past the goal it only stops
  // growing when the env is reset
  static constexpr bool kAlwaysBounded = false;
  // reward -1 on every step but the final one (Envs.MountainCar.cs:45)
  static constexpr int kStepReturn = -1;

  // Envs.MountainCar.cs:22-30
  static __device__ __forceinline__ void reset(float (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d0 = philox_block(key, env, 0, 0);
    s[0] = (float)(d0.uniform(0) * 0.2 - 0.6);
    s[1] = 0.0f;
  }

  // Envs.MountainCar.cs:32-50
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    float position = s[0], velocity = s[1];
    float force = (float)(action - 1);
    velocity += 0.001f * force - 0.0025f * CarTrig<MODE, BOUNDED, TABLE>::cos(3 * position);
    velocity = clampf(velocity, -0.07f, 0.07f);
    position += velocity;
    if (position < -1.2f) {
      position = -1.2f;
      velocity = 0.0f;
    }
    done = position >= 0.5f;
    reward = done ? 0.0f : -1.0f;
    s[0] = position; s[1] = velocity;
    steps++;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};
struct MountainCar : MountainCarT<false> {
  using Rollout = MountainCarT<true>;
};

template <bool TABLE>
struct MountainCarContinuousT {
  static constexpr bool kSinCosTable = TABLE;
  static constexpr bool kPrefetchNextTrip = true;
  static constexpr int NS = 2;
  using Real = float;
  static constexpr int W = 4;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 2;
  using Act = float;
  using Rew = double;  // `velocity * 0.1` is a double product in the reference
  using EpRet = double;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  static constexpr bool kAlwaysBounded = false;
  // rewards vary from step to step
  static constexpr int kStepReturn = kNoStepReturn;

  // Envs.MountainCarContinuous.cs:19-27
  static __device__ __forceinline__ void reset(float (&s)[NS], const RngKey& key, uint64_t env) {
    MountainCarT<false>::reset(s, key, env);
  }

  // Envs.MountainCarContinuous.cs:29-49
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    float position = s[0], velocity = s[1];
    float force = clampf(action, -1.0f, 1.0f);
    velocity += force * 0.0015f - 0.0025f * CarTrig<MODE, BOUNDED, TABLE>::cos(3 * position);
    velocity = clampf(velocity, -0.07f, 0.07f);
    position += velocity;
    if (position < -1.2f) {
      position = -1.2f;
      velocity = 0.0f;
    }
    done = position >= 0.45f;
    double r = (double)velocity * 0.1;
    if (done) r += 100.0;
    reward = r;
    s[0] = position; s[1] = velocity;
    steps++;
  }
  // Box([-1],[1]), Envs.MountainCarContinuous.cs:12
  static __device__ __forceinline__ Act sample(uint32_t w) { return box_sample(word_uniform(w), -1.0f, 1.0f); }
};
struct MountainCarContinuous : MountainCarContinuousT<false> {
  using Rollout = MountainCarContinuousT<true>;
};

template <bool TABLE>
struct PendulumT {
  template <int MODE>
  using PendulumTrig = typename ClassicTrigOf<MODE, TABLE>::type;
  static constexpr bool kPrefetchNextTrip = true;
  static constexpr bool kSinCosTable = TABLE;
  static constexpr int NS = 3;
  using Real = float;
  static constexpr int W = 4;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 3;
  using Act = float;
  using Rew = double;
  using EpRet = double;
  static constexpr bool kCheckAction = false;
  static __device__ __forceinline__ bool action_valid(Act) { return true; }
  // theta comes out of atan2 every step and the speed is clamped to 8 before it is
  // integrated (Envs.Pendulum.cs:40,46-47): every angle is below 4 for any finite state
  static constexpr bool kAlwaysBounded = true;
  // rewards vary from step to step
  static constexpr int kStepReturn = kNoStepReturn;

  // Envs.Pendulum.cs:19-28
  static __device__ __forceinline__ void reset(float (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d0 = philox_block(key, env, 0, 0);
    float theta = (float)(d0.uniform(0) * 2 * M_PI - M_PI);
    float thetaDot = (float)(d0.uniform(1) * 1 - 0.5);
    Trig<GX_MODE_EXACT, true>::sincos(theta, s[1], s[0]);  // (the polynomial routine in either form: a reset is rare)
    s[2] = thetaDot;
  }

  // Envs.Pendulum.cs:30-60
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    const float max_speed = 8.0f;
    const float max_torque = 2.0f;
    const float dt = 0.05f;
    const float g = 10.0f;
    const float m = 1.0f;
    const float l = 1.0f;

    using T = PendulumTrig<MODE>;
    float theta = T::atan2(s[1], s[0]);
    float thetaDot = s[2];
    float u = clampf(action, -max_torque, max_torque);

    float new_thetaDot =
        thetaDot + (-3 * g / (2 * l) * T::sin_plus_pi(theta) + 3.0f / (m * l * l) * u) * dt;
    new_thetaDot = clampf(new_thetaDot, -max_speed, max_speed);
    float new_theta = theta + new_thetaDot * dt;

    T::sincos(new_theta, s[1], s[0]);
    s[2] = new_thetaDot;
    steps++;

    // float*float promoted, then double products, summed left to right
    reward = -((double)(theta * theta) + 0.1 * (double)thetaDot * (double)thetaDot + 0.001 * (double)u * (double)u);
    done = false;
  }
  // Box([-2],[2]), Envs.Pendulum.cs:12
  static __device__ __forceinline__ Act sample(uint32_t w) { return box_sample(word_uniform(w), -2.0f, 2.0f); }
};
struct Pendulum : PendulumT<false> {
  using Rollout = PendulumT<true>;
};

hipError_t launch_classic_step(int env_id, const StepArgs& a, int mode, int policy, bool fused, bool bounded, dim3 grid,
                               hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_step_env<CartPole>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_step_env<MountainCar>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_step_env<MountainCarContinuous>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_PENDULUM: return launch_step_env<Pendulum>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_ACROBOT:
    case kEnvAcrobotAngles: return launch_acrobot_step(env_id, a, mode, policy, fused, bounded, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_reset_env<CartPole>(a, 0, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_reset_env<MountainCar>(a, 0, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_reset_env<MountainCarContinuous>(a, 0, grid, stream);
    case GX_ENV_PENDULUM: return launch_reset_env<Pendulum>(a, 0, grid, stream);
    case GX_ENV_ACROBOT:
    case kEnvAcrobotAngles: return launch_acrobot_reset(env_id, a, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic_sample(int env_id, const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_sample_env<CartPole>(a, out, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_sample_env<MountainCar>(a, out, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_sample_env<MountainCarContinuous>(a, out, grid, stream);
    case GX_ENV_PENDULUM: return launch_sample_env<Pendulum>(a, out, grid, stream);
    case GX_ENV_ACROBOT:
    case kEnvAcrobotAngles: return launch_acrobot_sample(a, out, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic_derive(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  if (env_id != kEnvAcrobotAngles) return hipErrorInvalidValue;
  return launch_acrobot_derive(a, grid, stream);
}

}  // namespace gx
