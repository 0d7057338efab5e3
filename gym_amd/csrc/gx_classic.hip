// Classic-control dynamics as gfx950 kernels: CartPole, MountainCar,
// MountainCarContinuous, Pendulum, Acrobot (the reference's semi-implicit
// Euler form).  Four consecutive envs per lane: every SoA row is read and
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

struct MountainCar {
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
    velocity += 0.001f * force - 0.0025f * Trig<MODE, BOUNDED>::cos(3 * position);
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

struct MountainCarContinuous {
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
    MountainCar::reset(s, key, env);
  }

  // Envs.MountainCarContinuous.cs:29-49
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    float position = s[0], velocity = s[1];
    float force = clampf(action, -1.0f, 1.0f);
    velocity += force * 0.0015f - 0.0025f * Trig<MODE, BOUNDED>::cos(3 * position);
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

struct Pendulum {
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
    Trig<GX_MODE_EXACT, true>::sincos(theta, s[1], s[0]);
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

    float theta = Trig<MODE, BOUNDED>::atan2(s[1], s[0]);
    float thetaDot = s[2];
    float u = clampf(action, -max_torque, max_torque);

    float new_thetaDot =
        thetaDot + (-3 * g / (2 * l) * Trig<MODE, BOUNDED>::sin_plus_pi(theta) + 3.0f / (m * l * l) * u) * dt;
    new_thetaDot = clampf(new_thetaDot, -max_speed, max_speed);
    float new_theta = theta + new_thetaDot * dt;

    Trig<MODE, BOUNDED>::sincos(new_theta, s[1], s[0]);
    s[2] = new_thetaDot;
    steps++;

    // float*float promoted, then double products, summed left to right
    reward = -((double)(theta * theta) + 0.1 * (double)thetaDot * (double)thetaDot + 0.001 * (double)u * (double)u);
    done = false;
  }
  // Box([-2],[2]), Envs.Pendulum.cs:12
  static __device__ __forceinline__ Act sample(uint32_t w) { return box_sample(word_uniform(w), -2.0f, 2.0f); }
};

#ifndef GX_ACROBOT_W
#define GX_ACROBOT_W 4
#endif
#ifndef GX_ACROBOT_ANGLES_LDS
#define GX_ACROBOT_ANGLES_LDS 1
#endif
#ifndef GX_ACROBOT_ANGLES_WAVES
#define GX_ACROBOT_ANGLES_WAVES 4
#endif
// Envs.Acrobot.cs:36-74: from the two angles and their velocities to the values after one step (the angles
// are not wrapped: the source keeps their cosines and sines).
template <int MODE, bool BOUNDED>
static __device__ __forceinline__ void acrobot_integrate(float& theta1, float& theta2, float& dtheta1, float& dtheta2, int32_t action) {
  using T = Trig<MODE, BOUNDED>;
  const float LINK_LENGTH_1 = 1.0f;
  const float LINK_MASS_1 = 1.0f;
  const float LINK_MASS_2 = 1.0f;
  const float LINK_COM_POS_1 = 0.5f;
  const float LINK_COM_POS_2 = 0.5f;
  const float LINK_MOI_1 = 1.0f;
  const float LINK_MOI_2 = 1.0f;
  const float MAX_VEL_1 = 4 * (float)M_PI;
  const float MAX_VEL_2 = 9 * (float)M_PI;
  const float dt = 0.2f;
  const float g = 9.8f;

  float u = action == 0 ? -1.0f : (action == 2 ? 1.0f : 0.0f);

  float sin2, cos2;
  T::sincos(theta2, sin2, cos2);

  float d1 = LINK_MASS_1 * LINK_COM_POS_1 * LINK_COM_POS_1 +
             LINK_MASS_2 * (LINK_LENGTH_1 * LINK_LENGTH_1 + LINK_COM_POS_2 * LINK_COM_POS_2 +
                            2 * LINK_LENGTH_1 * LINK_COM_POS_2 * cos2) +
             LINK_MOI_1 + LINK_MOI_2;
  float d2 = LINK_MASS_2 * (LINK_COM_POS_2 * LINK_COM_POS_2 + LINK_LENGTH_1 * LINK_COM_POS_2 * cos2) + LINK_MOI_2;
  float phi2 = LINK_MASS_2 * LINK_COM_POS_2 * g * T::cos_minus_half_pi(theta1 + theta2);
  float phi1 = -LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta2 * dtheta2 * sin2 -
               2 * LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta2 * dtheta1 * sin2 +
               (LINK_MASS_1 * LINK_COM_POS_1 + LINK_MASS_2 * LINK_LENGTH_1) * g * T::cos_minus_half_pi(theta1) +
               phi2;
  float ddtheta2, ddtheta1;
#ifdef GX_OLD_DIV  // A/B builds only: the hardware divide sequence
  constexpr bool kSharedReciprocals = false;
#else
  constexpr bool kSharedReciprocals = true;
#endif
  if (MODE == GX_MODE_EXACT && kSharedReciprocals) {
    // Four IEEE quotients over two divisors, d1 = 3.5 + cos2 in [2.5, 4.5] and the denominator below in
    // [0.57, 1.03]: each divisor's reciprocal is taken once in binary64 and every quotient is one product
    // rounded to float -- the same bits as the divide (see math::div_by_rcp), half the instructions.
    const double r_d1 = math::rcp_of_f32(d1);
    const float den = LINK_MASS_2 * LINK_COM_POS_2 * LINK_COM_POS_2 + LINK_MOI_2 - math::div_by_rcp(d2 * d2, r_d1);
    ddtheta2 = math::div_by_rcp(u + math::div_by_rcp(d2, r_d1) * phi1 - LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta1 * dtheta1 * sin2 - phi2,
                                math::rcp_of_f32(den));
    ddtheta1 = math::div_by_rcp(-(d2 * ddtheta2 + phi1), r_d1);
  } else {
    ddtheta2 = divide<MODE>(u + divide<MODE>(d2, d1) * phi1 - LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta1 * dtheta1 * sin2 - phi2,
                            LINK_MASS_2 * LINK_COM_POS_2 * LINK_COM_POS_2 + LINK_MOI_2 - divide<MODE>(d2 * d2, d1));
    ddtheta1 = divide<MODE>(-(d2 * ddtheta2 + phi1), d1);
  }

  dtheta1 += dt * ddtheta1;
  dtheta2 += dt * ddtheta2;
  dtheta1 = clampf(dtheta1, -MAX_VEL_1, MAX_VEL_1);
  dtheta2 = clampf(dtheta2, -MAX_VEL_2, MAX_VEL_2);
  theta1 += dt * dtheta1;
  theta2 += dt * dtheta2;
}

// Envs.Acrobot.cs:80-81: tip_y = -cos(theta1) - cos(theta1 + theta2) is only compared with 1.
// The addition formula on the float sines and cosines is within 2e-6 of it (their roundings
// plus the rounding of the float sum theta1 + theta2, below 16 in magnitude), so only a tip
// within 1e-5 of the threshold needs the expression evaluated as the source writes it.
template <int MODE, bool BOUNDED>
static __device__ __forceinline__ bool acrobot_tip_above(float theta1, float theta2, float c1n, float s1n, float c2n, float s2n) {
  using T = Trig<MODE, BOUNDED>;
  const float LINK_LENGTH_1 = 1.0f;
  const float LINK_LENGTH_2 = 1.0f;
  bool above;
  if (MODE == GX_MODE_EXACT) {
    const float tip_approx = -c1n - (c1n * c2n - s1n * s2n);
    above = tip_approx > 1.0f;
    if (fabsf(tip_approx - 1.0f) < 1e-5f) above = -LINK_LENGTH_1 * c1n - LINK_LENGTH_2 * T::cos(theta1 + theta2) > 1.0f;
  } else {
    above = -c1n - (c1n * c2n - s1n * s2n) > 1.0f;  // cos(theta1 + theta2) by the addition formula
  }
  return above;
}

struct Acrobot {
  static constexpr int NS = 6;
  using Real = float;
  static constexpr int W = GX_ACROBOT_W;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 6;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 3;
  static constexpr bool kCheckAction = true;
  static __device__ __forceinline__ bool action_valid(Act a) { return (uint32_t)a <= 2u; }
  // both angles come out of atan2 every step and the velocities are clamped to 4pi / 9pi
  // before they are integrated (Envs.Acrobot.cs:47-48,71-74): every angle is below 15
  static constexpr bool kAlwaysBounded = true;
  // reward -1 on every step but the final one (Envs.Acrobot.cs:88)
  static constexpr int kStepReturn = -1;

  // Envs.Acrobot.cs:22-28
  static __device__ __forceinline__ void reset(float (&s)[NS], const RngKey& key, uint64_t env) {
    Draws d0 = philox_block(key, env, 0, 0), d1 = philox_block(key, env, 0, 1);
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = (float)(d0.uniform(i) * 0.1 - 0.05);
    s[4] = (float)(d1.uniform(0) * 0.1 - 0.05);
    s[5] = (float)(d1.uniform(1) * 0.1 - 0.05);
  }

  // Envs.Acrobot.cs:30-91.  An action outside {0,1,2} throws in the reference
  // (Envs.Acrobot.cs:51); the host validates before launch where it can and
  // the kernel treats such a lane as torque 0 and raises the error flag.
  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    using T = Trig<MODE, BOUNDED>;
    float theta1 = T::atan2(s[1], s[0]);
    float theta2 = T::atan2(s[3], s[2]);
    float dtheta1 = s[4];
    float dtheta2 = s[5];
    acrobot_integrate<MODE, BOUNDED>(theta1, theta2, dtheta1, dtheta2, action);

    float c1n, s1n, c2n, s2n;
    T::sincos(theta1, s1n, c1n);
    T::sincos(theta2, s2n, c2n);
    s[0] = c1n; s[1] = s1n; s[2] = c2n; s[3] = s2n; s[4] = dtheta1; s[5] = dtheta2;
    steps++;

    done = acrobot_tip_above<MODE, BOUNDED>(theta1, theta2, c1n, s1n, c2n, s2n) || steps >= 500;
    reward = done ? 0.0f : -1.0f;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

// The same env in EXACT mode for batches whose rows the library owns, with the two angles kept next to the state.
// The source stores (cos, sin) of each angle as floats and starts every step by taking the angles back out of
// them with atan2 (Envs.Acrobot.cs:36-37).  Right after the step has evaluated that sine and cosine in binary64
// the same float angle costs a third of an atan2 (math::atan2_of_rounded_sincos), so the step leaves it in two
// more rows, 6 and 7, behind the six rows of the state image, and the next step starts from those: it reads
// rows 4..7 (the velocities and the angles) and rewrites all eight -- the bytes moved stay what they were, two
// arc tangents per env-step are gone.  Rows 0..3 are written for the caller (observation, gx_vec_get_state) and
// never read by a step; a state written from outside (gx_vec_set_state) is followed by derive().
struct AcrobotAngles {
  static constexpr int NS = 8;
  static constexpr int kImageRows = 6;      // rows of the state image; the rest is derived from them
  static constexpr int kWriteOnlyRows = 4;  // leading rows a step does not read
  static constexpr bool kRecomputeAddresses = true;
  static constexpr bool kStateInLds = GX_ACROBOT_ANGLES_LDS != 0;
  static constexpr int kMinWaves = GX_ACROBOT_ANGLES_WAVES;  // resident waves per SIMD the fused kernels are compiled for
  using Real = float;
  static constexpr int W = GX_ACROBOT_W;
  static constexpr bool kDerivedObs = false;
  static constexpr int NOBS = 6;
  using Act = int32_t;
  using Rew = float;
  using EpRet = int32_t;
  static constexpr int NACT = 3;
  static constexpr bool kCheckAction = true;
  static __device__ __forceinline__ bool action_valid(Act a) { return (uint32_t)a <= 2u; }
  static constexpr bool kAlwaysBounded = true;
  static constexpr int kStepReturn = -1;
  using T = Trig<GX_MODE_EXACT, true>;

  // the angles as the source's step would read them out of the stored cosines and sines (Envs.Acrobot.cs:36-37)
  static __device__ __forceinline__ void derive(float (&s)[NS]) {
    s[6] = T::atan2(s[1], s[0]);
    s[7] = T::atan2(s[3], s[2]);
  }
  // Envs.Acrobot.cs:22-28: six small uniforms, not a point of the unit circles
  static __device__ __forceinline__ void reset(float (&s)[NS], const RngKey& key, uint64_t env) {
    float head[Acrobot::NS];
    Acrobot::reset(head, key, env);
#pragma unroll
    for (int i = 0; i < Acrobot::NS; i++) s[i] = head[i];
    derive(s);
  }

  template <int MODE, bool BOUNDED>
  static __device__ __forceinline__ void step(float (&s)[NS], int32_t& steps, Act action, Rew& reward, bool& done) {
    static_assert(MODE == GX_MODE_EXACT && BOUNDED, "the kept angles are those of the EXACT arithmetic");
    float theta1 = s[6];
    float theta2 = s[7];
    float dtheta1 = s[4];
    float dtheta2 = s[5];
    acrobot_integrate<MODE, BOUNDED>(theta1, theta2, dtheta1, dtheta2, action);

    // T::sincos with the binary64 values kept for the angle
    double S1, C1, S2, C2;
    math::sincos_bounded((double)theta1, S1, C1);
    math::sincos_bounded((double)theta2, S2, C2);
    const float c1n = (float)C1, s1n = (float)S1, c2n = (float)C2, s2n = (float)S2;
    s[0] = c1n; s[1] = s1n; s[2] = c2n; s[3] = s2n; s[4] = dtheta1; s[5] = dtheta2;
    s[6] = math::atan2_of_rounded_sincos(theta1, S1, C1, s1n, c1n);
    s[7] = math::atan2_of_rounded_sincos(theta2, S2, C2, s2n, c2n);
    steps++;

    done = acrobot_tip_above<MODE, BOUNDED>(theta1, theta2, c1n, s1n, c2n, s2n) || steps >= 500;
    reward = done ? 0.0f : -1.0f;
  }
  static __device__ __forceinline__ Act sample(uint32_t w) { return word_below(w, NACT); }
};

hipError_t launch_classic_step(int env_id, const StepArgs& a, int mode, int policy, bool fused, bool bounded, dim3 grid,
                               hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_step_env<CartPole>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_step_env<MountainCar>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_step_env<MountainCarContinuous>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_PENDULUM: return launch_step_env<Pendulum>(a, mode, policy, fused, bounded, grid, stream);
    case GX_ENV_ACROBOT: return launch_step_env<Acrobot>(a, mode, policy, fused, bounded, grid, stream);
    case kEnvAcrobotAngles:
      if (mode != GX_MODE_EXACT) return hipErrorInvalidValue;
      return launch_step_env_exact<AcrobotAngles>(a, policy, fused, bounded, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  switch (env_id) {
    case GX_ENV_CARTPOLE: return launch_reset_env<CartPole>(a, 0, grid, stream);
    case GX_ENV_MOUNTAINCAR: return launch_reset_env<MountainCar>(a, 0, grid, stream);
    case GX_ENV_MOUNTAINCAR_CONT: return launch_reset_env<MountainCarContinuous>(a, 0, grid, stream);
    case GX_ENV_PENDULUM: return launch_reset_env<Pendulum>(a, 0, grid, stream);
    case GX_ENV_ACROBOT: return launch_reset_env<Acrobot>(a, 0, grid, stream);
    case kEnvAcrobotAngles: return launch_reset_env<AcrobotAngles>(a, 0, grid, stream);
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
    case kEnvAcrobotAngles: return launch_sample_env<Acrobot>(a, out, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_classic_derive(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  if (env_id != kEnvAcrobotAngles) return hipErrorInvalidValue;
  hipLaunchKernelGGL(classic_derive_kernel<AcrobotAngles>, grid, dim3(kClassicBlock), 0, stream, a);
  return hipGetLastError();
}

}  // namespace gx
