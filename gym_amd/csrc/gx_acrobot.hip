// Acrobot as the reference integrates it (semi-implicit Euler in binary32, Envs.Acrobot.cs), in the two forms the
// batches use: on six rows as the state image has them, and with the two angles kept in rows of their own.
// A translation unit of its own: these kernels are bound by vector-instruction issue with the scalar register
// file full, and gain 1-2 % when the compiler does not hoist loop-invariant scalar values out of the sweep
// (-mllvm -disable-machine-licm, see gym_amd/build.py), which the memory-bound classic-control kernels do not.
#pragma clang fp contract(off)

#include "gx_classic_kernel.h"

namespace gx {

#ifndef GX_ACROBOT_W
#define GX_ACROBOT_W 4
#endif
#ifndef GX_ACROBOT_ANGLES_LDS
#define GX_ACROBOT_ANGLES_LDS 1
#endif
#ifndef GX_ACROBOT_ANGLES_WAVES
#define GX_ACROBOT_ANGLES_WAVES 4
#endif
// EXACT: the table-driven routines (these kernels are bound by instruction issue, and five sin/cos evaluations are
// most of a step); FAST32: the binary32 ones.  Arguments are always bounded (kAlwaysBounded below).
template <int MODE>
struct AcrobotTrigOf {
  using type = Trig<MODE, true>;
};
template <>
struct AcrobotTrigOf<GX_MODE_EXACT> {
  using type = TrigTable;
};
#ifdef GX_ACROBOT_NO_TABLE  // A/B builds only: the polynomial routines with quadrant selection
template <int MODE>
using AcrobotTrig = Trig<MODE, true>;
constexpr bool kAcrobotSinCosTable = false;
#else
template <int MODE>
using AcrobotTrig = typename AcrobotTrigOf<MODE>::type;
constexpr bool kAcrobotSinCosTable = true;
#endif

// Envs.Acrobot.cs:36-74: from the two angles and their velocities to the values after one step (the angles
// are not wrapped: the source keeps their cosines and sines).
template <int MODE, bool BOUNDED>
static __device__ __forceinline__ void acrobot_integrate(float& theta1, float& theta2, float& dtheta1, float& dtheta2, int32_t action) {
  using T = AcrobotTrig<MODE>;
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
  using T = AcrobotTrig<MODE>;
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
  static constexpr bool kSinCosTable = kAcrobotSinCosTable;
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
    using T = AcrobotTrig<MODE>;
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
  // The fused rollouts are arithmetic from end to end: three waves with the states in registers beat four with the
  // states in LDS there (K = 64 at 16 Mi envs: 8.20 against 8.65 ms), while the single step, which waits on memory
  // as well, wants the waves (0.205 against 0.213 ms).
  static constexpr int kMinWavesRollout = 3;
  static constexpr bool kPrefetchNextTrip = true;
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
  static constexpr bool kSinCosTable = kAcrobotSinCosTable;
  static constexpr int kStepReturn = -1;
  using T = AcrobotTrig<GX_MODE_EXACT>;

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
#ifdef GX_ACROBOT_NO_TABLE
    math::sincos_bounded((double)theta1, S1, C1);
    math::sincos_bounded((double)theta2, S2, C2);
#else
    TrigTable::sincos64((double)theta1, S1, C1);
    TrigTable::sincos64((double)theta2, S2, C2);
#endif
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

hipError_t launch_acrobot_step(int env_id, const StepArgs& a, int mode, int policy, bool fused, bool bounded, dim3 grid,
                               hipStream_t stream) {
  if (env_id == kEnvAcrobotAngles) {
    if (mode != GX_MODE_EXACT) return hipErrorInvalidValue;
    return launch_step_env_exact<AcrobotAngles>(a, policy, fused, bounded, grid, stream);
  }
  return launch_step_env<Acrobot>(a, mode, policy, fused, bounded, grid, stream);
}
hipError_t launch_acrobot_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  return env_id == kEnvAcrobotAngles ? launch_reset_env<AcrobotAngles>(a, 0, grid, stream) : launch_reset_env<Acrobot>(a, 0, grid, stream);
}
hipError_t launch_acrobot_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  return launch_sample_env<Acrobot>(a, out, grid, stream);
}
hipError_t launch_acrobot_derive(const StepArgs& a, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(classic_derive_kernel<AcrobotAngles>, grid, dim3(kClassicBlock), 0, stream, a);
  return hipGetLastError();
}

}  // namespace gx
