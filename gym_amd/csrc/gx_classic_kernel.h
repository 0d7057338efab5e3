// Generic step / rollout / reset kernels for continuous-control envs.
//
// An env type E supplies: Real (float or double), NS (real components), W
// (envs per lane: 16 bytes of each SoA row per lane), Act, Rew, EpRet,
// reset(), step<MODE>(), sample() and, when the observation is not the state
// itself, observe().  One env per lane slot, grid-stride over the batch,
// single step and register-resident K-step rollout in one kernel, with
// auto-reset, TimeLimit and RecordEpisodeStatistics fused in.
#pragma once

#include "gx_device.h"
#include "gx_launch.h"

namespace gx {

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// ------------------------------------------------------------------ math ---
template <int MODE>
struct Trig;

// EXACT: Math.Sin/Cos/Atan2 take and return double; the C# source casts the
// result to float, which rounds once.
template <>
struct Trig<GX_MODE_EXACT> {
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) {
    double sd, cd;
    ::sincos((double)x, &sd, &cd);
    s = (float)sd;
    c = (float)cd;
  }
  static __device__ __forceinline__ float cos(float x) { return (float)::cos((double)x); }
  static __device__ __forceinline__ float sin(float x) { return (float)::sin((double)x); }
  // (float)Math.Cos(x - Math.PI / 2.0) with x a float: the subtraction is in double
  static __device__ __forceinline__ float cos_minus_half_pi(float x) { return (float)::cos((double)x - M_PI / 2.0); }
  // (float)Math.Sin(x + Math.PI)
  static __device__ __forceinline__ float sin_plus_pi(float x) { return (float)::sin((double)x + M_PI); }
  static __device__ __forceinline__ float atan2(float y, float x) { return (float)::atan2((double)y, (double)x); }
};

// FAST32: binary32 device trig, identities applied.
template <>
struct Trig<GX_MODE_FAST32> {
  static __device__ __forceinline__ void sincos(float x, float& s, float& c) { ::sincosf(x, &s, &c); }
  static __device__ __forceinline__ float cos(float x) { return ::cosf(x); }
  static __device__ __forceinline__ float sin(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float cos_minus_half_pi(float x) { return ::sinf(x); }
  static __device__ __forceinline__ float sin_plus_pi(float x) { return -::sinf(x); }
  static __device__ __forceinline__ float atan2(float y, float x) { return ::atan2f(y, x); }
};

// Math.Clamp(float, float, float)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------- vector row access ---
template <class T, int W>
struct VecOf;
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<int32_t, 4> { using type = int4; };
template <> struct VecOf<double, 4> { using type = double4; };
template <> struct VecOf<float, 2> { using type = float2; };
template <> struct VecOf<int32_t, 2> { using type = int2; };
template <> struct VecOf<double, 2> { using type = double2; };

template <class T, int W>
__device__ __forceinline__ void load_row(const T* base, uint64_t group, T (&out)[W]) {
  using V = typename VecOf<T, W>::type;
  V v = reinterpret_cast<const V*>(base)[group];
  out[0] = v.x; out[1] = v.y;
  if constexpr (W == 4) { out[2] = v.z; out[3] = v.w; }
}
template <class T, int W>
__device__ __forceinline__ void store_row(T* base, uint64_t group, const T (&in)[W]) {
  using V = typename VecOf<T, W>::type;
  V v;
  v.x = in[0]; v.y = in[1];
  if constexpr (W == 4) { v.z = in[2]; v.w = in[3]; }
  reinterpret_cast<V*>(base)[group] = v;
}
template <int W>
__device__ __forceinline__ void store_flags(uint8_t* base, uint64_t group, const bool (&b)[W]) {
  if constexpr (W == 4) {
    reinterpret_cast<uint32_t*>(base)[group] =
        (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
  } else {
    reinterpret_cast<uint16_t*>(base)[group] = (uint16_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8));
  }
}

// ---------------------------------------------------------------- kernels ---
template <class E, int MODE, int POLICY>
__global__ void __launch_bounds__(kBlock) classic_step_kernel(StepArgs a) {
  using Real = typename E::Real;
  using Act = typename E::Act;
  using Rew = typename E::Rew;
  using EpRet = typename E::EpRet;
  constexpr int NS = E::NS;
  constexpr int W = E::W;

  const bool autoreset = (a.flags & GX_AUTORESET) != 0;
  const bool record = (a.flags & GX_RECORD_STATS) != 0;
  const int32_t time_limit = a.time_limit;
  Real* const sreal = static_cast<Real*>(a.state_real);
  int32_t* const steps_row = a.state_int;  // these envs have one int row: the step counter
  StatAcc acc;

  const uint64_t ngroups = (a.num_envs + (W - 1)) / W;
  for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < ngroups; q += (uint64_t)gridDim.x * kBlock) {
    Real s[W][NS];
    int32_t steps[W];
    EpRet ep_ret[W] = {};
    int32_t ep_len[W] = {};
#pragma unroll
    for (int c = 0; c < NS; c++) {
      Real v[W];
      load_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
#pragma unroll
      for (int j = 0; j < W; j++) s[j][c] = v[j];
    }
    load_row<int32_t, W>(steps_row, q, steps);
    if (record) {
      load_row<EpRet, W>(static_cast<const EpRet*>(a.ep_return), q, ep_ret);
      if (a.ep_length) load_row<int32_t, W>(a.ep_length, q, ep_len);
    }

    Rew reward[W];
    bool done[W], trunc[W];
    for (int k = 0; k < a.k_steps; k++) {
      const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick + (uint64_t)k);
      Act act[W];
      if (POLICY == GX_POLICY_DEVICE_ACTIONS)
        load_row<Act, W>(static_cast<const Act*>(a.actions) + (uint64_t)k * a.action_stride, q, act);
#pragma unroll
      for (int j = 0; j < W; j++) {
        const uint64_t env = a.env_index_base + q * W + j;
        if (POLICY == GX_POLICY_RANDOM) act[j] = E::sample(philox_block(key, env, 2, 0));
        if (E::kCheckAction && POLICY == GX_POLICY_DEVICE_ACTIONS) {
          if (!E::action_valid(act[j]) && q * W + j < a.num_envs) atomicOr(a.error_flag, 1u);
        }
        bool d;
        E::template step<MODE>(s[j], steps[j], act[j], reward[j], d);
        // Wrappers.TimeLimit.cs:26-31; its counter always equals the env's own
        bool t = time_limit > 0 && steps[j] >= time_limit;
        d = d || t;
        done[j] = d;
        trunc[j] = t;
        if (record) {
          // EnvWrapper.cs:46-55
          ep_ret[j] += (EpRet)reward[j];
          ep_len[j] += 1;
          if (d) {
            const int len = a.ep_length ? ep_len[j] : steps[j];
            if (q * W + j < a.num_envs) acc.add((double)ep_ret[j], len);
            ep_ret[j] = 0;
            ep_len[j] = 0;
          }
        }
        if (autoreset && d) {
          E::reset(s[j], key, env);
          steps[j] = 0;
        }
      }
    }

#pragma unroll
    for (int c = 0; c < NS; c++) {
      Real v[W];
#pragma unroll
      for (int j = 0; j < W; j++) v[j] = s[j][c];
      store_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
    }
    store_row<int32_t, W>(steps_row, q, steps);
    if (record) {
      store_row<EpRet, W>(static_cast<EpRet*>(a.ep_return), q, ep_ret);
      if (a.ep_length) store_row<int32_t, W>(a.ep_length, q, ep_len);
    }
    store_row<Rew, W>(static_cast<Rew*>(a.reward), q, reward);
    store_flags<W>(a.done, q, done);
    if (a.truncated) store_flags<W>(a.truncated, q, trunc);
    if constexpr (E::kDerivedObs) {
      float o[W][E::NOBS];
#pragma unroll
      for (int j = 0; j < W; j++) E::observe(s[j], o[j]);
#pragma unroll
      for (int c = 0; c < E::NOBS; c++) {
        float v[W];
#pragma unroll
        for (int j = 0; j < W; j++) v[j] = o[j][c];
        store_row<float, W>(a.obs + (uint64_t)c * a.stride, q, v);
      }
    }
  }
  if (record) block_commit_stats(acc, a.stat_slots);
}

// Reset() of every env, or (when only_observe) just the derived observation
// rows of the current state.
template <class E>
__global__ void __launch_bounds__(kBlock) classic_reset_kernel(StepArgs a, int only_observe) {
  using Real = typename E::Real;
  constexpr int NS = E::NS;
  constexpr int W = E::W;
  Real* const sreal = static_cast<Real*>(a.state_real);
  const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick);
  const uint64_t ngroups = (a.num_envs + (W - 1)) / W;
  for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < ngroups; q += (uint64_t)gridDim.x * kBlock) {
    Real s[W][NS];
    if (only_observe) {
#pragma unroll
      for (int c = 0; c < NS; c++) {
        Real v[W];
        load_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
#pragma unroll
        for (int j = 0; j < W; j++) s[j][c] = v[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < W; j++) E::reset(s[j], key, a.env_index_base + q * W + j);
#pragma unroll
      for (int c = 0; c < NS; c++) {
        Real v[W];
#pragma unroll
        for (int j = 0; j < W; j++) v[j] = s[j][c];
        store_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
      }
      int32_t zero[W] = {};
      store_row<int32_t, W>(a.state_int, q, zero);
    }
    if constexpr (E::kDerivedObs) {
      float o[W][E::NOBS];
#pragma unroll
      for (int j = 0; j < W; j++) E::observe(s[j], o[j]);
#pragma unroll
      for (int c = 0; c < E::NOBS; c++) {
        float v[W];
#pragma unroll
        for (int j = 0; j < W; j++) v[j] = o[j][c];
        store_row<float, W>(a.obs + (uint64_t)c * a.stride, q, v);
      }
    }
  }
}

// Actions of the random policy for one tick, for harnesses that replay them.
template <class E>
__global__ void __launch_bounds__(kBlock) classic_sample_kernel(StepArgs a, void* out) {
  using Act = typename E::Act;
  const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick);
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock)
    static_cast<Act*>(out)[i] = E::sample(philox_block(key, a.env_index_base + i, 2, 0));
}

// -------------------------------------------------------------- dispatch ---
template <class E>
static hipError_t launch_step_env(const StepArgs& a, int mode, int policy, dim3 grid, hipStream_t stream) {
  if (mode == GX_MODE_EXACT) {
    if (policy == GX_POLICY_RANDOM)
      hipLaunchKernelGGL((classic_step_kernel<E, GX_MODE_EXACT, GX_POLICY_RANDOM>), grid, dim3(kBlock), 0, stream, a);
    else
      hipLaunchKernelGGL((classic_step_kernel<E, GX_MODE_EXACT, GX_POLICY_DEVICE_ACTIONS>), grid, dim3(kBlock), 0, stream, a);
  } else {
    if (policy == GX_POLICY_RANDOM)
      hipLaunchKernelGGL((classic_step_kernel<E, GX_MODE_FAST32, GX_POLICY_RANDOM>), grid, dim3(kBlock), 0, stream, a);
    else
      hipLaunchKernelGGL((classic_step_kernel<E, GX_MODE_FAST32, GX_POLICY_DEVICE_ACTIONS>), grid, dim3(kBlock), 0, stream, a);
  }
  return hipGetLastError();
}

template <class E>
static hipError_t launch_reset_env(const StepArgs& a, int only_observe, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(classic_reset_kernel<E>, grid, dim3(kBlock), 0, stream, a, only_observe);
  return hipGetLastError();
}

template <class E>
static hipError_t launch_sample_env(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(classic_sample_kernel<E>, grid, dim3(kBlock), 0, stream, a, out);
  return hipGetLastError();
}

}  // namespace gx
