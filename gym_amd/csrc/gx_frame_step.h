// The step / reset / sample kernels shared by the frame games (Pong, Breakout, SpaceInvaders):
// one env per lane, the game state in registers for K steps, TimeLimit, RecordEpisodeStatistics
// and auto-reset applied as in the other kernels.  A game supplies
//   struct G { struct State; static constexpr int kActions;
//              static void load(const StepArgs&, uint64_t i, State&); static void store(...);
//              static void prime(const StepArgs&, uint64_t i, State&);   // caller-supplied draws, where a game takes them
//              static void reset(State&, const RngKey&, uint64_t env);
//              static void step(State&, int action, const RngKey&, uint64_t env, float& reward, bool& done);
//              static int32_t steps(const State&);
//              static constexpr int kRecordDwords; static void pack(const State&, uint32_t* record); };
// pack() writes what the game's raster needs of a state as one short record: the raster's workgroups (several
// per frame) then read one cache line each instead of one line per state row.  The step and reset kernels write
// the records as they store the states; frame_pack_kernel rebuilds them after the states were written from outside.
#pragma once

#include "gx_device.h"

namespace gx {

// ROWS: the variant behind gx_vec_rollout_out (per-step reward / done / truncated rows, or none)
template <class G, int POLICY, bool ROWS = false>
__global__ void __launch_bounds__(kBlock) frame_step_kernel(StepArgs a) {
  const bool autoreset = (a.flags & GX_AUTORESET) != 0;
  const bool record = (a.flags & GX_RECORD_STATS) != 0;
  StatAcc acc;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock) {
    typename G::State s;
    G::load(a, i, s);
    G::prime(a, i, s);
    int32_t ep_ret = 0, ep_len = 0;
    if (record) {
      ep_ret = static_cast<const int32_t*>(a.ep_return)[i];
      if (a.ep_length) ep_len = a.ep_length[i];
    }
    const uint64_t env = a.env_index_base + i;
    float reward = 0.0f;
    bool done = false, trunc = false;
    Draws policy;
    for (int k = 0; k < a.k_steps; k++) {
      const uint64_t tick = a.tick + (uint64_t)k;
      const RngKey key = make_key(a.seed_lo, a.seed_hi, tick);
      int act;
      if (POLICY == GX_POLICY_RANDOM) {  // Discrete.Sample, Spaces.Discrete.cs:18
        if (k == 0 || (tick & 3) == 0) policy = policy_block(a.seed_lo, a.seed_hi, tick, env);
        act = word_below(policy_word(policy, tick), G::kActions);
      } else act = static_cast<const int32_t*>(a.actions)[(uint64_t)k * a.action_stride + i];
      G::step(s, act, key, env, reward, done);
      trunc = a.time_limit > 0 && G::steps(s) >= a.time_limit;  // Wrappers.TimeLimit.cs:26-31
      done = done || trunc;
      if (record) {  // EnvWrapper.cs:46-55
        ep_ret += (int32_t)reward;
        ep_len += 1;
        if (done) {
          acc.add((double)ep_ret, a.ep_length ? ep_len : G::steps(s));
          ep_ret = 0;
          ep_len = 0;
        }
      }
      if (autoreset && done) G::reset(s, key, env);
      if (ROWS && a.out_reward) {  // every step's results of a rollout, row k
        const uint64_t at = (uint64_t)k * a.stride + i;
        static_cast<float*>(a.out_reward)[at] = reward;
        a.out_done[at] = done;
        if (a.out_truncated) a.out_truncated[at] = trunc;
      }
    }
    G::store(a, i, s);
    if (a.render) G::pack(s, a.render + i * G::kRecordDwords);
    if (record) {
      static_cast<int32_t*>(a.ep_return)[i] = ep_ret;
      if (a.ep_length) a.ep_length[i] = ep_len;
    }
    if (!ROWS || !a.skip_last) {
      static_cast<float*>(a.reward)[i] = reward;
      a.done[i] = done;
      if (a.truncated) a.truncated[i] = trunc;
    }
  }
  if (record) block_commit_stats(acc, a.stat_slots);
}

template <class G>
__global__ void __launch_bounds__(kBlock) frame_reset_kernel(StepArgs a) {
  const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick);
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock) {
    typename G::State s;
    G::prime(a, i, s);
    G::reset(s, key, a.env_index_base + i);
    G::store(a, i, s);
    G::pack(s, a.render + i * G::kRecordDwords);
  }
}

template <class G>
__global__ void __launch_bounds__(kBlock) frame_pack_kernel(StepArgs a) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock) {
    typename G::State s;
    G::load(a, i, s);
    G::pack(s, a.render + i * G::kRecordDwords);
  }
}

template <class G>
__global__ void __launch_bounds__(kBlock) frame_sample_kernel(StepArgs a, int32_t* out) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock)
    out[i] = word_below(policy_word(policy_block(a.seed_lo, a.seed_hi, a.tick, a.env_index_base + i), a.tick), G::kActions);
}

template <class G>
hipError_t launch_frame_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) {
  if (a.out_reward || a.skip_last) {
    if (policy == GX_POLICY_RANDOM) hipLaunchKernelGGL((frame_step_kernel<G, GX_POLICY_RANDOM, true>), grid, dim3(kBlock), 0, stream, a);
    else hipLaunchKernelGGL((frame_step_kernel<G, GX_POLICY_DEVICE_ACTIONS, true>), grid, dim3(kBlock), 0, stream, a);
  } else if (policy == GX_POLICY_RANDOM) hipLaunchKernelGGL((frame_step_kernel<G, GX_POLICY_RANDOM>), grid, dim3(kBlock), 0, stream, a);
  else hipLaunchKernelGGL((frame_step_kernel<G, GX_POLICY_DEVICE_ACTIONS>), grid, dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}
template <class G>
hipError_t launch_frame_reset(const StepArgs& a, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(frame_reset_kernel<G>, grid, dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}
template <class G>
hipError_t launch_frame_pack(const StepArgs& a, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(frame_pack_kernel<G>, grid, dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}
template <class G>
hipError_t launch_frame_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(frame_sample_kernel<G>, grid, dim3(kBlock), 0, stream, a, static_cast<int32_t*>(out));
  return hipGetLastError();
}

}  // namespace gx
