// Table-driven environments (FrozenLake, CliffWalking, Taxi) and DummyEnv.
//
// The host compiles an env into a packed transition table
//   entry[(state * (A + 1) + column) * O + outcome]
//     bits  0..15  next state
//     bits 16..23  index into the reward table that follows the entries (a.table_lut values: eight for the
//                  built-in envs, up to 256 for a caller-supplied table)
//     bit  24      terminal (goal / hole / cliff / delivered)
// with A the size of the Discrete action space, column A the "any other
// action" case (Envs.FrozenLake.cs:56-62: a switch without default leaves the
// state alone) and O = 3 outcomes for the slippery extension, else 1.  A
// workgroup stages the table in LDS once and then grid-strides over the
// batch, four envs per lane, so the per-step traffic is the SoA rows only.
#include "gx_device.h"
#include "gx_launch.h"

namespace gx {

__device__ __forceinline__ void load4i(const int32_t* base, uint64_t quad, int32_t (&out)[4]) {
  int4 v = reinterpret_cast<const int4*>(base)[quad];
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
__device__ __forceinline__ void store4i(int32_t* base, uint64_t quad, const int32_t (&in)[4]) {
  reinterpret_cast<int4*>(base)[quad] = make_int4(in[0], in[1], in[2], in[3]);
}
__device__ __forceinline__ void store4f(float* base, uint64_t quad, const float (&in)[4]) {
  reinterpret_cast<float4*>(base)[quad] = make_float4(in[0], in[1], in[2], in[3]);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4f_nt(float* base, uint64_t quad, const float (&in)[4]) {
  f32x4 v = {in[0], in[1], in[2], in[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base) + quad);
}
__device__ __forceinline__ uint32_t pack4(const bool (&b)[4]) {
  return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}

// step k's results of a rollout into row k of the caller's per-step rows, when they are wanted
__device__ __forceinline__ void store_step_rows(const StepArgs& a, int k, uint64_t q, const float (&reward)[4], const bool (&done)[4],
                                                const bool (&trunc)[4]) {
  if (!a.out_reward) return;
  const uint64_t row = (uint64_t)k * a.stride;
  store4f_nt(static_cast<float*>(a.out_reward) + row, q, reward);
  __builtin_nontemporal_store(pack4(done), reinterpret_cast<uint32_t*>(a.out_done + row) + q);
  if (a.out_truncated) __builtin_nontemporal_store(pack4(trunc), reinterpret_cast<uint32_t*>(a.out_truncated + row) + q);
}

// Reset state: FrozenLake 0 (Envs.FrozenLake.cs:47), CliffWalking 36
// (Envs.CliffWalking.cs:31), Taxi `Next(500)` (Envs.Taxi.cs:20: -1, a uniform draw over the table's states).
__device__ __forceinline__ int32_t tabular_reset_state(int32_t start_state, int32_t n_states, const RngKey& key, uint64_t env) {
  if (start_state >= 0) return start_state;
  Draws d = philox_block(key, env, 0, 0);
  if (start_state == -1) return d.below(0, n_states);
  // extension (upstream Taxi-v3): passenger neither in the taxi nor at the destination
  int k = d.below(0, 300);
  int cell = k / 12, rem = k % 12, pass = rem / 3, d3 = rem % 3;
  int dest = d3 >= pass ? d3 + 1 : d3;
  return (cell * 5 + pass) * 4 + dest;
}

// SLIP: the table has three outcomes per (state, action) (the slippery extension).
// MULTI: the form behind rollouts and gx_vec_rollout_out (per-step rows, or no rows at all); the
// single-step form does not carry their pointers.
// ALIGNED: the batch's first global env index is a multiple of four, so a lane's four envs share one
// generator block; otherwise every env fetches its word from its own block (four calls in flight cost
// twenty registers, hence a variant of its own).
template <int POLICY, bool LDS_TABLE, bool SLIP, bool MULTI, bool ALIGNED>
__global__ void __launch_bounds__(kTabularBlock) tabular_step_kernel(StepArgs a) {
  extern __shared__ __align__(16) uint32_t lds_table[];
  const int A = a.table_actions, ncols = A + 1, O = SLIP ? 3 : 1;
  const int entries = a.table_states * ncols * O;
  const uint32_t* tab = a.table;
  if (LDS_TABLE) {
    if (POLICY == GX_POLICY_DEVICE_ACTIONS) {
      // sixteen bytes per lane and trip (the device copy of the table is padded to a multiple of four words): at
      // 64 Mi envs the Taxi stub steps in 0.283 instead of 0.303 ms, full Taxi-v3 in 0.402 instead of 0.413, the
      // 8x8 FrozenLake in 0.278 instead of 0.285
      const int chunks = (entries + a.table_lut + 3) >> 2;
      for (int i = threadIdx.x; i < chunks; i += blockDim.x)
        reinterpret_cast<uint4*>(lds_table)[i] = reinterpret_cast<const uint4*>(a.table)[i];
    } else {
      // with the actions sampled in the kernel the word-by-word copy is the faster one by 1-1.5 % (same runs)
      for (int i = threadIdx.x; i < entries + a.table_lut; i += blockDim.x) lds_table[i] = a.table[i];
    }
    __syncthreads();
    tab = lds_table;
  }
  const float* lut = reinterpret_cast<const float*>(tab + entries);

  const bool autoreset = (a.flags & GX_AUTORESET) != 0;
  const bool record = (a.flags & GX_RECORD_STATS) != 0;
  // every non-terminal transition of the table pays the same reward and auto-reset is on: the
  // open-episode return is that reward times the step counter, its row is not touched
  const bool derive_returns = a.derive_returns != 0;
  int32_t* const state_row = a.state_int;
  int32_t* const steps_row = a.state_int + a.stride;
  ThreadStats<int32_t> acc;  // every reward of a table is an integer: branch-free integer accumulation
  uint32_t bad_action = 0;   // device actions only: one outside the space (an error for caller-supplied tables)

  const uint64_t nquads = (a.num_envs + 3) >> 2;
  for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < nquads; it += (uint64_t)gridDim.x * blockDim.x) {
    // alternate launches sweep in opposite directions (memory-side cache reuse across steps)
    const uint64_t q = a.reverse ? nquads - 1 - it : it;
    int32_t st[4], steps[4], ep_ret[4] = {}, ep_len[4] = {};
    load4i(state_row, q, st);
    load4i(steps_row, q, steps);
    if (record) {
      if (derive_returns) {
#pragma unroll
        for (int j = 0; j < 4; j++) ep_ret[j] = a.ret_per_step * steps[j];
      } else {
        load4i(static_cast<const int32_t*>(a.ep_return), q, ep_ret);
      }
      if (a.ep_length) load4i(a.ep_length, q, ep_len);
    }
    float reward[4];
    bool done[4], trunc[4];
    // the env's word of the tick gives the random policy its action and a slippery env its outcome
    constexpr bool need_word = POLICY == GX_POLICY_RANDOM || SLIP;
    const uint64_t env0 = a.env_index_base + q * 4;
    for (int k = 0; k < a.k_steps; k++) {
      const uint64_t tick = a.tick + (uint64_t)k;
      const RngKey key = make_key(a.seed_lo, a.seed_hi, tick);
      int32_t act[4];
      if (POLICY == GX_POLICY_DEVICE_ACTIONS)
        load4i(static_cast<const int32_t*>(a.actions) + (uint64_t)k * a.action_stride, q, act);
      uint32_t word[4] = {0, 0, 0, 0};
      if (need_word) {
        if (ALIGNED) {
          const Draws d = tabular_block(a.seed_lo, a.seed_hi, tick, env0 >> 2);
#pragma unroll
          for (int j = 0; j < 4; j++) word[j] = d.w[j];
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) word[j] = tabular_word(a.seed_lo, a.seed_hi, tick, env0 + j);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint64_t env = env0 + j;
        const uint32_t w = word[j];
        if (POLICY == GX_POLICY_RANDOM) act[j] = word_below(w, A);
        const int col = (uint32_t)act[j] < (uint32_t)A ? act[j] : A;
        if (POLICY == GX_POLICY_DEVICE_ACTIONS) bad_action |= (uint32_t)(col == A && q * 4 + j < a.num_envs);
        const int o = (SLIP && col < A) ? word_outcome(w, A, O) : 0;
        // a state outside the table (only reachable through set_state) is clamped
        const int s_in = (uint32_t)st[j] < (uint32_t)a.table_states ? st[j] : 0;
        const uint32_t e = tab[(s_in * ncols + col) * O + o];
        st[j] = (int32_t)(e & 0xFFFFu);
        reward[j] = lut[(e >> 16) & 0xFFu];
        steps[j] += 1;
        bool d = ((e >> 24) & 1u) != 0 || steps[j] >= a.step_limit;
        bool t = a.time_limit > 0 && steps[j] >= a.time_limit;  // Wrappers.TimeLimit.cs:26-31
        d = d || t;
        done[j] = d;
        trunc[j] = t;
        if (record) {  // EnvWrapper.cs:46-55
          ep_ret[j] += (int32_t)reward[j];
          ep_len[j] += 1;
          acc.add(d && q * 4 + j < a.num_envs, ep_ret[j], a.ep_length ? ep_len[j] : steps[j]);
          ep_ret[j] = d ? 0 : ep_ret[j];
          ep_len[j] = d ? 0 : ep_len[j];
        }
        if (autoreset && d) {
          st[j] = tabular_reset_state(a.start_state, a.table_states, key, env);
          steps[j] = 0;
        }
      }
      if (MULTI) store_step_rows(a, k, q, reward, done, trunc);
    }
    store4i(state_row, q, st);
    store4i(steps_row, q, steps);
    if (record) {
      if (!derive_returns) store4i(static_cast<int32_t*>(a.ep_return), q, ep_ret);
      if (a.ep_length) store4i(a.ep_length, q, ep_len);
    }
    // write-only rows stream past the caches
    if (!MULTI || !a.skip_last) {
      store4f_nt(static_cast<float*>(a.reward), q, reward);
      __builtin_nontemporal_store(pack4(done), reinterpret_cast<uint32_t*>(a.done) + q);
      if (a.truncated) __builtin_nontemporal_store(pack4(trunc), reinterpret_cast<uint32_t*>(a.truncated) + q);
    }
  }
  if (record) block_commit_stats(acc.widen(), a.stat_slots);
  if (POLICY == GX_POLICY_DEVICE_ACTIONS && a.strict_actions && bad_action) atomicOr(a.error_flag, 1u);
}

// DummyEnv.cs:23-30: state += action; reward = state; done = state >= 10.
template <int POLICY>
__global__ void __launch_bounds__(kTabularBlock) dummy_step_kernel(StepArgs a) {  // a test fixture: one form serves all
  const bool autoreset = (a.flags & GX_AUTORESET) != 0;
  const bool record = (a.flags & GX_RECORD_STATS) != 0;
  int32_t* const state_row = a.state_int;
  int32_t* const steps_row = a.state_int + a.stride;
  StatAcc acc;
  const uint64_t nquads = (a.num_envs + 3) >> 2;
  for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < nquads; it += (uint64_t)gridDim.x * blockDim.x) {
    // alternate launches sweep in opposite directions (memory-side cache reuse across steps)
    const uint64_t q = a.reverse ? nquads - 1 - it : it;
    int32_t st[4], steps[4], ep_ret[4] = {}, ep_len[4] = {};
    load4i(state_row, q, st);
    load4i(steps_row, q, steps);
    if (record) {
      load4i(static_cast<const int32_t*>(a.ep_return), q, ep_ret);
      if (a.ep_length) load4i(a.ep_length, q, ep_len);
    }
    float reward[4];
    bool done[4], trunc[4];
    for (int k = 0; k < a.k_steps; k++) {
      const uint64_t tick = a.tick + (uint64_t)k;
      int32_t act[4];
      if (POLICY == GX_POLICY_DEVICE_ACTIONS)
        load4i(static_cast<const int32_t*>(a.actions) + (uint64_t)k * a.action_stride, q, act);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (POLICY == GX_POLICY_RANDOM)  // Discrete(3), DummyEnv.cs:14
          act[j] = word_below(tabular_word(a.seed_lo, a.seed_hi, tick, a.env_index_base + q * 4 + j), 3);
        st[j] += act[j];
        steps[j] += 1;
        reward[j] = (float)st[j];
        bool d = st[j] >= 10;
        bool t = a.time_limit > 0 && steps[j] >= a.time_limit;
        d = d || t;
        done[j] = d;
        trunc[j] = t;
        if (record) {
          ep_ret[j] += st[j];
          ep_len[j] += 1;
          if (d) {
            if (q * 4 + j < a.num_envs) acc.add((double)ep_ret[j], a.ep_length ? ep_len[j] : steps[j]);
            ep_ret[j] = 0;
            ep_len[j] = 0;
          }
        }
        if (autoreset && d) {
          st[j] = 0;  // DummyEnv.cs:17-21
          steps[j] = 0;
        }
      }
      store_step_rows(a, k, q, reward, done, trunc);
    }
    store4i(state_row, q, st);
    store4i(steps_row, q, steps);
    if (record) {
      store4i(static_cast<int32_t*>(a.ep_return), q, ep_ret);
      if (a.ep_length) store4i(a.ep_length, q, ep_len);
    }
    // write-only rows stream past the caches
    if (!a.skip_last) {
      store4f_nt(static_cast<float*>(a.reward), q, reward);
      __builtin_nontemporal_store(pack4(done), reinterpret_cast<uint32_t*>(a.done) + q);
      if (a.truncated) __builtin_nontemporal_store(pack4(trunc), reinterpret_cast<uint32_t*>(a.truncated) + q);
    }
  }
  if (record) block_commit_stats(acc, a.stat_slots);
}

__global__ void __launch_bounds__(kTabularBlock) tabular_reset_kernel(StepArgs a) {
  const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick);
  const uint64_t nquads = (a.num_envs + 3) >> 2;
  for (uint64_t q = (uint64_t)blockIdx.x * kTabularBlock + threadIdx.x; q < nquads; q += (uint64_t)gridDim.x * kTabularBlock) {
    int32_t st[4], zero[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) st[j] = tabular_reset_state(a.start_state, a.table_states, key, a.env_index_base + q * 4 + j);
    store4i(a.state_int, q, st);
    store4i(a.state_int + a.stride, q, zero);
  }
}

__global__ void __launch_bounds__(kTabularBlock) tabular_sample_kernel(StepArgs a, int32_t nact, int32_t* out) {
  for (uint64_t i = (uint64_t)blockIdx.x * kTabularBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kTabularBlock)
    out[i] = word_below(tabular_word(a.seed_lo, a.seed_hi, a.tick, a.env_index_base + i), nact);
}

hipError_t launch_tabular_step(int env_id, const StepArgs& a, int policy, dim3 grid, int block, size_t lds_bytes, hipStream_t stream) {
  if (env_id == GX_ENV_DUMMY) {
    if (policy == GX_POLICY_RANDOM)
      hipLaunchKernelGGL(dummy_step_kernel<GX_POLICY_RANDOM>, grid, dim3(block), 0, stream, a);
    else
      hipLaunchKernelGGL(dummy_step_kernel<GX_POLICY_DEVICE_ACTIONS>, grid, dim3(block), 0, stream, a);
    return hipGetLastError();
  }
  const bool lds = lds_bytes > 0, slip = a.table_outcomes > 1;
  const bool multi = a.k_steps > 1 || a.out_reward || a.skip_last;
  if (a.table_outcomes != 1 && a.table_outcomes != 3) return hipErrorInvalidValue;
  const bool aligned = (a.env_index_base & 3) == 0;
#define GX_TAB_LAUNCH(P, L_, S, M)                                                                                     \
  do {                                                                                                                 \
    if (aligned) hipLaunchKernelGGL((tabular_step_kernel<P, L_, S, M, true>), grid, dim3(block), lds_bytes, stream, a); \
    else hipLaunchKernelGGL((tabular_step_kernel<P, L_, S, M, false>), grid, dim3(block), lds_bytes, stream, a);        \
  } while (0)
#define GX_TAB_PICK_M(P, L_, S) do { if (multi) GX_TAB_LAUNCH(P, L_, S, true); else GX_TAB_LAUNCH(P, L_, S, false); } while (0)
#define GX_TAB_PICK_S(P, L_) do { if (slip) GX_TAB_PICK_M(P, L_, true); else GX_TAB_PICK_M(P, L_, false); } while (0)
#define GX_TAB_PICK_L(P) do { if (lds) GX_TAB_PICK_S(P, true); else GX_TAB_PICK_S(P, false); } while (0)
  if (policy == GX_POLICY_RANDOM) GX_TAB_PICK_L(GX_POLICY_RANDOM);
  else GX_TAB_PICK_L(GX_POLICY_DEVICE_ACTIONS);
#undef GX_TAB_PICK_L
#undef GX_TAB_PICK_S
#undef GX_TAB_PICK_M
#undef GX_TAB_LAUNCH
  return hipGetLastError();
}

hipError_t launch_tabular_reset(int env_id, const StepArgs& a, dim3 grid, hipStream_t stream) {
  (void)env_id;
  hipLaunchKernelGGL(tabular_reset_kernel, grid, dim3(kTabularBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_tabular_sample(int env_id, const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  const int nact = env_id == GX_ENV_DUMMY ? 3 : a.table_actions;
  hipLaunchKernelGGL(tabular_sample_kernel, grid, dim3(kTabularBlock), 0, stream, a, nact, static_cast<int32_t*>(out));
  return hipGetLastError();
}

}  // namespace gx
