// Generic step / rollout / reset kernels for continuous-control envs.
//
// An env type E supplies: Real (float or double), NS (real components), W
// (envs per lane: 16 bytes of each SoA row per lane), Act, Rew, EpRet,
// reset(), step<MODE, BOUNDED>(), sample() and, when the observation is not
// the state itself, observe().  One env per lane slot, grid-stride over the
// batch, single step and register-resident K-step rollout in one kernel, with
// auto-reset, TimeLimit and RecordEpisodeStatistics fused in.
//
// Two compile-time switches pick the kernel variant:
//   FUSED   : auto-reset and RecordEpisodeStatistics on, no TimeLimit -- the
//             throughput configuration; the wrapper flags become constants and
//             the rows that configuration never touches disappear.
//   BOUNDED : angles are inside the ranges the env's invariants guarantee (see
//             gx_trig.h); the host clears it for the one step that follows a
//             set_state and whenever auto-reset is off for an env whose angle
//             can then grow without bound.
#pragma once

#include <type_traits>

#include "gx_device.h"
#include "gx_launch.h"
#include "gx_trig.h"

namespace gx {

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

// Non-temporal forms for rows the kernels only write.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <class T, int W>
__device__ __forceinline__ void store_row_nt(T* base, uint64_t group, const T (&in)[W]) {
  constexpr int kDwords = (int)(sizeof(T) * W / 4);
  uint32_t d[kDwords];
  __builtin_memcpy(d, in, sizeof(T) * W);
  uint32_t* dst = reinterpret_cast<uint32_t*>(base) + group * kDwords;
  if constexpr (kDwords == 2) {
    u32x2 v = {d[0], d[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(dst));
  } else {
#pragma unroll
    for (int i = 0; i < kDwords; i += 4) {
      u32x4 v = {d[i], d[i + 1], d[i + 2], d[i + 3]};
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + i));
    }
  }
}
template <int W>
__device__ __forceinline__ void store_flags_nt(uint8_t* base, uint64_t group, const bool (&b)[W]) {
  if constexpr (W == 4) {
    const uint32_t v = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    __builtin_nontemporal_store(v, reinterpret_cast<uint32_t*>(base) + group);
  } else {
    const uint16_t v = (uint16_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8));
    __builtin_nontemporal_store(v, reinterpret_cast<uint16_t*>(base) + group);
  }
}

// Integer-valued returns in the kernels that sample their own actions (the fused rollouts and the
// in-kernel-policy step): every lane keeps its record in LDS (field-major, so a wave's access to
// one field is conflict-free) and adds to it with return-less LDS atomics when one of its envs
// finishes an episode.  No accumulator registers are carried through the kernel; at its end a lane
// reads its record back and the wave reduces as usual.  Measured against the register form:
// CartPole K = 64 rollout 3.54 -> 3.41 ms, in-kernel-policy step 0.1405 -> 0.1343 ms; the step with
// actions read from memory does not gain and keeps the registers.
struct LaneStatsLds {
  unsigned long long sum_length[kClassicBlock], sum_return[kClassicBlock], sum_return_sq[kClassicBlock];  // two's complement
  uint32_t episodes[kClassicBlock];
  int32_t min_return[kClassicBlock], max_return[kClassicBlock];
};
__device__ __forceinline__ void lane_stats_init(LaneStatsLds& w) {
  const int l = threadIdx.x;
  w.sum_length[l] = w.sum_return[l] = w.sum_return_sq[l] = 0ull;
  w.episodes[l] = 0u;
  w.min_return[l] = 0x7FFFFFFF;
  w.max_return[l] = (int32_t)0x80000000;
}
__device__ __forceinline__ void lane_stats_add(LaneStatsLds& w, int32_t ret, int32_t len) {
  const int l = threadIdx.x;
  atomicAdd(&w.episodes[l], 1u);
  atomicAdd(&w.sum_length[l], (unsigned long long)(long long)len);
  atomicAdd(&w.sum_return[l], (unsigned long long)(long long)ret);
  atomicAdd(&w.sum_return_sq[l], (unsigned long long)((long long)ret * ret));
  atomicMin(&w.min_return[l], ret);
  atomicMax(&w.max_return[l], ret);
}
__device__ __forceinline__ StatAcc lane_stats_read(const LaneStatsLds& w) {
  const int l = threadIdx.x;
  StatAcc a;
  a.episodes = w.episodes[l];
  a.sum_length = (long long)w.sum_length[l];
  a.sum_return = (double)(long long)w.sum_return[l];
  a.sum_return_sq = (double)(long long)w.sum_return_sq[l];
  if (a.episodes) { a.min_return = (double)w.min_return[l]; a.max_return = (double)w.max_return[l]; }
  return a;
}

// ---------------------------------------------------------------- kernels ---
// E::kStepReturn of an env whose rewards differ from step to step
constexpr int kNoStepReturn = 0x7FFFFFFF;

// An env may keep derived quantities next to its stored state while it is in registers: E::NR
// (>= E::NS) values per env, of which the first NS are the rows in memory and the rest are
// recomputed by E::derive() after a load (E::reset() fills them itself).
template <class E, class = void>
struct RegState {
  static constexpr int value = E::NS;
};
template <class E>
struct RegState<E, std::void_t<decltype(E::NR)>> {
  static constexpr int value = E::NR;
};

// Leading rows of an env that its step never reads: it rewrites them from the others (E::kWriteOnlyRows).
// They are not loaded and go out past the caches like the result rows.
template <class E, class = void>
struct WriteOnlyRows {
  static constexpr int value = 0;
};
template <class E>
struct WriteOnlyRows<E, std::void_t<decltype(E::kWriteOnlyRows)>> {
  static constexpr int value = E::kWriteOnlyRows;
};

template <class E, class = void>
struct RecomputeAddresses {
  static constexpr bool value = false;
};
template <class E>
struct RecomputeAddresses<E, std::void_t<decltype(E::kRecomputeAddresses)>> {
  static constexpr bool value = E::kRecomputeAddresses;
};

// E::kStateInLds: between its load and its store the real state of a lane's W envs lives in LDS
// ([row][env of the lane][lane], one dword per lane: conflict-free), and only the env being stepped is in
// registers -- for an env whose arithmetic needs most of the register file for itself.  A wave touches only its
// own lanes' words, so no barrier is involved.
template <class E, class = void>
struct StateInLds {
  static constexpr bool value = false;
};
template <class E>
struct StateInLds<E, std::void_t<decltype(E::kStateInLds)>> {
  static constexpr bool value = E::kStateInLds;
};

// E::kSinCosTable: the env's EXACT arithmetic evaluates sin and cos through the LDS table of gx_trig.h, which every
// kernel that runs its step, reset or derive fills first.
template <class E, class = void>
struct UsesSinCosTable {
  static constexpr bool value = false;
};
template <class E>
struct UsesSinCosTable<E, std::void_t<decltype(E::kSinCosTable)>> {
  static constexpr bool value = E::kSinCosTable;
};

// E::kPrefetchNextTrip: the single-step kernel with actions read from memory requests the rows of the lane's next
// trip before the arithmetic of the current one (24 registers more for a four-row env).  The requests then precede
// the current trip's stores in the wave's memory queue, so the next trip starts as soon as its own rows are there
// instead of after the stores before them have been acknowledged.  Measured at 16 Mi envs, kernel ms without / with:
// RK4 Acrobot 0.390 / 0.344, MountainCarContinuous 0.1474 / 0.1426 (binary64 0.1937 / 0.1908), Pendulum 0.1722 /
// 0.1702, Acrobot EXACT 0.1997 / 0.1976; CartPole and MountainCar lose 1 % and every env's in-kernel-policy form
// loses 0-4 % (no action row to wait for: their waves are not short of requests in flight), so those keep the plain form.
template <class E, class = void>
struct PrefetchNextTrip {
  static constexpr bool value = false;
};
template <class E>
struct PrefetchNextTrip<E, std::void_t<decltype(E::kPrefetchNextTrip)>> {
  static constexpr bool value = E::kPrefetchNextTrip;
};

// E::Rollout: the env as the fused rollouts with the in-kernel policy (the CACHED form) step it -- the same rows and
// rules with arithmetic routines chosen for a kernel that is bound by instruction issue, not by memory.
template <class E, bool CACHED, class = void>
struct StepEnv {
  using type = E;
};
template <class E>
struct StepEnv<E, true, std::void_t<typename E::Rollout>> {
  using type = typename E::Rollout;
};

// minimum resident waves per SIMD the fused kernels are compiled for (caps their registers): E::kMinWaves, else 1
#ifndef GX_FUSED_MIN_WAVES
#define GX_FUSED_MIN_WAVES 1
#endif
template <class E, class = void>
struct MinWaves {
  static constexpr int value = GX_FUSED_MIN_WAVES;
};
template <class E>
struct MinWaves<E, std::void_t<decltype(E::kMinWaves)>> {
  static constexpr int value = E::kMinWaves;
};

// E::kMinWavesRollout: the same for the CACHED form (fused rollouts with the in-kernel policy); with it an env that
// parks its states in LDS keeps them in registers in that form (E::kStateInLds then holds for the other forms only).
template <class E, class = void>
struct MinWavesRollout {
  static constexpr int value = MinWaves<E>::value;
  static constexpr bool unpark = false;
};
template <class E>
struct MinWavesRollout<E, std::void_t<decltype(E::kMinWavesRollout)>> {
  static constexpr int value = E::kMinWavesRollout;
  static constexpr bool unpark = true;
};

// CACHED (fused K-step rollouts with the random policy): the policy block of four ticks is kept
// in registers, so the generator runs once every four steps; the single-step kernels draw
// the block and use one word of it (16 registers per lane fewer, one more resident wave).
// ROWS (gx_vec_rollout_out): every step's reward / done / truncated go to the caller's per-step rows,
// or nothing at all is handed back (skip_last).  A separate variant because the extra pointers cost
// the plain kernels registers (the scalar file is full: they spill into vector registers) and with
// them a resident wave.
// SINGLE: exactly one step per launch (a.k_steps is 1), known at compile time; the form that requests a trip ahead.
template <class E, int MODE, int POLICY, bool FUSED, bool BOUNDED, bool CACHED = false, bool ROWS = false, bool SINGLE = false>
__global__ void __launch_bounds__(kClassicBlock, (FUSED && BOUNDED) ? (CACHED ? MinWavesRollout<E>::value : MinWaves<E>::value) : 1) classic_step_kernel(StepArgs a) {
  using ES = typename StepEnv<E, CACHED>::type;  // whose step() and reset() this kernel runs
  using Real = typename E::Real;
  using Act = typename E::Act;
  using Rew = typename E::Rew;
  using EpRet = typename E::EpRet;
  constexpr int NS = E::NS;
  constexpr int NR = RegState<E>::value;
  constexpr int W = E::W;
  constexpr int WO = WriteOnlyRows<E>::value;
  static_assert(ES::NS == NS && RegState<ES>::value == NR, "the rollout form steps the same rows");
  constexpr bool PARK = StateInLds<E>::value && !(CACHED && MinWavesRollout<E>::unpark);
  static_assert(!PARK || (NR == NS && !E::kDerivedObs), "the LDS form holds the stored rows only");
  __shared__ Real parked[PARK ? NS : 1][W][kClassicBlock];
  const int lane = threadIdx.x;
  if constexpr (UsesSinCosTable<ES>::value && MODE == GX_MODE_EXACT) sincos_lds_fill();

  const bool autoreset = FUSED ? true : (a.flags & GX_AUTORESET) != 0;
  const bool record = FUSED ? true : (a.flags & GX_RECORD_STATS) != 0;
  const int32_t time_limit = FUSED ? 0 : a.time_limit;
  int32_t* const ep_length_row = FUSED ? nullptr : a.ep_length;
  uint8_t* const truncated_row = FUSED ? nullptr : a.truncated;
  Real* const sreal = static_cast<Real*>(a.state_real);
  int32_t* const steps_row = a.state_int;  // these envs have one int row: the step counter
  // With auto-reset the return of the open episode of an env that pays a constant integer reward
  // on every non-final step is that constant times the step counter: the row is neither read
  // nor written (8 bytes per env-step less), the statistics come out the same.
  const bool derive_returns = FUSED && E::kStepReturn != kNoStepReturn && a.derive_returns != 0;
  // (an env that parks its states in LDS keeps the records in registers: LDS is what limits its resident waves --
  // Acrobot EXACT with in-kernel actions at 16 Mi envs: 0.218 -> 0.206 ms)
  constexpr bool kLdsStats = std::is_same<EpRet, int32_t>::value && POLICY == GX_POLICY_RANDOM && !PARK;
  __shared__ LaneStatsLds lane_stats;
  ThreadStats<EpRet> acc;
  if (kLdsStats && record) lane_stats_init(lane_stats);

  const uint64_t ngroups = (a.num_envs + (W - 1)) / W;
  const uint64_t it_stride = (uint64_t)gridDim.x * kClassicBlock;
  // consecutive launches sweep the batch in opposite directions: the rows a sweep touched
  // last are still in the 256 MiB memory-side cache when the next sweep starts there
  auto group_of = [&](uint64_t it) { return a.reverse ? ngroups - 1 - it : it; };
  // PF: what a trip reads (state rows, step counter, the first step's actions), requested one trip ahead
  constexpr bool PF = PrefetchNextTrip<E>::value && POLICY == GX_POLICY_DEVICE_ACTIONS && FUSED && BOUNDED && !CACHED && !ROWS && SINGLE;
  static_assert(!SINGLE || PF, "the single-step form exists for the envs that request a trip ahead");
  Real next_rows[PF ? NS : 1][W];
  int32_t next_steps[W];
  Act next_act[W];
  auto request = [&](uint64_t g) {
#pragma unroll
    for (int c = WO; c < (PF ? NS : 0); c++) load_row<Real, W>(sreal + (uint64_t)c * a.stride, g, next_rows[PF ? c : 0]);
    load_row<int32_t, W>(steps_row, g, next_steps);
    if (POLICY == GX_POLICY_DEVICE_ACTIONS) load_row<Act, W>(static_cast<const Act*>(a.actions), g, next_act);
  };
  const uint64_t it_first = (uint64_t)blockIdx.x * kClassicBlock + threadIdx.x;
  if constexpr (PF) {
    if (it_first < ngroups) request(group_of(it_first));
    // The first trip's rows are awaited here, outside the sweep: a wait at the loop's head would be counted for
    // this entry as well as for the back edge, where the stores of the trip before are still in flight behind the
    // rows, and would wait for those too.
#pragma unroll
    for (int j = 0; j < W; j++) {
#pragma unroll
      for (int c = WO; c < NS; c++) asm volatile("" : "+v"(next_rows[c][j]));
      asm volatile("" : "+v"(next_steps[j]));
      if (POLICY == GX_POLICY_DEVICE_ACTIONS) asm volatile("" : "+v"(next_act[j]));
    }
  }
  for (uint64_t it = it_first; it < ngroups; it += it_stride) {
    const uint64_t q = group_of(it);
    Real s[W][NR];
    int32_t steps[W];
    Act first_act[W];
    EpRet ep_ret[W] = {};
    int32_t ep_len[W] = {};
    if constexpr (PF) {
#pragma unroll
      for (int c = WO; c < NS; c++) {
#pragma unroll
        for (int j = 0; j < W; j++) {
          if constexpr (PARK) parked[c][j][lane] = next_rows[c][j];
          else s[j][c] = next_rows[c][j];
        }
      }
#pragma unroll
      for (int j = 0; j < W; j++) {
        steps[j] = next_steps[j];
        first_act[j] = next_act[j];
      }
      // what the trip reads besides: before the requests, so that waiting for it does not wait for them
      if (!derive_returns) load_row<EpRet, W>(static_cast<const EpRet*>(a.ep_return), q, ep_ret);
      // A lane without a further trip asks for group 0 and drops what comes: with the requests under a branch the
      // values of the two paths are merged by register copies behind them, which wait for the rows on the spot.
      request(it + it_stride < ngroups ? group_of(it + it_stride) : 0);
    } else {
#pragma unroll
      for (int c = 0; c < NS; c++) {
        Real v[W] = {};
        if (c >= WO) load_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
#pragma unroll
        for (int j = 0; j < W; j++) {
          if constexpr (PARK) {
            if (c >= WO) parked[c][j][lane] = v[j];
          } else {
            s[j][c] = v[j];
          }
        }
      }
      load_row<int32_t, W>(steps_row, q, steps);
    }
    if constexpr (NR > NS) {
#pragma unroll
      for (int j = 0; j < W; j++) E::derive(s[j]);
    }
    if (record) {
      if (derive_returns) {
#pragma unroll
        for (int j = 0; j < W; j++) ep_ret[j] = (EpRet)(E::kStepReturn * steps[j]);
      } else if (!PF) {
        load_row<EpRet, W>(static_cast<const EpRet*>(a.ep_return), q, ep_ret);
      }
      if (ep_length_row) load_row<int32_t, W>(ep_length_row, q, ep_len);
    }

    Rew reward[W];
    bool done[W], trunc[W];
    Draws policy[CACHED ? W : 1];
    const uint64_t env0 = a.env_index_base + q * W;
    const int k_steps = SINGLE ? 1 : a.k_steps;
    for (int k = 0; k < k_steps; k++) {
      const uint64_t tick = a.tick + (uint64_t)k;
      const RngKey key = make_key(a.seed_lo, a.seed_hi, tick);
      Act act[W];
      if (POLICY == GX_POLICY_DEVICE_ACTIONS) {
        if (PF && k == 0) {
#pragma unroll
          for (int j = 0; j < W; j++) act[j] = first_act[j];
        } else {
          load_row<Act, W>(static_cast<const Act*>(a.actions) + (uint64_t)k * a.action_stride, q, act);
        }
      }
      if (CACHED && POLICY == GX_POLICY_RANDOM && (k == 0 || (tick & 3) == 0)) {
#pragma unroll
        for (int j = 0; j < W; j++) policy[CACHED ? j : 0] = policy_block(a.seed_lo, a.seed_hi, tick, env0 + j);
      }
      uint32_t pending = 0;  // envs of this lane that finished in this step
      bool bad_action = false;
#pragma unroll
      for (int j = 0; j < W; j++) {
        if (POLICY == GX_POLICY_RANDOM) {
          if (!CACHED) policy[0] = policy_block(a.seed_lo, a.seed_hi, tick, env0 + j);
          act[j] = E::sample(policy_word(policy[CACHED ? j : 0], tick));
        }
        // an action the reference throws on: noted per lane, reported once after the lane's envs (a branch
        // per env would split their arithmetic into separate blocks)
        if (E::kCheckAction && POLICY == GX_POLICY_DEVICE_ACTIONS) bad_action |= !E::action_valid(act[j]) && q * W + j < a.num_envs;
        bool d;
        if constexpr (PARK) {
          Real cur[NR] = {};
#pragma unroll
          for (int c = WO; c < NS; c++) cur[c] = parked[c][j][lane];
          ES::template step<MODE, BOUNDED>(cur, steps[j], act[j], reward[j], d);
#pragma unroll
          for (int c = 0; c < NS; c++) parked[c][j][lane] = cur[c];
        } else {
          ES::template step<MODE, BOUNDED>(s[j], steps[j], act[j], reward[j], d);
        }
        // Wrappers.TimeLimit.cs:26-31; its counter always equals the env's own
        bool t = time_limit > 0 && steps[j] >= time_limit;
        d = d || t;
        done[j] = d;
        trunc[j] = t;
        if (record) {
          // EnvWrapper.cs:46-55
          ep_ret[j] += (EpRet)reward[j];
          ep_len[j] += 1;
          const bool finished = d && q * W + j < a.num_envs;
          if constexpr (kLdsStats) {
            if (finished) lane_stats_add(lane_stats, (int32_t)ep_ret[j], ep_length_row ? ep_len[j] : steps[j]);
          } else {
            acc.add(finished, ep_ret[j], ep_length_row ? ep_len[j] : steps[j]);
          }
          ep_ret[j] = d ? (EpRet)0 : ep_ret[j];
          ep_len[j] = d ? 0 : ep_len[j];
        }
        pending |= d ? (1u << j) : 0u;
      }
      if (E::kCheckAction && POLICY == GX_POLICY_DEVICE_ACTIONS) {
        if (bad_action) atomicOr(a.error_flag, 1u);
      }
      if (ROWS && a.out_reward) {  // every step's results of a rollout, row k (uniform branch)
        const uint64_t row = (uint64_t)k * a.stride;
        store_row_nt<Rew, W>(static_cast<Rew*>(a.out_reward) + row, q, reward);
        store_flags_nt<W>(a.out_done + row, q, done);
        if (a.out_truncated) store_flags_nt<W>(a.out_truncated + row, q, trunc);
      }
      // Auto-reset, compacted: each lane resets one finished env per round, so a
      // wave runs the RNG max-per-lane times instead of once per env slot.
      if (autoreset) {
        while (__any(pending != 0)) {
          if (pending) {
            const int jj = __ffs((int)pending) - 1;
            Real fresh[NR];
            ES::reset(fresh, key, env0 + (uint64_t)jj);
            if constexpr (PARK) {
#pragma unroll
              for (int c = 0; c < NS; c++) parked[c][jj][lane] = fresh[c];
            }
#pragma unroll
            for (int j = 0; j < W; j++) {
              if (jj == j) {
                if constexpr (!PARK) {
#pragma unroll
                  for (int c = 0; c < NR; c++) s[j][c] = fresh[c];
                }
                steps[j] = 0;
              }
            }
            pending &= pending - 1;
          }
        }
      }
    }

    // E::kRecomputeAddresses: the stores form their addresses from the group index again instead of keeping the
    // loads' address pairs (two registers per row) alive through the arithmetic
    uint64_t qs = q;
    if constexpr (RecomputeAddresses<E>::value) asm("" : "+v"(qs));
#pragma unroll
    for (int c = 0; c < NS; c++) {
      Real v[W];
#pragma unroll
      for (int j = 0; j < W; j++) v[j] = PARK ? parked[c][j][lane] : s[j][c];
      if (c >= WO) store_row<Real, W>(sreal + (uint64_t)c * a.stride, qs, v);
      else store_row_nt<Real, W>(sreal + (uint64_t)c * a.stride, qs, v);
    }
    store_row<int32_t, W>(steps_row, qs, steps);
    if (record) {
      if (!derive_returns) store_row<EpRet, W>(static_cast<EpRet*>(a.ep_return), qs, ep_ret);
      if (ep_length_row) store_row<int32_t, W>(ep_length_row, qs, ep_len);
    }
    // write-only rows: streamed past the caches, which stay with the rows the next step re-reads
    if (!ROWS || !a.skip_last) {
      store_row_nt<Rew, W>(static_cast<Rew*>(a.reward), qs, reward);
      store_flags_nt<W>(a.done, qs, done);
      if (truncated_row) store_flags_nt<W>(truncated_row, qs, trunc);
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
  if (record) {
    if constexpr (kLdsStats) block_commit_stats(lane_stats_read(lane_stats), a.stat_slots);
    else block_commit_stats(acc.widen(), a.stat_slots);
  }
}

// Reset() of every env, or (when only_observe) just the derived observation
// rows of the current state.
template <class E>
__global__ void __launch_bounds__(kClassicBlock) classic_reset_kernel(StepArgs a, int only_observe) {
  using Real = typename E::Real;
  constexpr int NS = E::NS;
  constexpr int NR = RegState<E>::value;
  constexpr int W = E::W;
  Real* const sreal = static_cast<Real*>(a.state_real);
  const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick);
  if constexpr (UsesSinCosTable<E>::value) sincos_lds_fill();
  const uint64_t ngroups = (a.num_envs + (W - 1)) / W;
  for (uint64_t q = (uint64_t)blockIdx.x * kClassicBlock + threadIdx.x; q < ngroups; q += (uint64_t)gridDim.x * kClassicBlock) {
    Real s[W][NR];
    if (only_observe) {
#pragma unroll
      for (int c = 0; c < NS; c++) {
        Real v[W];
        load_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
#pragma unroll
        for (int j = 0; j < W; j++) s[j][c] = v[j];
      }
      if constexpr (NR > NS) {
#pragma unroll
        for (int j = 0; j < W; j++) E::derive(s[j]);
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

// The rows behind the state image (E::kImageRows .. E::NS) from the rows of the image, after the image was
// written from outside.
template <class E>
__global__ void __launch_bounds__(kClassicBlock) classic_derive_kernel(StepArgs a) {
  using Real = typename E::Real;
  constexpr int NS = E::NS;
  constexpr int W = E::W;
  Real* const sreal = static_cast<Real*>(a.state_real);
  if constexpr (UsesSinCosTable<E>::value) sincos_lds_fill();
  const uint64_t ngroups = (a.num_envs + (W - 1)) / W;
  for (uint64_t q = (uint64_t)blockIdx.x * kClassicBlock + threadIdx.x; q < ngroups; q += (uint64_t)gridDim.x * kClassicBlock) {
    Real s[W][NS];
#pragma unroll
    for (int c = 0; c < NS; c++) {
      Real v[W] = {};
      if (c < E::kImageRows) load_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
#pragma unroll
      for (int j = 0; j < W; j++) s[j][c] = v[j];
    }
#pragma unroll
    for (int j = 0; j < W; j++) E::derive(s[j]);
#pragma unroll
    for (int c = E::kImageRows; c < NS; c++) {
      Real v[W];
#pragma unroll
      for (int j = 0; j < W; j++) v[j] = s[j][c];
      store_row<Real, W>(sreal + (uint64_t)c * a.stride, q, v);
    }
  }
}

// Actions of the random policy for one tick, for harnesses that replay them.
template <class E>
__global__ void __launch_bounds__(kClassicBlock) classic_sample_kernel(StepArgs a, void* out) {
  using Act = typename E::Act;
  for (uint64_t i = (uint64_t)blockIdx.x * kClassicBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kClassicBlock)
    static_cast<Act*>(out)[i] = E::sample(policy_word(policy_block(a.seed_lo, a.seed_hi, a.tick, a.env_index_base + i), a.tick));
}

// -------------------------------------------------------------- dispatch ---
// `fused` / `bounded` are requests: an env whose angles are bounded for every
// finite state (kAlwaysBounded) never instantiates the unbounded form.
template <class E, int MODE, int POLICY>
static void launch_step_variant(const StepArgs& a, bool fused, bool bounded, dim3 grid, hipStream_t stream) {
  if (E::kAlwaysBounded) bounded = true;
  constexpr bool kRandom = POLICY == GX_POLICY_RANDOM;
  if (a.out_reward || a.skip_last) {  // the per-step-rows variants
    if (fused && bounded) {
      hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, true, true, kRandom, true>), grid, dim3(kClassicBlock), 0, stream, a);
    } else if constexpr (E::kAlwaysBounded) {
      hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, false, true, false, true>), grid, dim3(kClassicBlock), 0, stream, a);
    } else {
      hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, false, false, false, true>), grid, dim3(kClassicBlock), 0, stream, a);
    }
  } else if (fused && bounded && kRandom && a.k_steps > 1) {
    hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, true, true, kRandom>), grid, dim3(kClassicBlock), 0, stream, a);
  } else if (fused && bounded) {
    if constexpr (PrefetchNextTrip<E>::value && POLICY == GX_POLICY_DEVICE_ACTIONS) {
      if (a.k_steps == 1) {
        hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, true, true, false, false, true>), grid, dim3(kClassicBlock), 0, stream, a);
        return;
      }
    }
    hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, true, true>), grid, dim3(kClassicBlock), 0, stream, a);
  } else if constexpr (E::kAlwaysBounded) {
    hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, false, true>), grid, dim3(kClassicBlock), 0, stream, a);
  } else {
    hipLaunchKernelGGL((classic_step_kernel<E, MODE, POLICY, false, false>), grid, dim3(kClassicBlock), 0, stream, a);
  }
}

template <class E>
static hipError_t launch_step_env(const StepArgs& a, int mode, int policy, bool fused, bool bounded, dim3 grid,
                                  hipStream_t stream) {
  if (mode == GX_MODE_EXACT) {
    if (policy == GX_POLICY_RANDOM) launch_step_variant<E, GX_MODE_EXACT, GX_POLICY_RANDOM>(a, fused, bounded, grid, stream);
    else launch_step_variant<E, GX_MODE_EXACT, GX_POLICY_DEVICE_ACTIONS>(a, fused, bounded, grid, stream);
  } else {
    if (policy == GX_POLICY_RANDOM) launch_step_variant<E, GX_MODE_FAST32, GX_POLICY_RANDOM>(a, fused, bounded, grid, stream);
    else launch_step_variant<E, GX_MODE_FAST32, GX_POLICY_DEVICE_ACTIONS>(a, fused, bounded, grid, stream);
  }
  return hipGetLastError();
}

// envs that exist in one numeric form only (the binary64 ones)
template <class E>
static hipError_t launch_step_env_exact(const StepArgs& a, int policy, bool fused, bool bounded, dim3 grid, hipStream_t stream) {
  if (policy == GX_POLICY_RANDOM) launch_step_variant<E, GX_MODE_EXACT, GX_POLICY_RANDOM>(a, fused, bounded, grid, stream);
  else launch_step_variant<E, GX_MODE_EXACT, GX_POLICY_DEVICE_ACTIONS>(a, fused, bounded, grid, stream);
  return hipGetLastError();
}

template <class E>
static hipError_t launch_reset_env(const StepArgs& a, int only_observe, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(classic_reset_kernel<E>, grid, dim3(kClassicBlock), 0, stream, a, only_observe);
  return hipGetLastError();
}

template <class E>
static hipError_t launch_sample_env(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(classic_sample_kernel<E>, grid, dim3(kClassicBlock), 0, stream, a, out);
  return hipGetLastError();
}

}  // namespace gx
