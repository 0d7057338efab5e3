// Shared device-side pieces of the batched environment engine: the counter
// RNG, the fused-wrapper bookkeeping and the per-block statistics reduction.
// gfx950 only; wavefront = 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gymx.h"

namespace gx {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
// The classic-control kernels run one wave per workgroup: measured on the CartPole step at
// 16 Mi envs, 64 / 128 / 256 / 512 threads take 0.142 / 0.146 / 0.1465 / 0.150 ms; with 128 or 256 threads
// MountainCar, MountainCarContinuous and Pendulum gain 1-3 %, CartPole and Acrobot lose as much.
// (-DGX_CLASSIC_BLOCK=n builds the other sizes for such comparisons.)
#ifndef GX_CLASSIC_BLOCK
#define GX_CLASSIC_BLOCK 64
#endif
constexpr int kClassicBlock = GX_CLASSIC_BLOCK;
// The tabular step kernels stage the transition table in LDS once per workgroup, so their
// workgroup size follows the table.  At 64 Mi envs, 256 / 128 threads: FrozenLake 4x4
// (352 bytes) 0.283 / 0.271 ms, 8x8 (1.3 KB) 0.279 / 0.266 ms, slippery 8x8 (3.9 KB)
// 0.343 / 0.360 ms, full Taxi-v3 (14 KB) 0.355 / 0.418 ms.
constexpr int kTabularBlock = 256, kTabularBlockSmallTable = 128;
constexpr size_t kSmallTableBytes = 2048;
// The host path stages a step's actions and results in pinned mapped memory up to this many
// bytes per step; beyond it the copies run on the copy engine (a separate stream).  CartPole,
// actions in, reward + done + observation out, ms per step staged / copy engine: 1 024 envs
// 0.023 / 0.056, 32 768 envs 0.053 / 0.075, 65 536 envs 0.087 / 0.094, 262 144 envs 0.339 / 0.185.
constexpr size_t kStageLimitBytes = 2u << 20;
constexpr int kEnvsPerThread = 4;    // one 16-byte vector per component per lane
constexpr int kStatFields = 6;

// ---------------------------------------------------------------------------
// Counter RNG.  The reference's System.Random instances are unseeded
// (Envs.CartPole.cs:19, Spaces.Box.cs:14) and Env.Seed is empty (Env.cs:33-36),
// so the engine defines its own stream:
//   w = word (d & 3) of Philox4x32-10(key = seed, counter = (env_lo, env_hi,
//       tick_lo, stream << 24 | tick_hi8 << 16 | d >> 2))          for draw d
//   NextDouble() stand-in: u = w * 2^-32;   Next(n) stand-in: (w * n) >> 32
// `tick` counts the Reset()/Step() calls made on the batch before the one that
// consumes the draw, so no per-env RNG state is stored or moved.
// Streams: 0 = Reset draws, 1 = draws made inside Step, 2 = random policy,
// 3 = the per-tick word of the table-driven envs, 4 / 5 = the deal and dealer
// words of Blackjack with upstream's rules (shared like stream 3).
// The policy stream takes one draw per env per tick, so four consecutive ticks
// share a block: tick t reads word t % 4 of the block whose counter holds
// t / 4, and a fused K-step rollout runs the generator once every four steps.
// The table-driven envs (FrozenLake, Taxi, CliffWalking, DummyEnv) are stepped
// four envs per lane and spend most of a step's arithmetic on the generator, so
// their word is shared the other way: stream 3, one block per four consecutive
// global env indices, env e reading word e % 4 of the block whose counter holds
// e / 4 -- one generator call per lane per step.  That word w gives the random
// policy its action and a slippery env the one draw its Step needs: with
// u = w * 2^-32 the action is floor(u * A) and the outcome is the next base-O
// digit of the same number, floor(frac(u * A) * O) -- independent of the
// action and uniform to 2^-32.
// ---------------------------------------------------------------------------
struct RngKey {
  uint32_t k0, k1;     // seed
  uint32_t tick_lo;
  uint32_t tick_hi8;   // bits 32..39 of the tick, pre-shifted to bit 16
};

__device__ __forceinline__ RngKey make_key(uint32_t seed_lo, uint32_t seed_hi, uint64_t tick) {
  RngKey k;
  k.k0 = seed_lo;
  k.k1 = seed_hi;
  k.tick_lo = (uint32_t)tick;
  k.tick_hi8 = ((uint32_t)(tick >> 32) & 0xFFu) << 16;
  return k;
}

// One Philox block = four 32-bit draws.
struct Draws {
  uint32_t w[4];
  // uniform double in [0,1) with 32 random bits (exact in binary64)
  __device__ __forceinline__ double uniform(int i) const { return (double)w[i] * 0x1.0p-32; }
  __device__ __forceinline__ int32_t below(int i, uint32_t n) const { return (int32_t)__umulhi(w[i], n); }
};

__device__ __forceinline__ Draws philox_block(const RngKey& key, uint64_t env, uint32_t stream, uint32_t block) {
  uint32_t c0 = (uint32_t)env, c1 = (uint32_t)(env >> 32), c2 = key.tick_lo;
  uint32_t c3 = (stream << 24) | key.tick_hi8 | block;
  uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Draws d;
  d.w[0] = c0; d.w[1] = c1; d.w[2] = c2; d.w[3] = c3;
  return d;
}

// Policy stream: the block shared by ticks 4m .. 4m+3, and tick t's word of it.
__device__ __forceinline__ Draws policy_block(uint32_t seed_lo, uint32_t seed_hi, uint64_t tick, uint64_t env) {
  return philox_block(make_key(seed_lo, seed_hi, tick >> 2), env, 2, 0);
}
__device__ __forceinline__ uint32_t policy_word(const Draws& d, uint64_t tick) {
  const uint32_t i = (uint32_t)tick & 3u;
  return i == 0 ? d.w[0] : (i == 1 ? d.w[1] : (i == 2 ? d.w[2] : d.w[3]));
}
// Streams whose block is shared by the four envs 4g .. 4g+3 at a tick (3: the per-tick word of the table-driven
// envs; 4 and 5: Blackjack's deal and dealer words), and env e's word of it.
__device__ __forceinline__ Draws tabular_block(uint32_t seed_lo, uint32_t seed_hi, uint64_t tick, uint64_t env_group, uint32_t stream = 3) {
  return philox_block(make_key(seed_lo, seed_hi, tick), env_group, stream, 0);
}
__device__ __forceinline__ uint32_t tabular_word(uint32_t seed_lo, uint32_t seed_hi, uint64_t tick, uint64_t env, uint32_t stream = 3) {
  const Draws d = tabular_block(seed_lo, seed_hi, tick, env >> 2, stream);
  const uint32_t i = (uint32_t)env & 3u;
  return i == 0 ? d.w[0] : (i == 1 ? d.w[1] : (i == 2 ? d.w[2] : d.w[3]));
}
// The next base-n digit of u = w * 2^-32: returns floor(n u) and leaves the fraction of n u in w.  Successive
// calls give independent draws below n out of one word, the k-th uniform to n^k * 2^-32.
__device__ __forceinline__ int32_t word_next_digit(uint32_t& w, uint32_t n) {
  const int32_t d = (int32_t)__umulhi(w, n);
  w *= n;
  return d;
}
// the outcome in [0, n_out) a table-driven env draws inside Step: the base-n_out digit that
// follows the action digit of its word of the tick (see the stream list above)
__device__ __forceinline__ int32_t word_outcome(uint32_t w, uint32_t n_act, uint32_t n_out) {
  return (int32_t)(__umulhi(w, n_act * n_out) - n_out * __umulhi(w, n_act));
}
// the two ways a draw is consumed: NextDouble() and Next(n)
__device__ __forceinline__ double word_uniform(uint32_t w) { return (double)w * 0x1.0p-32; }
__device__ __forceinline__ int32_t word_below(uint32_t w, uint32_t n) { return (int32_t)__umulhi(w, n); }

// Spaces.Box.cs:30 -- (float)(NextDouble() * (High - Low) + Low), the range
// being a binary32 subtraction promoted to double.
__device__ __forceinline__ float box_sample(double u, float low, float high) {
  float range = high - low;
  return (float)(u * (double)range + (double)low);
}

// ---------------------------------------------------------------------------
// Episode statistics (RecordEpisodeStatistics, EnvWrapper.cs:23-58): every
// thread accumulates the episodes its envs finished, a wave reduces them and
// adds the result to a slot only it writes, so no two waves ever meet on an
// address.  Integer valued returns stay exact in double, so those sums do not
// depend on order; the others are added in a fixed order (see below).
// ---------------------------------------------------------------------------
struct StatSlot {
  long long episodes;
  double sum_return;
  long long sum_length;
  double sum_return_sq;
  double min_return;
  double max_return;
  long long pad[2];  // 64 bytes: one slot per cache line half
};

struct StatAcc {
  long long episodes = 0;
  double sum_return = 0.0;
  long long sum_length = 0;
  double sum_return_sq = 0.0;
  double min_return = __builtin_huge_val();
  double max_return = -__builtin_huge_val();

  __device__ __forceinline__ void add(double ret, int len) {
    episodes += 1;
    sum_return += ret;
    sum_length += len;
    sum_return_sq += ret * ret;
    min_return = ret < min_return ? ret : min_return;
    max_return = ret > max_return ? ret : max_return;
  }
};

// Per-thread accumulators in front of StatAcc.
// Integer-valued returns (every env but MountainCarContinuous and Pendulum)
// accumulate in integer registers, branch-free.
template <class EpRet>
struct ThreadStats;

template <>
struct ThreadStats<int32_t> {
  int32_t episodes = 0, sum_length = 0;
  long long sum_return = 0, sum_return_sq = 0;
  int32_t min_return = 0x7FFFFFFF, max_return = (int32_t)0x80000000;
  __device__ __forceinline__ void add(bool d, int32_t ret, int32_t len) {
    episodes += d ? 1 : 0;
    sum_length += d ? len : 0;
    const int32_t r = d ? ret : 0;
    sum_return += r;
    sum_return_sq += (long long)r * r;
    min_return = d && ret < min_return ? ret : min_return;
    max_return = d && ret > max_return ? ret : max_return;
  }
  __device__ __forceinline__ StatAcc widen() const {
    StatAcc a;
    a.episodes = episodes; a.sum_return = (double)sum_return; a.sum_length = sum_length;
    a.sum_return_sq = (double)sum_return_sq;
    if (episodes) { a.min_return = (double)min_return; a.max_return = (double)max_return; }
    return a;
  }
};

template <>
struct ThreadStats<double> {
  StatAcc acc;
  __device__ __forceinline__ void add(bool d, double ret, int32_t len) {
    if (d) acc.add(ret, len);
  }
  __device__ __forceinline__ StatAcc widen() const { return acc; }
};

__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_down(lo, delta, 64);
  hi = __shfl_down(hi, delta, 64);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long shfl_down_i64(long long v, int delta) {
  int lo = (int)(v & 0xFFFFFFFFll), hi = (int)(v >> 32);
  lo = __shfl_down(lo, delta, 64);
  hi = __shfl_down(hi, delta, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// Called by every thread of the workgroup exactly once, at the end of the
// kernel.  `slots` has one entry per wave of the largest grid the batch ever
// launches: a wave folds its lanes with shuffles and adds the result to its own
// slot with return-less atomics, so it neither waits for the slot to be read
// nor meets its workgroup at a barrier before it retires.  Nobody else touches
// that slot during the launch and launches are stream-ordered, so the order of
// the floating-point additions into a slot is fixed.
__device__ __forceinline__ void block_commit_stats(const StatAcc& mine, StatSlot* slots) {
  StatAcc a = mine;
  // a wave that finished no episode has nothing to add
  if (!__any(a.episodes != 0)) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a.episodes += shfl_down_i64(a.episodes, d);
    a.sum_return += shfl_down_f64(a.sum_return, d);
    a.sum_length += shfl_down_i64(a.sum_length, d);
    a.sum_return_sq += shfl_down_f64(a.sum_return_sq, d);
    double mn = shfl_down_f64(a.min_return, d), mx = shfl_down_f64(a.max_return, d);
    a.min_return = mn < a.min_return ? mn : a.min_return;
    a.max_return = mx > a.max_return ? mx : a.max_return;
  }
  if ((threadIdx.x & 63) == 0) {
    StatSlot* g = slots + (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    atomicAdd(reinterpret_cast<unsigned long long*>(&g->episodes), (unsigned long long)a.episodes);
    atomicAdd(reinterpret_cast<unsigned long long*>(&g->sum_length), (unsigned long long)a.sum_length);
    unsafeAtomicAdd(&g->sum_return, a.sum_return);
    unsafeAtomicAdd(&g->sum_return_sq, a.sum_return_sq);
    unsafeAtomicMin(&g->min_return, a.min_return);
    unsafeAtomicMax(&g->max_return, a.max_return);
  }
}

// ---------------------------------------------------------------------------
// Arguments shared by the step / rollout kernels.  Component arrays are SoA,
// `stride` elements apart, 16-byte aligned, padded to a multiple of four envs.
// ---------------------------------------------------------------------------
struct StepArgs {
  void* state_real;        // [n_real][stride] float or double
  int32_t* state_int;      // [n_int][stride], last row = step counter
  const void* actions;     // int32/float [k_steps][action_stride]; unused for the random policy
  void* reward;            // [stride] float or double
  uint8_t* done;           // [stride]
  uint8_t* truncated;      // [stride] or nullptr
  void* ep_return;         // [stride] int32 or double; nullptr unless RECORD_STATS
  int32_t* ep_length;      // [stride]; nullptr when the step counter serves (AUTORESET)
  StatSlot* stat_slots;    // one per wave of the grid
  float* obs;              // derived observation rows, or nullptr
  const uint32_t* table;   // tabular envs: packed transition table
  uint32_t* error_flag;    // set non-zero by a kernel that meets an action the reference throws on
  uint32_t* render;        // frame envs: [num_envs][G::kRecordDwords], what the raster reads of each env's state
  // per-step outputs of a K-step rollout (gx_vec_rollout_out, GX_OUT_ALL): row k holds step k's
  // values; nullptr = not wanted
  void* out_reward;        // [k_steps][stride] float or double
  uint8_t* out_done;       // [k_steps][stride]
  uint8_t* out_truncated;  // [k_steps][stride] or nullptr
  // one launch only (gx_vec_set_rng_draws): uniforms that stand in for the generator in each env's next ball
  // reset (Pong, [2][stride]) or release of the ball (Breakout, [1][stride]); nullptr otherwise
  const double* forced_draws;
  uint64_t num_envs;
  uint64_t stride;
  uint64_t action_stride;
  uint64_t env_index_base;
  uint64_t tick;
  uint32_t seed_lo, seed_hi;
  uint32_t flags;          // gx_flags
  int32_t time_limit;      // 0 = none
  int32_t k_steps;         // 1 for a single step
  int32_t skip_last;       // GX_OUT_NONE: the reward / done / truncated rows of the view are not written
  uint32_t reverse;        // sweep the batch back to front (alternates launch to launch)
  int32_t table_states, table_actions, table_outcomes;
  int32_t table_lut;       // reward values that follow the packed entries (a multiple of 4, at least 8)
  uint32_t strict_actions; // tabular: an action outside the space raises error_flag (caller-supplied tables)
  int32_t start_state;
  int32_t step_limit;
  // 1: the open-episode return is not stored; it is ret_per_step x the step counter (envs whose
  // every non-final step pays the same integer reward, with auto-reset on)
  int32_t derive_returns;
  int32_t ret_per_step;
};

}  // namespace gx
