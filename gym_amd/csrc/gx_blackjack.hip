// Blackjack (Envs.Blackjack.cs): state (playerSum, dealerCard, usableAce), actions 0 stick / 1 hit.
//
// Int rows of the batch: row 0 the observation as the index of the reference's
// ObservationSpace Discrete(32*11*2), (playerSum*11 + dealerCard)*2 + usableAce; row 1 the
// dealer's face-down cards (raw sum in bits 0..7, "one of them is an ace" in bit 8; always 0
// under the reference's rules, whose kernel neither reads nor writes the row); row 2 the step
// counter.  Four envs per lane, 16 bytes per row per lane.
//
// Two rule sets, one kernel template:
//  - the reference's (Envs.Blackjack.cs:21-40): Reset draws Next(12,22), Next(1,11), Next(2)==1;
//    Step leaves the state alone, ends the episode at the tenth step and pays 1 there;
//  - GX_BLACKJACK_FULL, an extension: upstream Blackjack-v1 (natural = False, sab = False) with an
//    infinite deck.  Its draws come from three words per env and tick, each word one of the four
//    of a block that four consecutive envs share (one generator call per lane serves its four
//    envs), successive cards being successive base-13 digits of the word (word_next_digit):
//      stream 3  the action of the random policy (one binary digit), then the card of a hit;
//      stream 5  the dealer's first four draws after a stick (to 17 or more; the cards are added
//                to the face-down row, as upstream's dealer keeps them); a dealer who needs a
//                fifth draw, one hand in a thousand, goes on with the env's own stream 1;
//      stream 4  Reset: the dealer's two cards, then the player's two.
#include "gx_device.h"
#include "gx_launch.h"

namespace gx {

namespace {

__device__ __forceinline__ void load4i(const int32_t* base, uint64_t quad, int32_t (&out)[4]) {
  int4 v = reinterpret_cast<const int4*>(base)[quad];
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
__device__ __forceinline__ void store4i(int32_t* base, uint64_t quad, const int32_t (&in)[4]) {
  reinterpret_cast<int4*>(base)[quad] = make_int4(in[0], in[1], in[2], in[3]);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4f_nt(float* base, uint64_t quad, const float (&in)[4]) {
  f32x4 v = {in[0], in[1], in[2], in[3]};
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base) + quad);
}
__device__ __forceinline__ uint32_t pack4(const bool (&b)[4]) {
  return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
}

// rank index 0..12 of an infinite deck -> card value: ace = 1, 2..10, jack/queen/king = 10
__device__ __forceinline__ int32_t card_of(int32_t rank) { return rank + 1 < 10 ? rank + 1 : 10; }
__device__ __forceinline__ int32_t hand_value(int32_t raw, bool ace) { return raw + ((ace && raw + 10 <= 21) ? 10 : 0); }

constexpr uint32_t kStreamDeal = 4, kStreamDealer = 5;

// Envs.Blackjack.cs:23-25
__device__ __forceinline__ void blackjack_reset_reference(const RngKey& key, uint64_t env, int32_t& obs, int32_t& hidden) {
  const Draws d = philox_block(key, env, 0, 0);
  const int32_t sum = 12 + d.below(0, 10), dealer = 1 + d.below(1, 10), usable = d.below(2, 2) == 1;
  obs = (sum * 11 + dealer) * 2 + usable;
  hidden = 0;
}
// upstream's reset from the env's deal word
__device__ __forceinline__ void blackjack_deal(uint32_t word, int32_t& obs, int32_t& hidden) {
  const int32_t d0 = card_of(word_next_digit(word, 13)), d1 = card_of(word_next_digit(word, 13));
  const int32_t p0 = card_of(word_next_digit(word, 13)), p1 = card_of(word_next_digit(word, 13));
  const int32_t raw = p0 + p1;
  const bool usable = (p0 == 1 || p1 == 1) && raw + 10 <= 21;
  obs = ((raw + (usable ? 10 : 0)) * 11 + d0) * 2 + (usable ? 1 : 0);
  hidden = d1 | ((d1 == 1) << 8);
}
// the words of a lane's four envs from a shared-block stream: one generator call when the first env opens a block
template <bool ALIGNED>
__device__ __forceinline__ void lane_words(const StepArgs& a, uint64_t tick, uint64_t env0, uint32_t stream, uint32_t (&word)[4]) {
  if (ALIGNED) {
    const Draws d = tabular_block(a.seed_lo, a.seed_hi, tick, env0 >> 2, stream);
#pragma unroll
    for (int j = 0; j < 4; j++) word[j] = d.w[j];
  } else {
#pragma unroll
    for (int j = 0; j < 4; j++) word[j] = tabular_word(a.seed_lo, a.seed_hi, tick, env0 + j, stream);
  }
}

}  // namespace

// AHEAD (one step per launch, actions read from memory): a lane requests the rows of its next trip before the
// arithmetic of the current one, so that they precede the current trip's stores in the wave's memory queue.
template <int POLICY, bool FULL, bool ALIGNED, bool AHEAD = false>
__global__ void __launch_bounds__(kBlock) blackjack_step_kernel(StepArgs a) {
  const bool autoreset = (a.flags & GX_AUTORESET) != 0;
  const bool record = (a.flags & GX_RECORD_STATS) != 0;
  int32_t* const obs_row = a.state_int;
  int32_t* const hidden_row = a.state_int + a.stride;
  int32_t* const steps_row = a.state_int + 2 * a.stride;
  ThreadStats<int32_t> acc;  // rewards are -1, 0 or 1

  static_assert(!AHEAD || POLICY == GX_POLICY_DEVICE_ACTIONS, "the form that requests ahead reads its actions");
  const uint64_t nquads = (a.num_envs + 3) >> 2;
  const uint64_t it_stride = (uint64_t)gridDim.x * blockDim.x;
  // alternate launches sweep in opposite directions (memory-side cache reuse across steps)
  auto quad_of = [&](uint64_t it) { return a.reverse ? nquads - 1 - it : it; };
  int32_t next_obs[4], next_hidden[4] = {0, 0, 0, 0}, next_steps[4], next_act[4];
  auto request = [&](uint64_t g) {
    load4i(obs_row, g, next_obs);
    if (FULL) load4i(hidden_row, g, next_hidden);
    load4i(steps_row, g, next_steps);
    load4i(static_cast<const int32_t*>(a.actions), g, next_act);
  };
  const uint64_t it_first = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (AHEAD) {
    if (it_first < nquads) request(quad_of(it_first));
    // awaited here, outside the sweep: a wait at the loop's head would also be counted for the back edge, where the
    // stores of the trip before are in flight behind the rows
#pragma unroll
    for (int j = 0; j < 4; j++) {
      asm volatile("" : "+v"(next_obs[j]));
      if (FULL) asm volatile("" : "+v"(next_hidden[j]));
      asm volatile("" : "+v"(next_steps[j]));
      asm volatile("" : "+v"(next_act[j]));
    }
  }
  for (uint64_t it = it_first; it < nquads; it += it_stride) {
    const uint64_t q = quad_of(it);
    int32_t obs[4], hidden[4] = {0, 0, 0, 0}, steps[4], first_act[4] = {0, 0, 0, 0}, ep_ret[4] = {}, ep_len[4] = {};
    if constexpr (AHEAD) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        obs[j] = next_obs[j]; hidden[j] = next_hidden[j]; steps[j] = next_steps[j]; first_act[j] = next_act[j];
      }
      if (record && a.ep_length) load4i(a.ep_length, q, ep_len);
      // a lane without a further trip asks for quad 0 and drops what comes (a branch around the requests would end
      // in register copies that wait for the rows on the spot)
      request(it + it_stride < nquads ? quad_of(it + it_stride) : 0);
    } else {
      load4i(obs_row, q, obs);
      if (FULL) load4i(hidden_row, q, hidden);
      load4i(steps_row, q, steps);
      if (record && a.ep_length) load4i(a.ep_length, q, ep_len);
    }
    // Under either rule set a reward other than 0 comes with done (Envs.Blackjack.cs:36-37: 1 at the tenth step;
    // upstream: a bust or the settled hand), so the return of an open episode is 0 at every step boundary: the
    // ep_return row holds zeros from its allocation on and is neither read nor written (8 bytes per env-step).
    float reward[4];
    bool done[4], trunc[4];
    // the env's word of the tick gives the random policy its action and a hit its card
    constexpr bool need_word = POLICY == GX_POLICY_RANDOM || FULL;
    const uint64_t env0 = a.env_index_base + q * 4;
    const int k_steps = AHEAD ? 1 : a.k_steps;
    for (int k = 0; k < k_steps; k++) {
      const uint64_t tick = a.tick + (uint64_t)k;
      const RngKey key = make_key(a.seed_lo, a.seed_hi, tick);
      int32_t act[4] = {0, 0, 0, 0};
      if (AHEAD) {
#pragma unroll
        for (int j = 0; j < 4; j++) act[j] = first_act[j];
      } else if (POLICY == GX_POLICY_DEVICE_ACTIONS) {
        load4i(static_cast<const int32_t*>(a.actions) + (uint64_t)k * a.action_stride, q, act);
      }
      uint32_t word[4] = {0, 0, 0, 0}, dealer_word[4] = {0, 0, 0, 0}, deal_word[4] = {0, 0, 0, 0};
      if (need_word) lane_words<ALIGNED>(a, tick, env0, 3, word);
      if (FULL) {
        // with four envs per lane nearly every lane has one that sticks and one that ends: both words always
        lane_words<ALIGNED>(a, tick, env0, kStreamDealer, dealer_word);
        if (autoreset) lane_words<ALIGNED>(a, tick, env0, kStreamDeal, deal_word);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint64_t env = env0 + j;
        if (POLICY == GX_POLICY_RANDOM) act[j] = word_below(word[j], 2);  // Discrete(2), Envs.Blackjack.cs:12
        steps[j] += 1;
        bool d;
        int32_t r;
        if (!FULL) {  // Envs.Blackjack.cs:35-37
          d = steps[j] >= 10;
          r = d ? 1 : 0;
        } else {
          const bool usable = (obs[j] & 1) != 0;
          const int32_t pair = obs[j] >> 1, sum = pair / 11, dealer = pair - sum * 11;
          if (act[j] != 0) {
            const int32_t card = card_of(word_outcome(word[j], 2, 13));
            const int32_t raw = sum - (usable ? 10 : 0) + card;
            const bool usable2 = (usable || card == 1) && raw + 10 <= 21;
            const int32_t sum2 = raw + (usable2 ? 10 : 0);
            obs[j] = (sum2 * 11 + dealer) * 2 + (usable2 ? 1 : 0);
            d = sum2 > 21;
            r = d ? -1 : 0;
          } else {
            int32_t hsum = hidden[j] & 0xFF;
            bool hace = ((hidden[j] >> 8) & 1) != 0;
            int32_t raw = dealer + hsum;
            bool ace = dealer == 1 || hace;
            // the first four draws are the digits of the env's dealer word
            uint32_t dw = dealer_word[j];
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const int32_t card = card_of(word_next_digit(dw, 13));
              if (hand_value(raw, ace) < 17) {
                raw += card;
                hsum += card;
                ace = ace || card == 1;
                hace = hace || card == 1;
              }
            }
            // a fifth and later draws (a hand of seven or more cards): the env's own stream 1, four draws per call
            for (uint32_t block = 0; hand_value(raw, ace) < 17; block++) {
              const Draws dd = philox_block(key, env, 1, block);
#pragma unroll
              for (int i = 0; i < 4; i++) {
                if (hand_value(raw, ace) < 17) {
                  const int32_t card = card_of(dd.below(i, 13));
                  raw += card;
                  hsum += card;
                  ace = ace || card == 1;
                  hace = hace || card == 1;
                }
              }
            }
            hidden[j] = hsum | ((hace ? 1 : 0) << 8);
            const int32_t dv = hand_value(raw, ace);
            const int32_t ps = sum > 21 ? 0 : sum, ds = dv > 21 ? 0 : dv;
            d = true;
            r = (ps > ds) - (ps < ds);
          }
        }
        const bool t = a.time_limit > 0 && steps[j] >= a.time_limit;  // Wrappers.TimeLimit.cs:26-31
        d = d || t;
        reward[j] = (float)r;
        done[j] = d;
        trunc[j] = t;
        if (record) {  // EnvWrapper.cs:46-55
          ep_ret[j] += r;
          ep_len[j] += 1;
          acc.add(d && q * 4 + j < a.num_envs, ep_ret[j], a.ep_length ? ep_len[j] : steps[j]);
          ep_ret[j] = d ? 0 : ep_ret[j];
          ep_len[j] = d ? 0 : ep_len[j];
        }
        if (autoreset && d) {
          if (FULL) blackjack_deal(deal_word[j], obs[j], hidden[j]);
          else blackjack_reset_reference(key, env, obs[j], hidden[j]);
          steps[j] = 0;
        }
      }
      if (!AHEAD && a.out_reward) {  // per-step rows of a rollout
        const uint64_t row = (uint64_t)k * a.stride;
        store4f_nt(static_cast<float*>(a.out_reward) + row, q, reward);
        __builtin_nontemporal_store(pack4(done), reinterpret_cast<uint32_t*>(a.out_done + row) + q);
        if (a.out_truncated) __builtin_nontemporal_store(pack4(trunc), reinterpret_cast<uint32_t*>(a.out_truncated + row) + q);
      }
    }
    // under the reference's rules only a reset changes the observation
    if (FULL || autoreset) store4i(obs_row, q, obs);
    if (FULL) store4i(hidden_row, q, hidden);
    store4i(steps_row, q, steps);
    if (record && a.ep_length) store4i(a.ep_length, q, ep_len);
    if (AHEAD || !a.skip_last) {  // write-only rows stream past the caches
      store4f_nt(static_cast<float*>(a.reward), q, reward);
      __builtin_nontemporal_store(pack4(done), reinterpret_cast<uint32_t*>(a.done) + q);
      if (a.truncated) __builtin_nontemporal_store(pack4(trunc), reinterpret_cast<uint32_t*>(a.truncated) + q);
    }
  }
  if (record) block_commit_stats(acc.widen(), a.stat_slots);
}

template <bool FULL>
__global__ void __launch_bounds__(kBlock) blackjack_reset_kernel(StepArgs a) {
  const RngKey key = make_key(a.seed_lo, a.seed_hi, a.tick);
  const uint64_t nquads = (a.num_envs + 3) >> 2;
  for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < nquads; q += (uint64_t)gridDim.x * kBlock) {
    int32_t obs[4], hidden[4], zero[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t env = a.env_index_base + q * 4 + j;
      if (FULL) blackjack_deal(tabular_word(a.seed_lo, a.seed_hi, a.tick, env, kStreamDeal), obs[j], hidden[j]);
      else blackjack_reset_reference(key, env, obs[j], hidden[j]);
    }
    store4i(a.state_int, q, obs);
    store4i(a.state_int + a.stride, q, hidden);
    store4i(a.state_int + 2 * a.stride, q, zero);
  }
}

__global__ void __launch_bounds__(kBlock) blackjack_sample_kernel(StepArgs a, int32_t* out) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < a.num_envs; i += (uint64_t)gridDim.x * kBlock)
    out[i] = word_below(tabular_word(a.seed_lo, a.seed_hi, a.tick, a.env_index_base + i), 2);
}

hipError_t launch_blackjack_step(const StepArgs& a, int policy, dim3 grid, hipStream_t stream) {
  const bool full = (a.flags & GX_BLACKJACK_FULL) != 0, aligned = (a.env_index_base & 3) == 0;
#define GX_BJ_LAUNCH(P, F)                                                                                  \
  do {                                                                                                      \
    if (aligned) hipLaunchKernelGGL((blackjack_step_kernel<P, F, true>), grid, dim3(kBlock), 0, stream, a); \
    else hipLaunchKernelGGL((blackjack_step_kernel<P, F, false>), grid, dim3(kBlock), 0, stream, a);        \
  } while (0)
  if (policy == GX_POLICY_RANDOM) {
    if (full) GX_BJ_LAUNCH(GX_POLICY_RANDOM, true); else GX_BJ_LAUNCH(GX_POLICY_RANDOM, false);
  } else if (full && aligned && a.k_steps == 1 && !a.out_reward && !a.skip_last) {
    // upstream's rules, one step, actions from memory: 0.409 -> 0.399 ms at 64 Mi envs with the next trip requested
    // ahead (the reference's rules, which only move two rows, and the in-kernel policy do not gain)
    hipLaunchKernelGGL((blackjack_step_kernel<GX_POLICY_DEVICE_ACTIONS, true, true, true>), grid, dim3(kBlock), 0, stream, a);
  } else {
    if (full) GX_BJ_LAUNCH(GX_POLICY_DEVICE_ACTIONS, true); else GX_BJ_LAUNCH(GX_POLICY_DEVICE_ACTIONS, false);
  }
#undef GX_BJ_LAUNCH
  return hipGetLastError();
}

hipError_t launch_blackjack_reset(const StepArgs& a, dim3 grid, hipStream_t stream) {
  if (a.flags & GX_BLACKJACK_FULL) hipLaunchKernelGGL(blackjack_reset_kernel<true>, grid, dim3(kBlock), 0, stream, a);
  else hipLaunchKernelGGL(blackjack_reset_kernel<false>, grid, dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_blackjack_sample(const StepArgs& a, void* out, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL(blackjack_sample_kernel, grid, dim3(kBlock), 0, stream, a, static_cast<int32_t*>(out));
  return hipGetLastError();
}

}  // namespace gx
