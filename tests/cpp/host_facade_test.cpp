// Exercises include/gymx.hpp (the C++ host facade over the C ABI) on a GPU:
// the reference's own unit tests restated (Gymnasium.Tests/*.cs), the
// VectorEnv contract, the fused and the host-composed wrappers, and a replay
// of a CartPole batch against the CPU oracle (linked here as the checker).
// Prints one line per check; exit status 0 when all pass.
// Built by tests/cpp/facade_build.py.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "gymx.hpp"
#include "oracle/gym_oracle.h"

static int failures = 0;
#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
      failures++;                                                        \
    }                                                                    \
  } while (0)

using namespace gymx;

static void reference_unit_tests() {
  {  // CartPoleTests.cs:8-19
    DeviceEnv<CartPole> env;
    auto s = env.Reset();
    CHECK(s.size() == 4);
    for (int i = 0; i < 10; i++) {
      auto r = env.Step(1);
      CHECK(r.reward >= 0 && r.reward <= 1);
    }
    CHECK(env.ActionSpace.N == 2 && env.ObservationSpace.Dimension() == 4);
  }
  {  // AcrobotTests.cs:7-18, and the IndexOutOfRangeException of Envs.Acrobot.cs:51
    DeviceEnv<Acrobot> env;
    CHECK(env.Reset().size() == 6);
    CHECK(env.Step(1).reward <= 0);
    bool threw = false;
    try { env.Step(3); } catch (const std::out_of_range&) { threw = true; }
    CHECK(threw);
  }
  {  // MountainCarTests.cs:7-19, MountainCarContinuousTests.cs:7-19, PendulumTests.cs:7-18
    DeviceEnv<MountainCar> mc;
    auto s = mc.Reset();
    CHECK(mc.ObservationSpace.Contains({s[0], s[1]}));
    CHECK(mc.Step(2).reward <= 0);
    DeviceEnv<MountainCarContinuous> mcc;
    auto t = mcc.Reset();
    CHECK(mcc.ObservationSpace.Contains({t[0], t[1]}));
    mcc.Step(0.5f);
    DeviceEnv<Pendulum> p;
    CHECK(p.Reset().size() == 3 && p.Step(0.0f).reward <= 0);
  }
  {  // AtariEnvTests.cs:9-91
    DeviceEnv<Pong> pong;
    CHECK(pong.ActionSpace.N == 4 && pong.Reset().size() == 210 * 160 * 3);
    auto r = pong.Step(0);
    CHECK(r.state.size() == 210 * 160 * 3 && r.reward >= -1 && r.reward <= 1 && r.info.count("left_score") == 1);
    DeviceEnv<Breakout> bk;
    CHECK(bk.ActionSpace.N == 4 && bk.Reset().size() == 210 * 160 * 3);
    auto b = bk.Step(0);
    CHECK(b.reward >= 0 && b.info.at("lives") == 5 && b.info.at("bricks_remaining") == 108);
    DeviceEnv<SpaceInvaders> si;
    CHECK(si.ActionSpace.N == 6 && si.Reset().size() == 210 * 160 * 3);
    auto v = si.Step(0);
    CHECK(v.reward >= 0 && v.info.at("lives") == 3 && v.info.at("aliens_remaining") == 55 && v.info.at("player_x") == 76.0);
  }
  {  // Envs.Blackjack.cs:21-40: the state is a tuple that never changes; reward 1 at the tenth step
    DeviceEnv<Blackjack> env;
    CHECK(env.ActionSpace.N == 2 && env.ObservationSpace.N == 704);
    auto s0 = env.Reset();
    CHECK(std::get<0>(s0) >= 12 && std::get<0>(s0) <= 21 && std::get<1>(s0) >= 1 && std::get<1>(s0) <= 10);
    StepResult<std::tuple<int, int, bool>> r{};
    for (int i = 0; i < 10; i++) {
      r = env.Step(i & 1);
      CHECK(r.state == s0 && r.done == (i == 9) && r.reward == (i == 9 ? 1.0 : 0.0));
    }
  }
  {  // DummyEnvTests.cs:6-21
    DeviceEnv<DummyEnv> env;
    CHECK(env.Reset() == 0);
    auto r = env.Step(1);
    CHECK(r.state == 1 && r.reward == 1.0 && !r.done);
    for (int i = 0; i < 9; i++) r = env.Step(1);
    CHECK(r.done);
  }
  std::printf("reference unit tests restated: done\n");
}

static void registry_and_spaces() {
  GymnasiumRegistration::RegisterAll();  // GymnasiumRegistration.cs:8-30
  CHECK(EnvRegistry::List().size() == 12);
  auto env = EnvRegistry::Make<DeviceEnv<CartPole>>("CartPole-v1");
  CHECK(env->Reset().size() == 4);
  bool threw = false;
  try { EnvRegistry::Make("LunarLander-v2"); } catch (const std::invalid_argument&) { threw = true; }  // EnvRegistry.cs:28-31
  CHECK(threw);
  Box box({-1.0f, 0.0f}, {1.0f, 2.0f});
  for (int i = 0; i < 100; i++) CHECK(box.Contains(box.Sample()));
  CHECK(!box.Contains({2.0f, 1.0f}) && !box.Contains({0.0f}));
  threw = false;
  try { Box bad({0.0f}, {1.0f, 2.0f}); } catch (const std::invalid_argument&) { threw = true; }  // Spaces.Box.cs:18-19
  CHECK(threw);
  Discrete d(5);
  for (int i = 0; i < 100; i++) CHECK(d.Contains(d.Sample()));
  CHECK(!d.Contains(5) && !d.Contains(-1) && d.Shape() == std::vector<int>{1});
  std::printf("registry and spaces: done\n");
}

// VectorEnv over 1000 CartPoles replayed against the oracle: same seed, same actions.
static void vector_env_against_oracle() {
  const uint64_t n = 1000;
  BatchOptions o;
  o.seed = 42;
  o.autoreset = true;
  o.record_stats = true;
  VectorEnv<CartPole> vec(n, o);
  gxo_vec* ref = gxo_vec_create(GXO_CARTPOLE, n, 42, GXO_AUTORESET | GXO_RECORD_STATS, 0, nullptr, 0);
  auto states = vec.Reset();
  gxo_vec_reset(ref);
  CHECK(states.size() == n && vec.NumEnvs() == n);
  std::vector<float> want(4 * n);
  gxo_vec_get_obs(ref, want.data());
  bool same = true;
  for (uint64_t i = 0; i < n; i++)
    for (int c = 0; c < 4; c++) same = same && std::memcmp(&states[i][c], &want[c * n + i], 4) == 0;
  CHECK(same);
  std::vector<int> actions(n);
  std::vector<double> reward(n);
  std::vector<uint8_t> done(n), trunc(n);
  uint64_t x = 88172645463325252ull;
  for (int t = 0; t < 60; t++) {
    for (uint64_t i = 0; i < n; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      actions[i] = (int)(x & 1);
    }
    auto res = vec.Step(actions);
    std::vector<int32_t> a32(actions.begin(), actions.end());
    gxo_vec_step(ref, a32.data(), reward.data(), done.data(), trunc.data());
    gxo_vec_get_obs(ref, want.data());
    for (uint64_t i = 0; i < n; i++) {
      same = same && res[i].reward == reward[i] && res[i].done == (done[i] != 0);
      for (int c = 0; c < 4; c++) same = same && std::memcmp(&res[i].state[c], &want[c * n + i], 4) == 0;
    }
  }
  CHECK(same);
  gx_stats gs = vec.Device().Stats();
  gxo_stats os;
  gxo_vec_stats(ref, &os);
  CHECK(gs.episodes == os.episodes && gs.episodes > 0 && gs.sum_return == os.sum_return && gs.sum_length == os.sum_length);
  gxo_vec_destroy(ref);
  bool threw = false;
  try { vec.Step(std::vector<int>(n - 1)); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  std::printf("VectorEnv<CartPole> x 1000, 60 steps: bit-identical to the oracle, %lld episodes\n", (long long)gs.episodes);
}

// TimeLimit and RecordEpisodeStatistics: host-composed around one env (Wrappers.TimeLimit.cs,
// EnvWrapper.cs:23-58) and fused into the batch give the same episodes.
static void wrappers() {
  auto inner = std::make_shared<DeviceEnv<DummyEnv>>();
  auto limited = std::make_shared<TimeLimit<int, int>>(inner, 4);
  RecordEpisodeStatistics<int, int> rec(limited);
  rec.Reset();
  StepResult<int> r{};
  for (int i = 0; i < 4; i++) r = rec.Step(1);
  CHECK(r.done && r.info.count("TimeLimit.truncated") == 1);
  CHECK(rec.EpisodeRewards.size() == 1 && rec.EpisodeRewards[0] == 1 + 2 + 3 + 4 && rec.EpisodeLengths[0] == 4);
  rec.Step(1);
  rec.Reset();  // EnvWrapper.cs:34-38: the open episode is recorded by Reset()
  CHECK(rec.EpisodeRewards.size() == 2 && rec.EpisodeLengths[1] == 1);

  BatchOptions o;
  o.time_limit = 4;
  o.record_stats = true;
  VectorEnv<DummyEnv> vec(3, o);
  vec.Reset();
  std::vector<StepResult<int>> res;
  for (int i = 0; i < 4; i++) res = vec.Step({1, 1, 2});
  CHECK(res[0].done && res[0].info.count("TimeLimit.truncated") == 1 && res[2].state == 8);
  gx_stats s = vec.Device().Stats();
  CHECK(s.episodes == 3 && s.sum_length == 12 && s.sum_return == 10 + 10 + 20);
  std::printf("wrappers, host-composed and fused: done\n");
}

// The binary64 mode through gymx::Batch (the raw batch): doubles in, doubles out, checked against
// the oracle's binary64 restatement; the float-typed VectorEnv refuses it.
static void fp64_mode() {
  const uint64_t n = 257;
  BatchOptions o;
  o.seed = 11;
  o.mode = GX_MODE_FP64;
  Batch b(GX_ENV_MOUNTAINCAR, n, o);
  CHECK(b.info().real_dtype == GX_F64 && b.info().obs_dtype == GX_F64 && b.info().obs_bytes == n * 2 * 8);
  gxo_vec* ref = gxo_vec_create(GXO_MOUNTAINCAR, n, 11, GXO_FP64, 0, nullptr, 0);
  b.Reset();
  gxo_vec_reset(ref);
  std::vector<double> got(2 * n), want(2 * n), reward(n);
  std::vector<float> rew(n);
  std::vector<int32_t> actions(n);
  std::vector<uint8_t> done(n), odone(n);
  b.GetObs(got.data());
  gxo_vec_get_obs(ref, want.data());
  CHECK(std::memcmp(got.data(), want.data(), got.size() * 8) == 0);  // the reset formula has no transcendental
  double worst = 0;
  for (int t = 0; t < 50; t++) {
    for (uint64_t i = 0; i < n; i++) actions[i] = (int32_t)((i + (uint64_t)t) % 3);
    b.Step(actions.data(), got.data(), rew.data(), done.data(), nullptr);
    gxo_vec_step(ref, actions.data(), reward.data(), odone.data(), nullptr);
    gxo_vec_get_obs(ref, want.data());
    for (size_t k = 0; k < got.size(); k++) worst = std::max(worst, std::fabs(got[k] - want[k]));
    CHECK(done == odone);
  }
  CHECK(worst <= 1e-12);  // free-running for 50 steps: rounding differences stay near 1e-16 on this env
  gxo_vec_destroy(ref);
  bool threw = false;
  try { VectorEnv<MountainCar> v(4, o); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  std::printf("Batch(MountainCar, FP64) x 257, 50 steps: max abs difference %.3g\n", worst);
}

// Replaying what the reference drew: Pong's Reset() with the two NextDouble() values handed over
// (Envs.Pong.cs:56-59), against the source's formula and against the oracle given the same values.
static void supplied_draws() {
  const uint64_t n = 5;
  BatchOptions o;
  o.seed = 3;
  Batch b(GX_ENV_PONG, n, o);
  gxo_vec* ref = gxo_vec_create(GXO_PONG, n, 3, 0, 0, nullptr, 0);
  const std::vector<double> u = {0.5, 0.75, 0.0, 0.999, 0.25, /* vertical */ 0.0, 0.5, 0.125, 0.875, 1.0 / 3.0};
  b.SetRngDraws(u);
  CHECK(gxo_vec_set_rng_draws(ref, u.data()) == 0);
  b.Reset();
  gxo_vec_reset(ref);
  std::vector<uint8_t> got = b.GetState(), want(got.size());
  gxo_vec_get_state(ref, want.data());
  CHECK(got == want);
  const double* r = reinterpret_cast<const double*>(got.data());
  for (uint64_t i = 0; i < n; i++) {
    CHECK(r[0 * n + i] == 80.0 && r[1 * n + i] == 105.0);
    CHECK(r[2 * n + i] == (u[i] > 0.5 ? 3.0 : -3.0));
    CHECK(r[3 * n + i] == (u[n + i] - 0.5) * 2.0 * 3.0);
  }
  gxo_vec_destroy(ref);
  bool threw = false;
  try { b.SetRngDraws(std::vector<double>(2 * n, 1.0)); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);  // NextDouble() never returns 1
  std::printf("Batch(Pong) x 5: Reset() with supplied draws equals the source's formula and the oracle\n");
}

// A caller-supplied transition table through gymx::Batch: the reference's CliffWalking written out as P[s][a]
// (Envs.CliffWalking.cs:36-54) gives the source's known answers, the oracle's table env the same, and an
// action outside the space throws like the reference's out-of-range index.
static void caller_supplied_table() {
  const uint64_t n = 3;
  BatchOptions o;
  o.n_states = 48; o.n_actions = 4; o.start_state = 36; o.step_limit = 100;
  o.table.resize(48 * 4);
  for (int s = 0; s < 48; s++)
    for (int a = 0; a < 4; a++) {
      int row = s / 12, col = s % 12;
      if (a == 0) col = std::max(col - 1, 0);
      if (a == 1) row = std::min(row + 1, 3);
      if (a == 2) col = std::min(col + 1, 11);
      if (a == 3) row = std::max(row - 1, 0);
      const int ns = row * 12 + col;
      const bool cliff = ns >= 37 && ns <= 46, goal = ns == 47;
      o.table[(size_t)s * 4 + a] = gx_transition{ns, goal ? 0.0f : (cliff ? -100.0f : -1.0f), (cliff || goal) ? 1 : 0};
    }
  Batch b(GX_ENV_TABLE, n, o);
  CHECK(b.info().action_n == 4 && b.info().obs_dtype == GX_I32);
  gxo_vec* ref = gxo_vec_create_table(n, 0, 0, 0, 0, reinterpret_cast<const gxo_transition*>(o.table.data()), 48, 4, 1, 36, 100);
  CHECK(ref != nullptr);
  b.Reset();
  gxo_vec_reset(ref);
  // env 0 walks up, along the top of the cliff and down into the goal; env 1 steps onto the cliff; env 2 pushes left
  std::vector<int32_t> obs(n), act(n);
  std::vector<float> rew(n);
  std::vector<double> orew(n);
  std::vector<uint8_t> done(n), odone(n);
  double ret0 = 0;
  for (int t = 0; t < 13; t++) {
    act = {t == 0 ? 3 : (t < 12 ? 2 : 1), 2, 0};
    b.Step(act.data(), obs.data(), rew.data(), done.data(), nullptr);
    gxo_vec_step(ref, act.data(), orew.data(), odone.data(), nullptr);
    for (uint64_t i = 0; i < n; i++) CHECK((double)rew[i] == orew[i] && done[i] == odone[i]);
    ret0 += rew[0];
    if (t == 0) CHECK(obs[0] == 24 && obs[1] == 37 && rew[1] == -100.0f && done[1] && obs[2] == 36 && !done[2]);
  }
  CHECK(obs[0] == 47 && done[0] && ret0 == -12.0);  // 12 steps at -1, the goal pays 0
  bool threw = false;
  act = {0, 4, 0};
  try { b.Step(act.data(), obs.data(), rew.data(), done.data(), nullptr); } catch (const std::out_of_range&) { threw = true; }
  CHECK(threw);
  gxo_vec_destroy(ref);
  std::printf("Batch(caller-supplied table: CliffWalking as P[s][a]) x 3: the source's known answers, equal to the oracle\n");
}

int main() {
  int count = 0;
  if (gx_device_count(&count) != GX_OK || count == 0) {
    // no GPU: the facade must fail loudly, not fall back
    bool threw = false;
    try { Context c(0); } catch (const std::exception& e) { threw = true; std::printf("no GPU: %s\n", e.what()); }
    return threw ? 77 : 1;
  }
  reference_unit_tests();
  registry_and_spaces();
  vector_env_against_oracle();
  wrappers();
  fp64_mode();
  supplied_draws();
  caller_supplied_table();
  std::printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
  return failures ? 1 : 0;
}
