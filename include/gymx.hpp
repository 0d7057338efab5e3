// gymx.hpp -- C++17 host facade over the C ABI of libgymx.so (gymx.h).
//
// Header-only.  It mirrors the API surface of the reference (MCGPPeters/Gym,
// C#) member for member so that code written against `Env<TState,TAction>`,
// `VectorEnv<,>`, `EnvRegistry`, `Box` / `Discrete` and the two statistics
// wrappers reads the same here; each type cites the reference file it stands
// in for.  Everything that computes is behind the C ABI on the GPU; this
// header only moves arguments and results.
//
//   gymx::GymnasiumRegistration::RegisterAll();
//   auto env = gymx::EnvRegistry::Make("CartPole-v1");        // a batch of one
//   auto vec = gymx::VectorEnv<gymx::CartPole>(1 << 20, {.autoreset = true, .record_stats = true});
//   auto states = vec.Reset();  auto results = vec.Step(actions);
#ifndef GYMX_HPP
#define GYMX_HPP

#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <tuple>
#include <vector>

#include "gymx.h"

namespace gymx {

// GX_ERR_INVALID maps to the ArgumentException family of the reference
// (std::invalid_argument; an Acrobot action outside {0,1,2} throws IndexOutOfRangeException
// there, Envs.Acrobot.cs:51, std::out_of_range here), everything else to std::runtime_error.
[[noreturn]] inline void throw_last(int code, bool index_error = false) {
  const std::string msg = gx_last_error();
  if (code == GX_ERR_INVALID && index_error) throw std::out_of_range(msg);
  if (code == GX_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
inline void check(int code, bool index_error = false) {
  if (code != GX_OK) throw_last(code, index_error);
}

// ------------------------------------------------------------------ spaces ---
// Spaces.Space.cs:8-13
class Space {
 public:
  virtual ~Space() = default;
  virtual std::vector<int> Shape() const = 0;
};

// Spaces.Box.cs:8-44
class Box : public Space {
 public:
  std::vector<float> Low, High;
  Box(std::vector<float> low, std::vector<float> high) : Low(std::move(low)), High(std::move(high)), rng_(std::random_device{}()) {
    if (Low.size() != High.size()) throw std::invalid_argument("Low and high must have the same length.");  // Spaces.Box.cs:18-19
  }
  int Dimension() const { return (int)Low.size(); }
  std::vector<int> Shape() const override { return {Dimension()}; }
  // (float)(NextDouble() * (High[i] - Low[i]) + Low[i]), Spaces.Box.cs:25-33
  std::vector<float> Sample() {
    std::uniform_real_distribution<double> u(0.0, 1.0);
    std::vector<float> out(Low.size());
    for (size_t i = 0; i < Low.size(); i++) out[i] = (float)(u(rng_) * (double)(float)(High[i] - Low[i]) + (double)Low[i]);
    return out;
  }
  bool Contains(const std::vector<float>& x) const {
    if (x.size() != Low.size()) return false;
    for (size_t i = 0; i < x.size(); i++)
      if (!(x[i] >= Low[i] && x[i] <= High[i])) return false;
    return true;
  }

 private:
  std::mt19937_64 rng_;  // the reference owns an unseeded System.Random per space (Spaces.Box.cs:14)
};

// Spaces.Discrete.cs:8-20
class Discrete : public Space {
 public:
  int N;
  explicit Discrete(int n) : N(n), rng_(std::random_device{}()) {}
  std::vector<int> Shape() const override { return {1}; }
  int Sample() { return (int)(rng_() % (uint64_t)N); }
  bool Contains(int x) const { return x >= 0 && x < N; }

 private:
  std::mt19937_64 rng_;
};

// --------------------------------------------------------------------- env ---
using Info = std::map<std::string, double>;  // IDictionary<string, object> of the reference; every value it stores is numeric or bool

template <class TState>
struct StepResult {  // the tuple of Env.cs:18
  TState state;
  double reward;
  bool done;
  Info info;
};

// Env<TState,TAction>, Env.cs:7-52
template <class TState, class TAction>
class Env {
 public:
  using State = TState;
  using Action = TAction;
  virtual ~Env() = default;
  virtual TState Reset() = 0;
  virtual StepResult<TState> Step(TAction action) = 0;
  virtual void Render(const std::string& mode = "human") { (void)mode; }
  virtual void Close() {}
  virtual void Seed(int seed) { (void)seed; }  // an empty virtual in the reference, Env.cs:33-36
};

// ------------------------------------------------------------ device pieces ---
class Context {
 public:
  explicit Context(int device = 0) { check(gx_ctx_create(device, &h_)); }
  ~Context() { gx_ctx_destroy(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  gx_ctx* handle() const { return h_; }
  void Synchronize() { check(gx_ctx_synchronize(h_)); }
  static std::shared_ptr<Context> Default(int device = 0) {
    static std::map<int, std::shared_ptr<Context>> cache;
    auto it = cache.find(device);
    if (it == cache.end()) it = cache.emplace(device, std::make_shared<Context>(device)).first;
    return it->second;
  }

 private:
  gx_ctx* h_ = nullptr;
};

struct BatchOptions {
  uint64_t seed = 0;
  bool autoreset = false;      // GX_AUTORESET (the reference never resets, VectorEnv.cs:28-34)
  bool record_stats = false;   // RecordEpisodeStatistics fused, EnvWrapper.cs:23-58
  int time_limit = 0;          // TimeLimit maxSteps fused, Wrappers.TimeLimit.cs:8; 0 = none
  int mode = GX_MODE_EXACT;
  uint32_t extension_flags = 0;  // GX_SLIPPERY, GX_TAXI_FULL, GX_ACROBOT_RK4, GX_BLACKJACK_FULL
  std::string map_rows;        // FrozenLake rows joined by '/', "" = default
  uint64_t env_index_base = 0;
  int device = 0;
  // GX_ENV_TABLE only: the caller's transitions [n_states][n_actions][n_outcomes] -- what FrozenLake's map
  // argument (Envs.FrozenLake.cs:23) generalises to -- and the env's Reset() state and own step limit
  std::vector<gx_transition> table;
  int n_states = 0, n_actions = 0, n_outcomes = 1;
  int start_state = 0;         // -1: a uniform draw over the states (Envs.Taxi.cs:20)
  int step_limit = 0;          // 0: none
};

// One batch behind gx_vec_*; owns nothing but the handle.
class Batch {
 public:
  Batch(int env_id, uint64_t n, const BatchOptions& o) : ctx_(Context::Default(o.device)) {
    gx_desc d{};
    d.struct_size = sizeof d;
    d.env_id = env_id;
    d.num_envs = n;
    d.seed = o.seed;
    d.env_index_base = o.env_index_base;
    d.mode = o.mode;
    d.flags = o.extension_flags | (o.autoreset ? (uint32_t)GX_AUTORESET : 0u) | (o.record_stats ? (uint32_t)GX_RECORD_STATS : 0u);
    d.time_limit = o.time_limit;
    d.map_rows = o.map_rows.empty() ? nullptr : o.map_rows.c_str();
    if (env_id == GX_ENV_TABLE) {
      d.table = o.table.data();
      d.table_len = o.table.size();
      d.n_states = o.n_states;
      d.n_actions = o.n_actions;
      d.n_outcomes = o.n_outcomes;
      d.start_state = o.start_state;
      d.step_limit = o.step_limit;
    }
    check(gx_vec_create(ctx_->handle(), &d, &h_));
    check(gx_vec_info(h_, &info_));
  }
  ~Batch() { gx_vec_destroy(h_); }
  Batch(const Batch&) = delete;
  Batch& operator=(const Batch&) = delete;

  gx_vec* handle() const { return h_; }
  const gx_info& info() const { return info_; }
  uint64_t size() const { return info_.num_envs; }
  void Reset() { check(gx_vec_reset(h_)); }
  // host path: actions in, whatever is asked for out (VectorEnv.Step, VectorEnv.cs:28-34)
  void Step(const void* actions, void* obs, void* reward, uint8_t* done, uint8_t* truncated) {
    const bool throws_on_action = info_.env_id == GX_ENV_ACROBOT || info_.env_id == GX_ENV_TABLE;
    check(gx_vec_step_async(h_, actions, obs, reward, done, truncated), throws_on_action);
    check(gx_vec_wait(h_), throws_on_action);
  }
  void GetObs(void* obs) { check(gx_vec_get_obs(h_, obs, info_.obs_bytes)); }
  std::vector<uint8_t> GetState() {
    std::vector<uint8_t> s(info_.state_bytes);
    check(gx_vec_get_state(h_, s.data(), s.size()));
    return s;
  }
  void SetState(const std::vector<uint8_t>& s) { check(gx_vec_set_state(h_, s.data(), s.size())); }
  // Pong u[2][NumEnvs]: the two NextDouble() of each env's next ball reset (Envs.Pong.cs:163-164); Breakout
  // u[1][NumEnvs]: the one of its next release of the ball (Envs.Breakout.cs:100)
  void SetRngDraws(const std::vector<double>& u) { check(gx_vec_set_rng_draws(h_, u.empty() ? nullptr : u.data(), u.size())); }
  // zero-copy path
  gx_view View() { gx_view v{}; check(gx_vec_device_view(h_, &v)); return v; }
  void StepDevice(const void* d_actions, void* stream = nullptr) { check(gx_vec_step_device(h_, d_actions, stream)); }
  void StepRandom(void* stream = nullptr) { check(gx_vec_step_random(h_, stream)); }
  void Rollout(int k, int policy = GX_POLICY_RANDOM, const void* d_actions = nullptr, void* stream = nullptr) {
    check(gx_vec_rollout(h_, k, policy, d_actions, stream));
  }
  gx_stats Stats() { gx_stats s{}; check(gx_vec_stats(h_, &s)); return s; }
  void Synchronize() { ctx_->Synchronize(); }

 private:
  std::shared_ptr<Context> ctx_;
  gx_vec* h_ = nullptr;
  gx_info info_{};
};

// Per-environment traits: id, the reference's TState / TAction, how a state is read out of the
// SoA observation, the spaces.  One struct per reference class.
struct CartPole {  // Envs.CartPole.cs:10-83
  static constexpr int kEnvId = GX_ENV_CARTPOLE;
  static constexpr const char* kId = "CartPole-v1";
  using State = std::array<float, 4>;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(2); }
  static Box ObservationSpace() { return Box({-2.4f, -3.0f, -0.2095f, -3.0f}, {2.4f, 3.0f, 0.2095f, 3.0f}); }  // :13-16
};
struct MountainCar {  // Envs.MountainCar.cs:10-58
  static constexpr int kEnvId = GX_ENV_MOUNTAINCAR;
  static constexpr const char* kId = "MountainCar-v0";
  using State = std::array<float, 2>;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(3); }
  static Box ObservationSpace() { return Box({-1.2f, -0.07f}, {0.6f, 0.07f}); }
};
struct MountainCarContinuous {  // Envs.MountainCarContinuous.cs:10-57
  static constexpr int kEnvId = GX_ENV_MOUNTAINCAR_CONT;
  static constexpr const char* kId = "MountainCarContinuous-v0";
  using State = std::array<float, 2>;
  using Action = float;
  static Box ActionSpace() { return Box({-1.0f}, {1.0f}); }
  static Box ObservationSpace() { return Box({-1.2f, -0.07f}, {0.6f, 0.07f}); }
};
struct Pendulum {  // Envs.Pendulum.cs:10-68
  static constexpr int kEnvId = GX_ENV_PENDULUM;
  static constexpr const char* kId = "Pendulum-v1";
  using State = std::array<float, 3>;
  using Action = float;
  static Box ActionSpace() { return Box({-2.0f}, {2.0f}); }
  static Box ObservationSpace() { return Box({-1.0f, -1.0f, -8.0f}, {1.0f, 1.0f, 8.0f}); }
};
struct Acrobot {  // Envs.Acrobot.cs:10-99
  static constexpr int kEnvId = GX_ENV_ACROBOT;
  static constexpr const char* kId = "Acrobot-v1";
  using State = std::array<float, 6>;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(3); }
  static Box ObservationSpace() { return Box({-1, -1, -1, -1, -6, -6}, {1, 1, 1, 1, 6, 6}); }  // :13-16
};
struct FrozenLake {  // Envs.FrozenLake.cs:10-88
  static constexpr int kEnvId = GX_ENV_FROZENLAKE;
  static constexpr const char* kId = "FrozenLake-v1";
  using State = int;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(4); }
  static Discrete ObservationSpace() { return Discrete(16); }
};
struct Taxi {  // Envs.Taxi.cs:10-41
  static constexpr int kEnvId = GX_ENV_TAXI;
  static constexpr const char* kId = "Taxi-v3";
  using State = int;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(6); }
  static Discrete ObservationSpace() { return Discrete(500); }
};
struct CliffWalking {  // Envs.CliffWalking.cs:10-77
  static constexpr int kEnvId = GX_ENV_CLIFFWALKING;
  static constexpr const char* kId = "CliffWalking-v0";
  using State = int;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(4); }
  static Discrete ObservationSpace() { return Discrete(48); }
};
struct Blackjack {  // Envs.Blackjack.cs:10-48
  static constexpr int kEnvId = GX_ENV_BLACKJACK;
  static constexpr const char* kId = "Blackjack-v1";
  using State = std::tuple<int, int, bool>;  // (playerSum, dealerCard, usableAce)
  using Action = int;
  static Discrete ActionSpace() { return Discrete(2); }               // 0: stick, 1: hit
  static Discrete ObservationSpace() { return Discrete(32 * 11 * 2); }
};
struct DummyEnv {  // DummyEnv.cs:10-38
  static constexpr int kEnvId = GX_ENV_DUMMY;
  static constexpr const char* kId = "Dummy";
  using State = int;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(3); }
  static Discrete ObservationSpace() { return Discrete(20); }
};
constexpr int kFrameBytes = 210 * 160 * 3;
inline Box FrameSpace() { return Box(std::vector<float>(kFrameBytes, 0.0f), std::vector<float>(kFrameBytes, 255.0f)); }
struct Pong {  // Envs.Pong.cs:12-266
  static constexpr int kEnvId = GX_ENV_PONG;
  static constexpr const char* kId = "Pong-v4";
  using State = std::vector<uint8_t>;  // byte[]: the 210x160x3 frame
  using Action = int;
  static Discrete ActionSpace() { return Discrete(4); }
  static Box ObservationSpace() { return FrameSpace(); }
};
struct Breakout {  // Envs.Breakout.cs:12-360
  static constexpr int kEnvId = GX_ENV_BREAKOUT;
  static constexpr const char* kId = "Breakout-v4";
  using State = std::vector<uint8_t>;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(4); }
  static Box ObservationSpace() { return FrameSpace(); }
};
struct SpaceInvaders {  // Envs.SpaceInvaders.cs:12-541
  static constexpr int kEnvId = GX_ENV_SPACEINVADERS;
  static constexpr const char* kId = "SpaceInvaders-v4";
  using State = std::vector<uint8_t>;
  using Action = int;
  static Discrete ActionSpace() { return Discrete(6); }
  static Box ObservationSpace() { return FrameSpace(); }
};

namespace detail {
// env i of the observation the ABI returns: float[obs_dim][n] (SoA), int32[n], or uint8[n][100800]
template <class State>
struct StateOf;
template <size_t D>
struct StateOf<std::array<float, D>> {
  using Elem = float;
  static std::array<float, D> get(const std::vector<float>& obs, uint64_t n, uint64_t i) {
    std::array<float, D> s;
    for (size_t c = 0; c < D; c++) s[c] = obs[c * n + i];
    return s;
  }
};
template <>
struct StateOf<int> {
  using Elem = int32_t;
  static int get(const std::vector<int32_t>& obs, uint64_t, uint64_t i) { return obs[i]; }
};
template <>
struct StateOf<std::tuple<int, int, bool>> {  // the device holds the index of Discrete(32*11*2)
  using Elem = int32_t;
  static std::tuple<int, int, bool> get(const std::vector<int32_t>& obs, uint64_t, uint64_t i) {
    const int32_t v = obs[i];
    return {v / 22, (v / 2) % 11, (v & 1) != 0};
  }
};
template <>
struct StateOf<std::vector<uint8_t>> {
  using Elem = uint8_t;
  static std::vector<uint8_t> get(const std::vector<uint8_t>& obs, uint64_t, uint64_t i) {
    return std::vector<uint8_t>(obs.begin() + (ptrdiff_t)(i * kFrameBytes), obs.begin() + (ptrdiff_t)((i + 1) * kFrameBytes));
  }
};

// info dictionaries of the frame envs (Envs.Pong.cs:148-154, Envs.Breakout.cs:203-210,
// Envs.SpaceInvaders.cs:181-187), read from the state image
inline int popcount64(uint64_t x) { int c = 0; for (; x; x &= x - 1) c++; return c; }
inline Info frame_info(int env_id, const std::vector<uint8_t>& image, const gx_info& in, uint64_t i) {
  const uint64_t n = in.num_envs;
  const double* re = reinterpret_cast<const double*>(image.data());
  const int32_t* it = reinterpret_cast<const int32_t*>(image.data() + (size_t)in.n_real * n * 8);
  auto R = [&](int c) { return re[(uint64_t)c * n + i]; };
  auto I = [&](int c) { return (double)it[(uint64_t)c * n + i]; };
  auto U = [&](int c) { return (uint64_t)(uint32_t)it[(uint64_t)c * n + i]; };
  Info info;
  if (env_id == GX_ENV_PONG) {
    info = {{"left_score", I(0)}, {"right_score", I(1)}, {"ball_x", R(0)}, {"ball_y", R(1)}};
  } else if (env_id == GX_ENV_BREAKOUT) {
    const int bricks = popcount64(U(3) | (U(4) << 32)) + popcount64(U(5) | ((U(6) & 0xFFFu) << 32));
    info = {{"score", I(0)}, {"lives", I(1)}, {"ball_x", R(0)}, {"ball_y", R(1)}, {"bricks_remaining", (double)bricks}};
  } else if (env_id == GX_ENV_SPACEINVADERS) {
    info = {{"score", I(0)}, {"lives", I(1)}, {"aliens_remaining", (double)popcount64((U(6) | (U(7) << 32)) & ((1ull << 55) - 1))},
            {"player_x", R(0)}};
  }
  return info;
}
}  // namespace detail

// VectorEnv<TState,TAction>, VectorEnv.cs:10-35, as one device batch of E.
template <class E>
class VectorEnv {
 public:
  using State = typename E::State;
  using Action = typename E::Action;
  using Elem = typename detail::StateOf<State>::Elem;

  explicit VectorEnv(uint64_t num_envs, const BatchOptions& o = {}) : batch_(E::kEnvId, num_envs, o), opts_(o) {
    // the reference's state types are float-valued; a binary64 batch is driven through Batch directly
    if (batch_.info().obs_dtype == GX_F64)
      throw std::invalid_argument("VectorEnv materialises float states; use gymx::Batch for GX_MODE_FP64");
    obs_.resize(batch_.info().obs_bytes / sizeof(Elem));
    reward_f_.resize(num_envs);
    reward_d_.resize(num_envs);
    done_.resize(num_envs);
    trunc_.resize(num_envs);
  }
  uint64_t NumEnvs() const { return batch_.size(); }  // VectorEnv.cs:13
  Batch& Device() { return batch_; }

  std::vector<State> Reset() {  // VectorEnv.cs:20-26
    batch_.Reset();
    batch_.GetObs(obs_.data());
    return states();
  }
  std::vector<StepResult<State>> Step(const std::vector<Action>& actions) {  // VectorEnv.cs:28-34
    if (actions.size() != NumEnvs()) throw std::invalid_argument("Step needs one action per env");
    static_assert(sizeof(Action) == 4, "actions are int32 or float");
    const bool f64 = batch_.info().reward_dtype == GX_F64;
    batch_.Step(actions.data(), obs_.data(), f64 ? (void*)reward_d_.data() : (void*)reward_f_.data(), done_.data(),
                opts_.time_limit > 0 ? trunc_.data() : nullptr);
    std::vector<State> st = states();
    std::vector<uint8_t> image;
    const bool frames = batch_.info().obs_dtype == GX_U8;
    if (frames) image = batch_.GetState();
    std::vector<StepResult<State>> out;
    out.reserve(NumEnvs());
    for (uint64_t i = 0; i < NumEnvs(); i++) {
      Info info = frames ? detail::frame_info(E::kEnvId, image, batch_.info(), i) : Info{};
      if (opts_.time_limit > 0 && trunc_[i]) info["TimeLimit.truncated"] = 1.0;  // Wrappers.TimeLimit.cs:30
      out.push_back({std::move(st[i]), f64 ? reward_d_[i] : (double)reward_f_[i], done_[i] != 0, std::move(info)});
    }
    return out;
  }

 private:
  std::vector<State> states() const {
    std::vector<State> s;
    s.reserve(NumEnvs());
    for (uint64_t i = 0; i < NumEnvs(); i++) s.push_back(detail::StateOf<State>::get(obs_, NumEnvs(), i));
    return s;
  }
  Batch batch_;
  BatchOptions opts_;
  std::vector<Elem> obs_;
  std::vector<float> reward_f_;
  std::vector<double> reward_d_;
  std::vector<uint8_t> done_, trunc_;
};

// One env of the reference = a batch of one.
template <class E>
class DeviceEnv : public Env<typename E::State, typename E::Action> {
 public:
  using State = typename E::State;
  using Action = typename E::Action;
  explicit DeviceEnv(const BatchOptions& o = {}) : vec_(1, o), ActionSpace(E::ActionSpace()), ObservationSpace(E::ObservationSpace()) {}
  State Reset() override { return vec_.Reset()[0]; }
  StepResult<State> Step(Action action) override { return std::move(vec_.Step({action})[0]); }

 private:
  VectorEnv<E> vec_;

 public:
  decltype(E::ActionSpace()) ActionSpace;
  decltype(E::ObservationSpace()) ObservationSpace;
};

// ---------------------------------------------------------------- wrappers ---
// EnvWrapper.cs:8-21
template <class TState, class TAction>
class EnvWrapper : public Env<TState, TAction> {
 public:
  explicit EnvWrapper(std::shared_ptr<Env<TState, TAction>> env) : Inner(std::move(env)) {}
  TState Reset() override { return Inner->Reset(); }
  StepResult<TState> Step(TAction a) override { return Inner->Step(a); }
  void Render(const std::string& mode = "human") override { Inner->Render(mode); }
  void Close() override { Inner->Close(); }
  std::shared_ptr<Env<TState, TAction>> Inner;
};

// EnvWrapper.cs:23-58
template <class TState, class TAction>
class RecordEpisodeStatistics : public EnvWrapper<TState, TAction> {
 public:
  using EnvWrapper<TState, TAction>::EnvWrapper;
  std::vector<double> EpisodeRewards;
  std::vector<int> EpisodeLengths;
  TState Reset() override {
    if (currentLength_ > 0) {
      EpisodeRewards.push_back(currentReward_);
      EpisodeLengths.push_back(currentLength_);
    }
    currentReward_ = 0.0;
    currentLength_ = 0;
    return this->Inner->Reset();
  }
  StepResult<TState> Step(TAction a) override {
    StepResult<TState> r = this->Inner->Step(a);
    currentReward_ += r.reward;
    currentLength_++;
    if (r.done) {
      EpisodeRewards.push_back(currentReward_);
      EpisodeLengths.push_back(currentLength_);
      currentReward_ = 0.0;
      currentLength_ = 0;
    }
    return r;
  }

 private:
  double currentReward_ = 0.0;
  int currentLength_ = 0;
};

// Wrappers.TimeLimit.cs:6-34
template <class TState, class TAction>
class TimeLimit : public EnvWrapper<TState, TAction> {
 public:
  TimeLimit(std::shared_ptr<Env<TState, TAction>> env, int maxSteps) : EnvWrapper<TState, TAction>(std::move(env)), maxSteps_(maxSteps) {}
  TState Reset() override {
    steps_ = 0;
    return this->Inner->Reset();
  }
  StepResult<TState> Step(TAction a) override {
    StepResult<TState> r = this->Inner->Step(a);
    steps_++;
    if (steps_ >= maxSteps_) {
      r.done = true;
      r.info["TimeLimit.truncated"] = 1.0;
    }
    return r;
  }

 private:
  int maxSteps_, steps_ = 0;
};

// ---------------------------------------------------------------- registry ---
// EnvRegistry.cs:8-35: ids to factories of untyped envs (the reference stores Func<object>).
class EnvRegistry {
 public:
  using Factory = std::function<std::shared_ptr<void>()>;
  static void Register(const std::string& id, Factory f) { registry()[id] = std::move(f); }
  static std::shared_ptr<void> Make(const std::string& id) {
    auto it = registry().find(id);
    if (it == registry().end()) throw std::invalid_argument("Environment id '" + id + "' not found.");  // EnvRegistry.cs:28-31
    return it->second();
  }
  template <class T>
  static std::shared_ptr<T> Make(const std::string& id) { return std::static_pointer_cast<T>(Make(id)); }
  static std::vector<std::string> List() {
    std::vector<std::string> ids;
    for (auto& kv : registry()) ids.push_back(kv.first);
    return ids;
  }

 private:
  static std::map<std::string, Factory>& registry() {
    static std::map<std::string, Factory> r;
    return r;
  }
};

// GymnasiumRegistration.cs:8-30, the ids with a device kernel
struct GymnasiumRegistration {
  template <class E>
  static void Add() { EnvRegistry::Register(E::kId, [] { return std::static_pointer_cast<void>(std::make_shared<DeviceEnv<E>>()); }); }
  static void RegisterAll() {
    Add<CartPole>(); Add<MountainCar>(); Add<MountainCarContinuous>(); Add<Acrobot>(); Add<Pendulum>(); Add<FrozenLake>();
    Add<Taxi>(); Add<CliffWalking>(); Add<Blackjack>(); Add<Pong>(); Add<Breakout>(); Add<SpaceInvaders>();
  }
};

}  // namespace gymx
#endif
