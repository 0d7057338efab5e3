/*
 * gym_oracle.h -- CPU restatement of the reference's batched-step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * Parity status: the reference (C#, net8.0) cannot be compiled or run in the
 * build container and its own tests hold exactly one known-answer vector on
 * this path (Gymnasium.Tests/DummyEnvTests.cs:6-21, reproduced in
 * tests/test_oracle_kat.py).  For every other environment the reference's
 * tests assert only types, lengths and reward signs, so the numeric results
 * of this restatement are "parity unpinned" by reference outputs; they are
 * pinned instead by (i) hand-derived known answers read off the reference
 * source, and (ii) an independent numpy-float32 restatement
 * (oracle/np_oracle.py) that must agree bit-for-bit.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout, directory Gymnasium/).
 */
#ifndef GYM_ORACLE_H
#define GYM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Environment ids -- same numbering as include/gymx.h (kept in sync by
 * tests/test_abi.py). */
enum {
  GXO_CARTPOLE = 0,      /* Envs.CartPole.cs */
  GXO_MOUNTAINCAR = 1,   /* Envs.MountainCar.cs */
  GXO_MOUNTAINCAR_CONT = 2, /* Envs.MountainCarContinuous.cs */
  GXO_PENDULUM = 3,      /* Envs.Pendulum.cs */
  GXO_ACROBOT = 4,       /* Envs.Acrobot.cs */
  GXO_FROZENLAKE = 5,    /* Envs.FrozenLake.cs */
  GXO_TAXI = 6,          /* Envs.Taxi.cs */
  GXO_CLIFFWALKING = 7,  /* Envs.CliffWalking.cs */
  GXO_PONG = 8,          /* Envs.Pong.cs */
  GXO_DUMMY = 9,         /* DummyEnv.cs */
  GXO_BREAKOUT = 10,     /* Envs.Breakout.cs */
  GXO_SPACEINVADERS = 11, /* Envs.SpaceInvaders.cs */
  GXO_BLACKJACK = 12,    /* Envs.Blackjack.cs */
  GXO_TABLE = 13,        /* an env stepped from a transition table the caller supplies (gxo_vec_create_table): the
                            shape of FrozenLake.Step / CliffWalking.Step / Taxi.Step with the table for their bodies */
  GXO_NUM_ENVS = 14
};

/* Wrapper / engine flags. */
enum {
  GXO_AUTORESET = 1u,     /* engine extension: Reset() right after a done step */
  GXO_RECORD_STATS = 2u,  /* EnvWrapper.cs:23-58 RecordEpisodeStatistics */
  GXO_SLIPPERY = 4u,      /* extension: upstream FrozenLake is_slippery */
  GXO_TAXI_FULL = 8u,     /* extension: upstream Taxi-v3 dynamics */
  GXO_ACROBOT_RK4 = 16u,  /* extension: upstream Acrobot (RK4, fp64) */
  GXO_FP64 = 32u,         /* extension: the five classic-control envs computed and stored in binary64
                             (the engine's GX_MODE_FP64) */
  GXO_BLACKJACK_FULL = 64u /* extension: upstream Blackjack-v1 rules (the reference is a ten-step stub) */
};

#define GXO_PONG_FRAME_BYTES (210 * 160 * 3)

typedef struct gxo_vec gxo_vec;

typedef struct gxo_stats {
  int64_t episodes;
  double sum_return;
  int64_t sum_length;
  double sum_return_sq;
  double min_return;
  double max_return;
} gxo_stats;

/* ---- counter RNG (build-defined; the reference's System.Random is unseeded) */
void gxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint32_t gxo_rng_u32(uint64_t seed, uint64_t env_index, uint64_t tick, uint32_t stream, uint32_t draw);
double gxo_rng_uniform(uint64_t seed, uint64_t env_index, uint64_t tick, uint32_t stream, uint32_t draw);
int32_t gxo_rng_below(uint64_t seed, uint64_t env_index, uint64_t tick, uint32_t stream, uint32_t draw, int32_t n);

/* ---- space sampling (Spaces.Discrete.cs:18, Spaces.Box.cs:25-33) */
float gxo_box_sample(double u, float low, float high);

/* ---- batched engine shaped like VectorEnv.cs:10-35 */
/* One entry of a transition table, as include/gymx.h's gx_transition. */
typedef struct gxo_transition {
  int32_t next_state;
  float reward;
  int32_t done;
} gxo_transition;
/* GXO_TABLE: table[((s * n_actions + a) * n_outcomes + o)], n_outcomes 1 or 3 (equally likely, drawn per step from
 * the env's word of the tick like the slippery FrozenLake's), Reset() to start_state or, for -1, to a uniform draw
 * over the states, done also when the env's own step counter reaches step_limit (0: never).  An action outside
 * [0, n_actions) makes gxo_vec_step return -1.  The table is copied.  NULL when an argument is out of range. */
gxo_vec* gxo_vec_create_table(uint64_t num_envs, uint64_t seed, uint32_t flags, int32_t time_limit, uint64_t env_index_base,
                              const gxo_transition* table, int32_t n_states, int32_t n_actions, int32_t n_outcomes,
                              int32_t start_state, int32_t step_limit);
gxo_vec* gxo_vec_create(int env_id, uint64_t num_envs, uint64_t seed, uint32_t flags,
                        int32_t time_limit, const char* map_rows, uint64_t env_index_base);
void gxo_vec_destroy(gxo_vec* v);
uint64_t gxo_vec_num_envs(const gxo_vec* v);
int gxo_vec_obs_dim(const gxo_vec* v);      /* floats per env (Pong: bytes per env) */
int gxo_vec_n_real(const gxo_vec* v);       /* real-valued state components */
int gxo_vec_real_is_f64(const gxo_vec* v);
int gxo_vec_obs_is_f64(const gxo_vec* v);  /* classic control under GXO_FP64: double[obs_dim][N] */
int gxo_vec_n_int(const gxo_vec* v);        /* int32 state components (last one = steps) */
size_t gxo_vec_state_bytes(const gxo_vec* v);
uint64_t gxo_vec_tick(const gxo_vec* v);

void gxo_vec_reset(gxo_vec* v);
/* actions: int32[N] for Discrete action spaces, float[N] for Box.  Output
 * pointers may be NULL.  Returns 0, or -1 if an action is invalid where the
 * reference would throw (Envs.Acrobot.cs:45,51). */
int gxo_vec_step(gxo_vec* v, const void* actions, double* reward, uint8_t* done, uint8_t* truncated);
/* Same, with actions drawn from the env's action space using RNG stream 2;
 * actions_out (optional) receives them. */
int gxo_vec_step_random(gxo_vec* v, void* actions_out, double* reward, uint8_t* done, uint8_t* truncated);

/* SoA state image: n_real arrays of float (double for Pong) of length N, then
 * n_int arrays of int32 of length N.  Same layout as gx_vec_get_state. */
void gxo_vec_get_state(const gxo_vec* v, void* buf);
void gxo_vec_set_state(gxo_vec* v, const void* buf);
/* Pong: u[2][N] replaces the generator in each env's next ball reset (Envs.Pong.cs:58-59, 163-164), Breakout:
 * u[1][N] in its next release of the ball (Envs.Breakout.cs:100), during the next reset or step of the batch;
 * NULL withdraws.  -1 for another env. */
int gxo_vec_set_rng_draws(gxo_vec* v, const double* u);
/* Observation: classic control float[obs_dim][N]; tabular/Dummy int32[N];
 * Pong uint8[N][100800]. */
void gxo_vec_get_obs(const gxo_vec* v, void* buf);
void gxo_vec_stats(const gxo_vec* v, gxo_stats* out);
/* Per-env RecordEpisodeStatistics lists (EnvWrapper.cs:25-26). */
size_t gxo_vec_episode_count(const gxo_vec* v, uint64_t env);
void gxo_vec_episode_get(const gxo_vec* v, uint64_t env, size_t k, double* ret, int32_t* len);

/* Pong raster alone (Envs.Pong.cs:167-251) for frame tests. */
void gxo_pong_render(double ball_x, double ball_y, double left_y, double right_y,
                     int left_score, int right_score, uint8_t* frame);

/* Breakout raster alone (Envs.Breakout.cs:229-344).  bricks: bit row*18+col. */
void gxo_breakout_render(double ball_x, double ball_y, double paddle_x, int ball_stuck, int score, int lives,
                         const uint32_t bricks[4], uint8_t* frame);

/* SpaceInvaders raster alone (Envs.SpaceInvaders.cs:355-525) from one env's state image:
 * 13 reals, 9 ints in the order of gxo_vec_get_state. */
void gxo_invaders_render(const double* reals, const int32_t* ints, uint8_t* frame);

/* ---- CPU baseline: reference-shaped sequential loop over heap objects with a
 * virtual Step (VectorEnv.cs:28-34), single thread.  Runs `steps` batched
 * steps over `num_envs` envs with pre-drawn random actions and auto-reset,
 * returns elapsed seconds; *env_steps receives num_envs*steps. */
double gxo_bench_reference_shaped(int env_id, uint64_t num_envs, int steps, uint64_t seed,
                                  uint64_t* env_steps, double* checksum);

/* ---- CPU baseline on `threads` cores: contiguous env structs, no result lists. */
double gxo_bench_flat_threads(int env_id, uint64_t num_envs, int steps, int threads, uint64_t seed,
                              uint64_t* env_steps, double* checksum);

#ifdef __cplusplus
}
#endif
#endif
