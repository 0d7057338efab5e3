/*
 * gymx.h -- C ABI of the MI355X batched environment engine (libgymx.so).
 *
 * The reference (MCGPPeters/Gym, C#) has no native interface: its batched
 * path is `VectorEnv<TState,TAction>` looping over `Env<,>.Step`
 * (Gymnasium/VectorEnv.cs:10-35).  These entry points are what a C#
 * `[LibraryImport]` binding for a device-resident replacement of that class
 * calls; each one names the reference member it stands in for.
 *
 * Conventions
 *  - every function returns a gx_status; nothing throws or aborts across the
 *    boundary; gx_last_error() gives the thread-local message of the last
 *    failure on the calling thread;
 *  - handles are externally synchronised, like every object of the reference
 *    (no locks anywhere in Gymnasium/); distinct handles may be driven from
 *    distinct threads;
 *  - the library owns all device memory behind a handle; device pointers
 *    handed out by gx_vec_device_view stay valid until gx_vec_destroy;
 *  - "SoA image" = component arrays one after another, each of num_envs
 *    elements: first the n_real real-valued components (float, or double
 *    when real_is_f64), then the n_int int32 components (the last one is the
 *    per-env step counter).
 */
#ifndef GYMX_H
#define GYMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GX_VERSION 0x00010000u

typedef enum gx_status {
  GX_OK = 0,
  GX_ERR_INVALID = 1,      /* bad argument / invalid action (the reference throws) */
  GX_ERR_HIP = 2,          /* a HIP runtime call failed */
  GX_ERR_OOM = 3,          /* device or host allocation failed */
  GX_ERR_UNSUPPORTED = 4,  /* valid request this build cannot serve */
  GX_ERR_RCCL = 5          /* an RCCL call failed */
} gx_status;

/* Environment ids: the registrations of GymnasiumRegistration.cs:10-28 that
 * are on the device path, plus DummyEnv (DummyEnv.cs) for the reference's
 * only known-answer test. */
typedef enum gx_env_id {
  GX_ENV_CARTPOLE = 0,         /* "CartPole-v1"              Envs.CartPole.cs */
  GX_ENV_MOUNTAINCAR = 1,      /* "MountainCar-v0"           Envs.MountainCar.cs */
  GX_ENV_MOUNTAINCAR_CONT = 2, /* "MountainCarContinuous-v0" Envs.MountainCarContinuous.cs */
  GX_ENV_PENDULUM = 3,         /* "Pendulum-v1"              Envs.Pendulum.cs */
  GX_ENV_ACROBOT = 4,          /* "Acrobot-v1"               Envs.Acrobot.cs */
  GX_ENV_FROZENLAKE = 5,       /* "FrozenLake-v1"            Envs.FrozenLake.cs */
  GX_ENV_TAXI = 6,             /* "Taxi-v3"                  Envs.Taxi.cs */
  GX_ENV_CLIFFWALKING = 7,     /* "CliffWalking-v0"          Envs.CliffWalking.cs */
  GX_ENV_PONG = 8,             /* "Pong-v4"                  Envs.Pong.cs */
  GX_ENV_DUMMY = 9,            /* DummyEnv.cs */
  GX_ENV_BREAKOUT = 10,        /* "Breakout-v4"              Envs.Breakout.cs */
  GX_ENV_SPACEINVADERS = 11,   /* "SpaceInvaders-v4"         Envs.SpaceInvaders.cs */
  GX_ENV_BLACKJACK = 12,       /* "Blackjack-v1"             Envs.Blackjack.cs */
  GX_ENV_TABLE = 13,           /* a table-driven env whose transitions the caller supplies (gx_desc.table): what
                                  FrozenLake's map argument (Envs.FrozenLake.cs:23) generalises to */
  GX_ENV_COUNT = 14
} gx_env_id;

/* Numeric mode of the continuous-control kernels. */
typedef enum gx_mode {
  GX_MODE_EXACT = 0,  /* operation-for-operation binary32 as the C# source; sin/cos/atan2 in
                         binary64 rounded once, no contraction (default) */
  GX_MODE_FAST32 = 1, /* binary32 device trig, fused multiply-add allowed */
  GX_MODE_FP64 = 2    /* extension: the same formulas with every float read as double (state rows,
                         observation and arithmetic in binary64; actions and rewards keep their
                         types).  The reference has no such form: parity is against a CPU build of
                         the same formulas, to 1e-9 relative per step.  Ignored by the tabular and
                         frame envs and by GX_ACROBOT_RK4, which is binary64 already. */
} gx_mode;

typedef enum gx_flags {
  GX_AUTORESET = 1u,    /* Reset() an env in the same step that returns done (the reference's
                           VectorEnv.Step, VectorEnv.cs:28-34, never resets; default off) */
  GX_RECORD_STATS = 2u, /* fuse RecordEpisodeStatistics (EnvWrapper.cs:23-58) into the step */
  GX_SLIPPERY = 4u,     /* extension: upstream FrozenLake is_slippery (the reference is deterministic) */
  GX_TAXI_FULL = 8u,    /* extension: upstream Taxi-v3 transitions (the reference is a stub) */
  GX_ACROBOT_RK4 = 16u, /* extension: upstream Acrobot, classical RK4 in binary64
                           (the reference integrates with semi-implicit Euler in binary32) */
  GX_BLACKJACK_FULL = 64u /* extension: upstream Blackjack-v1 rules, infinite deck, natural = False, sab = False
                           (the reference's Step is a ten-step stub, Envs.Blackjack.cs:34-39) */
} gx_flags;

typedef enum gx_policy {
  GX_POLICY_DEVICE_ACTIONS = 0, /* actions read from a device array */
  GX_POLICY_RANDOM = 1          /* ActionSpace.Sample() per env per step from the counter RNG
                                   (Spaces.Discrete.cs:18, Spaces.Box.cs:25-33) */
} gx_policy;

typedef enum gx_dtype { GX_F32 = 0, GX_F64 = 1, GX_I32 = 2, GX_U8 = 3 } gx_dtype;

typedef struct gx_ctx gx_ctx;
typedef struct gx_vec gx_vec;
typedef struct gx_comm gx_comm;

/* One entry of a caller-supplied transition table (GX_ENV_TABLE): taking action a in state s moves the env
 * to next_state, pays reward and ends the episode when done is non-zero -- the three things the bodies of
 * FrozenLake.Step, CliffWalking.Step and Taxi.Step compute (Envs.FrozenLake.cs:52-69, Envs.CliffWalking.cs:36-54,
 * Envs.Taxi.cs:25-33). */
typedef struct gx_transition {
  int32_t next_state; /* 0 .. n_states-1 */
  float reward;
  int32_t done;       /* non-zero: terminal */
} gx_transition;

/* Construction parameters of a batch (the reference has no configuration
 * beyond constructor arguments: Envs.FrozenLake.cs:23 map, TimeLimit maxSteps
 * Wrappers.TimeLimit.cs:11). */
typedef struct gx_desc {
  uint32_t struct_size;    /* sizeof(gx_desc), for forward compatibility */
  int32_t env_id;          /* gx_env_id */
  uint64_t num_envs;       /* VectorEnv.NumEnvs, VectorEnv.cs:13 */
  uint64_t seed;           /* key of the per-env counter RNG */
  uint64_t env_index_base; /* global index of env 0: RNG streams are keyed by the global
                              index so results do not depend on how a batch is sharded */
  int32_t mode;            /* gx_mode */
  uint32_t flags;          /* gx_flags */
  int32_t time_limit;      /* TimeLimit maxSteps (Wrappers.TimeLimit.cs:8); 0 = no wrapper */
  int32_t reserved;
  const char* map_rows;    /* FrozenLake rows joined by '/', NULL = default 4x4 map
                              (Envs.FrozenLake.cs:25-30), "8x8" = upstream's 8x8 map */
  /* ---- GX_ENV_TABLE only (ignored, and may be left out of struct_size, for every other env) ----
   * The library compiles the table into the packed form its built-in table-driven envs use and stages it in LDS
   * (up to 64 KiB packed: (n_actions + 1) * n_states * n_outcomes + n_rewards words; larger tables are read through
   * L2).  Copied at gx_vec_create; the caller's array is not kept. */
  const gx_transition* table; /* [n_states][n_actions][n_outcomes], entry ((s * n_actions + a) * n_outcomes + o) */
  uint64_t table_len;      /* n_states * n_actions * n_outcomes (checked) */
  int32_t n_states;        /* 1 .. 65536; ObservationSpace is Discrete(n_states) */
  int32_t n_actions;       /* 1 .. 255; ActionSpace is Discrete(n_actions).  An action outside it is an error as where the
                              reference throws: GX_ERR_INVALID from gx_vec_step_async before any env is stepped; on the
                              zero-copy path from the gx_vec_wait that follows (the env it was given to takes action 0) */
  int32_t n_outcomes;      /* 1 (0 reads as 1): deterministic.  3: every step draws one of three equally likely outcomes,
                              independent of the action, from the env's word of the tick (how GX_SLIPPERY compiles
                              FrozenLake: outcome o of action a there is action (a + 3 + o) mod 4) */
  int32_t start_state;     /* what Reset() sets: this state (Envs.FrozenLake.cs:47, Envs.CliffWalking.cs:31), or -1: a
                              uniform draw over the states (Envs.Taxi.cs:20, Next(500)) */
  int32_t step_limit;      /* the env's own `steps >= limit` term of done (Envs.FrozenLake.cs:65, Envs.Taxi.cs:29);
                              0 = none.  Independent of time_limit, the TimeLimit wrapper's */
  int32_t reserved2;
  /* Rewards: at most 256 distinct values, finite.  With GX_RECORD_STATS they must be integers of magnitude at most
   * 2^20 (every table-driven env of the reference pays integers; the fused statistics add them exactly), else
   * GX_ERR_UNSUPPORTED. */
} gx_desc;
/* sizeof(gx_desc) of callers built before the GX_ENV_TABLE fields existed; still accepted for the other envs */
#define GX_DESC_SIZE_V1 56u

/* Static facts about a batch (what an agent reads from ActionSpace /
 * ObservationSpace, e.g. Envs.CartPole.cs:12-16). */
typedef struct gx_info {
  uint64_t num_envs;
  int32_t env_id;
  int32_t n_real;          /* real-valued state components */
  int32_t real_dtype;      /* GX_F32 or GX_F64 */
  int32_t n_int;           /* int32 state components, the last is the step counter */
  int32_t obs_dim;         /* elements per env in the observation */
  int32_t obs_dtype;       /* GX_F32 (classic control; GX_F64 under GX_MODE_FP64), GX_I32 (tabular),
                              GX_U8 (Pong, Breakout, SpaceInvaders) */
  int32_t action_dtype;    /* GX_I32 (Discrete) or GX_F32 (Box of dimension 1) */
  int32_t action_n;        /* Discrete.N, or 0 for Box */
  float action_low, action_high; /* Box bounds when action_dtype == GX_F32 */
  int32_t reward_dtype;    /* GX_F32, or GX_F64 where the reference computes a double */
  int32_t obs_is_state;    /* 1: the observation is the real state image itself */
  uint64_t state_bytes;    /* size of the SoA image for get/set_state */
  uint64_t obs_bytes;      /* size of a host observation buffer */
} gx_info;

/* Device pointers for on-GPU agents (zero-copy path). */
typedef struct gx_view {
  uint64_t num_envs;
  uint64_t stride;    /* elements between consecutive component arrays: a multiple of 4, >= num_envs, and for
                         large batches more than num_envs rounded up (rows are kept off power-of-two spacings) */
  void* state_real;   /* [n_real][stride] float or double; for reading -- a state is written with gx_vec_set_state
                         (Acrobot in GX_MODE_EXACT keeps two rows of derived angles behind these, which a write
                         that bypasses gx_vec_set_state would leave stale) */
  int32_t* state_int; /* [n_int][stride] */
  void* obs;          /* classic: == state_real (or float [obs_dim][stride] when it is derived);
                         tabular: == state_int; Pong, Breakout, SpaceInvaders: uint8 [num_envs][100800] */
  void* reward;       /* [num_envs] float or double (gx_info.reward_dtype) */
  uint8_t* done;      /* [num_envs] */
  uint8_t* truncated; /* [num_envs], NULL unless time_limit > 0 */
} gx_view;

/* Aggregate of RecordEpisodeStatistics.EpisodeRewards / EpisodeLengths
 * (EnvWrapper.cs:25-26) over all envs of a batch. */
typedef struct gx_stats {
  int64_t episodes;
  double sum_return;
  int64_t sum_length;
  double sum_return_sq;
  double min_return; /* +inf when episodes == 0 */
  double max_return; /* -inf when episodes == 0 */
} gx_stats;

/* ---- library ---------------------------------------------------------- */
uint32_t gx_version(void);
const char* gx_last_error(void);
int gx_device_count(int* count);

/* ---- context: one per GPU -------------------------------------------- */
int gx_ctx_create(int device, gx_ctx** out);
/* Fails with GX_ERR_INVALID (and frees nothing) while a batch or communicator created on the
 * context is alive: gx_vec_destroy synchronises the context's streams, so destroy batches first. */
int gx_ctx_destroy(gx_ctx* ctx);
int gx_ctx_synchronize(gx_ctx* ctx);
/* The HIP stream (hipStream_t) step kernels are launched on when the caller
 * passes no stream of its own. */
int gx_ctx_stream(gx_ctx* ctx, void** stream);

/* ---- batch: replaces `new VectorEnv<,>(List<Env<,>>)`, VectorEnv.cs:15 -- */
int gx_vec_create(gx_ctx* ctx, const gx_desc* desc, gx_vec** out);
int gx_vec_destroy(gx_vec* vec); /* Env.Close(), Env.cs:28 */
int gx_vec_info(const gx_vec* vec, gx_info* out);

/* VectorEnv.Reset(), VectorEnv.cs:20-26 (through RecordEpisodeStatistics.Reset
 * EnvWrapper.cs:32-42 and TimeLimit.Reset Wrappers.TimeLimit.cs:17-21 when fused). */
int gx_vec_reset(gx_vec* vec);

/* The reference keeps state private and has no setter; these exist for the
 * replay harness and double as checkpoint/resume.  Synchronous.  The step
 * counter of the image is also TimeLimit's counter and, with GX_AUTORESET, the
 * length of the open episode; the return collected so far is kept.
 * Components of the SoA image (reals | ints), named after the reference's fields:
 *   CartPole   x, xDot, theta, thetaDot | steps          MountainCar(+Cont)  position, velocity | steps
 *   Pendulum   cos, sin, thetaDot | steps                Acrobot  cos1, sin1, cos2, sin2, thetaDot1, thetaDot2 | steps
 *   Acrobot+RK4 (f64) theta1, theta2, dtheta1, dtheta2 | steps   (set_state: finite values, |theta| <= 1024,
 *                  else GX_ERR_INVALID; the next step wraps the angles to [-pi, pi])
 *   (the five above hold doubles under GX_MODE_FP64)
 *   tabular, Dummy, GX_ENV_TABLE   - | state, steps
 *   Blackjack        - | (playerSum*11 + dealerCard)*2 + usableAce (the index of ObservationSpace Discrete(704),
 *                  Envs.Blackjack.cs:13), dealer's face-down cards (raw sum | "an ace among them" << 8; 0 unless
 *                  GX_BLACKJACK_FULL), steps
 *   Pong (f64)     ballX, ballY, ballVelX, ballVelY, leftPaddleY, rightPaddleY | leftScore, rightScore, steps
 *   Breakout (f64) ballX, ballY, ballVelX, ballVelY, paddleX | score, lives, ballStuck,
 *                  bricks[0..3] (bit row*18+col of word (row*18+col)/32 set = brick present), steps
 *   SpaceInvaders (f64) playerX, alienFormationX, alienFormationY, playerBullets[0..2] (x, y),
 *                  alienBullets[0..1] (x, y) | score, lives, frameCounter, alienMoveRight, playerBullets.Count,
 *                  alienBullets.Count, aliens bits 0..31, aliens bits 32..54 (bit row*11+col), steps.
 *                  The reference's alien bullet list is unbounded; a game never holds more than one
 *                  (fired every 60 steps at most, gone within 40), the image holds two. */
int gx_vec_get_state(gx_vec* vec, void* host_soa, size_t bytes);
int gx_vec_set_state(gx_vec* vec, const void* host_soa, size_t bytes);
/* The draws the reference's unseeded Random makes inside Step(), supplied by the caller (a recording of the
 * reference) instead of drawn from the engine's generator.  u is host double[k][num_envs], every value in [0, 1)
 * as _random.NextDouble() gives them (else GX_ERR_INVALID), count = k * num_envs:
 *   Pong, k = 2      ResetBall(), Envs.Pong.cs:159-165, and the same two lines of Reset(), :58-59: u[i] decides the
 *                    horizontal direction (> 0.5: towards the agent's paddle), u[num_envs + i] the vertical speed;
 *   Breakout, k = 1  the horizontal speed when FIRE releases the ball, Envs.Breakout.cs:100.
 * They serve the next launch only -- gx_vec_reset, or the next step or rollout, in which env i's first such event
 * (at most one per step) takes them and any later one draws from the generator again; envs without the event in
 * that launch ignore them.  u = NULL or count = 0 withdraws values not used yet.  GX_ERR_UNSUPPORTED for the other
 * envs, whose recorded runs are replayed by writing the recorded states with gx_vec_set_state.  Synchronous. */
int gx_vec_set_rng_draws(gx_vec* vec, const double* u, size_t count);
/* Observation of every env as Reset()/Step() would return it: classic control
 * float[obs_dim][num_envs] (double under GX_MODE_FP64); tabular int32[num_envs]; Pong, Breakout and SpaceInvaders uint8[num_envs][100800].
 * Synchronous. */
int gx_vec_get_obs(gx_vec* vec, void* host_obs, size_t bytes);

/* VectorEnv.Step(List<TAction>), VectorEnv.cs:28-34 -- host path.  h_actions is
 * int32[num_envs] or float[num_envs].  Output pointers may be NULL (that copy
 * is skipped).  Buffers are caller-owned, should be pinned, and must stay
 * alive and untouched until gx_vec_wait returns.  Copies run on a side stream
 * ordered against the compute stream by events. */
int gx_vec_step_async(gx_vec* vec, const void* h_actions, void* h_obs, void* h_reward, uint8_t* h_done,
                      uint8_t* h_truncated);
/* Also waits for the batch's most recent step launch that can raise the invalid-action flag, on
 * whichever stream it was enqueued (through an event of the library's own recorded behind it: a
 * caller's stream handle is never kept, so the caller may destroy its stream once it has
 * synchronised it), so an error is reported by the wait that follows the offending step. */
int gx_vec_wait(gx_vec* vec);

/* VectorEnv.Step -- zero-copy path: actions already on the device, results
 * read through gx_vec_device_view.  d_actions is int32 or float [gx_view.stride]
 * (num_envs rounded up to a multiple of four: the kernels read whole 16-byte
 * vectors, so the padding must be addressable; its values are ignored), 16-byte
 * aligned.  `stream` is a hipStream_t or NULL for the context's stream; the call
 * only enqueues. */
int gx_vec_step_device(gx_vec* vec, const void* d_actions, void* stream);
/* One step with actions sampled on the device (GX_POLICY_RANDOM). */
int gx_vec_step_random(gx_vec* vec, void* stream);
int gx_vec_device_view(gx_vec* vec, gx_view* out);

/* K fused steps with the state held in registers.  policy RANDOM samples
 * actions in the kernel; DEVICE_ACTIONS reads d_actions[K][stride].  The
 * reward/done/truncated arrays of the view hold the last step's values. */
int gx_vec_rollout(gx_vec* vec, int k_steps, int policy, const void* d_actions, void* stream);

/* The same with a choice of what the rollout hands back (VectorEnv.Step returns every step's
 * tuple, VectorEnv.cs:28-34): GX_OUT_LAST as gx_vec_rollout; GX_OUT_ALL additionally writes every
 * step's reward, done and (with a time limit, optional) truncated flag into caller-owned device
 * rows, row k = step k: d_reward_rows [k_steps][stride] float or double (gx_info.reward_dtype),
 * d_done_rows / d_truncated_rows uint8 [k_steps][stride], 16-byte aligned (5 or 9 more bytes per
 * env-step; the state still never leaves registers); GX_OUT_NONE writes no reward/done rows at
 * all (statistics only), the rows of the view keep their old contents. */
typedef enum gx_rollout_out { GX_OUT_LAST = 0, GX_OUT_ALL = 1, GX_OUT_NONE = 2 } gx_rollout_out;
int gx_vec_rollout_out(gx_vec* vec, int k_steps, int policy, const void* d_actions, int out, void* d_reward_rows,
                       uint8_t* d_done_rows, uint8_t* d_truncated_rows, void* stream);

/* Draws of the random policy for one tick, written to a device array
 * (int32 or float [num_envs]); for harnesses that replay the same actions
 * through another engine. */
int gx_vec_sample_actions(gx_vec* vec, uint64_t tick, void* d_actions, void* stream);
int gx_vec_tick(const gx_vec* vec, uint64_t* tick);

/* RecordEpisodeStatistics aggregate of this batch (device reduction + copy).  Runs on the
 * context's stream: steps enqueued on a caller's own stream must have been synchronised (or
 * ordered before the context's stream) by the caller.  The same holds for gx_vec_get_state,
 * gx_vec_get_obs and gx_vec_reset. */
int gx_vec_stats(gx_vec* vec, gx_stats* out);

/* ---- one entry point that allocates nothing ------------------------------ */
/* No counterpart in the reference.  For callers that cannot allocate device memory around a
 * step: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * gx_rows_step is gx_vec_rollout (k_steps = 1: gx_vec_step_device / gx_vec_step_random) followed
 * by gx_vec_stats, on SoA rows the CALLER owns, for the five classic-control envs in binary32
 * with GX_AUTORESET | GX_RECORD_STATS fused.  It calls no hipMalloc/hipFree or any other
 * allocating routine and only enqueues on `stream`.  The per-wave statistics records its kernels
 * need live in a fixed 2.5 MiB + 320 B array inside the library, one copy per GPU, held until the
 * process ends; that bounds a call at GX_ROWS_MAX_ENVS envs (larger: GX_ERR_INVALID).  Every
 * call on a GPU shares that GPU's array (cleared at the start of each call), so calls that overlap
 * on one GPU -- two streams, two threads -- race: serialise them. */
#define GX_ROWS_MAX_ENVS (1ull << 24)
typedef struct gx_rows {
  uint32_t struct_size;    /* sizeof(gx_rows) */
  int32_t env_id;          /* GX_ENV_CARTPOLE .. GX_ENV_ACROBOT */
  uint64_t num_envs;       /* 1 .. GX_ROWS_MAX_ENVS */
  uint64_t stride;         /* elements between component rows: a multiple of 4, >= num_envs */
  uint64_t seed;           /* as gx_desc */
  uint64_t env_index_base; /* as gx_desc */
  uint64_t tick;           /* the batch's tick before this call (gx_vec_tick); the caller adds k_steps */
  int32_t mode;            /* GX_MODE_EXACT or GX_MODE_FAST32 */
  int32_t state_bounded;   /* non-zero: every state is a Reset() draw or the result of an earlier step with
                              auto-reset (what gx_vec_* tracks itself); 0 after states were written from outside */
  float* state_real;       /* in/out [n_real][stride], the SoA image documented at gx_vec_get_state */
  int32_t* steps;          /* in/out [stride], the per-env step counter */
  void* ep_return;         /* in/out [stride], the open episode's return: double for MountainCarContinuous and
                              Pendulum, int32 for the others.  May be NULL for CartPole, MountainCar (only with
                              state_bounded != 0) and Acrobot: their open-episode return is then taken as the
                              reward per step times the step counter, as gx_vec_* does until a gx_vec_set_state */
  const void* actions;     /* in [k_steps][stride] int32 or float, or NULL: ActionSpace.Sample() in the kernel */
  void* reward;            /* out [stride] float (double for MountainCarContinuous, Pendulum): last step's */
  uint8_t* done;           /* out [stride]: last step's */
  gx_stats* stats;         /* out, DEVICE memory: the episodes that finished during this call */
  uint32_t* error;         /* out, device, may be NULL: set non-zero when Acrobot met an action outside {0,1,2} */
} gx_rows;
/* All pointers are device pointers, 16-byte aligned.  Padding elements [num_envs, stride) of the
 * rows must be addressable; their values are ignored on input and unspecified on output. */
int gx_rows_step(gx_ctx* ctx, const gx_rows* rows, int k_steps, void* stream);

/* ---- multi-GPU: episode statistics only -------------------------------- */
/* unique_id: 128 bytes, produced by rank 0 with gx_comm_unique_id and passed
 * to the other ranks by the host program. */
int gx_comm_unique_id(void* id128);
int gx_comm_init(gx_ctx* ctx, const void* id128, int rank, int world, gx_comm** out);
int gx_comm_destroy(gx_comm* comm);
/* All-reduce (sum / min / max per field) of the batch's statistics over RCCL. */
int gx_stats_allreduce(gx_comm* comm, gx_vec* vec, gx_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* GYMX_H */
