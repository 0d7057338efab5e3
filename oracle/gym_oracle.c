/*
 * gym_oracle.c -- CPU restatement of the reference's batched-step path.
 *
 * TEST INFRASTRUCTURE ONLY (see gym_oracle.h for the parity status).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -std=c11 (oracle/Makefile).
 * -ffp-contract=off matters: CoreCLR on x64 evaluates `float op float` with
 * scalar SSE and rounds to binary32 after every operation and never fuses
 * a*b+c, so this file must not either.  Math.Sin/Cos/Atan2 are double->double
 * calls into the platform libm on Linux, which is the glibc linked here.
 *
 * Typing rules followed throughout (C# 12 / net8.0):
 *   float op float   -> float (one rounding per op)
 *   int   op float   -> float
 *   x op double      -> double          (double literals, Math.PI)
 *   (float)expr      -> one rounding of the double value
 *   const float expressions are folded in binary32, operation by operation.
 * The C compiler applies the same rules to the same expressions, so the
 * source below keeps the reference's expression shapes instead of pre-folded
 * literals.
 */
#define _GNU_SOURCE
#include "gym_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ======================================================================
 * Counter-based RNG (build-defined).
 *
 * The reference owns an unseeded `new Random()` per env and per space
 * (Envs.CartPole.cs:19, Spaces.Box.cs:14, Spaces.Discrete.cs:11) and
 * Env.Seed is an empty virtual (Env.cs:33-36), so its draws cannot be
 * reproduced.  The build defines instead
 *     w = word (draw & 3) of Philox4x32-10(key = seed, ctr = (env_lo, env_hi,
 *               tick_lo, stream<<24 | tick_hi8<<16 | draw>>2))
 *     u = w * 2^-32  in [0,1)            (NextDouble stand-in, 32 random bits)
 *     k = (w * n) >> 32  in [0,n)        (Next(n) stand-in)
 * with `tick` = number of Reset()/Step() calls made on the batch before the
 * one that consumes the draw.  Streams: 0 = Reset draws, 1 = draws an env
 * makes inside Step, 2 = action-space samples of the random policy, 3 = the
 * per-tick word of the table-driven envs, 4 / 5 = Blackjack's deal and dealer words.
 * ====================================================================== */
void gxo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t gxo_rng_u32(uint64_t seed, uint64_t env_index, uint64_t tick, uint32_t stream, uint32_t draw) {
  uint32_t ctr[4], key[2], out[4];
  if (stream == 2) {
    /* the policy stream takes one draw per env per tick: four consecutive ticks share a
     * block (counter tick / 4), tick t reading word t % 4, so that a fused K-step rollout
     * runs the generator once every four steps */
    draw = (uint32_t)(tick & 3u);
    tick >>= 2;
  }
  if (stream >= 3) {
    /* the per-tick word of the table-driven envs (FrozenLake, Taxi, CliffWalking, DummyEnv; stream 3) and the
     * deal and dealer words of Blackjack under upstream's rules (streams 4, 5): four consecutive envs share a
     * block (counter env / 4), env e reading word e % 4 */
    draw = (uint32_t)(env_index & 3u);
    env_index >>= 2;
  }
  ctr[0] = (uint32_t)env_index;
  ctr[1] = (uint32_t)(env_index >> 32);
  ctr[2] = (uint32_t)tick;
  ctr[3] = (stream << 24) | ((uint32_t)((tick >> 32) & 0xFFu) << 16) | ((draw >> 2) & 0xFFFFu);
  key[0] = (uint32_t)seed;
  key[1] = (uint32_t)(seed >> 32);
  gxo_philox4x32_10(ctr, key, out);
  return out[draw & 3u];
}

double gxo_rng_uniform(uint64_t seed, uint64_t env_index, uint64_t tick, uint32_t stream, uint32_t draw) {
  return (double)gxo_rng_u32(seed, env_index, tick, stream, draw) * 0x1.0p-32;
}

/* Random.Next(n) stand-in (Spaces.Discrete.cs:18, Envs.Taxi.cs:20): the draw
 * scaled into [0,n) by multiply-shift. */
int32_t gxo_rng_below(uint64_t seed, uint64_t env_index, uint64_t tick, uint32_t stream, uint32_t draw, int32_t n) {
  uint64_t w = gxo_rng_u32(seed, env_index, tick, stream, draw);
  return (int32_t)((w * (uint64_t)(uint32_t)n) >> 32);
}

/* Spaces.Box.cs:30 -- `(float)(NextDouble() * (High[i] - Low[i]) + Low[i])`:
 * the range is a binary32 subtraction, promoted to double for the rest. */
float gxo_box_sample(double u, float low, float high) {
  float range = high - low;
  return (float)(u * (double)range + (double)low);
}

/* ======================================================================
 * Per-env objects with a virtual Reset/Step, like Env<TState,TAction>
 * (Env.cs:7-52).  The RNG context replaces the private `_rng` field.
 * ====================================================================== */
typedef struct rng_ctx {
  uint64_t seed;
  uint64_t env_index;
  uint64_t tick;
} rng_ctx;

typedef struct env_obj env_obj;
typedef struct env_vtbl {
  void (*reset)(env_obj*, const rng_ctx*);
  /* returns 0, or -1 where the reference throws */
  int (*step)(env_obj*, const void* action, const rng_ctx*, double* reward, int* done);
  void (*export_state)(const env_obj*, void* reals, int32_t* ints);
  void (*import_state)(env_obj*, const void* reals, const int32_t* ints);
  size_t size;
} env_vtbl;
struct env_obj {
  const env_vtbl* vt;
};

static inline float clampf(float v, float lo, float hi) {
  /* Math.Clamp(float,float,float): value<min -> min, value>max -> max */
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}

/* ---------------------------------------------------------------- CartPole */
typedef struct {
  env_obj base;
  float s[4];
  int32_t steps;
} cartpole_t;

/* Envs.CartPole.cs:22-32 */
static void cartpole_reset(env_obj* e, const rng_ctx* r) {
  cartpole_t* c = (cartpole_t*)e;
  for (uint32_t i = 0; i < 4; i++)
    c->s[i] = (float)(gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, i) * 0.1 - 0.05);
  c->steps = 0;
}

/* Envs.CartPole.cs:34-75 */
static int cartpole_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  cartpole_t* c = (cartpole_t*)e;
  int a = *(const int32_t*)action;
  const float gravity = 9.8f;
  const float masscart = 1.0f;
  const float masspole = 0.1f;
  const float total_mass = masscart + masspole;
  const float length = 0.5f;
  const float polemass_length = masspole * length;
  const float force_mag = 10.0f;
  const float tau = 0.02f;
  const float theta_threshold_radians = 12 * (float)M_PI / 180;
  const float x_threshold = 2.4f;

  float x = c->s[0], xDot = c->s[1], theta = c->s[2], thetaDot = c->s[3];
  float force = a == 1 ? force_mag : -force_mag;
  float costheta = (float)cos((double)theta);
  float sintheta = (float)sin((double)theta);

  float temp = (force + polemass_length * thetaDot * thetaDot * sintheta) / total_mass;
  float thetaacc = (gravity * sintheta - costheta * temp) /
                   (length * (4.0f / 3.0f - masspole * costheta * costheta / total_mass));
  float xacc = temp - polemass_length * thetaacc * costheta / total_mass;

  x += tau * xDot;
  xDot += tau * xacc;
  theta += tau * thetaDot;
  thetaDot += tau * thetaacc;
  c->s[0] = x; c->s[1] = xDot; c->s[2] = theta; c->s[3] = thetaDot;
  c->steps++;

  int d = x < -x_threshold || x > x_threshold || theta < -theta_threshold_radians ||
          theta > theta_threshold_radians || c->steps >= 500;
  *done = d;
  *reward = !d ? 1.0 : 0.0;
  return 0;
}

static void f32x_export(const float* s, int n, int32_t steps, void* reals, int32_t* ints) {
  memcpy(reals, s, sizeof(float) * (size_t)n);
  ints[0] = steps;
}

static void cartpole_export(const env_obj* e, void* reals, int32_t* ints) {
  const cartpole_t* c = (const cartpole_t*)e;
  f32x_export(c->s, 4, c->steps, reals, ints);
}
static void cartpole_import(env_obj* e, const void* reals, const int32_t* ints) {
  cartpole_t* c = (cartpole_t*)e;
  memcpy(c->s, reals, sizeof c->s);
  c->steps = ints[0];
}
static const env_vtbl cartpole_vt = {cartpole_reset, cartpole_step, cartpole_export, cartpole_import, sizeof(cartpole_t)};

/* ------------------------------------------------------------- MountainCar */
typedef struct {
  env_obj base;
  float s[2];
  int32_t steps;
} mcar_t;

/* Envs.MountainCar.cs:22-30, Envs.MountainCarContinuous.cs:19-27 */
static void mcar_reset(env_obj* e, const rng_ctx* r) {
  mcar_t* c = (mcar_t*)e;
  c->s[0] = (float)(gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, 0) * 0.2 - 0.6);
  c->s[1] = 0.0f;
  c->steps = 0;
}

/* Envs.MountainCar.cs:32-50 */
static int mcar_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  mcar_t* c = (mcar_t*)e;
  int a = *(const int32_t*)action;
  float position = c->s[0], velocity = c->s[1];
  float force = (float)(a - 1);
  velocity += 0.001f * force - 0.0025f * (float)cos((double)(3 * position));
  velocity = clampf(velocity, -0.07f, 0.07f);
  position += velocity;
  if (position < -1.2f) {
    position = -1.2f;
    velocity = 0.0f;
  }
  int d = position >= 0.5f;
  *done = d;
  *reward = d ? 0.0 : -1.0;
  c->s[0] = position; c->s[1] = velocity;
  c->steps++;
  return 0;
}

/* Envs.MountainCarContinuous.cs:29-49 */
static int mcc_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  mcar_t* c = (mcar_t*)e;
  float act = *(const float*)action;
  float position = c->s[0], velocity = c->s[1];
  float force = clampf(act, -1.0f, 1.0f);
  velocity += force * 0.0015f - 0.0025f * (float)cos((double)(3 * position));
  velocity = clampf(velocity, -0.07f, 0.07f);
  position += velocity;
  if (position < -1.2f) {
    position = -1.2f;
    velocity = 0.0f;
  }
  int d = position >= 0.45f;
  double rew = (double)velocity * 0.1;
  if (d && position >= 0.45f) rew += 100.0;
  *done = d;
  *reward = rew;
  c->s[0] = position; c->s[1] = velocity;
  c->steps++;
  return 0;
}

static void mcar_export(const env_obj* e, void* reals, int32_t* ints) {
  const mcar_t* c = (const mcar_t*)e;
  f32x_export(c->s, 2, c->steps, reals, ints);
}
static void mcar_import(env_obj* e, const void* reals, const int32_t* ints) {
  mcar_t* c = (mcar_t*)e;
  memcpy(c->s, reals, sizeof c->s);
  c->steps = ints[0];
}
static const env_vtbl mcar_vt = {mcar_reset, mcar_step, mcar_export, mcar_import, sizeof(mcar_t)};
static const env_vtbl mcc_vt = {mcar_reset, mcc_step, mcar_export, mcar_import, sizeof(mcar_t)};

/* ---------------------------------------------------------------- Pendulum */
typedef struct {
  env_obj base;
  float s[3];
  int32_t steps;
} pendulum_t;

/* Envs.Pendulum.cs:19-28 */
static void pendulum_reset(env_obj* e, const rng_ctx* r) {
  pendulum_t* c = (pendulum_t*)e;
  float theta = (float)(gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, 0) * 2 * M_PI - M_PI);
  float thetaDot = (float)(gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, 1) * 1 - 0.5);
  c->s[0] = (float)cos((double)theta);
  c->s[1] = (float)sin((double)theta);
  c->s[2] = thetaDot;
  c->steps = 0;
}

/* Envs.Pendulum.cs:30-60 */
static int pendulum_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  pendulum_t* c = (pendulum_t*)e;
  float act = *(const float*)action;
  const float max_speed = 8.0f;
  const float max_torque = 2.0f;
  const float dt = 0.05f;
  const float g = 10.0f;
  const float m = 1.0f;
  const float l = 1.0f;

  float theta = (float)atan2((double)c->s[1], (double)c->s[0]);
  float thetaDot = c->s[2];
  float u = clampf(act, -max_torque, max_torque);

  float new_thetaDot =
      thetaDot + (-3 * g / (2 * l) * (float)sin((double)theta + M_PI) + 3.0f / (m * l * l) * u) * dt;
  new_thetaDot = clampf(new_thetaDot, -max_speed, max_speed);
  float new_theta = theta + new_thetaDot * dt;

  c->s[0] = (float)cos((double)new_theta);
  c->s[1] = (float)sin((double)new_theta);
  c->s[2] = new_thetaDot;
  c->steps++;

  /* `theta * theta` is float*float (binary32 product) promoted to double;
   * `0.1 * thetaDot * thetaDot` is ((0.1*thetaDot)*thetaDot) in double. */
  *reward = -((double)(theta * theta) + 0.1 * (double)thetaDot * (double)thetaDot +
              0.001 * (double)u * (double)u);
  *done = 0;
  return 0;
}

static void pendulum_export(const env_obj* e, void* reals, int32_t* ints) {
  const pendulum_t* c = (const pendulum_t*)e;
  f32x_export(c->s, 3, c->steps, reals, ints);
}
static void pendulum_import(env_obj* e, const void* reals, const int32_t* ints) {
  pendulum_t* c = (pendulum_t*)e;
  memcpy(c->s, reals, sizeof c->s);
  c->steps = ints[0];
}
static const env_vtbl pendulum_vt = {pendulum_reset, pendulum_step, pendulum_export, pendulum_import, sizeof(pendulum_t)};

/* ----------------------------------------------------------------- Acrobot */
typedef struct {
  env_obj base;
  float s[6];
  int32_t steps;
} acrobot_t;

/* Envs.Acrobot.cs:22-28 -- all six slots get a small uniform draw. */
static void acrobot_reset(env_obj* e, const rng_ctx* r) {
  acrobot_t* c = (acrobot_t*)e;
  for (uint32_t i = 0; i < 6; i++)
    c->s[i] = (float)(gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, i) * 0.1 - 0.05);
  c->steps = 0;
}

/* Envs.Acrobot.cs:30-91 -- semi-implicit Euler in binary32, no angle wrap. */
static int acrobot_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  acrobot_t* c = (acrobot_t*)e;
  int a = *(const int32_t*)action;
  const float LINK_LENGTH_1 = 1.0f;
  const float LINK_LENGTH_2 = 1.0f;
  const float LINK_MASS_1 = 1.0f;
  const float LINK_MASS_2 = 1.0f;
  const float LINK_COM_POS_1 = 0.5f;
  const float LINK_COM_POS_2 = 0.5f;
  const float LINK_MOI_1 = 1.0f;
  const float LINK_MOI_2 = 1.0f;
  const float MAX_VEL_1 = 4 * (float)M_PI;
  const float MAX_VEL_2 = 9 * (float)M_PI;
  const float dt = 0.2f;
  const float g = 9.8f;
  const float torque[3] = {-1.0f, 0.0f, 1.0f};
  if (a < 0 || a > 2) return -1; /* IndexOutOfRangeException, Envs.Acrobot.cs:51 */

  float theta1 = (float)atan2((double)c->s[1], (double)c->s[0]);
  float theta2 = (float)atan2((double)c->s[3], (double)c->s[2]);
  float dtheta1 = c->s[4];
  float dtheta2 = c->s[5];
  float u = torque[a];

  float d1 = LINK_MASS_1 * LINK_COM_POS_1 * LINK_COM_POS_1 +
             LINK_MASS_2 * (LINK_LENGTH_1 * LINK_LENGTH_1 + LINK_COM_POS_2 * LINK_COM_POS_2 +
                            2 * LINK_LENGTH_1 * LINK_COM_POS_2 * (float)cos((double)theta2)) +
             LINK_MOI_1 + LINK_MOI_2;
  float d2 = LINK_MASS_2 * (LINK_COM_POS_2 * LINK_COM_POS_2 +
                            LINK_LENGTH_1 * LINK_COM_POS_2 * (float)cos((double)theta2)) +
             LINK_MOI_2;
  float phi2 = LINK_MASS_2 * LINK_COM_POS_2 * g * (float)cos((double)(theta1 + theta2) - M_PI / 2.0);
  float phi1 = -LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta2 * dtheta2 * (float)sin((double)theta2) -
               2 * LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta2 * dtheta1 * (float)sin((double)theta2) +
               (LINK_MASS_1 * LINK_COM_POS_1 + LINK_MASS_2 * LINK_LENGTH_1) * g *
                   (float)cos((double)theta1 - M_PI / 2.0) +
               phi2;
  float ddtheta2 = (u + d2 / d1 * phi1 -
                    LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta1 * dtheta1 * (float)sin((double)theta2) - phi2) /
                   (LINK_MASS_2 * LINK_COM_POS_2 * LINK_COM_POS_2 + LINK_MOI_2 - d2 * d2 / d1);
  float ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;

  dtheta1 += dt * ddtheta1;
  dtheta2 += dt * ddtheta2;
  dtheta1 = clampf(dtheta1, -MAX_VEL_1, MAX_VEL_1);
  dtheta2 = clampf(dtheta2, -MAX_VEL_2, MAX_VEL_2);
  theta1 += dt * dtheta1;
  theta2 += dt * dtheta2;

  c->s[0] = (float)cos((double)theta1);
  c->s[1] = (float)sin((double)theta1);
  c->s[2] = (float)cos((double)theta2);
  c->s[3] = (float)sin((double)theta2);
  c->s[4] = dtheta1;
  c->s[5] = dtheta2;
  c->steps++;

  float tip_y = -LINK_LENGTH_1 * (float)cos((double)theta1) - LINK_LENGTH_2 * (float)cos((double)(theta1 + theta2));
  int d = tip_y > 1.0f || c->steps >= 500;
  *done = d;
  *reward = d ? 0.0 : -1.0;
  return 0;
}

static void acrobot_export(const env_obj* e, void* reals, int32_t* ints) {
  const acrobot_t* c = (const acrobot_t*)e;
  f32x_export(c->s, 6, c->steps, reals, ints);
}
static void acrobot_import(env_obj* e, const void* reals, const int32_t* ints) {
  acrobot_t* c = (acrobot_t*)e;
  memcpy(c->s, reals, sizeof c->s);
  c->steps = ints[0];
}
static const env_vtbl acrobot_vt = {acrobot_reset, acrobot_step, acrobot_export, acrobot_import, sizeof(acrobot_t)};

/* ------------------------------------------- Acrobot, upstream RK4 in fp64
 * EXTENSION (parity unpinned by the reference, which integrates with
 * semi-implicit Euler in binary32, Envs.Acrobot.cs:69-74).  Follows the
 * published Gymnasium "book" dynamics with the same constants: state
 * (theta1, theta2, dtheta1, dtheta2) in double, one classical RK4 step of
 * dt = 0.2, wrap of both angles to [-pi, pi), velocity bounds 4pi / 9pi,
 * terminal when -cos(t1) - cos(t1 + t2) > 1 or steps >= 500.  Reset draws
 * four uniforms in [-0.1, 0.1).  Every a*b+c below is written with fma() so
 * that a device build, which contracts, can match bit-for-bit wherever the
 * transcendental results agree. */
typedef struct {
  env_obj base;
  double s[4];
  int32_t steps;
} acrobot64_t;

static void acrobot64_reset(env_obj* e, const rng_ctx* r) {
  acrobot64_t* c = (acrobot64_t*)e;
  for (uint32_t i = 0; i < 4; i++)
    c->s[i] = fma(gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, i), 0.2, -0.1);
  c->steps = 0;
}

static inline void acrobot64_dsdt(const double y[4], double u, double dy[4]) {
  const double g = 9.8;
  double t1 = y[0], t2 = y[1], w1 = y[2], w2 = y[3];
  double c2 = cos(t2), s2 = sin(t2);
  /* m1*lc1^2 + m2*(l1^2 + lc2^2 + 2*l1*lc2*cos t2) + I1 + I2 with unit masses
   * and lengths, lc = 0.5, I = 1 */
  double d1 = c2 + 3.5;
  double d2 = fma(0.5, c2, 1.25);
  double phi2 = (0.5 * g) * cos(t1 + t2 - M_PI / 2.0);
  double phi1 = fma(1.5 * g, cos(t1 - M_PI / 2.0), phi2);
  phi1 = fma(-(w2 * s2), w1, phi1);
  phi1 = fma(-0.5 * (w2 * s2), w2, phi1);
  double r = d2 / d1;
  double num = fma(r, phi1, u);
  num = fma(-0.5 * (w1 * s2), w1, num) - phi2;
  double den = fma(-d2, r, 1.25);
  double a2 = num / den;
  double a1 = -fma(d2, a2, phi1) / d1;
  dy[0] = w1; dy[1] = w2; dy[2] = a1; dy[3] = a2;
}

static inline double wrap_pi(double x) {
  /* upstream wrap(x, -pi, pi): repeated +-2pi until inside [-pi, pi] */
  const double two_pi = 2.0 * M_PI;
  while (x > M_PI) x -= two_pi;
  while (x < -M_PI) x += two_pi;
  return x;
}

static int acrobot64_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  acrobot64_t* c = (acrobot64_t*)e;
  int a = *(const int32_t*)action;
  if (a < 0 || a > 2) return -1;
  const double dt = 0.2;
  double u = (double)(a - 1);
  double y0[4] = {c->s[0], c->s[1], c->s[2], c->s[3]};
  double k1[4], k2[4], k3[4], k4[4], y[4];
  acrobot64_dsdt(y0, u, k1);
  for (int i = 0; i < 4; i++) y[i] = fma(0.5 * dt, k1[i], y0[i]);
  acrobot64_dsdt(y, u, k2);
  for (int i = 0; i < 4; i++) y[i] = fma(0.5 * dt, k2[i], y0[i]);
  acrobot64_dsdt(y, u, k3);
  for (int i = 0; i < 4; i++) y[i] = fma(dt, k3[i], y0[i]);
  acrobot64_dsdt(y, u, k4);
  for (int i = 0; i < 4; i++) {
    double sum = fma(2.0, k2[i], k1[i]);
    sum = fma(2.0, k3[i], sum) + k4[i];
    y[i] = fma(dt / 6.0, sum, y0[i]);
  }
  y[0] = wrap_pi(y[0]);
  y[1] = wrap_pi(y[1]);
  const double mv1 = 4.0 * M_PI, mv2 = 9.0 * M_PI;
  y[2] = y[2] < -mv1 ? -mv1 : (y[2] > mv1 ? mv1 : y[2]);
  y[3] = y[3] < -mv2 ? -mv2 : (y[3] > mv2 ? mv2 : y[3]);
  for (int i = 0; i < 4; i++) c->s[i] = y[i];
  c->steps++;
  int d = (-cos(y[0]) - cos(y[1] + y[0]) > 1.0) || c->steps >= 500;
  *done = d;
  *reward = d ? 0.0 : -1.0;
  return 0;
}

static void acrobot64_export(const env_obj* e, void* reals, int32_t* ints) {
  const acrobot64_t* c = (const acrobot64_t*)e;
  memcpy(reals, c->s, sizeof c->s);
  ints[0] = c->steps;
}
static void acrobot64_import(env_obj* e, const void* reals, const int32_t* ints) {
  acrobot64_t* c = (acrobot64_t*)e;
  memcpy(c->s, reals, sizeof c->s);
  c->steps = ints[0];
}
static const env_vtbl acrobot64_vt = {acrobot64_reset, acrobot64_step, acrobot64_export, acrobot64_import, sizeof(acrobot64_t)};

/* ------------------------------- classic control in binary64 (GXO_FP64)
 * EXTENSION (parity unpinned by the reference, which computes in binary32):
 * the same five sources with every `float` read as `double`, every literal
 * taken at its binary64 value and the (float) casts dropped; statement and
 * operand order as in the source.  Reset formulas are the source's double
 * expressions without the final rounding to float. */
typedef struct {
  env_obj base;
  double s[6];
  int32_t steps;
  int n; /* components in use, set at creation */
} classic64_t;

/* Math.Sin and Math.Cos are separate calls into the C library; kept out of line so the compiler
 * does not merge a sin/cos pair of one argument into sincos(), whose results differ from theirs
 * in the last place for about one argument in 700 (glibc 2.35). */
static double __attribute__((noinline)) libm_sin(double x) { return sin(x); }
static double __attribute__((noinline)) libm_cos(double x) { return cos(x); }

static inline double clampd(double v, double lo, double hi) {
  if (v < lo) return lo;
  if (v > hi) return hi;
  return v;
}

static void classic64_export(const env_obj* e, void* reals, int32_t* ints) {
  const classic64_t* c = (const classic64_t*)e;
  memcpy(reals, c->s, sizeof(double) * (size_t)c->n);
  ints[0] = c->steps;
}
static void classic64_import(env_obj* e, const void* reals, const int32_t* ints) {
  classic64_t* c = (classic64_t*)e;
  memcpy(c->s, reals, sizeof(double) * (size_t)c->n);
  c->steps = ints[0];
}

/* Envs.CartPole.cs:22-32 */
static void cartpole64_reset(env_obj* e, const rng_ctx* r) {
  classic64_t* c = (classic64_t*)e;
  for (uint32_t i = 0; i < 4; i++) c->s[i] = gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, i) * 0.1 - 0.05;
  c->steps = 0;
}

/* Envs.CartPole.cs:34-75 */
static int cartpole64_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  classic64_t* c = (classic64_t*)e;
  int a = *(const int32_t*)action;
  const double gravity = 9.8, masscart = 1.0, masspole = 0.1;
  const double total_mass = masscart + masspole;
  const double length = 0.5;
  const double polemass_length = masspole * length;
  const double force_mag = 10.0, tau = 0.02;
  const double theta_threshold_radians = 12 * M_PI / 180;
  const double x_threshold = 2.4;

  double x = c->s[0], xDot = c->s[1], theta = c->s[2], thetaDot = c->s[3];
  double force = a == 1 ? force_mag : -force_mag;
  double costheta = libm_cos(theta);
  double sintheta = libm_sin(theta);

  double temp = (force + polemass_length * thetaDot * thetaDot * sintheta) / total_mass;
  double thetaacc = (gravity * sintheta - costheta * temp) /
                    (length * (4.0 / 3.0 - masspole * costheta * costheta / total_mass));
  double xacc = temp - polemass_length * thetaacc * costheta / total_mass;

  x += tau * xDot;
  xDot += tau * xacc;
  theta += tau * thetaDot;
  thetaDot += tau * thetaacc;
  c->s[0] = x; c->s[1] = xDot; c->s[2] = theta; c->s[3] = thetaDot;
  c->steps++;

  int d = x < -x_threshold || x > x_threshold || theta < -theta_threshold_radians ||
          theta > theta_threshold_radians || c->steps >= 500;
  *done = d;
  *reward = !d ? 1.0 : 0.0;
  return 0;
}

/* Envs.MountainCar.cs:22-30, Envs.MountainCarContinuous.cs:19-27 */
static void mcar64_reset(env_obj* e, const rng_ctx* r) {
  classic64_t* c = (classic64_t*)e;
  c->s[0] = gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, 0) * 0.2 - 0.6;
  c->s[1] = 0.0;
  c->steps = 0;
}

/* Envs.MountainCar.cs:32-50 */
static int mcar64_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  classic64_t* c = (classic64_t*)e;
  int a = *(const int32_t*)action;
  double position = c->s[0], velocity = c->s[1];
  double force = (double)(a - 1);
  velocity += 0.001 * force - 0.0025 * libm_cos(3 * position);
  velocity = clampd(velocity, -0.07, 0.07);
  position += velocity;
  if (position < -1.2) {
    position = -1.2;
    velocity = 0.0;
  }
  int d = position >= 0.5;
  *done = d;
  *reward = d ? 0.0 : -1.0;
  c->s[0] = position; c->s[1] = velocity;
  c->steps++;
  return 0;
}

/* Envs.MountainCarContinuous.cs:29-49 */
static int mcc64_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  classic64_t* c = (classic64_t*)e;
  double act = (double)*(const float*)action; /* the action stays a float, as the API has it */
  double position = c->s[0], velocity = c->s[1];
  double force = clampd(act, -1.0, 1.0);
  velocity += force * 0.0015 - 0.0025 * libm_cos(3 * position);
  velocity = clampd(velocity, -0.07, 0.07);
  position += velocity;
  if (position < -1.2) {
    position = -1.2;
    velocity = 0.0;
  }
  int d = position >= 0.45;
  double rew = velocity * 0.1;
  if (d) rew += 100.0;
  *done = d;
  *reward = rew;
  c->s[0] = position; c->s[1] = velocity;
  c->steps++;
  return 0;
}

/* Envs.Pendulum.cs:19-28 */
static void pendulum64_reset(env_obj* e, const rng_ctx* r) {
  classic64_t* c = (classic64_t*)e;
  double theta = gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, 0) * 2 * M_PI - M_PI;
  double thetaDot = gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, 1) * 1 - 0.5;
  c->s[0] = libm_cos(theta);
  c->s[1] = libm_sin(theta);
  c->s[2] = thetaDot;
  c->steps = 0;
}

/* Envs.Pendulum.cs:30-60 */
static int pendulum64_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  classic64_t* c = (classic64_t*)e;
  double act = (double)*(const float*)action;
  const double max_speed = 8.0, max_torque = 2.0, dt = 0.05, g = 10.0, m = 1.0, l = 1.0;

  double theta = atan2(c->s[1], c->s[0]);
  double thetaDot = c->s[2];
  double u = clampd(act, -max_torque, max_torque);

  double new_thetaDot = thetaDot + (-3 * g / (2 * l) * libm_sin(theta + M_PI) + 3.0 / (m * l * l) * u) * dt;
  new_thetaDot = clampd(new_thetaDot, -max_speed, max_speed);
  double new_theta = theta + new_thetaDot * dt;

  c->s[0] = libm_cos(new_theta);
  c->s[1] = libm_sin(new_theta);
  c->s[2] = new_thetaDot;
  c->steps++;

  *reward = -(theta * theta + 0.1 * thetaDot * thetaDot + 0.001 * u * u);
  *done = 0;
  return 0;
}

/* Envs.Acrobot.cs:22-28 */
static void acrobot_e64_reset(env_obj* e, const rng_ctx* r) {
  classic64_t* c = (classic64_t*)e;
  for (uint32_t i = 0; i < 6; i++) c->s[i] = gxo_rng_uniform(r->seed, r->env_index, r->tick, 0, i) * 0.1 - 0.05;
  c->steps = 0;
}

/* Envs.Acrobot.cs:30-91, semi-implicit Euler */
static int acrobot_e64_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  classic64_t* c = (classic64_t*)e;
  int a = *(const int32_t*)action;
  const double LINK_LENGTH_1 = 1.0, LINK_LENGTH_2 = 1.0, LINK_MASS_1 = 1.0, LINK_MASS_2 = 1.0;
  const double LINK_COM_POS_1 = 0.5, LINK_COM_POS_2 = 0.5, LINK_MOI_1 = 1.0, LINK_MOI_2 = 1.0;
  const double MAX_VEL_1 = 4 * M_PI, MAX_VEL_2 = 9 * M_PI;
  const double dt = 0.2, g = 9.8;
  const double torque[3] = {-1.0, 0.0, 1.0};
  if (a < 0 || a > 2) return -1; /* Envs.Acrobot.cs:51 */

  double theta1 = atan2(c->s[1], c->s[0]);
  double theta2 = atan2(c->s[3], c->s[2]);
  double dtheta1 = c->s[4];
  double dtheta2 = c->s[5];
  double u = torque[a];

  double d1 = LINK_MASS_1 * LINK_COM_POS_1 * LINK_COM_POS_1 +
              LINK_MASS_2 * (LINK_LENGTH_1 * LINK_LENGTH_1 + LINK_COM_POS_2 * LINK_COM_POS_2 +
                             2 * LINK_LENGTH_1 * LINK_COM_POS_2 * libm_cos(theta2)) +
              LINK_MOI_1 + LINK_MOI_2;
  double d2 = LINK_MASS_2 * (LINK_COM_POS_2 * LINK_COM_POS_2 + LINK_LENGTH_1 * LINK_COM_POS_2 * libm_cos(theta2)) + LINK_MOI_2;
  double phi2 = LINK_MASS_2 * LINK_COM_POS_2 * g * libm_cos(theta1 + theta2 - M_PI / 2.0);
  double phi1 = -LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta2 * dtheta2 * libm_sin(theta2) -
                2 * LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta2 * dtheta1 * libm_sin(theta2) +
                (LINK_MASS_1 * LINK_COM_POS_1 + LINK_MASS_2 * LINK_LENGTH_1) * g * libm_cos(theta1 - M_PI / 2.0) + phi2;
  double ddtheta2 = (u + d2 / d1 * phi1 - LINK_MASS_2 * LINK_LENGTH_1 * LINK_COM_POS_2 * dtheta1 * dtheta1 * libm_sin(theta2) - phi2) /
                    (LINK_MASS_2 * LINK_COM_POS_2 * LINK_COM_POS_2 + LINK_MOI_2 - d2 * d2 / d1);
  double ddtheta1 = -(d2 * ddtheta2 + phi1) / d1;

  dtheta1 += dt * ddtheta1;
  dtheta2 += dt * ddtheta2;
  dtheta1 = clampd(dtheta1, -MAX_VEL_1, MAX_VEL_1);
  dtheta2 = clampd(dtheta2, -MAX_VEL_2, MAX_VEL_2);
  theta1 += dt * dtheta1;
  theta2 += dt * dtheta2;

  c->s[0] = libm_cos(theta1);
  c->s[1] = libm_sin(theta1);
  c->s[2] = libm_cos(theta2);
  c->s[3] = libm_sin(theta2);
  c->s[4] = dtheta1;
  c->s[5] = dtheta2;
  c->steps++;

  double tip_y = -LINK_LENGTH_1 * libm_cos(theta1) - LINK_LENGTH_2 * libm_cos(theta1 + theta2);
  int d = tip_y > 1.0 || c->steps >= 500;
  *done = d;
  *reward = d ? 0.0 : -1.0;
  return 0;
}

static const env_vtbl cartpole64_vt = {cartpole64_reset, cartpole64_step, classic64_export, classic64_import, sizeof(classic64_t)};
static const env_vtbl mcar64_vt = {mcar64_reset, mcar64_step, classic64_export, classic64_import, sizeof(classic64_t)};
static const env_vtbl mcc64_vt = {mcar64_reset, mcc64_step, classic64_export, classic64_import, sizeof(classic64_t)};
static const env_vtbl pendulum64_vt = {pendulum64_reset, pendulum64_step, classic64_export, classic64_import, sizeof(classic64_t)};
static const env_vtbl acrobot_e64_vt = {acrobot_e64_reset, acrobot_e64_step, classic64_export, classic64_import, sizeof(classic64_t)};

/* ---------------------------------------------------- grid worlds (tabular) */
#define GRID_MAX_CELLS 4096
typedef struct grid_desc {
  int nrow, ncol, start, goal;
  int step_limit;
  uint8_t hole[GRID_MAX_CELLS]; /* FrozenLake holes / CliffWalking cliff */
  int is_cliff;
  int slippery;
} grid_desc;

typedef struct {
  env_obj base;
  const grid_desc* g;
  int32_t state;
  int32_t steps;
} grid_t;

/* Envs.FrozenLake.cs:45-50, Envs.CliffWalking.cs:29-34 */
static void grid_reset(env_obj* e, const rng_ctx* r) {
  (void)r;
  grid_t* c = (grid_t*)e;
  c->state = c->g->start;
  c->steps = 0;
}

/* Envs.FrozenLake.cs:52-69, Envs.CliffWalking.cs:36-54 */
static int grid_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  grid_t* c = (grid_t*)e;
  const grid_desc* g = c->g;
  int a = *(const int32_t*)action;
  if (g->slippery && a >= 0 && a <= 3) {
    /* EXTENSION (upstream is_slippery; the reference declares _rng and never
     * uses it, Envs.FrozenLake.cs:16): the executed action is a-1, a or a+1
     * (mod 4) with probability 1/3 each.  The outcome is drawn from the env's
     * word w of the tick (stream 3): with u = w / 2^32, Discrete(4).Sample() of that tick
     * is floor(4u) and the outcome is the next base-3 digit, floor(frac(4u) * 3)
     * = floor(12u) - 3 floor(4u): one generator call serves both, the two are
     * independent, and the outcome does not depend on the action passed in. */
    uint64_t w = gxo_rng_u32(r->seed, r->env_index, r->tick, 3, 0);
    int k = (int)((w * 12u) >> 32) - 3 * (int)((w * 4u) >> 32);
    a = (a + 3 + k) & 3;
  }
  int row = c->state / g->ncol;
  int col = c->state % g->ncol;
  switch (a) {
    case 0: col = col - 1 > 0 ? col - 1 : 0; break;
    case 1: row = row + 1 < g->nrow - 1 ? row + 1 : g->nrow - 1; break;
    case 2: col = col + 1 < g->ncol - 1 ? col + 1 : g->ncol - 1; break;
    case 3: row = row - 1 > 0 ? row - 1 : 0; break;
    default: break;
  }
  c->state = row * g->ncol + col;
  c->steps++;
  int at_goal = c->state == g->goal;
  int in_hole = g->hole[c->state];
  *done = at_goal || in_hole || c->steps >= g->step_limit;
  if (g->is_cliff) {
    double rew = in_hole ? -100.0 : -1.0;
    if (at_goal) rew = 0.0;
    *reward = rew;
  } else {
    *reward = at_goal ? 1.0 : 0.0;
  }
  return 0;
}

static void grid_export(const env_obj* e, void* reals, int32_t* ints) {
  (void)reals;
  const grid_t* c = (const grid_t*)e;
  ints[0] = c->state;
  ints[1] = c->steps;
}
static void grid_import(env_obj* e, const void* reals, const int32_t* ints) {
  (void)reals;
  grid_t* c = (grid_t*)e;
  c->state = ints[0];
  c->steps = ints[1];
}
static const env_vtbl grid_vt = {grid_reset, grid_step, grid_export, grid_import, sizeof(grid_t)};

/* -------------------------------------------------------------------- Taxi */
typedef struct {
  env_obj base;
  int32_t state;
  int32_t steps;
  int full;
} taxi_t;

/* Envs.Taxi.cs:18-23 */
static void taxi_reset(env_obj* e, const rng_ctx* r) {
  taxi_t* c = (taxi_t*)e;
  if (!c->full) {
    c->state = gxo_rng_below(r->seed, r->env_index, r->tick, 0, 0, 500);
  } else {
    /* EXTENSION: upstream draws only states whose passenger is not already
     * at the destination and not in the taxi: 25 cells * 4 pass * 3 dest. */
    int k = gxo_rng_below(r->seed, r->env_index, r->tick, 0, 0, 300);
    int cell = k / 12, rem = k % 12, pass = rem / 3, d3 = rem % 3;
    int dest = d3 >= pass ? d3 + 1 : d3;
    c->state = (cell * 5 + pass) * 4 + dest;
  }
  c->steps = 0;
}

/* EXTENSION: upstream Taxi-v3 transition (5x5 grid, walls, 4 depots). */
static void taxi_full_transition(int s, int a, int* ns, double* rew, int* term) {
  static const int loc_r[4] = {0, 0, 4, 4};
  static const int loc_c[4] = {0, 4, 0, 3};
  /* wall_e[r][c] = 1 when a wall blocks moving east out of (r,c) */
  static const uint8_t wall_e[5][5] = {
      {0, 1, 0, 0, 1}, {0, 1, 0, 0, 1}, {0, 0, 0, 0, 1}, {1, 0, 1, 0, 1}, {1, 0, 1, 0, 1}};
  int dest = s % 4, pass = (s / 4) % 5, col = (s / 20) % 5, row = s / 100;
  double r = -1.0;
  int t = 0;
  switch (a) {
    case 0: row = row + 1 < 4 ? row + 1 : 4; break;          /* south */
    case 1: row = row - 1 > 0 ? row - 1 : 0; break;          /* north */
    case 2: if (!wall_e[row][col]) col = col + 1; break;     /* east */
    case 3: if (col > 0 && !wall_e[row][col - 1]) col = col - 1; break; /* west */
    case 4: /* pickup */
      if (pass < 4 && row == loc_r[pass] && col == loc_c[pass]) pass = 4;
      else r = -10.0;
      break;
    case 5: /* dropoff */
      if (pass == 4 && row == loc_r[dest] && col == loc_c[dest]) {
        pass = dest; r = 20.0; t = 1;
      } else if (pass == 4 && ((row == 0 && col == 0) || (row == 0 && col == 4) ||
                               (row == 4 && col == 0) || (row == 4 && col == 3))) {
        for (int k = 0; k < 4; k++)
          if (row == loc_r[k] && col == loc_c[k]) pass = k;
      } else {
        r = -10.0;
      }
      break;
    default: break;
  }
  *ns = ((row * 5 + col) * 5 + pass) * 4 + dest;
  *rew = r;
  *term = t;
}

/* Envs.Taxi.cs:25-33 -- the reference is a stub: the state never changes. */
static int taxi_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  taxi_t* c = (taxi_t*)e;
  int a = *(const int32_t*)action;
  c->steps++;
  if (!c->full) {
    *done = c->steps >= 200;
    *reward = -1.0;
  } else {
    int ns, t;
    double rew;
    taxi_full_transition(c->state, a, &ns, &rew, &t);
    c->state = ns;
    *done = t || c->steps >= 200;
    *reward = rew;
  }
  return 0;
}

static void taxi_export(const env_obj* e, void* reals, int32_t* ints) {
  (void)reals;
  const taxi_t* c = (const taxi_t*)e;
  ints[0] = c->state;
  ints[1] = c->steps;
}
static void taxi_import(env_obj* e, const void* reals, const int32_t* ints) {
  (void)reals;
  taxi_t* c = (taxi_t*)e;
  c->state = ints[0];
  c->steps = ints[1];
}
static const env_vtbl taxi_vt = {taxi_reset, taxi_step, taxi_export, taxi_import, sizeof(taxi_t)};

/* ------------------------------------------------- caller-supplied tables */
/* What FrozenLake's map argument (Envs.FrozenLake.cs:23) generalises to: the body of Step is a table lookup,
 * the rest -- steps++, the step-limit term of done, Reset -- is as in Envs.FrozenLake.cs:45-69 / Envs.Taxi.cs:18-33. */
typedef struct table_desc {
  gxo_transition* t;
  int32_t n_states, n_actions, n_outcomes, start_state, step_limit;
} table_desc;

typedef struct {
  env_obj base;
  const table_desc* d;
  int32_t state;
  int32_t steps;
} table_t;

static void table_reset(env_obj* e, const rng_ctx* r) {
  table_t* c = (table_t*)e;
  /* Envs.FrozenLake.cs:47 a fixed state; Envs.Taxi.cs:20 `_rng.Next(n)` */
  c->state = c->d->start_state >= 0 ? c->d->start_state : gxo_rng_below(r->seed, r->env_index, r->tick, 0, 0, c->d->n_states);
  c->steps = 0;
}

static int table_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  table_t* c = (table_t*)e;
  const table_desc* d = c->d;
  int a = *(const int32_t*)action;
  if (a < 0 || a >= d->n_actions) return -1;
  int o = 0;
  if (d->n_outcomes > 1) {
    /* the digit that follows the action digit of the env's word of the tick (stream 3), as in grid_step */
    uint64_t w = gxo_rng_u32(r->seed, r->env_index, r->tick, 3, 0);
    o = (int)((w * (uint64_t)(d->n_actions * d->n_outcomes)) >> 32) - d->n_outcomes * (int)((w * (uint64_t)d->n_actions) >> 32);
  }
  /* a state outside the table (only reachable through set_state) is read as state 0 */
  int s = c->state >= 0 && c->state < d->n_states ? c->state : 0;
  const gxo_transition* t = &d->t[((size_t)s * d->n_actions + a) * d->n_outcomes + o];
  c->state = t->next_state;
  c->steps++;
  *done = t->done != 0 || (d->step_limit > 0 && c->steps >= d->step_limit);
  *reward = (double)t->reward;
  return 0;
}

static void table_export(const env_obj* e, void* reals, int32_t* ints) {
  (void)reals;
  const table_t* c = (const table_t*)e;
  ints[0] = c->state;
  ints[1] = c->steps;
}
static void table_import(env_obj* e, const void* reals, const int32_t* ints) {
  (void)reals;
  table_t* c = (table_t*)e;
  c->state = ints[0];
  c->steps = ints[1];
}
static const env_vtbl table_vt = {table_reset, table_step, table_export, table_import, sizeof(table_t)};

/* ------------------------------------------------------------------- Dummy */
typedef struct {
  env_obj base;
  int32_t state;
  int32_t steps;
} dummy_t;

/* DummyEnv.cs:17-21 */
static void dummy_reset(env_obj* e, const rng_ctx* r) {
  (void)r;
  dummy_t* c = (dummy_t*)e;
  c->state = 0;
  c->steps = 0;
}
/* DummyEnv.cs:23-30 */
static int dummy_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  (void)r;
  dummy_t* c = (dummy_t*)e;
  c->state += *(const int32_t*)action;
  c->steps++;
  *reward = (double)c->state;
  *done = c->state >= 10;
  return 0;
}
static void dummy_export(const env_obj* e, void* reals, int32_t* ints) {
  (void)reals;
  const dummy_t* c = (const dummy_t*)e;
  ints[0] = c->state;
  ints[1] = c->steps;
}
static void dummy_import(env_obj* e, const void* reals, const int32_t* ints) {
  (void)reals;
  dummy_t* c = (dummy_t*)e;
  c->state = ints[0];
  c->steps = ints[1];
}
static const env_vtbl dummy_vt = {dummy_reset, dummy_step, dummy_export, dummy_import, sizeof(dummy_t)};

/* --------------------------------------------------------------- Blackjack */
/* Envs.Blackjack.cs: state (playerSum, dealerCard, usableAce), Discrete(2) actions (0 stick, 1 hit),
 * ObservationSpace Discrete(32*11*2).  The state image holds the observation as that space's
 * index, (playerSum*11 + dealerCard)*2 + usableAce, then the dealer's face-down cards (raw sum in
 * bits 0..7, "one of them is an ace" in bit 8; 0 under the reference's rules, which never deal
 * them), then the step counter. */
typedef struct {
  env_obj base;
  int32_t player_sum, dealer_card, usable_ace;
  int32_t hidden_sum, hidden_ace; /* EXTENSION only */
  int32_t steps;
  int full;
} blackjack_t;

/* the 13 ranks of an infinite deck: ace = 1, 2..10, and jack/queen/king count 10 */
static inline int bj_card(int rank_index) { return rank_index + 1 < 10 ? rank_index + 1 : 10; }

/* Envs.Blackjack.cs:21-30 */
/* floor(13 u) for u = *w / 2^32, leaving the fraction of 13 u in *w: successive base-13 digits of one word */
static int bj_next_digit13(uint32_t* w) {
  uint64_t p = (uint64_t)*w * 13u;
  *w = (uint32_t)p;
  return (int)(p >> 32);
}

static void blackjack_reset(env_obj* e, const rng_ctx* r) {
  blackjack_t* c = (blackjack_t*)e;
  if (!c->full) {
    c->player_sum = 12 + gxo_rng_below(r->seed, r->env_index, r->tick, 0, 0, 10);  /* Next(12, 22) */
    c->dealer_card = 1 + gxo_rng_below(r->seed, r->env_index, r->tick, 0, 1, 10);  /* Next(1, 11) */
    c->usable_ace = gxo_rng_below(r->seed, r->env_index, r->tick, 0, 2, 2) == 1;   /* Next(2) == 1 */
    c->hidden_sum = 0;
    c->hidden_ace = 0;
  } else {
    /* EXTENSION (upstream Blackjack-v1): the dealer's two cards, then the player's two: the first four base-13
     * digits of the env's deal word of the tick (stream 4) */
    uint32_t w = gxo_rng_u32(r->seed, r->env_index, r->tick, 4, 0);
    int d0 = bj_card(bj_next_digit13(&w));
    int d1 = bj_card(bj_next_digit13(&w));
    int p0 = bj_card(bj_next_digit13(&w));
    int p1 = bj_card(bj_next_digit13(&w));
    c->dealer_card = d0;
    c->hidden_sum = d1;
    c->hidden_ace = d1 == 1;
    int raw = p0 + p1;
    c->usable_ace = (p0 == 1 || p1 == 1) && raw + 10 <= 21;
    c->player_sum = raw + (c->usable_ace ? 10 : 0);
  }
  c->steps = 0;
}

/* Envs.Blackjack.cs:32-40 -- the reference is a stub: ten steps, then reward 1. */
static int blackjack_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  blackjack_t* c = (blackjack_t*)e;
  int a = *(const int32_t*)action;
  c->steps++;
  if (!c->full) {
    *done = c->steps >= 10;
    *reward = *done ? 1.0 : 0.0;
    return 0;
  }
  /* EXTENSION: upstream rules (natural = False, sab = False).  Any action other than 0 hits. */
  if (a != 0) {
    /* the card is the base-13 digit that follows the action digit of the env's word of the tick
     * (stream 3): floor(26u) - 13 floor(2u) with u = w / 2^32, as the slippery outcome of FrozenLake */
    uint64_t w = gxo_rng_u32(r->seed, r->env_index, r->tick, 3, 0);
    int card = bj_card((int)((w * 26u) >> 32) - 13 * (int)((w * 2u) >> 32));
    int raw = c->player_sum - (c->usable_ace ? 10 : 0) + card;
    c->usable_ace = (c->usable_ace || card == 1) && raw + 10 <= 21;
    c->player_sum = raw + (c->usable_ace ? 10 : 0);
    *done = c->player_sum > 21;
    *reward = *done ? -1.0 : 0.0;
  } else {
    /* the dealer draws to 17 or more, then the scores are compared: draws 0..3 are the first four base-13 digits
     * of the env's dealer word of the tick (stream 5), later ones draws 0, 1, ... of the env's stream 1 */
    int raw = c->dealer_card + c->hidden_sum;
    int ace = c->dealer_card == 1 || c->hidden_ace;
    uint32_t dw = gxo_rng_u32(r->seed, r->env_index, r->tick, 5, 0);
    uint32_t k = 0;
    for (;;) {
      int sum = raw + ((ace && raw + 10 <= 21) ? 10 : 0);
      if (sum >= 17) break;
      int card = k < 4 ? bj_card(bj_next_digit13(&dw)) : bj_card(gxo_rng_below(r->seed, r->env_index, r->tick, 1, k - 4, 13));
      k++;
      raw += card;
      ace = ace || card == 1;
      c->hidden_sum += card;
      c->hidden_ace = c->hidden_ace || card == 1;
    }
    int dealer = raw + ((ace && raw + 10 <= 21) ? 10 : 0);
    int ps = c->player_sum > 21 ? 0 : c->player_sum, ds = dealer > 21 ? 0 : dealer;
    *done = 1;
    *reward = (double)((ps > ds) - (ps < ds));
  }
  return 0;
}

static void blackjack_export(const env_obj* e, void* reals, int32_t* ints) {
  (void)reals;
  const blackjack_t* c = (const blackjack_t*)e;
  ints[0] = (c->player_sum * 11 + c->dealer_card) * 2 + c->usable_ace;
  ints[1] = c->hidden_sum | (c->hidden_ace << 8);
  ints[2] = c->steps;
}
static void blackjack_import(env_obj* e, const void* reals, const int32_t* ints) {
  (void)reals;
  blackjack_t* c = (blackjack_t*)e;
  c->usable_ace = ints[0] & 1;
  c->dealer_card = (ints[0] >> 1) % 11;
  c->player_sum = (ints[0] >> 1) / 11;
  c->hidden_sum = ints[1] & 0xFF;
  c->hidden_ace = (ints[1] >> 8) & 1;
  c->steps = ints[2];
}
static const env_vtbl blackjack_vt = {blackjack_reset, blackjack_step, blackjack_export, blackjack_import, sizeof(blackjack_t)};

/* -------------------------------------------------------------------- Pong */
#define SCREEN_WIDTH 160
#define SCREEN_HEIGHT 210
#define CHANNELS 3
#define PADDLE_SPEED 4.0
#define BALL_SPEED 3.0
#define PADDLE_HEIGHT 15
#define PADDLE_WIDTH 2
#define BALL_SIZE 2

typedef struct {
  env_obj base;
  double ballX, ballY, ballVelX, ballVelY, leftPaddleY, rightPaddleY;
  int32_t leftScore, rightScore, steps;
  /* recorded NextDouble() values for the next ball reset (gxo_vec_set_rng_draws); used once */
  int forced;
  double force0, force1;
} pong_t;

/* Envs.Pong.cs:159-165; `reset_stream` = 0 from Reset, 1 from inside Step */
static void pong_reset_ball(pong_t* c, const rng_ctx* r, uint32_t stream) {
  double u0, u1;
  if (c->forced) {
    u0 = c->force0;
    u1 = c->force1;
    c->forced = 0;
  } else {
    u0 = gxo_rng_uniform(r->seed, r->env_index, r->tick, stream, 0);
    u1 = gxo_rng_uniform(r->seed, r->env_index, r->tick, stream, 1);
  }
  c->ballX = SCREEN_WIDTH / 2.0;
  c->ballY = SCREEN_HEIGHT / 2.0;
  c->ballVelX = u0 > 0.5 ? BALL_SPEED : -BALL_SPEED;
  c->ballVelY = (u1 - 0.5) * 2.0 * BALL_SPEED;
}

/* Envs.Pong.cs:54-70 */
static void pong_reset(env_obj* e, const rng_ctx* r) {
  pong_t* c = (pong_t*)e;
  pong_reset_ball(c, r, 0);
  c->leftPaddleY = SCREEN_HEIGHT / 2.0 - PADDLE_HEIGHT / 2.0;
  c->rightPaddleY = SCREEN_HEIGHT / 2.0 - PADDLE_HEIGHT / 2.0;
  c->leftScore = 0;
  c->rightScore = 0;
  c->steps = 0;
}

static inline double dmax(double a, double b) { return a >= b ? a : b; }
static inline double dmin(double a, double b) { return a <= b ? a : b; }

/* Envs.Pong.cs:72-157 (game state only; the frame is drawn by pong_render) */
static int pong_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  pong_t* c = (pong_t*)e;
  int a = *(const int32_t*)action;
  c->steps++;
  double rew = 0.0;
  switch (a) {
    case 2: c->rightPaddleY = dmax(0, c->rightPaddleY - PADDLE_SPEED); break;
    case 3: c->rightPaddleY = dmin(SCREEN_HEIGHT - PADDLE_HEIGHT, c->rightPaddleY + PADDLE_SPEED); break;
    default: break;
  }
  double ballCenterY = c->ballY + BALL_SIZE / 2.0;
  double leftPaddleCenterY = c->leftPaddleY + PADDLE_HEIGHT / 2.0;
  if (ballCenterY < leftPaddleCenterY - 2)
    c->leftPaddleY = dmax(0, c->leftPaddleY - PADDLE_SPEED * 0.8);
  else if (ballCenterY > leftPaddleCenterY + 2)
    c->leftPaddleY = dmin(SCREEN_HEIGHT - PADDLE_HEIGHT, c->leftPaddleY + PADDLE_SPEED * 0.8);

  c->ballX += c->ballVelX;
  c->ballY += c->ballVelY;

  if (c->ballY <= 0 || c->ballY >= SCREEN_HEIGHT - BALL_SIZE) {
    c->ballVelY = -c->ballVelY;
    c->ballY = dmax(0, dmin(SCREEN_HEIGHT - BALL_SIZE, c->ballY));
  }
  if (c->ballX <= PADDLE_WIDTH && c->ballY + BALL_SIZE >= c->leftPaddleY &&
      c->ballY <= c->leftPaddleY + PADDLE_HEIGHT) {
    c->ballVelX = fabs(c->ballVelX);
    double rel = (c->leftPaddleY + PADDLE_HEIGHT / 2.0) - (c->ballY + BALL_SIZE / 2.0);
    c->ballVelY = -rel * 0.1;
  }
  if (c->ballX + BALL_SIZE >= SCREEN_WIDTH - PADDLE_WIDTH && c->ballY + BALL_SIZE >= c->rightPaddleY &&
      c->ballY <= c->rightPaddleY + PADDLE_HEIGHT) {
    c->ballVelX = -fabs(c->ballVelX);
    double rel = (c->rightPaddleY + PADDLE_HEIGHT / 2.0) - (c->ballY + BALL_SIZE / 2.0);
    c->ballVelY = -rel * 0.1;
  }
  if (c->ballX < 0) {
    c->rightScore++;
    rew = 1.0;
    pong_reset_ball(c, r, 1);
  } else if (c->ballX > SCREEN_WIDTH) {
    c->leftScore++;
    rew = -1.0;
    pong_reset_ball(c, r, 1);
  }
  *done = c->leftScore >= 21 || c->rightScore >= 21 || c->steps >= 10000;
  *reward = rew;
  return 0;
}

/* Envs.Pong.cs:192-209 */
static void draw_rectangle(uint8_t* screen, int x, int y, int width, int height) {
  for (int dy = 0; dy < height; dy++)
    for (int dx = 0; dx < width; dx++) {
      int px = x + dx, py = y + dy;
      if (px >= 0 && px < SCREEN_WIDTH && py >= 0 && py < SCREEN_HEIGHT) {
        int index = (py * SCREEN_WIDTH + px) * CHANNELS;
        screen[index] = 255;
        screen[index + 1] = 255;
        screen[index + 2] = 255;
      }
    }
}

/* Envs.Pong.cs:223-251; glyph rows from the table at :228-237 */
static void draw_digit(uint8_t* screen, char digit, int x, int y) {
  static const uint8_t pat[10][5][3] = {
      {{1, 1, 1}, {1, 0, 1}, {1, 0, 1}, {1, 0, 1}, {1, 1, 1}},
      {{0, 1, 0}, {1, 1, 0}, {0, 1, 0}, {0, 1, 0}, {1, 1, 1}},
      {{1, 1, 1}, {0, 0, 1}, {1, 1, 1}, {1, 0, 0}, {1, 1, 1}},
      {{1, 1, 1}, {0, 0, 1}, {1, 1, 1}, {0, 0, 1}, {1, 1, 1}},
      {{1, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 0, 1}, {0, 0, 1}},
      {{1, 1, 1}, {1, 0, 0}, {1, 1, 1}, {0, 0, 1}, {1, 1, 1}},
      {{1, 1, 1}, {1, 0, 0}, {1, 1, 1}, {1, 0, 1}, {1, 1, 1}},
      {{1, 1, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}},
      {{1, 1, 1}, {1, 0, 1}, {1, 1, 1}, {1, 0, 1}, {1, 1, 1}},
      {{1, 1, 1}, {1, 0, 1}, {1, 1, 1}, {0, 0, 1}, {1, 1, 1}}};
  if (digit < '0' || digit > '9') return; /* '-' of a negative score: blank glyph */
  for (int dy = 0; dy < 5; dy++)
    for (int dx = 0; dx < 3; dx++)
      if (pat[digit - '0'][dy][dx]) draw_rectangle(screen, x + dx, y + dy, 1, 1);
}

/* Envs.Pong.cs:211-221 */
static void draw_score(uint8_t* screen, int score, int centerX, int y) {
  char str[16];
  int len = snprintf(str, sizeof str, "%d", score);
  int startX = centerX - (len * 6) / 2;
  for (int i = 0; i < len; i++) draw_digit(screen, str[i], startX + i * 8, y);
}

/* Envs.Pong.cs:167-190 */
void gxo_pong_render(double ball_x, double ball_y, double left_y, double right_y, int left_score,
                     int right_score, uint8_t* screen) {
  memset(screen, 0, GXO_PONG_FRAME_BYTES);
  draw_rectangle(screen, 0, (int)left_y, PADDLE_WIDTH, PADDLE_HEIGHT);
  draw_rectangle(screen, SCREEN_WIDTH - PADDLE_WIDTH, (int)right_y, PADDLE_WIDTH, PADDLE_HEIGHT);
  draw_rectangle(screen, (int)ball_x, (int)ball_y, BALL_SIZE, BALL_SIZE);
  for (int y = 0; y < SCREEN_HEIGHT; y += 8) draw_rectangle(screen, SCREEN_WIDTH / 2 - 1, y, 2, 4);
  draw_score(screen, left_score, SCREEN_WIDTH / 4, 20);
  draw_score(screen, right_score, 3 * SCREEN_WIDTH / 4, 20);
}

static void pong_export(const env_obj* e, void* reals, int32_t* ints) {
  const pong_t* c = (const pong_t*)e;
  double* d = (double*)reals;
  d[0] = c->ballX; d[1] = c->ballY; d[2] = c->ballVelX; d[3] = c->ballVelY;
  d[4] = c->leftPaddleY; d[5] = c->rightPaddleY;
  ints[0] = c->leftScore; ints[1] = c->rightScore; ints[2] = c->steps;
}
static void pong_import(env_obj* e, const void* reals, const int32_t* ints) {
  pong_t* c = (pong_t*)e;
  const double* d = (const double*)reals;
  c->ballX = d[0]; c->ballY = d[1]; c->ballVelX = d[2]; c->ballVelY = d[3];
  c->leftPaddleY = d[4]; c->rightPaddleY = d[5];
  c->leftScore = ints[0]; c->rightScore = ints[1]; c->steps = ints[2];
}
static const env_vtbl pong_vt = {pong_reset, pong_step, pong_export, pong_import, sizeof(pong_t)};

/* ---------------------------------------------------------------- Breakout */
#define BK_PADDLE_SPEED 5.0
#define BK_BALL_SPEED 3.0
#define BK_PADDLE_HEIGHT 4
#define BK_PADDLE_WIDTH 16
#define BK_BALL_SIZE 2
#define BK_BRICK_WIDTH 8
#define BK_BRICK_HEIGHT 6
#define BK_BRICK_ROWS 6
#define BK_BRICK_COLS 18
#define BK_BRICK_START_Y 60

typedef struct {
  env_obj base;
  double ballX, ballY, ballVelX, ballVelY, paddleX;
  int32_t score, lives, ballStuck;
  uint8_t bricks[BK_BRICK_ROWS][BK_BRICK_COLS];
  int32_t steps;
  /* recorded NextDouble() value for the next release of the ball (gxo_vec_set_rng_draws); used once */
  int forced;
  double force0;
} breakout_t;

/* Envs.Breakout.cs:62-87 (no random draw) */
static void breakout_reset(env_obj* e, const rng_ctx* r) {
  (void)r;
  breakout_t* c = (breakout_t*)e;
  for (int row = 0; row < BK_BRICK_ROWS; row++)
    for (int col = 0; col < BK_BRICK_COLS; col++) c->bricks[row][col] = 1;
  c->paddleX = SCREEN_WIDTH / 2.0 - BK_PADDLE_WIDTH / 2.0;
  c->ballX = SCREEN_WIDTH / 2.0;
  c->ballY = SCREEN_HEIGHT - 30;
  c->ballVelX = 0;
  c->ballVelY = 0;
  c->ballStuck = 1;
  c->score = 0;
  c->lives = 5;
  c->steps = 0;
}

/* Envs.Breakout.cs:89-213 (game state only; the frame is drawn by gxo_breakout_render) */
static int breakout_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  breakout_t* c = (breakout_t*)e;
  int a = *(const int32_t*)action;
  c->steps++;
  double rew = 0.0;
  switch (a) {
    case 1:
      if (c->ballStuck) {
        c->ballStuck = 0;
        /* the only random draw of the env: stream 1 (inside Step), draw 0 */
        double u = c->forced ? c->force0 : gxo_rng_uniform(r->seed, r->env_index, r->tick, 1, 0);
        c->forced = 0;
        c->ballVelX = (u - 0.5) * 2.0 * BK_BALL_SPEED;
        c->ballVelY = -BK_BALL_SPEED;
      }
      break;
    case 2:
      c->paddleX = dmin(SCREEN_WIDTH - BK_PADDLE_WIDTH, c->paddleX + BK_PADDLE_SPEED);
      if (c->ballStuck) c->ballX = c->paddleX + BK_PADDLE_WIDTH / 2.0;
      break;
    case 3:
      c->paddleX = dmax(0, c->paddleX - BK_PADDLE_SPEED);
      if (c->ballStuck) c->ballX = c->paddleX + BK_PADDLE_WIDTH / 2.0;
      break;
    default: break;
  }
  if (!c->ballStuck) {
    c->ballX += c->ballVelX;
    c->ballY += c->ballVelY;
    if (c->ballX <= 0 || c->ballX >= SCREEN_WIDTH - BK_BALL_SIZE) {
      c->ballVelX = -c->ballVelX;
      c->ballX = dmax(0, dmin(SCREEN_WIDTH - BK_BALL_SIZE, c->ballX));
    }
    if (c->ballY <= 0) {
      c->ballVelY = -c->ballVelY;
      c->ballY = 0;
    }
    if (c->ballY + BK_BALL_SIZE >= SCREEN_HEIGHT - 20 && c->ballY + BK_BALL_SIZE <= SCREEN_HEIGHT - 16 &&
        c->ballX + BK_BALL_SIZE >= c->paddleX && c->ballX <= c->paddleX + BK_PADDLE_WIDTH) {
      c->ballVelY = -fabs(c->ballVelY);
      double relativeIntersectX = (c->ballX + BK_BALL_SIZE / 2.0) - (c->paddleX + BK_PADDLE_WIDTH / 2.0);
      double normalizedIntersectX = relativeIntersectX / (BK_PADDLE_WIDTH / 2.0);
      c->ballVelX = normalizedIntersectX * BK_BALL_SPEED * 0.75;
    }
    /* (int) truncates toward zero: a ball with 54 < y < 60 lands in brick row 0 */
    int brickRow = (int)((c->ballY - BK_BRICK_START_Y) / BK_BRICK_HEIGHT);
    int brickCol = (int)(c->ballX / BK_BRICK_WIDTH);
    if (brickRow >= 0 && brickRow < BK_BRICK_ROWS && brickCol >= 0 && brickCol < BK_BRICK_COLS &&
        c->bricks[brickRow][brickCol]) {
      c->bricks[brickRow][brickCol] = 0;
      c->ballVelY = -c->ballVelY;
      rew = (BK_BRICK_ROWS - brickRow) + 1;
      c->score += (int)rew;
    }
    if (c->ballY > SCREEN_HEIGHT) {
      c->lives--;
      if (c->lives > 0) {
        c->ballX = c->paddleX + BK_PADDLE_WIDTH / 2.0;
        c->ballY = SCREEN_HEIGHT - 30;
        c->ballVelX = 0;
        c->ballVelY = 0;
        c->ballStuck = 1;
      }
    }
  }
  int allBricksGone = 1;
  for (int row = 0; row < BK_BRICK_ROWS && allBricksGone; row++)
    for (int col = 0; col < BK_BRICK_COLS && allBricksGone; col++)
      if (c->bricks[row][col]) allBricksGone = 0;
  if (allBricksGone) rew += 100;
  *done = c->lives <= 0 || allBricksGone || c->steps >= 10000;
  *reward = rew;
  return 0;
}

/* Envs.Breakout.cs:287-304 */
static void draw_rect_rgb(uint8_t* screen, int x, int y, int width, int height, uint8_t r, uint8_t g, uint8_t b) {
  for (int dy = 0; dy < height; dy++)
    for (int dx = 0; dx < width; dx++) {
      int px = x + dx, py = y + dy;
      if (px >= 0 && px < SCREEN_WIDTH && py >= 0 && py < SCREEN_HEIGHT) {
        int index = (py * SCREEN_WIDTH + px) * CHANNELS;
        screen[index] = r;
        screen[index + 1] = g;
        screen[index + 2] = b;
      }
    }
}

/* Envs.Breakout.cs:229-285, 306-344.  `bricks` holds bit row*18+col of word
 * (row*18+col)/32; the glyph table (:321-330) equals Pong's (Envs.Pong.cs:228-237). */
void gxo_breakout_render(double ball_x, double ball_y, double paddle_x, int ball_stuck, int score, int lives,
                         const uint32_t bricks[4], uint8_t* screen) {
  static const uint8_t rgb[BK_BRICK_ROWS][3] = {{255, 0, 0}, {255, 165, 0}, {255, 255, 0},
                                                {0, 255, 0}, {0, 0, 255},   {128, 0, 128}};
  memset(screen, 0, GXO_PONG_FRAME_BYTES);
  for (int row = 0; row < BK_BRICK_ROWS; row++)
    for (int col = 0; col < BK_BRICK_COLS; col++) {
      int bit = row * BK_BRICK_COLS + col;
      if ((bricks[bit >> 5] >> (bit & 31)) & 1u)
        draw_rect_rgb(screen, col * BK_BRICK_WIDTH, BK_BRICK_START_Y + row * BK_BRICK_HEIGHT, BK_BRICK_WIDTH - 1,
                      BK_BRICK_HEIGHT - 1, rgb[row][0], rgb[row][1], rgb[row][2]);
    }
  draw_rectangle(screen, (int)paddle_x, SCREEN_HEIGHT - 20, BK_PADDLE_WIDTH, BK_PADDLE_HEIGHT);
  if (!ball_stuck)
    draw_rectangle(screen, (int)ball_x, (int)ball_y, BK_BALL_SIZE, BK_BALL_SIZE);
  else
    draw_rectangle(screen, (int)(paddle_x + BK_PADDLE_WIDTH / 2.0 - BK_BALL_SIZE / 2.0), SCREEN_HEIGHT - 30,
                   BK_BALL_SIZE, BK_BALL_SIZE);
  char str[16];
  int len = snprintf(str, sizeof str, "%d", score);
  for (int i = 0; i < len; i++) draw_digit(screen, str[i], 10 + i * 6, 10);
  for (int i = 0; i < lives; i++) draw_rectangle(screen, 10 + i * 8, 25, 6, 3);
}

/* state image: reals ballX, ballY, ballVelX, ballVelY, paddleX; ints score, lives,
 * ballStuck, four words of brick bits, steps */
static void breakout_export(const env_obj* e, void* reals, int32_t* ints) {
  const breakout_t* c = (const breakout_t*)e;
  double* d = (double*)reals;
  d[0] = c->ballX; d[1] = c->ballY; d[2] = c->ballVelX; d[3] = c->ballVelY; d[4] = c->paddleX;
  ints[0] = c->score; ints[1] = c->lives; ints[2] = c->ballStuck;
  uint32_t w[4] = {0, 0, 0, 0};
  for (int row = 0; row < BK_BRICK_ROWS; row++)
    for (int col = 0; col < BK_BRICK_COLS; col++) {
      int bit = row * BK_BRICK_COLS + col;
      if (c->bricks[row][col]) w[bit >> 5] |= 1u << (bit & 31);
    }
  for (int k = 0; k < 4; k++) ints[3 + k] = (int32_t)w[k];
  ints[7] = c->steps;
}
static void breakout_import(env_obj* e, const void* reals, const int32_t* ints) {
  breakout_t* c = (breakout_t*)e;
  const double* d = (const double*)reals;
  c->ballX = d[0]; c->ballY = d[1]; c->ballVelX = d[2]; c->ballVelY = d[3]; c->paddleX = d[4];
  c->score = ints[0]; c->lives = ints[1]; c->ballStuck = ints[2] != 0;
  for (int row = 0; row < BK_BRICK_ROWS; row++)
    for (int col = 0; col < BK_BRICK_COLS; col++) {
      int bit = row * BK_BRICK_COLS + col;
      c->bricks[row][col] = (uint8_t)(((uint32_t)ints[3 + (bit >> 5)] >> (bit & 31)) & 1u);
    }
  c->steps = ints[7];
}
static const env_vtbl breakout_vt = {breakout_reset, breakout_step, breakout_export, breakout_import, sizeof(breakout_t)};

/* ----------------------------------------------------------- SpaceInvaders */
#define SI_PLAYER_SPEED 3.0
#define SI_BULLET_SPEED 6.0
#define SI_ALIEN_SPEED 0.5
#define SI_ALIEN_DROP_SPEED 8.0
#define SI_PLAYER_WIDTH 8
#define SI_PLAYER_HEIGHT 6
#define SI_ALIEN_WIDTH 6
#define SI_ALIEN_HEIGHT 6
#define SI_BULLET_WIDTH 2
#define SI_BULLET_HEIGHT 4
#define SI_ALIEN_ROWS 5
#define SI_ALIEN_COLS 11
#define SI_MAX_PLAYER_BULLETS 3 /* Envs.SpaceInvaders.cs:195 */
/* The reference's alien bullet list is unbounded, but a bullet is fired at most every 60 steps
 * (:136) and leaves the screen within 40 (4.2 px per step from y >= 46), so a game never holds
 * more than one; the state image has room for two and a third is dropped. */
#define SI_MAX_ALIEN_BULLETS 2

typedef struct {
  env_obj base;
  double playerX, formX, formY;
  double pbx[SI_MAX_PLAYER_BULLETS], pby[SI_MAX_PLAYER_BULLETS];
  double abx[SI_MAX_ALIEN_BULLETS], aby[SI_MAX_ALIEN_BULLETS];
  int32_t score, lives, frameCounter, moveRight, npb, nab;
  uint8_t aliens[SI_ALIEN_ROWS][SI_ALIEN_COLS];
  int32_t steps;
} invaders_t;

/* Envs.SpaceInvaders.cs:69-96 (no random draw) */
static void invaders_reset(env_obj* e, const rng_ctx* r) {
  (void)r;
  invaders_t* c = (invaders_t*)e;
  for (int row = 0; row < SI_ALIEN_ROWS; row++)
    for (int col = 0; col < SI_ALIEN_COLS; col++) c->aliens[row][col] = 1;
  c->playerX = SCREEN_WIDTH / 2.0 - SI_PLAYER_WIDTH / 2.0;
  c->formX = 10;
  c->formY = 40;
  c->moveRight = 1;
  c->npb = 0;
  c->nab = 0;
  for (int i = 0; i < SI_MAX_PLAYER_BULLETS; i++) c->pbx[i] = c->pby[i] = 0;
  for (int i = 0; i < SI_MAX_ALIEN_BULLETS; i++) c->abx[i] = c->aby[i] = 0;
  c->score = 0;
  c->lives = 3;
  c->steps = 0;
  c->frameCounter = 0;
}

/* List<T>.RemoveAt: later elements move down; the vacated slot reads 0 in the state image */
static void si_remove(double* x, double* y, int32_t* n, int i) {
  for (int k = i; k + 1 < *n; k++) { x[k] = x[k + 1]; y[k] = y[k + 1]; }
  (*n)--;
  x[*n] = 0; y[*n] = 0;
}

/* Envs.SpaceInvaders.cs:192-199 */
static void si_fire_player(invaders_t* c) {
  if (c->npb < SI_MAX_PLAYER_BULLETS) {
    c->pbx[c->npb] = c->playerX + SI_PLAYER_WIDTH / 2.0;
    c->pby[c->npb] = SCREEN_HEIGHT - 40;
    c->npb++;
  }
}

/* Envs.SpaceInvaders.cs:98-190 (game state only; the frame is drawn by gxo_invaders_render) */
static int invaders_step(env_obj* e, const void* action, const rng_ctx* r, double* reward, int* done) {
  invaders_t* c = (invaders_t*)e;
  int a = *(const int32_t*)action;
  c->steps++;
  c->frameCounter++;
  double rew = 0.0;
  switch (a) {
    case 1: si_fire_player(c); break;
    case 2: c->playerX = dmin(SCREEN_WIDTH - SI_PLAYER_WIDTH, c->playerX + SI_PLAYER_SPEED); break;
    case 3: c->playerX = dmax(0, c->playerX - SI_PLAYER_SPEED); break;
    case 4:
      c->playerX = dmin(SCREEN_WIDTH - SI_PLAYER_WIDTH, c->playerX + SI_PLAYER_SPEED);
      si_fire_player(c);
      break;
    case 5:
      c->playerX = dmax(0, c->playerX - SI_PLAYER_SPEED);
      si_fire_player(c);
      break;
    default: break;
  }
  /* UpdateBullets, :226-257 */
  for (int i = c->npb - 1; i >= 0; i--) {
    c->pby[i] -= SI_BULLET_SPEED;
    if (c->pby[i] < 0) si_remove(c->pbx, c->pby, &c->npb, i);
  }
  for (int i = c->nab - 1; i >= 0; i--) {
    c->aby[i] += SI_BULLET_SPEED * 0.7;
    if (c->aby[i] > SCREEN_HEIGHT) si_remove(c->abx, c->aby, &c->nab, i);
  }
  /* UpdateAliens every 10 frames, :259-279 */
  if (c->frameCounter % 10 == 0) {
    if (c->moveRight) {
      c->formX += SI_ALIEN_SPEED;
      if (c->formX + SI_ALIEN_COLS * 12 > SCREEN_WIDTH - 10) {
        c->moveRight = 0;
        c->formY += SI_ALIEN_DROP_SPEED;
      }
    } else {
      c->formX -= SI_ALIEN_SPEED;
      if (c->formX < 10) {
        c->moveRight = 1;
        c->formY += SI_ALIEN_DROP_SPEED;
      }
    }
  }
  /* alien shooting, :136-139 and :201-224: draw 0 decides, draw 1 picks the column (stream 1) */
  if (c->frameCounter % 60 == 0 && gxo_rng_uniform(r->seed, r->env_index, r->tick, 1, 0) < 0.3) {
    int srow[SI_ALIEN_COLS], scol[SI_ALIEN_COLS], ns = 0;
    for (int col = 0; col < SI_ALIEN_COLS; col++)
      for (int row = SI_ALIEN_ROWS - 1; row >= 0; row--)
        if (c->aliens[row][col]) { srow[ns] = row; scol[ns] = col; ns++; break; }
    if (ns > 0) {
      int k = gxo_rng_below(r->seed, r->env_index, r->tick, 1, 1, ns);
      if (c->nab < SI_MAX_ALIEN_BULLETS) {
        c->abx[c->nab] = c->formX + scol[k] * 12 + SI_ALIEN_WIDTH / 2.0;
        c->aby[c->nab] = c->formY + srow[k] * 10 + SI_ALIEN_HEIGHT;
        c->nab++;
      }
    }
  }
  /* CheckCollisions, :281-339 */
  for (int i = c->npb - 1; i >= 0; i--) {
    double bx = c->pbx[i], by = c->pby[i];
    int hit = 0;
    for (int row = 0; row < SI_ALIEN_ROWS && !hit; row++)
      for (int col = 0; col < SI_ALIEN_COLS && !hit; col++)
        if (c->aliens[row][col]) {
          double alienX = c->formX + col * 12, alienY = c->formY + row * 10;
          if (bx >= alienX && bx <= alienX + SI_ALIEN_WIDTH && by >= alienY && by <= alienY + SI_ALIEN_HEIGHT) {
            c->aliens[row][col] = 0;
            si_remove(c->pbx, c->pby, &c->npb, i);
            int points = row == 0 ? 30 : (row <= 2 ? 20 : 10);
            c->score += points;
            rew += points;
            hit = 1;
          }
        }
  }
  for (int i = c->nab - 1; i >= 0; i--) {
    double bx = c->abx[i], by = c->aby[i];
    if (bx >= c->playerX && bx <= c->playerX + SI_PLAYER_WIDTH && by >= SCREEN_HEIGHT - 40 &&
        by <= SCREEN_HEIGHT - 40 + SI_PLAYER_HEIGHT) {
      si_remove(c->abx, c->aby, &c->nab, i);
      c->lives--;
      rew -= 50;
    }
  }
  if (c->formY + (SI_ALIEN_ROWS * 8) > SCREEN_HEIGHT - 40) c->lives = 0;
  int allAliensGone = 1;
  for (int row = 0; row < SI_ALIEN_ROWS && allAliensGone; row++)
    for (int col = 0; col < SI_ALIEN_COLS && allAliensGone; col++)
      if (c->aliens[row][col]) allAliensGone = 0;
  if (allAliensGone) {
    rew += 100;
    for (int row = 0; row < SI_ALIEN_ROWS; row++)
      for (int col = 0; col < SI_ALIEN_COLS; col++) c->aliens[row][col] = 1;
    c->formY = 40;
    c->formX = 10;
  }
  *done = c->lives <= 0 || c->steps >= 10000;
  *reward = rew;
  return 0;
}

/* Envs.SpaceInvaders.cs:437-459 */
static void si_draw_player(uint8_t* screen, int x, int y) {
  static const uint8_t sprite[6][8] = {{0, 0, 0, 1, 1, 0, 0, 0}, {0, 0, 1, 1, 1, 1, 0, 0}, {0, 0, 1, 1, 1, 1, 0, 0},
                                       {0, 1, 1, 1, 1, 1, 1, 0}, {1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 1, 1, 1}};
  for (int dy = 0; dy < 6; dy++)
    for (int dx = 0; dx < 8; dx++)
      if (sprite[dy][dx]) draw_rect_rgb(screen, x + dx, y + dy, 1, 1, 0, 255, 0);
}

/* Envs.SpaceInvaders.cs:355-435, 480-525.  state image as gxo_vec_get_state lays it out:
 * reals playerX, formX, formY, 3 player bullets (x, y), 2 alien bullets (x, y);
 * ints score, lives, frameCounter, moveRight, npb, nab, aliens bits 0..31, bits 32..54, steps */
void gxo_invaders_render(const double* reals, const int32_t* ints, uint8_t* screen) {
  static const uint8_t alien[6][6] = {{0, 1, 0, 0, 1, 0}, {0, 0, 1, 1, 0, 0}, {0, 1, 1, 1, 1, 0},
                                      {1, 1, 0, 0, 1, 1}, {1, 1, 1, 1, 1, 1}, {0, 1, 0, 0, 1, 0}};
  static const uint8_t rgb[SI_ALIEN_ROWS][3] = {{255, 0, 255}, {0, 255, 255}, {255, 255, 0}, {0, 255, 0}, {255, 0, 0}};
  const double playerX = reals[0], formX = reals[1], formY = reals[2];
  const uint64_t bits = (uint64_t)(uint32_t)ints[6] | ((uint64_t)(uint32_t)ints[7] << 32);
  memset(screen, 0, GXO_PONG_FRAME_BYTES);
  for (int row = 0; row < SI_ALIEN_ROWS; row++)
    for (int col = 0; col < SI_ALIEN_COLS; col++)
      if ((bits >> (row * SI_ALIEN_COLS + col)) & 1u) {
        int x = (int)(formX + col * 12), y = (int)(formY + row * 10);
        for (int dy = 0; dy < 6; dy++)
          for (int dx = 0; dx < 6; dx++)
            if (alien[dy][dx]) draw_rect_rgb(screen, x + dx, y + dy, 1, 1, rgb[row][0], rgb[row][1], rgb[row][2]);
      }
  si_draw_player(screen, (int)playerX, SCREEN_HEIGHT - 40);
  int npb = ints[4] < 0 ? 0 : (ints[4] > SI_MAX_PLAYER_BULLETS ? SI_MAX_PLAYER_BULLETS : ints[4]);
  int nab = ints[5] < 0 ? 0 : (ints[5] > SI_MAX_ALIEN_BULLETS ? SI_MAX_ALIEN_BULLETS : ints[5]);
  for (int i = 0; i < npb; i++)
    draw_rect_rgb(screen, (int)reals[3 + 2 * i], (int)reals[4 + 2 * i], SI_BULLET_WIDTH, SI_BULLET_HEIGHT, 255, 255, 0);
  for (int i = 0; i < nab; i++)
    draw_rect_rgb(screen, (int)reals[9 + 2 * i], (int)reals[10 + 2 * i], SI_BULLET_WIDTH, SI_BULLET_HEIGHT, 255, 0, 0);
  char str[16];
  int len = snprintf(str, sizeof str, "%d", ints[0]);
  for (int i = 0; i < len; i++) draw_digit(screen, str[i], 10 + i * 6, 10);
  for (int i = 0; i < ints[1]; i++) si_draw_player(screen, (SCREEN_WIDTH - 40) - i * 10, 10);
  for (int x = 0; x < SCREEN_WIDTH; x++) draw_rect_rgb(screen, x, SCREEN_HEIGHT - 20, 1, 2, 0, 255, 0);
}

static void invaders_export(const env_obj* e, void* reals, int32_t* ints) {
  const invaders_t* c = (const invaders_t*)e;
  double* d = (double*)reals;
  d[0] = c->playerX; d[1] = c->formX; d[2] = c->formY;
  for (int i = 0; i < SI_MAX_PLAYER_BULLETS; i++) { d[3 + 2 * i] = c->pbx[i]; d[4 + 2 * i] = c->pby[i]; }
  for (int i = 0; i < SI_MAX_ALIEN_BULLETS; i++) { d[9 + 2 * i] = c->abx[i]; d[10 + 2 * i] = c->aby[i]; }
  ints[0] = c->score; ints[1] = c->lives; ints[2] = c->frameCounter; ints[3] = c->moveRight;
  ints[4] = c->npb; ints[5] = c->nab;
  uint64_t bits = 0;
  for (int row = 0; row < SI_ALIEN_ROWS; row++)
    for (int col = 0; col < SI_ALIEN_COLS; col++)
      if (c->aliens[row][col]) bits |= (uint64_t)1 << (row * SI_ALIEN_COLS + col);
  ints[6] = (int32_t)(uint32_t)bits;
  ints[7] = (int32_t)(uint32_t)(bits >> 32);
  ints[8] = c->steps;
}
static void invaders_import(env_obj* e, const void* reals, const int32_t* ints) {
  invaders_t* c = (invaders_t*)e;
  const double* d = (const double*)reals;
  c->playerX = d[0]; c->formX = d[1]; c->formY = d[2];
  for (int i = 0; i < SI_MAX_PLAYER_BULLETS; i++) { c->pbx[i] = d[3 + 2 * i]; c->pby[i] = d[4 + 2 * i]; }
  for (int i = 0; i < SI_MAX_ALIEN_BULLETS; i++) { c->abx[i] = d[9 + 2 * i]; c->aby[i] = d[10 + 2 * i]; }
  c->score = ints[0]; c->lives = ints[1]; c->frameCounter = ints[2]; c->moveRight = ints[3] != 0;
  c->npb = ints[4] < 0 ? 0 : (ints[4] > SI_MAX_PLAYER_BULLETS ? SI_MAX_PLAYER_BULLETS : ints[4]);
  c->nab = ints[5] < 0 ? 0 : (ints[5] > SI_MAX_ALIEN_BULLETS ? SI_MAX_ALIEN_BULLETS : ints[5]);
  const uint64_t bits = (uint64_t)(uint32_t)ints[6] | ((uint64_t)(uint32_t)ints[7] << 32);
  for (int row = 0; row < SI_ALIEN_ROWS; row++)
    for (int col = 0; col < SI_ALIEN_COLS; col++)
      c->aliens[row][col] = (uint8_t)((bits >> (row * SI_ALIEN_COLS + col)) & 1u);
  c->steps = ints[8];
}
static const env_vtbl invaders_vt = {invaders_reset, invaders_step, invaders_export, invaders_import, sizeof(invaders_t)};

/* ======================================================================
 * Wrappers and the batched driver.
 * ====================================================================== */
typedef struct episode_list {
  double* ret;
  int32_t* len;
  size_t n, cap;
} episode_list;

static void episode_push(episode_list* l, double ret, int32_t len) {
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 4;
    l->ret = (double*)realloc(l->ret, l->cap * sizeof(double));
    l->len = (int32_t*)realloc(l->len, l->cap * sizeof(int32_t));
  }
  l->ret[l->n] = ret;
  l->len[l->n] = len;
  l->n++;
}

struct gxo_vec {
  int env_id;
  uint64_t n;
  uint64_t seed;
  uint32_t flags;
  int32_t time_limit; /* Wrappers.TimeLimit.cs maxSteps; 0 = wrapper absent */
  uint64_t env_index_base;
  uint64_t tick;
  env_obj** envs;
  grid_desc* grid;
  table_desc* table;
  int n_real, real_is_f64, obs_is_f64, n_int, obs_dim, action_is_float;
  /* Wrappers.TimeLimit.cs:9 `_steps` per env */
  int32_t* tl_steps;
  /* EnvWrapper.cs:27-28 `_currentReward`, `_currentLength` per env */
  double* cur_reward;
  int32_t* cur_length;
  episode_list* episodes;
};

static int parse_map(grid_desc* g, const char* rows) {
  /* rows separated by '\n' or '/'; Envs.FrozenLake.cs:31-42 */
  int nrow = 0, ncol = 0, col = 0;
  g->goal = -1;
  memset(g->hole, 0, sizeof g->hole);
  for (const char* p = rows;; p++) {
    if (*p == '\n' || *p == '/' || *p == 0) {
      if (col > 0) {
        if (nrow == 0) ncol = col;
        else if (col != ncol) return -1;
        nrow++;
      }
      col = 0;
      if (*p == 0) break;
      continue;
    }
    if (nrow > 0 && col >= ncol) return -1;
    int idx = nrow * ncol + col; /* ncol is still 0 while the first row is read */
    if (idx >= GRID_MAX_CELLS) return -1;
    if (*p == 'G') g->goal = idx;
    if (*p == 'H') g->hole[idx] = 1;
    col++;
  }
  if (nrow == 0 || ncol <= 0 || nrow * ncol > GRID_MAX_CELLS) return -1;
  g->nrow = nrow;
  g->ncol = ncol;
  return 0;
}

static const char* const FROZENLAKE_4X4 = "SFFF/FHFH/FFFH/HFFG"; /* Envs.FrozenLake.cs:25-30 */

static gxo_vec* vec_create(int env_id, uint64_t num_envs, uint64_t seed, uint32_t flags, int32_t time_limit,
                           const char* map_rows, uint64_t env_index_base, table_desc* table);

gxo_vec* gxo_vec_create(int env_id, uint64_t num_envs, uint64_t seed, uint32_t flags, int32_t time_limit,
                        const char* map_rows, uint64_t env_index_base) {
  if (env_id == GXO_TABLE) return NULL; /* gxo_vec_create_table */
  return vec_create(env_id, num_envs, seed, flags, time_limit, map_rows, env_index_base, NULL);
}

gxo_vec* gxo_vec_create_table(uint64_t num_envs, uint64_t seed, uint32_t flags, int32_t time_limit, uint64_t env_index_base,
                              const gxo_transition* table, int32_t n_states, int32_t n_actions, int32_t n_outcomes,
                              int32_t start_state, int32_t step_limit) {
  if (n_outcomes == 0) n_outcomes = 1;
  if (!table || n_states < 1 || n_states > 65536 || n_actions < 1 || n_actions > 255 || (n_outcomes != 1 && n_outcomes != 3) ||
      start_state < -1 || start_state >= n_states || step_limit < 0)
    return NULL;
  const size_t len = (size_t)n_states * n_actions * n_outcomes;
  for (size_t i = 0; i < len; i++)
    if (table[i].next_state < 0 || table[i].next_state >= n_states) return NULL;
  table_desc* d = (table_desc*)calloc(1, sizeof *d);
  d->t = (gxo_transition*)malloc(len * sizeof *d->t);
  memcpy(d->t, table, len * sizeof *d->t);
  d->n_states = n_states; d->n_actions = n_actions; d->n_outcomes = n_outcomes;
  d->start_state = start_state; d->step_limit = step_limit;
  return vec_create(GXO_TABLE, num_envs, seed, flags, time_limit, NULL, env_index_base, d);
}

static gxo_vec* vec_create(int env_id, uint64_t num_envs, uint64_t seed, uint32_t flags, int32_t time_limit,
                           const char* map_rows, uint64_t env_index_base, table_desc* table) {
  if (env_id < 0 || env_id >= GXO_NUM_ENVS) return NULL;
  gxo_vec* v = (gxo_vec*)calloc(1, sizeof *v);
  v->table = table;
  v->env_id = env_id;
  v->n = num_envs;
  v->seed = seed;
  v->flags = flags;
  v->time_limit = time_limit;
  v->env_index_base = env_index_base;
  const env_vtbl* vt = NULL;
  switch (env_id) {
    case GXO_CARTPOLE: vt = &cartpole_vt; v->n_real = 4; v->n_int = 1; v->obs_dim = 4; break;
    case GXO_MOUNTAINCAR: vt = &mcar_vt; v->n_real = 2; v->n_int = 1; v->obs_dim = 2; break;
    case GXO_MOUNTAINCAR_CONT: vt = &mcc_vt; v->n_real = 2; v->n_int = 1; v->obs_dim = 2; v->action_is_float = 1; break;
    case GXO_PENDULUM: vt = &pendulum_vt; v->n_real = 3; v->n_int = 1; v->obs_dim = 3; v->action_is_float = 1; break;
    case GXO_ACROBOT:
      if (flags & GXO_ACROBOT_RK4) { vt = &acrobot64_vt; v->n_real = 4; v->real_is_f64 = 1; v->n_int = 1; v->obs_dim = 6; }
      else { vt = &acrobot_vt; v->n_real = 6; v->n_int = 1; v->obs_dim = 6; }
      break;
    case GXO_FROZENLAKE:
    case GXO_CLIFFWALKING: vt = &grid_vt; v->n_real = 0; v->n_int = 2; v->obs_dim = 1; break;
    case GXO_TAXI: vt = &taxi_vt; v->n_real = 0; v->n_int = 2; v->obs_dim = 1; break;
    case GXO_DUMMY: vt = &dummy_vt; v->n_real = 0; v->n_int = 2; v->obs_dim = 1; break;
    case GXO_PONG: vt = &pong_vt; v->n_real = 6; v->real_is_f64 = 1; v->n_int = 3; v->obs_dim = GXO_PONG_FRAME_BYTES; break;
    case GXO_BREAKOUT: vt = &breakout_vt; v->n_real = 5; v->real_is_f64 = 1; v->n_int = 8; v->obs_dim = GXO_PONG_FRAME_BYTES; break;
    case GXO_SPACEINVADERS: vt = &invaders_vt; v->n_real = 13; v->real_is_f64 = 1; v->n_int = 9; v->obs_dim = GXO_PONG_FRAME_BYTES; break;
    case GXO_BLACKJACK: vt = &blackjack_vt; v->n_real = 0; v->n_int = 3; v->obs_dim = 1; break;
    case GXO_TABLE: vt = &table_vt; v->n_real = 0; v->n_int = 2; v->obs_dim = 1; break;
  }
  int classic64 = 0;
  if ((flags & GXO_FP64) && env_id <= GXO_ACROBOT && !(env_id == GXO_ACROBOT && (flags & GXO_ACROBOT_RK4))) {
    static const env_vtbl* const vt64[5] = {&cartpole64_vt, &mcar64_vt, &mcc64_vt, &pendulum64_vt, &acrobot_e64_vt};
    vt = vt64[env_id];
    v->real_is_f64 = 1;
    v->obs_is_f64 = 1;
    classic64 = 1;
  }
  if (env_id == GXO_FROZENLAKE || env_id == GXO_CLIFFWALKING) {
    grid_desc* g = (grid_desc*)calloc(1, sizeof *g);
    if (env_id == GXO_FROZENLAKE) {
      if (parse_map(g, map_rows && *map_rows ? map_rows : FROZENLAKE_4X4) != 0) {
        free(g); free(v); return NULL;
      }
      g->start = 0;
      g->step_limit = 100; /* Envs.FrozenLake.cs:65 */
      g->is_cliff = 0;
      g->slippery = (flags & GXO_SLIPPERY) != 0;
    } else {
      /* Envs.CliffWalking.cs:15-27 */
      g->nrow = 4; g->ncol = 12; g->start = 36; g->goal = 47;
      for (int c = 1; c < 11; c++) g->hole[36 + c] = 1;
      g->step_limit = 100; /* Envs.CliffWalking.cs:49 */
      g->is_cliff = 1;
    }
    v->grid = g;
  }
  v->envs = (env_obj**)calloc(num_envs ? num_envs : 1, sizeof(env_obj*));
  for (uint64_t i = 0; i < num_envs; i++) {
    env_obj* e = (env_obj*)calloc(1, vt->size);
    e->vt = vt;
    if (v->grid) ((grid_t*)e)->g = v->grid;
    if (v->table) ((table_t*)e)->d = v->table;
    if (env_id == GXO_TAXI) ((taxi_t*)e)->full = (flags & GXO_TAXI_FULL) != 0;
    if (env_id == GXO_BLACKJACK) ((blackjack_t*)e)->full = (flags & GXO_BLACKJACK_FULL) != 0;
    if (classic64) ((classic64_t*)e)->n = v->n_real;
    v->envs[i] = e;
  }
  v->tl_steps = (int32_t*)calloc(num_envs ? num_envs : 1, sizeof(int32_t));
  v->cur_reward = (double*)calloc(num_envs ? num_envs : 1, sizeof(double));
  v->cur_length = (int32_t*)calloc(num_envs ? num_envs : 1, sizeof(int32_t));
  v->episodes = (episode_list*)calloc(num_envs ? num_envs : 1, sizeof(episode_list));
  if (env_id == GXO_PONG || env_id == GXO_BREAKOUT || env_id == GXO_SPACEINVADERS) {
    /* The Pong, Breakout and SpaceInvaders constructors call Reset() (Envs.Pong.cs:47-52,
     * Envs.Breakout.cs:54-60, Envs.SpaceInvaders.cs:59-67).  Pong's draw is unobservable once the batch is
     * reset, so construction leaves tick at 0. */
    for (uint64_t i = 0; i < num_envs; i++) {
      rng_ctx r = {seed, env_index_base + i, 0};
      vt->reset(v->envs[i], &r);
    }
  }
  return v;
}

void gxo_vec_destroy(gxo_vec* v) {
  if (!v) return;
  for (uint64_t i = 0; i < v->n; i++) {
    free(v->envs[i]);
    free(v->episodes[i].ret);
    free(v->episodes[i].len);
  }
  free(v->envs);
  free(v->grid);
  if (v->table) free(v->table->t);
  free(v->table);
  free(v->tl_steps);
  free(v->cur_reward);
  free(v->cur_length);
  free(v->episodes);
  free(v);
}

uint64_t gxo_vec_num_envs(const gxo_vec* v) { return v->n; } /* VectorEnv.cs:13 */
int gxo_vec_obs_dim(const gxo_vec* v) { return v->obs_dim; }
int gxo_vec_n_real(const gxo_vec* v) { return v->n_real; }
int gxo_vec_real_is_f64(const gxo_vec* v) { return v->real_is_f64; }
int gxo_vec_obs_is_f64(const gxo_vec* v) { return v->obs_is_f64; }
int gxo_vec_n_int(const gxo_vec* v) { return v->n_int; }
uint64_t gxo_vec_tick(const gxo_vec* v) { return v->tick; }
size_t gxo_vec_state_bytes(const gxo_vec* v) {
  return (size_t)v->n * ((size_t)v->n_real * (v->real_is_f64 ? 8 : 4) + (size_t)v->n_int * 4);
}

/* Reset through the wrapper chain RecordEpisodeStatistics(TimeLimit(env)):
 * EnvWrapper.cs:32-42 then Wrappers.TimeLimit.cs:17-21 then the env. */
static void wrapped_reset(gxo_vec* v, uint64_t i, const rng_ctx* r) {
  if (v->flags & GXO_RECORD_STATS) {
    if (v->cur_length[i] > 0) episode_push(&v->episodes[i], v->cur_reward[i], v->cur_length[i]);
    v->cur_reward[i] = 0;
    v->cur_length[i] = 0;
  }
  v->tl_steps[i] = 0;
  v->envs[i]->vt->reset(v->envs[i], r);
}

/* VectorEnv.cs:20-26 */
/* Recorded draws serve one pass over the batch (one Reset or one Step of every env): what a pass did not use
 * is dropped. */
static void pong_drop_draws(gxo_vec* v) {
  for (uint64_t i = 0; i < v->n; i++) {
    if (v->env_id == GXO_PONG) ((pong_t*)v->envs[i])->forced = 0;
    if (v->env_id == GXO_BREAKOUT) ((breakout_t*)v->envs[i])->forced = 0;
  }
}

/* The _random.NextDouble() values of env i's next in-Step draw, in place of the restatement's generator, for the
 * next gxo_vec_reset or step only: Pong u[2][N], the ball reset (Envs.Pong.cs:58-59, 163-164); Breakout u[1][N],
 * the release of the ball (Envs.Breakout.cs:100).  Returns -1 for another env. */
int gxo_vec_set_rng_draws(gxo_vec* v, const double* u) {
  if (v->env_id != GXO_PONG && v->env_id != GXO_BREAKOUT) return -1;
  for (uint64_t i = 0; i < v->n; i++) {
    if (v->env_id == GXO_PONG) {
      pong_t* c = (pong_t*)v->envs[i];
      c->forced = u != NULL;
      if (u) {
        c->force0 = u[i];
        c->force1 = u[v->n + i];
      }
    } else {
      breakout_t* c = (breakout_t*)v->envs[i];
      c->forced = u != NULL;
      if (u) c->force0 = u[i];
    }
  }
  return 0;
}

void gxo_vec_reset(gxo_vec* v) {
  for (uint64_t i = 0; i < v->n; i++) {
    rng_ctx r = {v->seed, v->env_index_base + i, v->tick};
    wrapped_reset(v, i, &r);
  }
  pong_drop_draws(v);
  v->tick++;
}

static int vec_step_impl(gxo_vec* v, const void* actions, int random_policy, void* actions_out,
                         double* reward, uint8_t* done, uint8_t* truncated) {
  /* VectorEnv.cs:28-34: sequential loop, one virtual Step per env */
  for (uint64_t i = 0; i < v->n; i++) {
    rng_ctx r = {v->seed, v->env_index_base + i, v->tick};
    int32_t ai = 0;
    float af = 0.0f;
    const void* ap;
    if (random_policy) {
      double u;
      switch (v->env_id) {
        case GXO_MOUNTAINCAR_CONT: /* Box([-1],[1]) Envs.MountainCarContinuous.cs:12 */
          u = gxo_rng_uniform(r.seed, r.env_index, r.tick, 2, 0);
          af = gxo_box_sample(u, -1.0f, 1.0f);
          break;
        case GXO_PENDULUM: /* Box([-2],[2]) Envs.Pendulum.cs:12 */
          u = gxo_rng_uniform(r.seed, r.env_index, r.tick, 2, 0);
          af = gxo_box_sample(u, -2.0f, 2.0f);
          break;
        default: {
          static const int nact[GXO_NUM_ENVS] = {2, 3, 0, 0, 3, 4, 6, 4, 4, 3, 4, 6, 2, 0};
          const int tabular = v->env_id == GXO_FROZENLAKE || v->env_id == GXO_TAXI || v->env_id == GXO_CLIFFWALKING || v->env_id == GXO_DUMMY ||
                              v->env_id == GXO_BLACKJACK || v->env_id == GXO_TABLE;
          ai = gxo_rng_below(r.seed, r.env_index, r.tick, tabular ? 3 : 2, 0, v->table ? v->table->n_actions : nact[v->env_id]);
        }
      }
      if (v->action_is_float) {
        ap = &af;
        if (actions_out) ((float*)actions_out)[i] = af;
      } else {
        ap = &ai;
        if (actions_out) ((int32_t*)actions_out)[i] = ai;
      }
    } else {
      ap = v->action_is_float ? (const void*)((const float*)actions + i) : (const void*)((const int32_t*)actions + i);
    }
    double rew = 0;
    int d = 0, trunc = 0;
    if (v->envs[i]->vt->step(v->envs[i], ap, &r, &rew, &d) != 0) return -1;
    /* Wrappers.TimeLimit.cs:23-33 */
    v->tl_steps[i]++;
    if (v->time_limit > 0 && v->tl_steps[i] >= v->time_limit) {
      d = 1;
      trunc = 1;
    }
    /* EnvWrapper.cs:44-57 */
    if (v->flags & GXO_RECORD_STATS) {
      v->cur_reward[i] += rew;
      v->cur_length[i]++;
      if (d) {
        episode_push(&v->episodes[i], v->cur_reward[i], v->cur_length[i]);
        v->cur_reward[i] = 0;
        v->cur_length[i] = 0;
      }
    }
    if (reward) reward[i] = rew;
    if (done) done[i] = (uint8_t)d;
    if (truncated) truncated[i] = (uint8_t)trunc;
    if (d && (v->flags & GXO_AUTORESET)) wrapped_reset(v, i, &r);
  }
  pong_drop_draws(v);
  v->tick++;
  return 0;
}

int gxo_vec_step(gxo_vec* v, const void* actions, double* reward, uint8_t* done, uint8_t* truncated) {
  return vec_step_impl(v, actions, 0, NULL, reward, done, truncated);
}
int gxo_vec_step_random(gxo_vec* v, void* actions_out, double* reward, uint8_t* done, uint8_t* truncated) {
  return vec_step_impl(v, NULL, 1, actions_out, reward, done, truncated);
}

void gxo_vec_get_state(const gxo_vec* v, void* buf) {
  size_t rsz = v->real_is_f64 ? 8 : 4;
  char* reals = (char*)buf;
  int32_t* ints = (int32_t*)(reals + (size_t)v->n * v->n_real * rsz);
  double tmp[16];
  int32_t itmp[12];
  for (uint64_t i = 0; i < v->n; i++) {
    v->envs[i]->vt->export_state(v->envs[i], tmp, itmp);
    for (int k = 0; k < v->n_real; k++)
      memcpy(reals + ((size_t)k * v->n + i) * rsz, (char*)tmp + (size_t)k * rsz, rsz);
    for (int k = 0; k < v->n_int; k++) ints[(size_t)k * v->n + i] = itmp[k];
  }
}

void gxo_vec_set_state(gxo_vec* v, const void* buf) {
  size_t rsz = v->real_is_f64 ? 8 : 4;
  const char* reals = (const char*)buf;
  const int32_t* ints = (const int32_t*)(reals + (size_t)v->n * v->n_real * rsz);
  double tmp[16];
  int32_t itmp[12];
  for (uint64_t i = 0; i < v->n; i++) {
    for (int k = 0; k < v->n_real; k++)
      memcpy((char*)tmp + (size_t)k * rsz, reals + ((size_t)k * v->n + i) * rsz, rsz);
    for (int k = 0; k < v->n_int; k++) itmp[k] = ints[(size_t)k * v->n + i];
    v->envs[i]->vt->import_state(v->envs[i], tmp, itmp);
    /* the TimeLimit counter equals the env's own step counter at all times, and so does the
     * length of the open episode when every done step resets the env; the return collected
     * so far is kept */
    v->tl_steps[i] = itmp[v->n_int - 1];
    if (v->flags & GXO_AUTORESET) v->cur_length[i] = itmp[v->n_int - 1];
  }
}

void gxo_vec_get_obs(const gxo_vec* v, void* buf) {
  double tmp[16];
  int32_t itmp[12];
  for (uint64_t i = 0; i < v->n; i++) {
    v->envs[i]->vt->export_state(v->envs[i], tmp, itmp);
    if (v->env_id == GXO_PONG) {
      gxo_pong_render(tmp[0], tmp[1], tmp[4], tmp[5], itmp[0], itmp[1],
                      (uint8_t*)buf + (size_t)i * GXO_PONG_FRAME_BYTES);
    } else if (v->env_id == GXO_BREAKOUT) {
      gxo_breakout_render(tmp[0], tmp[1], tmp[4], itmp[2], itmp[0], itmp[1], (const uint32_t*)&itmp[3],
                          (uint8_t*)buf + (size_t)i * GXO_PONG_FRAME_BYTES);
    } else if (v->env_id == GXO_SPACEINVADERS) {
      gxo_invaders_render(tmp, itmp, (uint8_t*)buf + (size_t)i * GXO_PONG_FRAME_BYTES);
    } else if (v->env_id == GXO_ACROBOT && v->real_is_f64 && !v->obs_is_f64) {
      float* o = (float*)buf;
      o[0 * v->n + i] = (float)cos(tmp[0]);
      o[1 * v->n + i] = (float)sin(tmp[0]);
      o[2 * v->n + i] = (float)cos(tmp[1]);
      o[3 * v->n + i] = (float)sin(tmp[1]);
      o[4 * v->n + i] = (float)tmp[2];
      o[5 * v->n + i] = (float)tmp[3];
    } else if (v->obs_is_f64) {
      for (int k = 0; k < v->n_real; k++) ((double*)buf)[(size_t)k * v->n + i] = tmp[k];
    } else if (v->n_real > 0) {
      for (int k = 0; k < v->n_real; k++) ((float*)buf)[(size_t)k * v->n + i] = ((float*)tmp)[k];
    } else {
      ((int32_t*)buf)[i] = itmp[0];
    }
  }
}

void gxo_vec_stats(const gxo_vec* v, gxo_stats* out) {
  gxo_stats s = {0, 0.0, 0, 0.0, INFINITY, -INFINITY};
  for (uint64_t i = 0; i < v->n; i++) {
    const episode_list* l = &v->episodes[i];
    for (size_t k = 0; k < l->n; k++) {
      s.episodes++;
      s.sum_return += l->ret[k];
      s.sum_length += l->len[k];
      s.sum_return_sq += l->ret[k] * l->ret[k];
      if (l->ret[k] < s.min_return) s.min_return = l->ret[k];
      if (l->ret[k] > s.max_return) s.max_return = l->ret[k];
    }
  }
  *out = s;
}

size_t gxo_vec_episode_count(const gxo_vec* v, uint64_t env) { return v->episodes[env].n; }
void gxo_vec_episode_get(const gxo_vec* v, uint64_t env, size_t k, double* ret, int32_t* len) {
  *ret = v->episodes[env].ret[k];
  *len = v->episodes[env].len[k];
}

/* ======================================================================
 * CPU baseline, shaped like the reference: a list of heap objects, one
 * virtual Step per env per batch step, a freshly allocated result list per
 * call (VectorEnv.cs:28-34), single thread.  It omits the per-env-step
 * Dictionary allocation and tuple boxing of the reference
 * (Envs.CartPole.cs:73, VectorEnv.cs:30-32), so it flatters the reference.
 * ====================================================================== */
typedef struct step_result {
  double state[16];
  double reward;
  int done;
  void* info;
} step_result;

double gxo_bench_reference_shaped(int env_id, uint64_t num_envs, int steps, uint64_t seed, uint64_t* env_steps,
                                  double* checksum) {
  gxo_vec* v = gxo_vec_create(env_id, num_envs, seed, 0, 0, NULL, 0);
  if (!v) return -1.0;
  gxo_vec_reset(v);
  void* actions = malloc(num_envs * 4);
  static const int nact[GXO_NUM_ENVS] = {2, 3, 0, 0, 3, 4, 6, 4, 4, 3, 4, 6, 2};
  for (uint64_t i = 0; i < num_envs; i++) {
    if (v->action_is_float) {
      float lim = env_id == GXO_PENDULUM ? 2.0f : 1.0f;
      ((float*)actions)[i] = gxo_box_sample(gxo_rng_uniform(seed, i, 0, 2, 0), -lim, lim);
    } else {
      ((int32_t*)actions)[i] = gxo_rng_below(seed, i, 0, 2, 0, nact[env_id]);
    }
  }
  /* every Pong object owns its 210x160x3 `_screen` (Envs.Pong.cs:49) */
  uint8_t* screens = (env_id == GXO_PONG || env_id == GXO_BREAKOUT || env_id == GXO_SPACEINVADERS) ? (uint8_t*)malloc(num_envs * (size_t)GXO_PONG_FRAME_BYTES) : NULL;
  double acc = 0.0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int s = 0; s < steps; s++) {
    step_result* results = (step_result*)malloc(num_envs * sizeof(step_result));
    for (uint64_t i = 0; i < num_envs; i++) {
      rng_ctx r = {seed, i, v->tick};
      env_obj* e = v->envs[i];
      const void* ap = (const char*)actions + i * 4;
      int32_t ints[12];
      e->vt->step(e, ap, &r, &results[i].reward, &results[i].done);
      e->vt->export_state(e, results[i].state, ints);
      results[i].info = NULL;
      if (results[i].done) e->vt->reset(e, &r); /* the caller's episode loop */
      if (screens) { /* Envs.Pong.cs:146 RenderScreen() on every step */
        const double* st = results[i].state;
        e->vt->export_state(e, results[i].state, ints);
        if (env_id == GXO_PONG)
          gxo_pong_render(st[0], st[1], st[4], st[5], ints[0], ints[1], screens + i * (size_t)GXO_PONG_FRAME_BYTES);
        else if (env_id == GXO_BREAKOUT)
          gxo_breakout_render(st[0], st[1], st[4], ints[2], ints[0], ints[1], (const uint32_t*)&ints[3],
                              screens + i * (size_t)GXO_PONG_FRAME_BYTES);
        else
          gxo_invaders_render(st, ints, screens + i * (size_t)GXO_PONG_FRAME_BYTES);
        acc += screens[i * (size_t)GXO_PONG_FRAME_BYTES + 60];
      }
    }
    v->tick++;
    acc += results[num_envs - 1].reward + results[0].state[0];
    free(results);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (env_steps) *env_steps = num_envs * (uint64_t)steps;
  if (checksum) *checksum = acc;
  free(actions);
  free(screens);
  gxo_vec_destroy(v);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ======================================================================
 * CPU baseline, all cores: the same env code without what the reference's
 * VectorEnv loop adds -- env structs in one contiguous array per thread (no
 * heap object per env), no per-call result list, `threads` POSIX threads each
 * stepping its own slice for `steps` batched steps.  An upper bound on what
 * the host could do with this arithmetic, reported next to the 1-core figure.
 * ====================================================================== */
typedef struct soa_job {
  const env_vtbl* vt;
  char* envs;         /* count structs of vt->size bytes */
  const char* actions;
  uint64_t first, count, seed;
  int steps;
  double acc;
} soa_job;

static void* soa_worker(void* p) {
  soa_job* j = (soa_job*)p;
  double acc = 0.0;
  for (int s = 0; s < j->steps; s++) {
    for (uint64_t i = 0; i < j->count; i++) {
      env_obj* e = (env_obj*)(j->envs + i * j->vt->size);
      rng_ctx r = {j->seed, j->first + i, (uint64_t)s + 1};
      double reward;
      int done;
      j->vt->step(e, j->actions + (j->first + i) * 4, &r, &reward, &done);
      acc += reward;
      if (done) j->vt->reset(e, &r);
    }
  }
  j->acc = acc;
  return NULL;
}

double gxo_bench_flat_threads(int env_id, uint64_t num_envs, int steps, int threads, uint64_t seed, uint64_t* env_steps,
                              double* checksum) {
  gxo_vec* v = gxo_vec_create(env_id, threads > 0 ? (uint64_t)threads : 1, seed, 0, 0, NULL, 0);
  if (!v || threads <= 0 || v->grid) { /* grid envs share a map object; not needed for the baseline */
    gxo_vec_destroy(v);
    return -1.0;
  }
  const env_vtbl* vt = v->envs[0]->vt;
  static const int nact[GXO_NUM_ENVS] = {2, 3, 0, 0, 3, 4, 6, 4, 4, 3, 4, 6};
  char* actions = (char*)malloc(num_envs * 4);
  for (uint64_t i = 0; i < num_envs; i++) {
    if (v->action_is_float) {
      float lim = env_id == GXO_PENDULUM ? 2.0f : 1.0f;
      ((float*)actions)[i] = gxo_box_sample(gxo_rng_uniform(seed, i, 0, 2, 0), -lim, lim);
    } else {
      ((int32_t*)actions)[i] = gxo_rng_below(seed, i, 0, 2, 0, nact[env_id]);
    }
  }
  soa_job* jobs = (soa_job*)calloc((size_t)threads, sizeof *jobs);
  pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof *tid);
  uint64_t per = (num_envs + (uint64_t)threads - 1) / (uint64_t)threads;
  for (int t = 0; t < threads; t++) {
    uint64_t first = (uint64_t)t * per, count = first >= num_envs ? 0 : (num_envs - first < per ? num_envs - first : per);
    jobs[t].vt = vt; jobs[t].first = first; jobs[t].count = count; jobs[t].seed = seed; jobs[t].steps = steps;
    jobs[t].actions = actions;
    jobs[t].envs = (char*)calloc(count ? count : 1, vt->size);
    for (uint64_t i = 0; i < count; i++) {
      env_obj* e = (env_obj*)(jobs[t].envs + i * vt->size);
      memcpy(e, v->envs[0], vt->size); /* vtable pointer and constructor state */
      rng_ctx r = {seed, first + i, 0};
      vt->reset(e, &r);
    }
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; t++) pthread_create(&tid[t], NULL, soa_worker, &jobs[t]);
  double acc = 0.0;
  for (int t = 0; t < threads; t++) {
    pthread_join(tid[t], NULL);
    acc += jobs[t].acc;
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  for (int t = 0; t < threads; t++) free(jobs[t].envs);
  free(jobs); free(tid); free(actions);
  gxo_vec_destroy(v);
  if (env_steps) *env_steps = num_envs * (uint64_t)steps;
  if (checksum) *checksum = acc;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
