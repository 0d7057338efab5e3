"""Second, independent restatement of the reference step path in numpy.

TEST INFRASTRUCTURE ONLY.  It exists to pin oracle/gym_oracle.c: the
reference's own tests hold no numeric vectors for these environments, so two
restatements written separately from the reference source (this one is
vectorised over the batch and structured around arrays, the C one around
per-env objects) must agree bit-for-bit (tests/test_oracle_cross.py).

numpy float32 arithmetic rounds to binary32 after every operation, which is
what CoreCLR does for C# `float`.  Math.Sin/Cos/Atan2 are evaluated with
Python's math module (the platform libm, as CoreCLR uses on Linux) on the
float64 value and rounded once to float32 -- numpy's own vectorised sin/cos
are a different implementation and are deliberately not used.

Citations are into the reference's Gymnasium/ directory.
"""
import math

import numpy as np

f32 = np.float32
f64 = np.float64
M32 = np.uint64(0xFFFFFFFF)


# --------------------------------------------------------------- counter RNG
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 on uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(x, np.uint64) for x in (c0, c1, c2, c3))
    k0 = np.uint64(k0)
    k1 = np.uint64(k1)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & M32
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c0, c1, c2, c3


def rng_u32(seed, env, tick, stream, draw):
    """word (draw & 3) of the Philox block (draw >> 2) of this (env, tick, stream); the policy
    stream (2) has one draw per tick, word tick % 4 of the block of tick // 4; the per-tick word of
    the table-driven envs (stream 3) and Blackjack's deal and dealer words (4, 5) are word env % 4 of the
    block of env // 4"""
    if stream == 2:
        draw, tick = tick & 3, tick >> 2
    env = np.asarray(env, np.uint64)
    if stream >= 3:
        sel = (env & np.uint64(3)).astype(np.int64)
        env = env >> np.uint64(2)
        tick = int(tick)
        c3 = np.uint64((stream << 24) | (((tick >> 32) & 0xFF) << 16))
        o = philox4x32_10(env & M32, env >> np.uint64(32), np.full_like(env, tick & 0xFFFFFFFF),
                          np.full_like(env, c3), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        return np.choose(sel, o)
    tick = int(tick)
    c3 = np.uint64((stream << 24) | (((tick >> 32) & 0xFF) << 16) | ((draw >> 2) & 0xFFFF))
    o = philox4x32_10(env & M32, env >> np.uint64(32), np.full_like(env, tick & 0xFFFFFFFF),
                      np.full_like(env, c3), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return o[draw & 3]


def rng_uniform(seed, env, tick, stream, draw):
    return rng_u32(seed, env, tick, stream, draw).astype(f64) * 2.0 ** -32


def rng_below(seed, env, tick, stream, draw, n):
    return ((rng_u32(seed, env, tick, stream, draw) * np.uint64(n)) >> np.uint64(32)).astype(np.int32)


# -------------------------------------------------------------- libm helpers
def _map1(fn, x):
    x = np.asarray(x, f64)
    return np.array([fn(float(v)) for v in x.ravel()], f64).reshape(x.shape)


def cos32(x64):
    return _map1(math.cos, x64).astype(f32)


def sin32(x64):
    return _map1(math.sin, x64).astype(f32)


def atan2_32(y, x):
    y = np.asarray(y, f64)
    x = np.asarray(x, f64)
    return np.array([math.atan2(float(a), float(b)) for a, b in zip(y.ravel(), x.ravel())], f64).reshape(y.shape).astype(f32)


def clamp(v, lo, hi):
    # Math.Clamp: value < min -> min; value > max -> max; else value
    return np.where(v < lo, lo, np.where(v > hi, hi, v)).astype(v.dtype)


# ---------------------------------------------------------------- CartPole
def cartpole_reset(seed, env, tick):
    """Envs.CartPole.cs:22-32"""
    return np.stack([(rng_uniform(seed, env, tick, 0, i) * 0.1 - 0.05).astype(f32) for i in range(4)])


def cartpole_step(s, steps, action):
    """Envs.CartPole.cs:34-75.  s: float32 [4, N]"""
    gravity, masscart, masspole = f32(9.8), f32(1.0), f32(0.1)
    total_mass = f32(masscart + masspole)
    length = f32(0.5)
    polemass_length = f32(masspole * length)
    force_mag, tau = f32(10.0), f32(0.02)
    thr = f32(f32(f32(12) * f32(math.pi)) / f32(180))
    x_thr = f32(2.4)
    x, xd, th, thd = (s[i].astype(f32) for i in range(4))
    force = np.where(np.asarray(action) == 1, force_mag, -force_mag).astype(f32)
    c = cos32(th.astype(f64))
    sn = sin32(th.astype(f64))
    temp = (force + polemass_length * thd * thd * sn) / total_mass
    thacc = (gravity * sn - c * temp) / (length * (f32(f32(4.0) / f32(3.0)) - masspole * c * c / total_mass))
    xacc = temp - polemass_length * thacc * c / total_mass
    x = x + tau * xd
    xd = xd + tau * xacc
    th = th + tau * thd
    thd = thd + tau * thacc
    steps = steps + 1
    done = (x < -x_thr) | (x > x_thr) | (th < -thr) | (th > thr) | (steps >= 500)
    reward = np.where(done, 0.0, 1.0)
    return np.stack([x, xd, th, thd]).astype(f32), steps, reward, done


# ------------------------------------------------------------- MountainCar
def mountaincar_reset(seed, env, tick):
    """Envs.MountainCar.cs:22-30 / Envs.MountainCarContinuous.cs:19-27"""
    p = (rng_uniform(seed, env, tick, 0, 0) * 0.2 - 0.6).astype(f32)
    return np.stack([p, np.zeros_like(p)])


def _mcar_tail(p, v, goal):
    v = clamp(v, f32(-0.07), f32(0.07))
    p = (p + v).astype(f32)
    wall = p < f32(-1.2)
    p = np.where(wall, f32(-1.2), p).astype(f32)
    v = np.where(wall, f32(0.0), v).astype(f32)
    return p, v, p >= f32(goal)


def mountaincar_step(s, steps, action):
    """Envs.MountainCar.cs:32-50"""
    p, v = s[0].astype(f32), s[1].astype(f32)
    force = (np.asarray(action, np.int64) - 1).astype(f32)
    v = v + (f32(0.001) * force - f32(0.0025) * cos32((f32(3) * p).astype(f64)))
    p, v, done = _mcar_tail(p, v.astype(f32), 0.5)
    return np.stack([p, v]), steps + 1, np.where(done, 0.0, -1.0), done


def mountaincar_cont_step(s, steps, action):
    """Envs.MountainCarContinuous.cs:29-49"""
    p, v = s[0].astype(f32), s[1].astype(f32)
    force = clamp(np.asarray(action, f32), f32(-1.0), f32(1.0))
    v = v + (force * f32(0.0015) - f32(0.0025) * cos32((f32(3) * p).astype(f64)))
    p, v, done = _mcar_tail(p, v.astype(f32), 0.45)
    reward = v.astype(f64) * 0.1
    reward = np.where(done, reward + 100.0, reward)
    return np.stack([p, v]), steps + 1, reward, done


# ---------------------------------------------------------------- Pendulum
def pendulum_reset(seed, env, tick):
    """Envs.Pendulum.cs:19-28"""
    th = (rng_uniform(seed, env, tick, 0, 0) * 2 * math.pi - math.pi).astype(f32)
    thd = (rng_uniform(seed, env, tick, 0, 1) * 1 - 0.5).astype(f32)
    return np.stack([cos32(th.astype(f64)), sin32(th.astype(f64)), thd])


def pendulum_step(s, steps, action):
    """Envs.Pendulum.cs:30-60"""
    dt = f32(0.05)
    th = atan2_32(s[1], s[0])
    thd = s[2].astype(f32)
    u = clamp(np.asarray(action, f32), f32(-2.0), f32(2.0))
    k1 = f32(f32(-3 * f32(10.0)) / f32(2 * f32(1.0)))  # -3*g/(2*l)
    k2 = f32(f32(3.0) / f32(1.0))  # 3.0f/(m*l*l)
    nthd = thd + (k1 * sin32(th.astype(f64) + math.pi) + k2 * u) * dt
    nthd = clamp(nthd.astype(f32), f32(-8.0), f32(8.0))
    nth = (th + nthd * dt).astype(f32)
    reward = -((th * th).astype(f64) + 0.1 * thd.astype(f64) * thd.astype(f64) + 0.001 * u.astype(f64) * u.astype(f64))
    ns = np.stack([cos32(nth.astype(f64)), sin32(nth.astype(f64)), nthd])
    return ns, steps + 1, reward, np.zeros(th.shape, bool)


# ----------------------------------------------------------------- Acrobot
def acrobot_reset(seed, env, tick):
    """Envs.Acrobot.cs:22-28"""
    return np.stack([(rng_uniform(seed, env, tick, 0, i) * 0.1 - 0.05).astype(f32) for i in range(6)])


def acrobot_step(s, steps, action):
    """Envs.Acrobot.cs:30-91"""
    action = np.asarray(action)
    if ((action < 0) | (action > 2)).any():
        raise IndexError("torque[action]")  # Envs.Acrobot.cs:51
    one, half, g, dt = f32(1.0), f32(0.5), f32(9.8), f32(0.2)
    mv1 = f32(f32(4) * f32(math.pi))
    mv2 = f32(f32(9) * f32(math.pi))
    t1 = atan2_32(s[1], s[0])
    t2 = atan2_32(s[3], s[2])
    w1, w2 = s[4].astype(f32), s[5].astype(f32)
    u = (action.astype(np.int64) - 1).astype(f32)
    c2 = cos32(t2.astype(f64))
    s2 = sin32(t2.astype(f64))
    lc2 = f32(half * half)  # lc^2 = 0.25
    d1 = f32(one * half * half) + one * (f32(one * one + lc2) + f32(2 * one * half) * c2) + one + one
    d2 = one * (lc2 + f32(one * half) * c2) + one
    phi2 = f32(f32(one * half) * g) * cos32((t1 + t2).astype(f64) - math.pi / 2.0)
    phi1 = (f32(-one * one * half) * w2 * w2 * s2 - f32(2 * one * one * half) * w2 * w1 * s2
            + f32(f32(one * half + one * one) * g) * cos32(t1.astype(f64) - math.pi / 2.0) + phi2)
    a2 = (u + d2 / d1 * phi1 - f32(one * one * half) * w1 * w1 * s2 - phi2) / (f32(f32(one * half * half) + one) - d2 * d2 / d1)
    a1 = -(d2 * a2 + phi1) / d1
    w1 = clamp((w1 + dt * a1).astype(f32), -mv1, mv1)
    w2 = clamp((w2 + dt * a2).astype(f32), -mv2, mv2)
    t1 = (t1 + dt * w1).astype(f32)
    t2 = (t2 + dt * w2).astype(f32)
    ct1 = cos32(t1.astype(f64))
    ns = np.stack([ct1, sin32(t1.astype(f64)), cos32(t2.astype(f64)), sin32(t2.astype(f64)), w1, w2]).astype(f32)
    steps = steps + 1
    tip = -one * ct1 - one * cos32((t1 + t2).astype(f64))
    done = (tip > one) | (steps >= 500)
    return ns, steps, np.where(done, 0.0, -1.0), done


# ------------------------------------------- classic control in binary64
# Extension (the engine's FP64 mode; the reference computes in binary32): the
# same sources with every float read as double and the (float) casts dropped.
def _cos(x):
    return _map1(math.cos, x)


def _sin(x):
    return _map1(math.sin, x)


def _atan2(y, x):
    y = np.asarray(y, f64)
    x = np.asarray(x, f64)
    return np.array([math.atan2(float(a), float(b)) for a, b in zip(y.ravel(), x.ravel())], f64).reshape(y.shape)


def cartpole64_reset(seed, env, tick):
    return np.stack([rng_uniform(seed, env, tick, 0, i) * 0.1 - 0.05 for i in range(4)])


def cartpole64_step(s, steps, action):
    """Envs.CartPole.cs:34-75 in double"""
    total_mass = 1.0 + 0.1
    pml = 0.1 * 0.5
    thr = 12 * math.pi / 180
    x, xd, th, thd = (s[i].astype(f64) for i in range(4))
    force = np.where(np.asarray(action) == 1, 10.0, -10.0)
    c, sn = _cos(th), _sin(th)
    temp = (force + pml * thd * thd * sn) / total_mass
    thacc = (9.8 * sn - c * temp) / (0.5 * (4.0 / 3.0 - 0.1 * c * c / total_mass))
    xacc = temp - pml * thacc * c / total_mass
    x = x + 0.02 * xd
    xd = xd + 0.02 * xacc
    th = th + 0.02 * thd
    thd = thd + 0.02 * thacc
    steps = steps + 1
    done = (x < -2.4) | (x > 2.4) | (th < -thr) | (th > thr) | (steps >= 500)
    return np.stack([x, xd, th, thd]), steps, np.where(done, 0.0, 1.0), done


def mountaincar64_reset(seed, env, tick):
    p = rng_uniform(seed, env, tick, 0, 0) * 0.2 - 0.6
    return np.stack([p, np.zeros_like(p)])


def _mcar64_tail(p, v, goal):
    v = clamp(v, -0.07, 0.07)
    p = p + v
    wall = p < -1.2
    p = np.where(wall, -1.2, p)
    v = np.where(wall, 0.0, v)
    return p, v, p >= goal


def mountaincar64_step(s, steps, action):
    """Envs.MountainCar.cs:32-50 in double"""
    p, v = s[0].astype(f64), s[1].astype(f64)
    force = (np.asarray(action, np.int64) - 1).astype(f64)
    v = v + (0.001 * force - 0.0025 * _cos(3 * p))
    p, v, done = _mcar64_tail(p, v, 0.5)
    return np.stack([p, v]), steps + 1, np.where(done, 0.0, -1.0), done


def mountaincar_cont64_step(s, steps, action):
    """Envs.MountainCarContinuous.cs:29-49 in double; the action stays a float"""
    p, v = s[0].astype(f64), s[1].astype(f64)
    force = clamp(np.asarray(action, f32).astype(f64), -1.0, 1.0)
    v = v + (force * 0.0015 - 0.0025 * _cos(3 * p))
    p, v, done = _mcar64_tail(p, v, 0.45)
    reward = v * 0.1
    reward = np.where(done, reward + 100.0, reward)
    return np.stack([p, v]), steps + 1, reward, done


def pendulum64_reset(seed, env, tick):
    th = rng_uniform(seed, env, tick, 0, 0) * 2 * math.pi - math.pi
    thd = rng_uniform(seed, env, tick, 0, 1) * 1 - 0.5
    return np.stack([_cos(th), _sin(th), thd])


def pendulum64_step(s, steps, action):
    """Envs.Pendulum.cs:30-60 in double"""
    dt = 0.05
    th = _atan2(s[1], s[0])
    thd = s[2].astype(f64)
    u = clamp(np.asarray(action, f32).astype(f64), -2.0, 2.0)
    nthd = thd + (-3 * 10.0 / (2 * 1.0) * _sin(th + math.pi) + 3.0 / (1.0 * 1.0 * 1.0) * u) * dt
    nthd = clamp(nthd, -8.0, 8.0)
    nth = th + nthd * dt
    reward = -(th * th + 0.1 * thd * thd + 0.001 * u * u)
    return np.stack([_cos(nth), _sin(nth), nthd]), steps + 1, reward, np.zeros(th.shape, bool)


def acrobot64_euler_reset(seed, env, tick):
    return np.stack([rng_uniform(seed, env, tick, 0, i) * 0.1 - 0.05 for i in range(6)])


def acrobot64_euler_step(s, steps, action):
    """Envs.Acrobot.cs:30-91 in double (semi-implicit Euler, as the reference integrates)"""
    action = np.asarray(action)
    if ((action < 0) | (action > 2)).any():
        raise IndexError("torque[action]")
    g, dt = 9.8, 0.2
    mv1, mv2 = 4 * math.pi, 9 * math.pi
    t1 = _atan2(s[1], s[0])
    t2 = _atan2(s[3], s[2])
    w1, w2 = s[4].astype(f64), s[5].astype(f64)
    u = (action.astype(np.int64) - 1).astype(f64)
    c2, s2 = _cos(t2), _sin(t2)
    d1 = 1.0 * 0.5 * 0.5 + 1.0 * (1.0 * 1.0 + 0.5 * 0.5 + 2 * 1.0 * 0.5 * c2) + 1.0 + 1.0
    d2 = 1.0 * (0.5 * 0.5 + 1.0 * 0.5 * c2) + 1.0
    phi2 = 1.0 * 0.5 * g * _cos(t1 + t2 - math.pi / 2.0)
    phi1 = (-1.0 * 1.0 * 0.5 * w2 * w2 * s2 - 2 * 1.0 * 1.0 * 0.5 * w2 * w1 * s2
            + (1.0 * 0.5 + 1.0 * 1.0) * g * _cos(t1 - math.pi / 2.0) + phi2)
    a2 = (u + d2 / d1 * phi1 - 1.0 * 1.0 * 0.5 * w1 * w1 * s2 - phi2) / (1.0 * 0.5 * 0.5 + 1.0 - d2 * d2 / d1)
    a1 = -(d2 * a2 + phi1) / d1
    w1 = clamp(w1 + dt * a1, -mv1, mv1)
    w2 = clamp(w2 + dt * a2, -mv2, mv2)
    t1 = t1 + dt * w1
    t2 = t2 + dt * w2
    ct1 = _cos(t1)
    ns = np.stack([ct1, _sin(t1), _cos(t2), _sin(t2), w1, w2])
    steps = steps + 1
    tip = -1.0 * ct1 - 1.0 * _cos(t1 + t2)
    done = (tip > 1.0) | (steps >= 500)
    return ns, steps, np.where(done, 0.0, -1.0), done


# -------------------------------------------------------------- grid worlds
def parse_map(rows):
    """Envs.FrozenLake.cs:31-42 -> (nrow, ncol, goal, hole mask)"""
    nrow, ncol = len(rows), len(rows[0])
    flat = "".join(rows)
    goal = flat.rfind("G")  # the loop keeps the last 'G'
    holes = np.array([ch == "H" for ch in flat], bool)
    return nrow, ncol, goal, holes


def grid_move(state, action, nrow, ncol):
    row, col = state // ncol, state % ncol
    a = np.asarray(action)
    col = np.where(a == 0, np.maximum(col - 1, 0), col)
    row = np.where(a == 1, np.minimum(row + 1, nrow - 1), row)
    col = np.where(a == 2, np.minimum(col + 1, ncol - 1), col)
    row = np.where(a == 3, np.maximum(row - 1, 0), row)
    return (row * ncol + col).astype(np.int32)


def slip_outcome(seed, env, tick, n_act=4, n_out=3):
    """Extension (upstream is_slippery): the outcome drawn inside Step is the base-n_out digit that
    follows the action digit of the env's word of the tick (stream 3): floor(w n_act n_out / 2^32) mod n_out."""
    w = rng_u32(seed, env, tick, 3, 0)
    hi = (w * np.uint64(n_act * n_out)) >> np.uint64(32)
    return (hi % np.uint64(n_out)).astype(np.int32)


def frozenlake_step(state, steps, action, rows=("SFFF", "FHFH", "FFFH", "HFFG"), slippery=None):
    """Envs.FrozenLake.cs:52-69.  slippery = (seed, env, tick) turns on the is_slippery extension: an
    action in 0..3 is executed as a-1, a or a+1 (mod 4) by slip_outcome."""
    nrow, ncol, goal, holes = parse_map(rows)
    action = np.asarray(action)
    if slippery is not None:
        k = slip_outcome(*slippery)
        action = np.where((action >= 0) & (action <= 3), (action + 3 + k) & 3, action)
    ns = grid_move(state, action, nrow, ncol)
    steps = steps + 1
    done = (ns == goal) | holes[ns] | (steps >= 100)
    return ns, steps, np.where(ns == goal, 1.0, 0.0), done


def cliffwalking_step(state, steps, action):
    """Envs.CliffWalking.cs:36-54"""
    ns = grid_move(state, action, 4, 12)
    steps = steps + 1
    cliff = (ns >= 37) & (ns <= 46)
    done = (ns == 47) | cliff | (steps >= 100)
    reward = np.where(ns == 47, 0.0, np.where(cliff, -100.0, -1.0))
    return ns, steps, reward, done


def taxi_step(state, steps, action):
    """Envs.Taxi.cs:25-33 (a stub in the reference)"""
    steps = steps + 1
    return state, steps, np.full(state.shape, -1.0), steps >= 200


def table_step(state, steps, action, table, n_outcomes=1, step_limit=0, rng=None):
    """An env stepped from a caller-supplied transition table (include/gymx.h GX_ENV_TABLE): the lookup stands for the
    body of FrozenLake.Step / Taxi.Step, the rest is Envs.FrozenLake.cs:64-68.  table: structured array
    [n_states, n_actions, n_outcomes] of (next_state, reward, done); rng = (seed, env, tick) when n_outcomes > 1."""
    table = np.asarray(table).reshape(table.shape[0], table.shape[1], -1)
    a = np.asarray(action)
    assert ((a >= 0) & (a < table.shape[1])).all()
    o = slip_outcome(*rng, n_act=table.shape[1], n_out=n_outcomes) if n_outcomes > 1 else np.zeros(a.shape, np.int32)
    s = np.where((state >= 0) & (state < table.shape[0]), state, 0)
    t = table[s, a, o]
    steps = steps + 1
    done = (t["done"] != 0) | ((steps >= step_limit) if step_limit > 0 else False)
    return t["next_state"].astype(np.int32), steps, t["reward"].astype(f64), done


def dummy_step(state, steps, action):
    """DummyEnv.cs:23-30"""
    ns = (state + np.asarray(action, np.int32)).astype(np.int32)
    return ns, steps + 1, ns.astype(f64), ns >= 10


# --------------------------------------------------------------- Blackjack
DECK = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 10], np.int32)  # ace, 2..10, jack, queen, king


def blackjack_reset(seed, env, tick):
    """Envs.Blackjack.cs:21-30: Next(12, 22), Next(1, 11), Next(2) == 1 -> (sum, dealer, usable)"""
    return (12 + rng_below(seed, env, tick, 0, 0, 10), 1 + rng_below(seed, env, tick, 0, 1, 10),
            (rng_below(seed, env, tick, 0, 2, 2) == 1).astype(np.int32))


def blackjack_step(steps):
    """Envs.Blackjack.cs:32-40 (a stub in the reference): the state stays, ten steps, then reward 1"""
    steps = steps + 1
    done = steps >= 10
    return steps, np.where(done, 1.0, 0.0), done


def _hand_value(hand):
    """upstream sum_hand / usable_ace over a list of cards -> (value, usable)"""
    raw = sum(hand)
    usable = 1 in hand and raw + 10 <= 21
    return (raw + 10 if usable else raw), int(usable)


def _digits13(w, count):
    """the first `count` base-13 digits of w / 2^32, per element"""
    w = np.asarray(w, np.uint64)
    out = []
    for _ in range(count):
        p = w * np.uint64(13)
        out.append((p >> np.uint64(32)).astype(np.int64))
        w = p & M32
    return out


def blackjack_full_deal(seed, env, tick):
    """Extension (upstream Blackjack-v1 reset): dealer's two cards, then the player's two: the first four
    base-13 digits of the env's deal word (stream 4).  Returns per-env [dealer hand, player hand] as
    Python lists of cards."""
    env = np.atleast_1d(np.asarray(env, np.uint64))
    cards = [DECK[d] for d in _digits13(rng_u32(seed, env, tick, 4, 0), 4)]
    return [[[int(cards[0][i]), int(cards[1][i])], [int(cards[2][i]), int(cards[3][i])]] for i in range(len(env))]


def blackjack_full_step(hands, action, seed, env, tick):
    """Extension: upstream step (natural = False, sab = False) on the hands of blackjack_full_deal,
    updated in place; any non-zero action hits.  The hit card is the base-13 digit after the action
    digit of the env's word of the tick (stream 3); the dealer's first four draws are the base-13 digits
    of the env's dealer word (stream 5), later ones stream 1, draws 0, 1, ..."""
    env = np.atleast_1d(np.asarray(env, np.uint64))
    w = rng_u32(seed, env, tick, 3, 0)
    hit_rank = ((w * np.uint64(26)) >> np.uint64(32)) % np.uint64(13)
    dealer_ranks = _digits13(rng_u32(seed, env, tick, 5, 0), 4)
    reward = np.zeros(len(env))
    done = np.zeros(len(env), bool)
    for i, (dealer, player) in enumerate(hands):
        if action[i] != 0:
            player.append(int(DECK[int(hit_rank[i])]))
            if _hand_value(player)[0] > 21:
                reward[i], done[i] = -1.0, True
        else:
            k = 0
            while _hand_value(dealer)[0] < 17:
                rank = dealer_ranks[k][i] if k < 4 else rng_below(seed, env[i:i + 1], tick, 1, k - 4, 13)[0]
                dealer.append(int(DECK[int(rank)]))
                k += 1
            p, d = _hand_value(player)[0], _hand_value(dealer)[0]
            p, d = (0 if p > 21 else p), (0 if d > 21 else d)
            reward[i], done[i] = float(p > d) - float(p < d), True
    return reward, done


def blackjack_image(hands, steps):
    """hands -> the three int rows of the state image: observation index, face-down cards, steps"""
    obs, hidden = [], []
    for dealer, player in hands:
        v, u = _hand_value(player)
        obs.append((v * 11 + dealer[0]) * 2 + u)
        hidden.append(sum(dealer[1:]) | (int(1 in dealer[1:]) << 8))
    return np.array([obs, hidden, list(steps)], np.int32)


# -------------------------------------------------------------------- Pong
def pong_reset_ball(seed, env, tick, stream):
    """Envs.Pong.cs:159-165"""
    n = np.asarray(env).shape
    vx = np.where(rng_uniform(seed, env, tick, stream, 0) > 0.5, 3.0, -3.0)
    vy = (rng_uniform(seed, env, tick, stream, 1) - 0.5) * 2.0 * 3.0
    return np.full(n, 80.0), np.full(n, 105.0), vx, vy


def pong_step(st, ints, action, seed, env, tick):
    """Envs.Pong.cs:72-157.  st: float64 [6, N] (ballX, ballY, velX, velY,
    leftY, rightY); ints: int32 [3, N] (leftScore, rightScore, steps)."""
    bx, by, vx, vy, ly, ry = (st[i].astype(f64).copy() for i in range(6))
    ls, rs, steps = (ints[i].astype(np.int32).copy() for i in range(3))
    a = np.asarray(action)
    steps = steps + 1
    ry = np.where(a == 2, np.maximum(0.0, ry - 4.0), ry)
    ry = np.where(a == 3, np.minimum(195.0, ry + 4.0), ry)
    bc, lc = by + 1.0, ly + 7.5
    up, down = bc < lc - 2, bc > lc + 2
    ly = np.where(up, np.maximum(0.0, ly - 4.0 * 0.8), np.where(down, np.minimum(195.0, ly + 4.0 * 0.8), ly))
    bx = bx + vx
    by = by + vy
    wall = (by <= 0) | (by >= 208)
    vy = np.where(wall, -vy, vy)
    by = np.where(wall, np.maximum(0.0, np.minimum(208.0, by)), by)
    hitl = (bx <= 2) & (by + 2 >= ly) & (by <= ly + 15)
    vx = np.where(hitl, np.abs(vx), vx)
    vy = np.where(hitl, -((ly + 7.5) - (by + 1.0)) * 0.1, vy)
    hitr = (bx + 2 >= 158) & (by + 2 >= ry) & (by <= ry + 15)
    vx = np.where(hitr, -np.abs(vx), vx)
    vy = np.where(hitr, -((ry + 7.5) - (by + 1.0)) * 0.1, vy)
    sr = bx < 0
    sl = (~sr) & (bx > 160)
    rs = rs + sr.astype(np.int32)
    ls = ls + sl.astype(np.int32)
    reward = np.where(sr, 1.0, np.where(sl, -1.0, 0.0))
    nbx, nby, nvx, nvy = pong_reset_ball(seed, env, tick, 1)
    rb = sr | sl
    bx, by, vx, vy = (np.where(rb, n, o) for n, o in ((nbx, bx), (nby, by), (nvx, vx), (nvy, vy)))
    done = (ls >= 21) | (rs >= 21) | (steps >= 10000)
    return np.stack([bx, by, vx, vy, ly, ry]), np.stack([ls, rs, steps]), reward, done


_GLYPHS = [
    ["111", "101", "101", "101", "111"], ["010", "110", "010", "010", "111"],
    ["111", "001", "111", "100", "111"], ["111", "001", "111", "001", "111"],
    ["101", "101", "111", "001", "001"], ["111", "100", "111", "001", "111"],
    ["111", "100", "111", "101", "111"], ["111", "001", "001", "001", "001"],
    ["111", "101", "111", "101", "111"], ["111", "101", "111", "001", "111"],
]  # Envs.Pong.cs:228-237


def pong_frame(bx, by, ly, ry, ls, rs):
    """Envs.Pong.cs:167-251 as a boolean mask (every drawn pixel is white, so
    draw order is irrelevant), expanded to HWC bytes."""
    m = np.zeros((210, 160), bool)

    def rect(x, y, w, h):
        m[max(y, 0):max(min(y + h, 210), 0), max(x, 0):max(min(x + w, 160), 0)] = True

    rect(0, int(ly), 2, 15)
    rect(158, int(ry), 2, 15)
    rect(int(bx), int(by), 2, 2)
    for y in range(0, 210, 8):
        rect(79, y, 2, 4)
    for score, cx in ((ls, 40), (rs, 120)):
        txt = str(int(score))
        x0 = cx - (len(txt) * 6) // 2
        for i, ch in enumerate(txt):
            if not ch.isdigit():
                continue
            for dy, row in enumerate(_GLYPHS[int(ch)]):
                for dx, bit in enumerate(row):
                    if bit == "1":
                        rect(x0 + 8 * i + dx, 20 + dy, 1, 1)
    return np.repeat((m.astype(np.uint8) * 255).reshape(-1), 3)


# ---------------------------------------------------------------- Breakout
def breakout_unpack(words):
    """int32 [4, N] brick words -> bool [N, 6, 18] (bit row*18+col)"""
    w = np.asarray(words).astype(np.int64) & 0xFFFFFFFF
    bits = np.arange(108)
    return ((w[bits >> 5] >> (bits & 31)[:, None]) & 1).astype(bool).T.reshape(-1, 6, 18)


def breakout_pack(bricks):
    flat = np.asarray(bricks, bool).reshape(-1, 108)
    out = np.zeros((4, flat.shape[0]), np.int64)
    for b in range(108):
        out[b >> 5] |= flat[:, b].astype(np.int64) << (b & 31)
    return out.astype(np.uint32).view(np.int32)


def breakout_step(st, ints, action, seed, env, tick):
    """Envs.Breakout.cs:89-213.  st: float64 [5, N] (ballX, ballY, velX, velY,
    paddleX); ints: int32 [8, N] (score, lives, stuck, 4 brick words, steps)."""
    bx, by, vx, vy, px = (st[i].astype(f64).copy() for i in range(5))
    score, lives = ints[0].astype(np.int32).copy(), ints[1].astype(np.int32).copy()
    stuck = ints[2] != 0
    bricks = breakout_unpack(ints[3:7])
    steps = ints[7].astype(np.int32) + 1
    a = np.asarray(action)
    n = bx.shape[0]
    fire = (a == 1) & stuck
    vx = np.where(fire, (rng_uniform(seed, env, tick, 1, 0) - 0.5) * 2.0 * 3.0, vx)
    vy = np.where(fire, -3.0, vy)
    stuck = stuck & ~fire
    px = np.where(a == 2, np.minimum(144.0, px + 5.0), np.where(a == 3, np.maximum(0.0, px - 5.0), px))
    bx = np.where(((a == 2) | (a == 3)) & stuck, px + 8.0, bx)
    mv = ~stuck
    bx = np.where(mv, bx + vx, bx)
    by = np.where(mv, by + vy, by)
    wall = mv & ((bx <= 0) | (bx >= 158))
    vx = np.where(wall, -vx, vx)
    bx = np.where(wall, np.maximum(0.0, np.minimum(158.0, bx)), bx)
    top = mv & (by <= 0)
    vy = np.where(top, -vy, vy)
    by = np.where(top, 0.0, by)
    pad = mv & (by + 2 >= 190) & (by + 2 <= 194) & (bx + 2 >= px) & (bx <= px + 16)
    vy = np.where(pad, -np.abs(vy), vy)
    vx = np.where(pad, ((bx + 1.0) - (px + 8.0)) / 8.0 * 3.0 * 0.75, vx)
    row = np.trunc((by - 60) / 6).astype(np.int64)
    col = np.trunc(bx / 8).astype(np.int64)
    inb = mv & (row >= 0) & (row < 6) & (col >= 0) & (col < 18)
    idx = np.arange(n)
    hit = inb & bricks[idx, np.clip(row, 0, 5), np.clip(col, 0, 17)]
    bricks[idx[hit], row[hit], col[hit]] = False
    vy = np.where(hit, -vy, vy)
    reward = np.where(hit, (6 - row) + 1, 0).astype(f64)
    score = score + reward.astype(np.int32)
    fell = mv & (by > 210)
    lives = lives - fell.astype(np.int32)
    rs = fell & (lives > 0)
    bx = np.where(rs, px + 8.0, bx)
    by = np.where(rs, 180.0, by)
    vx = np.where(rs, 0.0, vx)
    vy = np.where(rs, 0.0, vy)
    stuck = stuck | rs
    gone = ~bricks.reshape(n, -1).any(axis=1)
    reward = reward + np.where(gone, 100.0, 0.0)
    done = (lives <= 0) | gone | (steps >= 10000)
    new_ints = np.concatenate([np.stack([score, lives, stuck.astype(np.int32)]), breakout_pack(bricks), steps[None]])
    return np.stack([bx, by, vx, vy, px]), new_ints.astype(np.int32), reward, done


_BRICK_RGB = [(255, 0, 0), (255, 165, 0), (255, 255, 0), (0, 255, 0), (0, 0, 255), (128, 0, 128)]  # Envs.Breakout.cs:248-253


def breakout_frame(bx, by, px, stuck, score, lives, words):
    """Envs.Breakout.cs:229-344, painter's order kept: bricks, paddle, ball, score, lives."""
    f = np.zeros((210, 160, 3), np.uint8)

    def rect(x, y, w, h, rgb):
        f[max(y, 0):max(min(y + h, 210), 0), max(x, 0):max(min(x + w, 160), 0)] = rgb

    bricks = breakout_unpack(np.asarray(words, np.int64).reshape(4, 1).astype(np.uint32).view(np.int32))[0]
    for r in range(6):
        for c in range(18):
            if bricks[r, c]:
                rect(c * 8, 60 + r * 6, 7, 5, _BRICK_RGB[r])
    white = (255, 255, 255)
    rect(int(px), 190, 16, 4, white)
    if not stuck:
        rect(int(bx), int(by), 2, 2, white)
    else:
        rect(int(px + 8.0 - 1.0), 180, 2, 2, white)
    for i, ch in enumerate(str(int(score))):
        if not ch.isdigit():
            continue
        for dy, rowbits in enumerate(_GLYPHS[int(ch)]):
            for dx, bit in enumerate(rowbits):
                if bit == "1":
                    rect(10 + i * 6 + dx, 10 + dy, 1, 1, white)
    for i in range(int(lives)):
        rect(10 + i * 8, 25, 6, 3, white)
    return f.reshape(-1)


# ----------------------------------------------------------- SpaceInvaders
class InvadersGame:
    """Envs.SpaceInvaders.cs:12-541 for one env, written with Python lists where the source uses
    List<(double,double)> (so removal and append order are the source's own).  Draws come from
    the counter RNG: stream 1, draw 0 = NextDouble() < 0.3, draw 1 = Next(shooters.Count)."""

    def __init__(self):
        self.reset()

    def reset(self):  # :69-96
        self.aliens = [[True] * 11 for _ in range(5)]
        self.player_x = 160 / 2.0 - 8 / 2.0
        self.form_x, self.form_y, self.move_right = 10.0, 40.0, True
        self.player_bullets, self.alien_bullets = [], []
        self.score, self.lives, self.steps, self.frame = 0, 3, 0, 0

    def _fire(self):  # :192-199
        if len(self.player_bullets) < 3:
            self.player_bullets.append((self.player_x + 8 / 2.0, 210.0 - 40))

    def step(self, action, seed, env, tick):  # :98-190
        self.steps += 1
        self.frame += 1
        reward = 0.0
        if action == 1:
            self._fire()
        elif action == 2:
            self.player_x = min(160.0 - 8, self.player_x + 3.0)
        elif action == 3:
            self.player_x = max(0.0, self.player_x - 3.0)
        elif action == 4:
            self.player_x = min(160.0 - 8, self.player_x + 3.0)
            self._fire()
        elif action == 5:
            self.player_x = max(0.0, self.player_x - 3.0)
            self._fire()
        for i in range(len(self.player_bullets) - 1, -1, -1):  # :229-241
            x, y = self.player_bullets[i]
            y -= 6.0
            if y < 0:
                del self.player_bullets[i]
            else:
                self.player_bullets[i] = (x, y)
        for i in range(len(self.alien_bullets) - 1, -1, -1):  # :244-256
            x, y = self.alien_bullets[i]
            y += 6.0 * 0.7
            if y > 210:
                del self.alien_bullets[i]
            else:
                self.alien_bullets[i] = (x, y)
        if self.frame % 10 == 0:  # :259-279
            if self.move_right:
                self.form_x += 0.5
                if self.form_x + 11 * 12 > 160 - 10:
                    self.move_right = False
                    self.form_y += 8.0
            else:
                self.form_x -= 0.5
                if self.form_x < 10:
                    self.move_right = True
                    self.form_y += 8.0
        e = np.array([env], np.uint64)
        if self.frame % 60 == 0 and float(rng_uniform(seed, e, tick, 1, 0)[0]) < 0.3:  # :136-139, :201-224
            shooters = []
            for col in range(11):
                for row in range(4, -1, -1):
                    if self.aliens[row][col]:
                        shooters.append((row, col))
                        break
            if shooters:
                row, col = shooters[int(rng_below(seed, e, tick, 1, 1, len(shooters))[0])]
                if len(self.alien_bullets) < 2:  # capacity of the state image (never reached in play)
                    self.alien_bullets.append((self.form_x + col * 12 + 6 / 2.0, self.form_y + row * 10 + 6))
        for i in range(len(self.player_bullets) - 1, -1, -1):  # :286-323
            bx, by = self.player_bullets[i]
            hit = False
            for row in range(5):
                for col in range(11):
                    if self.aliens[row][col]:
                        ax, ay = self.form_x + col * 12, self.form_y + row * 10
                        if ax <= bx <= ax + 6 and ay <= by <= ay + 6:
                            self.aliens[row][col] = False
                            del self.player_bullets[i]
                            pts = {0: 30, 1: 20, 2: 20}.get(row, 10)
                            self.score += pts
                            reward += pts
                            hit = True
                            break
                if hit:
                    break
        for i in range(len(self.alien_bullets) - 1, -1, -1):  # :326-336
            bx, by = self.alien_bullets[i]
            if self.player_x <= bx <= self.player_x + 8 and 210 - 40 <= by <= 210 - 40 + 6:
                del self.alien_bullets[i]
                self.lives -= 1
                reward -= 50
        if self.form_y + 5 * 8 > 210 - 40:  # :145-148
            self.lives = 0
        if not any(any(r) for r in self.aliens):  # :151-174
            reward += 100
            self.aliens = [[True] * 11 for _ in range(5)]
            self.form_y, self.form_x = 40.0, 10.0
        return reward, self.lives <= 0 or self.steps >= 10000

    def image(self):
        """(13 reals, 9 ints) as gx_vec_get_state lays one env out"""
        pb = self.player_bullets + [(0.0, 0.0)] * (3 - len(self.player_bullets))
        ab = self.alien_bullets + [(0.0, 0.0)] * (2 - len(self.alien_bullets))
        reals = [self.player_x, self.form_x, self.form_y] + [v for b in pb for v in b] + [v for b in ab for v in b]
        bits = 0
        for row in range(5):
            for col in range(11):
                if self.aliens[row][col]:
                    bits |= 1 << (row * 11 + col)
        ints = np.array([self.score, self.lives, self.frame, int(self.move_right), len(self.player_bullets), len(self.alien_bullets),
                         bits & 0xFFFFFFFF, bits >> 32, self.steps], np.int64).astype(np.uint32).view(np.int32)
        return np.array(reals, f64), ints

    def load(self, reals, ints):
        self.player_x, self.form_x, self.form_y = (float(v) for v in reals[:3])
        npb, nab = min(max(int(ints[4]), 0), 3), min(max(int(ints[5]), 0), 2)
        self.player_bullets = [(float(reals[3 + 2 * i]), float(reals[4 + 2 * i])) for i in range(npb)]
        self.alien_bullets = [(float(reals[9 + 2 * i]), float(reals[10 + 2 * i])) for i in range(nab)]
        self.score, self.lives, self.frame, self.move_right = int(ints[0]), int(ints[1]), int(ints[2]), bool(ints[3])
        bits = (int(ints[6]) & 0xFFFFFFFF) | ((int(ints[7]) & 0xFFFFFFFF) << 32)
        self.aliens = [[bool((bits >> (row * 11 + col)) & 1) for col in range(11)] for row in range(5)]
        self.steps = int(ints[8])

    _ALIEN = ["010010", "001100", "011110", "110011", "111111", "010010"]  # :416-423
    _PLAYER = ["00011000", "00111100", "00111100", "01111110", "11111111", "11111111"]  # :440-447
    _ROW_RGB = [(255, 0, 255), (0, 255, 255), (255, 255, 0), (0, 255, 0), (255, 0, 0)]  # :374-378

    def frame_bytes(self):  # :355-411, painter's order
        f = np.zeros((210, 160, 3), np.uint8)

        def rect(x, y, w, h, rgb):
            f[max(y, 0):max(min(y + h, 210), 0), max(x, 0):max(min(x + w, 160), 0)] = rgb

        def sprite(rows, x, y, rgb):
            for dy, r in enumerate(rows):
                for dx, bit in enumerate(r):
                    if bit == "1":
                        rect(x + dx, y + dy, 1, 1, rgb)

        for row in range(5):
            for col in range(11):
                if self.aliens[row][col]:
                    sprite(self._ALIEN, int(self.form_x + col * 12), int(self.form_y + row * 10), self._ROW_RGB[row])
        sprite(self._PLAYER, int(self.player_x), 210 - 40, (0, 255, 0))
        for x, y in self.player_bullets:
            rect(int(x), int(y), 2, 4, (255, 255, 0))
        for x, y in self.alien_bullets:
            rect(int(x), int(y), 2, 4, (255, 0, 0))
        for i, ch in enumerate(str(int(self.score))):
            if ch.isdigit():
                sprite(_GLYPHS[int(ch)], 10 + i * 6, 10, (255, 255, 255))
        for i in range(self.lives):
            sprite(self._PLAYER, 160 - 40 - i * 10, 10, (0, 255, 0))
        for x in range(160):
            rect(x, 210 - 20, 1, 2, (0, 255, 0))
        return f.reshape(-1)


# ---------------------------------------------------------------- wrappers
class RecordEpisodeStatistics:
    """EnvWrapper.cs:23-58 for one env index; `step` takes the inner result."""

    def __init__(self):
        self.episode_rewards, self.episode_lengths = [], []
        self._r, self._n = 0.0, 0

    def reset(self):
        if self._n > 0:
            self.episode_rewards.append(self._r)
            self.episode_lengths.append(self._n)
        self._r, self._n = 0.0, 0

    def step(self, reward, done):
        self._r += reward
        self._n += 1
        if done:
            self.episode_rewards.append(self._r)
            self.episode_lengths.append(self._n)
            self._r, self._n = 0.0, 0


class TimeLimit:
    """Wrappers.TimeLimit.cs:6-34; `step` returns (done, truncated)."""

    def __init__(self, max_steps):
        self.max_steps, self._steps = max_steps, 0

    def reset(self):
        self._steps = 0

    def step(self, done):
        self._steps += 1
        if self._steps >= self.max_steps:
            return True, True
        return bool(done), False
