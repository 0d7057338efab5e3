"""ctypes binding of the CPU oracle (oracle/gym_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the gym_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgym_oracle.so")

(CARTPOLE, MOUNTAINCAR, MOUNTAINCAR_CONT, PENDULUM, ACROBOT, FROZENLAKE, TAXI, CLIFFWALKING, PONG, DUMMY,
 BREAKOUT, SPACEINVADERS, BLACKJACK, TABLE) = range(14)
AUTORESET, RECORD_STATS, SLIPPERY, TAXI_FULL, ACROBOT_RK4, FP64, BLACKJACK_FULL = 1, 2, 4, 8, 16, 32, 64
PONG_FRAME_BYTES = 210 * 160 * 3
FLOAT_ACTION = {MOUNTAINCAR_CONT, PENDULUM}


class Transition(C.Structure):
    """gxo_transition: one entry of a caller-supplied table"""
    _fields_ = [("next_state", C.c_int32), ("reward", C.c_float), ("done", C.c_int32)]


TRANSITION_DTYPE = np.dtype([("next_state", np.int32), ("reward", np.float32), ("done", np.int32)])


class Stats(C.Structure):
    _fields_ = [
        ("episodes", C.c_int64),
        ("sum_return", C.c_double),
        ("sum_length", C.c_int64),
        ("sum_return_sq", C.c_double),
        ("min_return", C.c_double),
        ("max_return", C.c_double),
    ]


def build():
    """Compile the oracle if the shared object is missing or stale."""
    src = os.path.join(_HERE, "gym_oracle.c")
    hdr = os.path.join(_HERE, "gym_oracle.h")
    if os.path.exists(_SO) and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    try:
        build()  # no-op when the shared object is newer than its sources
    except (OSError, subprocess.CalledProcessError):
        if not os.path.exists(_SO):
            raise
    L = C.CDLL(_SO)
    vp, u64, u32, i32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_double
    L.gxo_philox4x32_10.argtypes = [vp, vp, vp]
    L.gxo_rng_u32.restype = u32
    L.gxo_rng_u32.argtypes = [u64, u64, u64, u32, u32]
    L.gxo_rng_uniform.restype = dbl
    L.gxo_rng_uniform.argtypes = [u64, u64, u64, u32, u32]
    L.gxo_rng_below.restype = i32
    L.gxo_rng_below.argtypes = [u64, u64, u64, u32, u32, i32]
    L.gxo_box_sample.restype = C.c_float
    L.gxo_box_sample.argtypes = [dbl, C.c_float, C.c_float]
    L.gxo_vec_create.restype = vp
    L.gxo_vec_create.argtypes = [C.c_int, u64, u64, u32, i32, C.c_char_p, u64]
    L.gxo_vec_create_table.restype = vp
    L.gxo_vec_create_table.argtypes = [u64, u64, u32, i32, u64, vp, i32, i32, i32, i32, i32]
    L.gxo_vec_destroy.argtypes = [vp]
    for name in ("num_envs", "tick"):
        getattr(L, "gxo_vec_" + name).restype = u64
        getattr(L, "gxo_vec_" + name).argtypes = [vp]
    for name in ("obs_dim", "n_real", "real_is_f64", "obs_is_f64", "n_int"):
        getattr(L, "gxo_vec_" + name).restype = C.c_int
        getattr(L, "gxo_vec_" + name).argtypes = [vp]
    L.gxo_vec_state_bytes.restype = C.c_size_t
    L.gxo_vec_state_bytes.argtypes = [vp]
    L.gxo_vec_reset.argtypes = [vp]
    L.gxo_vec_step.restype = C.c_int
    L.gxo_vec_step.argtypes = [vp, vp, vp, vp, vp]
    L.gxo_vec_step_random.restype = C.c_int
    L.gxo_vec_step_random.argtypes = [vp, vp, vp, vp, vp]
    L.gxo_vec_get_state.argtypes = [vp, vp]
    L.gxo_vec_set_state.argtypes = [vp, vp]
    L.gxo_vec_set_rng_draws.argtypes = [vp, vp]
    L.gxo_vec_set_rng_draws.restype = C.c_int
    L.gxo_vec_get_obs.argtypes = [vp, vp]
    L.gxo_vec_stats.argtypes = [vp, C.POINTER(Stats)]
    L.gxo_vec_episode_count.restype = C.c_size_t
    L.gxo_vec_episode_count.argtypes = [vp, u64]
    L.gxo_vec_episode_get.argtypes = [vp, u64, C.c_size_t, C.POINTER(dbl), C.POINTER(i32)]
    L.gxo_pong_render.argtypes = [dbl, dbl, dbl, dbl, C.c_int, C.c_int, vp]
    L.gxo_invaders_render.argtypes = [vp, vp, vp]
    L.gxo_breakout_render.argtypes = [dbl, dbl, dbl, C.c_int, C.c_int, C.c_int, vp, vp]
    L.gxo_bench_reference_shaped.restype = dbl
    L.gxo_bench_reference_shaped.argtypes = [C.c_int, u64, C.c_int, u64, C.POINTER(u64), C.POINTER(dbl)]
    L.gxo_bench_flat_threads.restype = dbl
    L.gxo_bench_flat_threads.argtypes = [C.c_int, u64, C.c_int, C.c_int, u64, C.POINTER(u64), C.POINTER(dbl)]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleVec:
    """Batched CPU engine shaped like the reference VectorEnv (VectorEnv.cs:10-35)."""

    def __init__(self, env_id, num_envs, seed=0, flags=0, time_limit=0, map_rows=None, env_index_base=0, table=None,
                 n_outcomes=1, start_state=0, step_limit=0):
        """table (env_id TABLE only): array of TRANSITION_DTYPE, shape [n_states, n_actions] or
        [n_states, n_actions, n_outcomes]"""
        L = lib()
        rows = None if map_rows is None else "/".join(map_rows).encode()
        if env_id == TABLE:
            t = np.ascontiguousarray(table, dtype=TRANSITION_DTYPE)
            self._h = L.gxo_vec_create_table(num_envs, seed, flags, time_limit, env_index_base, _ptr(t), t.shape[0], t.shape[1],
                                             n_outcomes, start_state, step_limit)
        else:
            self._h = L.gxo_vec_create(env_id, num_envs, seed, flags, time_limit, rows, env_index_base)
        if not self._h:
            raise ValueError("oracle: cannot create env %r" % (env_id,))
        self.env_id = env_id
        self.n = num_envs
        self.n_real = L.gxo_vec_n_real(self._h)
        self.real_is_f64 = bool(L.gxo_vec_real_is_f64(self._h))
        self.obs_is_f64 = bool(L.gxo_vec_obs_is_f64(self._h))
        self.n_int = L.gxo_vec_n_int(self._h)
        self.obs_dim = L.gxo_vec_obs_dim(self._h)
        self.action_dtype = np.float32 if env_id in FLOAT_ACTION else np.int32

    def __del__(self):
        if getattr(self, "_h", None):
            lib().gxo_vec_destroy(self._h)
            self._h = None

    @property
    def tick(self):
        return lib().gxo_vec_tick(self._h)

    def reset(self):
        lib().gxo_vec_reset(self._h)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=self.action_dtype)
        assert a.shape == (self.n,)
        reward = np.empty(self.n, np.float64)
        done = np.empty(self.n, np.uint8)
        trunc = np.empty(self.n, np.uint8)
        rc = lib().gxo_vec_step(self._h, _ptr(a), _ptr(reward), _ptr(done), _ptr(trunc))
        if rc != 0:
            raise IndexError("oracle: invalid action")
        return reward, done, trunc

    def step_random(self):
        a = np.empty(self.n, self.action_dtype)
        reward = np.empty(self.n, np.float64)
        done = np.empty(self.n, np.uint8)
        trunc = np.empty(self.n, np.uint8)
        rc = lib().gxo_vec_step_random(self._h, _ptr(a), _ptr(reward), _ptr(done), _ptr(trunc))
        assert rc == 0
        return a, reward, done, trunc

    def state_bytes(self):
        return lib().gxo_vec_state_bytes(self._h)

    def get_state_raw(self):
        buf = np.empty(self.state_bytes(), np.uint8)
        lib().gxo_vec_get_state(self._h, _ptr(buf))
        return buf

    def set_state_raw(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        assert buf.size == self.state_bytes()
        lib().gxo_vec_set_state(self._h, _ptr(buf))

    def set_rng_draws(self, u):
        """Pong u[2][N] (ball reset), Breakout u[1][N] (release of the ball): stands in for the generator in each env's
        next such event, for the next reset / step only"""
        if u is None:
            assert lib().gxo_vec_set_rng_draws(self._h, None) == 0
            return
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.size == (2 if self.env_id == PONG else 1) * self.n
        assert lib().gxo_vec_set_rng_draws(self._h, _ptr(u)) == 0

    def split_state(self, buf):
        """raw SoA image -> (reals [n_real, N], ints [n_int, N])"""
        rdt = np.float64 if self.real_is_f64 else np.float32
        nb = self.n * self.n_real * np.dtype(rdt).itemsize
        reals = buf[:nb].view(rdt).reshape(self.n_real, self.n)
        ints = buf[nb:].view(np.int32).reshape(self.n_int, self.n)
        return reals, ints

    def get_state(self):
        return self.split_state(self.get_state_raw())

    def set_state(self, reals, ints):
        rdt = np.float64 if self.real_is_f64 else np.float32
        r = np.ascontiguousarray(reals, dtype=rdt).reshape(-1).view(np.uint8)
        i = np.ascontiguousarray(ints, dtype=np.int32).reshape(-1).view(np.uint8)
        self.set_state_raw(np.concatenate([r, i]))

    def get_obs(self):
        if self.env_id in (PONG, BREAKOUT, SPACEINVADERS):
            out = np.empty((self.n, PONG_FRAME_BYTES), np.uint8)
        elif self.n_real > 0:
            out = np.empty((self.obs_dim, self.n), np.float64 if self.obs_is_f64 else np.float32)
        else:
            out = np.empty(self.n, np.int32)
        lib().gxo_vec_get_obs(self._h, _ptr(out))
        return out

    def stats(self):
        s = Stats()
        lib().gxo_vec_stats(self._h, C.byref(s))
        return s

    def episodes(self, env):
        L = lib()
        out = []
        for k in range(L.gxo_vec_episode_count(self._h, env)):
            r, n = C.c_double(), C.c_int32()
            L.gxo_vec_episode_get(self._h, env, k, C.byref(r), C.byref(n))
            out.append((r.value, n.value))
        return out


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().gxo_philox4x32_10(_ptr(c), _ptr(k), _ptr(o))
    return [int(x) for x in o]


def rng_u32(seed, env, tick, stream, draw):
    return lib().gxo_rng_u32(seed, env, tick, stream, draw)


def rng_uniform(seed, env, tick, stream, draw):
    return lib().gxo_rng_uniform(seed, env, tick, stream, draw)


def rng_below(seed, env, tick, stream, draw, n):
    return lib().gxo_rng_below(seed, env, tick, stream, draw, n)


def pong_render(ball_x, ball_y, left_y, right_y, left_score, right_score):
    out = np.empty(PONG_FRAME_BYTES, np.uint8)
    lib().gxo_pong_render(ball_x, ball_y, left_y, right_y, left_score, right_score, _ptr(out))
    return out


def breakout_render(ball_x, ball_y, paddle_x, ball_stuck, score, lives, bricks):
    """bricks: four uint32 words, bit row*18+col"""
    out = np.empty(PONG_FRAME_BYTES, np.uint8)
    w = np.ascontiguousarray(bricks, dtype=np.uint32)
    assert w.shape == (4,)
    lib().gxo_breakout_render(ball_x, ball_y, paddle_x, int(ball_stuck), score, lives, _ptr(w), _ptr(out))
    return out


def invaders_render(reals, ints):
    """reals: 13 float64, ints: 9 int32 (one env's state image)"""
    out = np.empty(PONG_FRAME_BYTES, np.uint8)
    r = np.ascontiguousarray(reals, dtype=np.float64)
    i = np.ascontiguousarray(ints, dtype=np.int32)
    assert r.shape == (13,) and i.shape == (9,)
    lib().gxo_invaders_render(_ptr(r), _ptr(i), _ptr(out))
    return out


def bench_flat_threads(env_id, num_envs, steps, threads, seed=0):
    n = C.c_uint64()
    chk = C.c_double()
    secs = lib().gxo_bench_flat_threads(env_id, num_envs, steps, threads, seed, C.byref(n), C.byref(chk))
    return secs, n.value


def bench_reference_shaped(env_id, num_envs, steps, seed=0):
    n = C.c_uint64()
    chk = C.c_double()
    secs = lib().gxo_bench_reference_shaped(env_id, num_envs, steps, seed, C.byref(n), C.byref(chk))
    return secs, n.value
