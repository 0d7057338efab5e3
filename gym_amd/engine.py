"""Thin object layer over the C ABI: Context (one GPU) and DeviceBatch (one
device-resident batch of environments).  Device buffers are exposed as torch
tensors through the zero-copy view; torch is used for memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_NP = {L.F32: np.float32, L.F64: np.float64, L.I32: np.int32, L.U8: np.uint8}
# gx_transition as a numpy record
TRANSITION_DTYPE = np.dtype([("next_state", np.int32), ("reward", np.float32), ("done", np.int32)])


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        L.check(L.load().gx_ctx_create(device, C.byref(self._h)))
        self.device = device

    def synchronize(self):
        L.check(L.load().gx_ctx_synchronize(self._h))

    @property
    def stream(self):
        s = C.c_void_p()
        L.check(L.load().gx_ctx_stream(self._h, C.byref(s)))
        return s.value

    def close(self):
        """gx_ctx_destroy refuses while a batch or communicator of the context is alive; the handle is kept then, so
        that a later close() can succeed, and the refusal is raised instead of leaking the streams silently"""
        if self._h:
            L.check(L.load().gx_ctx_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceBatch:
    """A batch of `num_envs` environments resident in HBM (VectorEnv.cs:10-35)."""

    def __init__(self, ctx, env_id, num_envs, seed=0, flags=0, time_limit=0, mode=L.MODE_EXACT, map_rows=None,
                 env_index_base=0, table=None, n_outcomes=1, start_state=0, step_limit=0):
        """table (env_id L.ENV_TABLE only): transitions as a numpy structured array of TRANSITION_DTYPE, shape
        [n_states, n_actions] or [n_states, n_actions, n_outcomes]; start_state -1 resets to a uniform draw over the
        states; step_limit 0 = the env never ends by its own step counter"""
        lib = L.load()
        self.ctx = ctx
        rows = None
        if map_rows is not None:
            rows = map_rows.encode() if isinstance(map_rows, str) else "/".join(map_rows).encode()
        d = L.GxDesc(C.sizeof(L.GxDesc), env_id, num_envs, seed, env_index_base, mode, flags, time_limit, 0, rows)
        if table is not None:
            t = np.ascontiguousarray(table, dtype=TRANSITION_DTYPE)
            if t.ndim not in (2, 3):
                raise ValueError("table must have shape [n_states, n_actions] or [n_states, n_actions, n_outcomes]")
            d.table, d.table_len = t.ctypes.data, t.size
            d.n_states, d.n_actions, d.n_outcomes = t.shape[0], t.shape[1], n_outcomes
            d.start_state, d.step_limit = start_state, step_limit
        self._h = C.c_void_p()
        L.check(lib.gx_vec_create(ctx._h, C.byref(d), C.byref(self._h)))
        self.info = L.GxInfo()
        L.check(lib.gx_vec_info(self._h, C.byref(self.info)))
        self.n = num_envs
        self.env_id = env_id
        self.flags = flags
        self.action_dtype = _NP[self.info.action_dtype]
        self.reward_dtype = _NP[self.info.reward_dtype]
        self.real_dtype = _NP[self.info.real_dtype]
        self.obs_dtype = _NP[self.info.obs_dtype]

    def close(self):
        if getattr(self, "_h", None):
            L.load().gx_vec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- control -------------------------------------------------------
    @property
    def tick(self):
        t = C.c_uint64()
        L.check(L.load().gx_vec_tick(self._h, C.byref(t)))
        return t.value

    def reset(self):
        L.check(L.load().gx_vec_reset(self._h))

    def get_state_raw(self):
        buf = np.empty(self.info.state_bytes, np.uint8)
        L.check(L.load().gx_vec_get_state(self._h, _ptr(buf), buf.size))
        return buf

    def set_state_raw(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        L.check(L.load().gx_vec_set_state(self._h, _ptr(buf), buf.size))

    def set_rng_draws(self, u):
        """The NextDouble() values of each env's next in-Step draw, in [0, 1), for the next reset / step / rollout launch
        only: Pong u[2][num_envs] (ball reset, Envs.Pong.cs:163-164), Breakout u[1][num_envs] (release of the ball,
        Envs.Breakout.cs:100); None withdraws values not used yet"""
        if u is None:
            L.check(L.load().gx_vec_set_rng_draws(self._h, None, 0))
            return
        u = np.ascontiguousarray(u, np.float64)
        L.check(L.load().gx_vec_set_rng_draws(self._h, _ptr(u), u.size))

    def split_state(self, buf):
        nb = self.n * self.info.n_real * np.dtype(self.real_dtype).itemsize
        reals = buf[:nb].view(self.real_dtype).reshape(self.info.n_real, self.n)
        ints = buf[nb:].view(np.int32).reshape(self.info.n_int, self.n)
        return reals, ints

    def get_state(self):
        return self.split_state(self.get_state_raw())

    def set_state(self, reals, ints):
        r = np.ascontiguousarray(reals, self.real_dtype).reshape(-1).view(np.uint8)
        i = np.ascontiguousarray(ints, np.int32).reshape(-1).view(np.uint8)
        self.set_state_raw(np.concatenate([r, i]))

    def obs_shape(self):
        if self.info.obs_dtype == L.U8:  # Pong, Breakout frames
            return (self.n, self.info.obs_dim)
        if self.info.obs_dtype == L.I32:
            return (self.n,)
        return (self.info.obs_dim, self.n)

    def get_obs(self):
        out = np.empty(self.obs_shape(), self.obs_dtype)
        L.check(L.load().gx_vec_get_obs(self._h, _ptr(out), out.nbytes))
        return out

    # ---- host path -----------------------------------------------------
    def step_host(self, actions, want_obs=False, want_trunc=False):
        """VectorEnv.Step through pageable/pinned host buffers; returns numpy arrays."""
        a = np.ascontiguousarray(actions, self.action_dtype)
        if a.shape != (self.n,):
            raise ValueError("actions must have shape (%d,)" % self.n)
        reward = np.empty(self.n, self.reward_dtype)
        done = np.empty(self.n, np.uint8)
        trunc = np.empty(self.n, np.uint8) if want_trunc else None
        obs = np.empty(self.obs_shape(), self.obs_dtype) if want_obs else None
        lib = L.load()
        L.check(lib.gx_vec_step_async(self._h, _ptr(a), _ptr(obs), _ptr(reward), _ptr(done), _ptr(trunc)))
        L.check(lib.gx_vec_wait(self._h))
        return obs, reward, done, trunc

    def wait(self):
        L.check(L.load().gx_vec_wait(self._h))

    # ---- zero-copy path --------------------------------------------------
    def step_device(self, d_actions_ptr, stream=None):
        L.check(L.load().gx_vec_step_device(self._h, C.c_void_p(d_actions_ptr), C.c_void_p(stream or 0)))

    def step_random(self, stream=None):
        L.check(L.load().gx_vec_step_random(self._h, C.c_void_p(stream or 0)))

    def rollout(self, k_steps, policy=L.POLICY_RANDOM, d_actions_ptr=None, stream=None):
        L.check(L.load().gx_vec_rollout(self._h, k_steps, policy, C.c_void_p(d_actions_ptr or 0), C.c_void_p(stream or 0)))

    def rollout_out(self, k_steps, out, policy=L.POLICY_RANDOM, d_actions_ptr=None, reward_rows=None, done_rows=None,
                    truncated_rows=None, stream=None):
        """K fused steps; out = L.OUT_ALL writes every step's reward / done / truncated into the caller's device
        rows [k_steps][stride] (device addresses), L.OUT_NONE writes none, L.OUT_LAST is rollout()."""
        L.check(L.load().gx_vec_rollout_out(self._h, k_steps, policy, C.c_void_p(d_actions_ptr or 0), out,
                                            C.c_void_p(reward_rows or 0), C.c_void_p(done_rows or 0),
                                            C.c_void_p(truncated_rows or 0), C.c_void_p(stream or 0)))

    def sample_actions(self, tick, d_actions_ptr, stream=None):
        L.check(L.load().gx_vec_sample_actions(self._h, tick, C.c_void_p(d_actions_ptr), C.c_void_p(stream or 0)))

    def view(self):
        v = L.GxView()
        L.check(L.load().gx_vec_device_view(self._h, C.byref(v)))
        return v

    def stats(self):
        s = L.GxStats()
        L.check(L.load().gx_vec_stats(self._h, C.byref(s)))
        return s


def rows_step(ctx, env_id, num_envs, stride, seed, tick, state_real, steps, reward, done, stats, k_steps=1,
              actions=None, ep_return=None, error=None, env_index_base=0, mode=L.MODE_EXACT, state_bounded=True,
              stream=None):
    """gx_rows_step: K fused steps + the statistics of the episodes they finish, on device rows the
    caller owns (every pointer argument is a device address as an int); allocates nothing."""
    r = L.GxRows(C.sizeof(L.GxRows), env_id, num_envs, stride, seed, env_index_base, tick, mode, 1 if state_bounded else 0,
                 state_real, steps, ep_return, actions, reward, done, stats, error)
    L.check(L.load().gx_rows_step(ctx._h, C.byref(r), k_steps, C.c_void_p(stream or 0)))


class _DevicePtr:
    """Minimal __cuda_array_interface__ carrier so torch can wrap a library-owned row."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}
        self._owner = owner  # keeps the batch (and its device memory) alive


_TYPESTR = {L.F32: "<f4", L.F64: "<f8", L.I32: "<i4", L.U8: "|u1"}


def torch_views(batch):
    """Zero-copy torch tensors over a batch's device rows (for agents that live on the GPU):
    dict with state_real [n_real, stride], state_int [n_int, stride], reward [stride],
    done [stride], obs (Pong, Breakout: [num_envs, 100800]).  Valid until the batch is closed.  The
    tensors carry no stream ordering: either launch the steps on torch's current stream (pass
    its pointer as `stream`) or synchronise the context before reading."""
    import torch

    v = batch.view()
    info = batch.info
    out = {}

    def wrap(ptr, shape, dt):
        return torch.as_tensor(_DevicePtr(ptr, shape, _TYPESTR[dt], batch), device="cuda")

    if info.n_real:
        out["state_real"] = wrap(v.state_real, (info.n_real, v.stride), info.real_dtype)
    out["state_int"] = wrap(v.state_int, (info.n_int, v.stride), L.I32)
    out["reward"] = wrap(v.reward, (v.stride,), info.reward_dtype)
    out["done"] = wrap(v.done, (v.stride,), L.U8)
    if v.truncated:
        out["truncated"] = wrap(v.truncated, (v.stride,), L.U8)
    if info.obs_dtype == L.U8:
        out["obs"] = wrap(v.obs, (v.num_envs, info.obs_dim), L.U8)
    elif not info.obs_is_state:
        out["obs"] = wrap(v.obs, (info.obs_dim, v.stride), L.F32)
    return out


class Comm:
    """RCCL communicator for the episode-statistics all-reduce (hosts without torch.distributed).
    Rank 0 creates the 128-byte id with Comm.unique_id() and hands it to the other ranks."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        L.check(L.load().gx_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def __init__(self, ctx, unique_id, rank, world):
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        L.check(L.load().gx_comm_init(ctx._h, C.cast(buf, C.c_void_p), rank, world, C.byref(self._h)))

    def allreduce_stats(self, batch):
        s = L.GxStats()
        L.check(L.load().gx_stats_allreduce(self._h, batch._h, C.byref(s)))
        return s

    def close(self):
        if getattr(self, "_h", None):
            L.load().gx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
