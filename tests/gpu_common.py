"""Shared helpers for the GPU parity tests: every call goes through the C ABI
(gym_amd._lib -> libgymx.so); the oracle is only the checker."""
import numpy as np

from gym_amd import _lib as L
from gym_amd.engine import Context, DeviceBatch

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = Context(0)
    return _ctx


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({1: np.uint8, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def ulp_mismatch(a, b):
    """(count of elements that differ in bits, max relative error over them)"""
    a = np.asarray(a); b = np.asarray(b)
    diff = bits(a) != bits(b)
    n = int(diff.sum())
    if n == 0:
        return 0, 0.0
    x, y = a[diff].astype(np.float64), b[diff].astype(np.float64)
    rel = np.abs(x - y) / np.maximum(np.abs(y), 1e-30)
    return n, float(rel.max())


N_ACT = {L.ENV_CARTPOLE: 2, L.ENV_MOUNTAINCAR: 3, L.ENV_ACROBOT: 3, L.ENV_FROZENLAKE: 4, L.ENV_TAXI: 6,
         L.ENV_CLIFFWALKING: 4, L.ENV_PONG: 4, L.ENV_DUMMY: 3, L.ENV_BREAKOUT: 4, L.ENV_SPACEINVADERS: 6, L.ENV_BLACKJACK: 2}


def random_actions(rng, env_id, n, wide=False):
    if env_id in (L.ENV_MOUNTAINCAR_CONT, L.ENV_PENDULUM):
        return (rng.random(n) * 6 - 3).astype(np.float32)  # beyond the Box, exercises the clamp
    if wide and env_id != L.ENV_ACROBOT:
        return rng.integers(-1, N_ACT[env_id] + 2, n).astype(np.int32)
    return rng.integers(0, N_ACT[env_id], n).astype(np.int32)


def make_pair(oracle, env_id, n, seed=1, flags=0, time_limit=0, map_rows=None, base=0, mode=L.MODE_EXACT):
    g = DeviceBatch(ctx(), env_id, n, seed=seed, flags=flags, time_limit=time_limit, map_rows=map_rows,
                    env_index_base=base, mode=mode)
    o_rows = map_rows
    if map_rows == "8x8":
        o_rows = ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"]
    o_flags = flags | (oracle.FP64 if mode == L.MODE_FP64 else 0)  # the oracle spells that mode as a flag
    o = oracle.OracleVec(env_id, n, seed=seed, flags=o_flags, time_limit=time_limit, map_rows=o_rows, env_index_base=base)
    return g, o
