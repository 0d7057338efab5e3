"""Transition tables of the reference's table-driven envs written out as gx_transition arrays, and random tables:
what the tests feed through the caller-supplied-table door (GX_ENV_TABLE) of the engine and of the oracle."""
import numpy as np

DT = np.dtype([("next_state", np.int32), ("reward", np.float32), ("done", np.int32)])
MAP4 = ["SFFF", "FHFH", "FFFH", "HFFG"]  # Envs.FrozenLake.cs:25-30
MAP8 = ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"]


def _move(s, a, nrow, ncol):
    """Envs.FrozenLake.cs:54-63 / Envs.CliffWalking.cs:38-47"""
    row, col = divmod(s, ncol)
    if a == 0:
        col = max(col - 1, 0)
    elif a == 1:
        row = min(row + 1, nrow - 1)
    elif a == 2:
        col = min(col + 1, ncol - 1)
    elif a == 3:
        row = max(row - 1, 0)
    return row * ncol + col


def frozenlake(rows=MAP4, slippery=False):
    """Envs.FrozenLake.cs:52-69 as P[s][a][o]; slippery: outcome o of action a is action (a + 3 + o) mod 4"""
    nrow, ncol = len(rows), len(rows[0])
    flat = "".join(rows)
    goal = flat.rfind("G")
    n_out = 3 if slippery else 1
    t = np.zeros((nrow * ncol, 4, n_out), DT)
    for s in range(nrow * ncol):
        for a in range(4):
            for o in range(n_out):
                ns = _move(s, (a + 3 + o) & 3 if slippery else a, nrow, ncol)
                t[s, a, o] = (ns, 1.0 if ns == goal else 0.0, int(ns == goal or flat[ns] == "H"))
    return t


def cliffwalking():
    """Envs.CliffWalking.cs:36-54: 4x12, start 36, goal 47, cliff 37..46"""
    t = np.zeros((48, 4), DT)
    for s in range(48):
        for a in range(4):
            ns = _move(s, a, 4, 12)
            cliff = 37 <= ns <= 46
            t[s, a] = (ns, 0.0 if ns == 47 else (-100.0 if cliff else -1.0), int(ns == 47 or cliff))
    return t


def taxi_stub():
    """Envs.Taxi.cs:25-33: the state never changes, every step pays -1"""
    t = np.zeros((500, 6), DT)
    t["next_state"] = np.arange(500, dtype=np.int32)[:, None]
    t["reward"] = -1.0
    return t


def random_table(rng, n_states, n_actions, n_outcomes=1, rewards=(-3, -1, 0, 1, 2, 7, 10, -20, 50, 4, 5, 6), p_done=0.05):
    t = np.zeros((n_states, n_actions, n_outcomes), DT)
    t["next_state"] = rng.integers(0, n_states, t.shape)
    t["reward"] = rng.choice(np.asarray(rewards, np.float32), t.shape)
    t["done"] = rng.random(t.shape) < p_done
    return t
