"""Generates tests/golden/*.npz from the CPU oracle (oracle/gym_oracle.c).

The reference is C# and cannot run here, and its own tests hold no numeric vectors for these
envs, so these fixtures are outputs of the oracle (itself pinned by hand-derived known answers
and by the independent numpy restatement).  Per env: 256 start states x 64 steps of random
actions -- initial SoA state, the actions, every reward / done flag and the final state; for
Pong additionally 8 frames as bit masks (a frame is 0/255 with equal channels), for Breakout
and SpaceInvaders as indices into a small palette.  The binary64 mode (an extension of this
build) has its own fixtures, *_fp64.npz: 64 envs x 32 steps with the state after every step, so
that a checker can re-synchronise per step.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

N, T, SEED = 256, 64, 20250614
# black, white and the six brick rows (Envs.Breakout.cs:248-253)
BREAKOUT_PALETTE = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [255, 165, 0], [255, 255, 0], [0, 255, 0], [0, 0, 255],
                             [128, 0, 128], [255, 0, 255], [0, 255, 255]], np.uint8)  # + SpaceInvaders' magenta, cyan


def palette_index(frames):
    """uint8 [k, 100800] HWC frames -> uint8 [k, 33600] palette indices (every pixel must be a palette colour)"""
    px = frames.reshape(frames.shape[0], -1, 3).astype(np.int32)
    key = (px[..., 0] << 16) | (px[..., 1] << 8) | px[..., 2]
    pal = (BREAKOUT_PALETTE[:, 0].astype(np.int32) << 16) | (BREAKOUT_PALETTE[:, 1].astype(np.int32) << 8) | BREAKOUT_PALETTE[:, 2]
    idx = np.full(key.shape, 255, np.uint8)
    for i, c in enumerate(pal):
        idx[key == c] = i
    assert (idx != 255).all()
    return idx

CASES = [
    ("cartpole", O.CARTPOLE, 2, {}), ("mountaincar", O.MOUNTAINCAR, 3, {}), ("mountaincar_cont", O.MOUNTAINCAR_CONT, None, {}),
    ("pendulum", O.PENDULUM, None, {}), ("acrobot", O.ACROBOT, 3, {}), ("frozenlake", O.FROZENLAKE, 4, {}),
    ("frozenlake_8x8", O.FROZENLAKE, 4, {"map_rows": ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"]}),
    # extensions of this build (the reference's FrozenLake is deterministic, its Taxi a stub): pinned by the
    # oracle's own definition only
    ("frozenlake_8x8_slippery", O.FROZENLAKE, 4, {"flags": O.SLIPPERY, "map_rows": ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"]}),
    ("taxi_full", O.TAXI, 6, {"flags": O.TAXI_FULL}),
    # Blackjack with auto-reset, so that the 64 steps hold many games: the reference's ten-step stub, and
    # upstream's rules (an extension, pinned by the oracle's own definition only)
    ("blackjack", O.BLACKJACK, 2, {"flags": O.AUTORESET}),
    ("blackjack_full", O.BLACKJACK, 2, {"flags": O.BLACKJACK_FULL | O.AUTORESET}),
    ("taxi", O.TAXI, 6, {}), ("cliffwalking", O.CLIFFWALKING, 4, {}), ("pong", O.PONG, 4, {}),
    ("breakout", O.BREAKOUT, 4, {}),
    ("spaceinvaders", O.SPACEINVADERS, 6, {}),
    # an env stepped from a caller-supplied table (include/gymx.h GX_ENV_TABLE): 300 states x 5 actions x 3 equally
    # likely outcomes, twelve distinct rewards, Reset() to a uniform draw, the env's own limit of 40 steps; the table
    # itself is part of the fixture
    ("table_random", O.TABLE, 5, {"flags": O.AUTORESET, "table": "random", "n_outcomes": 3, "start_state": -1, "step_limit": 40}),
]


def random_table(rng, n_states, n_actions, n_outcomes):
    t = np.zeros((n_states, n_actions, n_outcomes), O.TRANSITION_DTYPE)
    t["next_state"] = rng.integers(0, n_states, t.shape)
    t["reward"] = rng.choice(np.array([-3, -1, 0, 1, 2, 7, 10, -20, 50, 4, 5, 6], np.float32), t.shape)
    t["done"] = rng.random(t.shape) < 0.05
    return t


CASES_FP64 = [("cartpole_fp64", O.CARTPOLE, 2), ("mountaincar_fp64", O.MOUNTAINCAR, 3), ("mountaincar_cont_fp64", O.MOUNTAINCAR_CONT, None),
              ("pendulum_fp64", O.PENDULUM, None), ("acrobot_fp64", O.ACROBOT, 3)]


def main():
    for name, eid, nact, kw in CASES:
        rng = np.random.default_rng(sum(map(ord, name)))
        steps = {"pong": 400, "breakout": 600, "spaceinvaders": 900}.get(name, T)
        kw = dict(kw)
        table = None
        if kw.get("table") == "random":
            table = kw["table"] = random_table(rng, 300, nact, kw["n_outcomes"])
        v = O.OracleVec(eid, N, seed=SEED, **kw)
        v.reset()
        init = v.get_state_raw().copy()
        if nact is None:
            actions = (rng.random((steps, N)) * 5 - 2.5).astype(np.float32)
        else:
            actions = rng.integers(0, nact, (steps, N)).astype(np.int32)
        rewards = np.empty((steps, N), np.float64)
        dones = np.empty((steps, N), np.uint8)
        frames = []
        for t in range(steps):
            rewards[t], dones[t], _ = v.step(actions[t])
            if name == "pong" and t % 50 == 49:
                f = v.get_obs()[:8]
                assert ((f == 0) | (f == 255)).all() and (f[:, 0::3] == f[:, 1::3]).all() and (f[:, 0::3] == f[:, 2::3]).all()
                frames.append(np.packbits(f[:, 0::3] == 255, axis=1))
            if name in ("breakout", "spaceinvaders") and t % 150 == 149:
                frames.append(palette_index(v.get_obs()[:8]))
        out = dict(init=init, actions=actions, rewards=rewards, dones=dones, final=v.get_state_raw(), seed=np.int64(SEED))
        if kw.get("map_rows"):
            out["map_rows"] = np.array("/".join(kw["map_rows"]))
        if kw.get("flags"):
            out["flags"] = np.int64(kw["flags"])
        if table is not None:
            out.update(table_next_state=table["next_state"], table_reward=table["reward"], table_done=table["done"],
                       start_state=np.int64(kw["start_state"]), step_limit=np.int64(kw["step_limit"]))
        if frames:
            out["frame_index" if name in ("breakout", "spaceinvaders") else "frame_masks"] = np.stack(frames)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: getattr(x, "shape", None) for k, x in out.items()})


def main_fp64():
    n, steps = 64, 32
    for name, eid, nact in CASES_FP64:
        rng = np.random.default_rng(sum(map(ord, name)))
        v = O.OracleVec(eid, n, seed=SEED, flags=O.FP64)
        v.reset()
        init = v.get_state_raw().copy()
        if nact is None:
            actions = (rng.random((steps, n)) * 5 - 2.5).astype(np.float32)
        else:
            actions = rng.integers(0, nact, (steps, n)).astype(np.int32)
        rewards = np.empty((steps, n), np.float64)
        dones = np.empty((steps, n), np.uint8)
        states = []
        for t in range(steps):
            rewards[t], dones[t], _ = v.step(actions[t])
            states.append(v.get_state_raw().copy())
        out = dict(init=init, actions=actions, rewards=rewards, dones=dones, states=np.stack(states), final=states[-1],
                   seed=np.int64(SEED))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: getattr(x, "shape", None) for k, x in out.items()})


if __name__ == "__main__":
    if "--fp64-only" in sys.argv:
        main_fp64()
        sys.exit(0)
    if "--only" in sys.argv:  # python tests/golden/make_golden.py --only name [name ...]
        CASES[:] = [c for c in CASES if c[0] in sys.argv]
        main()
        sys.exit(0)
    main()
    main_fp64()
