"""Replays recordings of the reference environments (made by csharp/ReplayRecorder.cs on a machine with a
.NET SDK and committed as tests/golden/csharp/*.gxr) through the oracle (CPU) and through the C ABI (GPU).
None can be made in the build container -- the reference is C# and there is no `dotnet` -- so until someone
commits a file those two tests are skipped, and parity with the reference itself stays unpinned.  The
format, the loader and the replay rules are exercised meanwhile on recordings written by the same code
from the oracle, which stands in for the reference there and proves nothing about it."""
import glob
import os

import numpy as np
import pytest

import reference_recording as RR

HERE = os.path.dirname(os.path.abspath(__file__))
RECORDED = sorted(glob.glob(os.path.join(HERE, "golden", "csharp", "*.gxr")))
NEED = "no recording of the reference under tests/golden/csharp/ (csharp/ReplayRecorder.cs makes them; needs a .NET SDK)"

# (oracle env constant, action count or None for a Box action, limit of the Box)
SELF_TEST = [("CARTPOLE", 2, 0), ("PENDULUM", None, 2.0), ("ACROBOT", 3, 0), ("CLIFFWALKING", 4, 0), ("BLACKJACK", 2, 0),
             ("PONG", 4, 0), ("BREAKOUT", 4, 0)]


def oracle_callables(oracle, rec):
    v = oracle.OracleVec(rec.env_id, rec.n, seed=1)
    v.reset()

    def step(a):
        r, d, _ = v.step(a)
        return r, d
    return v, v.set_state_raw, step, v.get_state_raw, v.get_obs


def draws_setter(engine, rec):
    """Pong's ball-reset values and Breakout's release value can be handed to the engine (gx_vec_set_rng_draws / the
    oracle's twin)"""
    return engine.set_rng_draws if rec.env_id in (RR.PONG, RR.BREAKOUT) else None


def make_self_recording(oracle, name, nact, limit, n=96, steps=80):
    eid = getattr(oracle, name)
    src = oracle.OracleVec(eid, n, seed=77)
    src.reset()
    rng = np.random.default_rng(len(name))
    frames = name in ("PONG", "BREAKOUT")
    if name == "PONG":  # far enough into the game that points are scored while recording
        for _ in range(40):
            src.step(rng.integers(0, 4, n).astype(np.int32))
    if nact is None:
        actions = [((rng.random(n) * 2 - 1) * 1.25 * limit).astype(np.float32) for _ in range(steps)]
    else:
        actions = [rng.integers(0, nact, n).astype(np.int32) for _ in range(steps)]
        if name == "BREAKOUT":
            actions[3][:] = 1  # FIRE releases the ball with a drawn velocity

    def step(a):
        r, d, _ = src.step(a)
        return r, d
    return RR.record_from(step, src.get_state_raw, src.get_obs, eid, n, src.n_real, src.real_is_f64, src.n_int, nact is None,
                          actions, frame_every=20 if frames else 0)


@pytest.mark.parametrize("name,nact,limit", SELF_TEST, ids=[c[0].lower() for c in SELF_TEST])
def test_format_round_trip_and_replay_rules(oracle, tmp_path, name, nact, limit):
    rec = make_self_recording(oracle, name, nact, limit)
    path = str(tmp_path / (name.lower() + ".gxr"))
    RR.write(path, rec)
    back = RR.read(path)
    assert (back.env_id, back.n, back.steps, back.n_real, back.real_is_f64, back.n_int) == \
           (rec.env_id, rec.n, rec.steps, rec.n_real, rec.real_is_f64, rec.n_int)
    assert np.array_equal(back.init, rec.init) and all(np.array_equal(a, b) for a, b in zip(back.images, rec.images))
    assert sorted(back.frames) == sorted(rec.frames)
    v, set_state, step, get_state, get_obs = oracle_callables(oracle, back)
    rep = RR.replay(back, set_state, step, get_state, get_obs)
    assert rep["real_elements_differing"] == 0
    if name in ("PONG", "BREAKOUT"):
        assert rep["unpinned_events"] > 0 and rep["frames"] == 4
    if name in ("PONG", "BREAKOUT"):  # with the drawn values handed over nothing is left out
        full = RR.replay(back, set_state, step, get_state, get_obs, v.set_rng_draws)
        assert full["unpinned_events"] == 0 and full["replayed_draws"] == rep["unpinned_events"]
        assert full["real_elements_differing"] == 0 and full["real_elements"] > rep["real_elements"]
    # a corrupted reward or state must be caught
    back.rewards[5][3] += 1.0
    with pytest.raises(AssertionError):
        RR.replay(back, set_state, step, get_state, get_obs)
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 7)
    with pytest.raises(ValueError):
        RR.read(path)


@pytest.mark.gpu
@pytest.mark.parametrize("name,nact,limit", SELF_TEST, ids=[c[0].lower() for c in SELF_TEST])
def test_replay_rules_through_the_abi(oracle, name, nact, limit):
    from gpu_common import ctx
    from gym_amd.engine import DeviceBatch

    rec = make_self_recording(oracle, name, nact, limit)
    g = DeviceBatch(ctx(), rec.env_id, rec.n, seed=1)
    g.reset()

    def step(a):
        _, r, d, _ = g.step_host(a)
        return r, d
    rep = RR.replay(rec, g.set_state_raw, step, g.get_state_raw, g.get_obs, draws_setter(g, rec))
    assert rep["real_elements_differing"] <= 2, rep
    if name in ("PONG", "BREAKOUT"):
        assert rep["unpinned_events"] == 0 and rep["replayed_draws"] > 0, rep


@pytest.mark.skipif(not RECORDED, reason=NEED)
@pytest.mark.parametrize("path", RECORDED or [None], ids=[os.path.basename(p) for p in RECORDED] or ["none"])
def test_oracle_reproduces_the_reference(oracle, path):
    rec = RR.read(path)
    v, set_state, step, get_state, get_obs = oracle_callables(oracle, rec)
    rep = RR.replay(rec, set_state, step, get_state, get_obs, draws_setter(v, rec))
    print(os.path.basename(path), rep)


@pytest.mark.gpu
@pytest.mark.skipif(not RECORDED, reason=NEED)
@pytest.mark.parametrize("path", RECORDED or [None], ids=[os.path.basename(p) for p in RECORDED] or ["none"])
def test_gpu_reproduces_the_reference(path):
    from gpu_common import ctx
    from gym_amd.engine import DeviceBatch

    rec = RR.read(path)
    g = DeviceBatch(ctx(), rec.env_id, rec.n, seed=1)
    g.reset()

    def step(a):
        _, r, d, _ = g.step_host(a)
        return r, d
    rep = RR.replay(rec, g.set_state_raw, step, g.get_state_raw, g.get_obs, draws_setter(g, rec))
    print(os.path.basename(path), rep)
