"""The single-step kernels that request a lane's next trip ahead (MountainCarContinuous, Pendulum, Acrobot in EXACT
mode, MountainCarContinuous in binary64, RK4 Acrobot, Blackjack under upstream's rules; actions read from memory): with a grid of three workgroups
every lane makes several trips, the last of them with nothing left to request, and the batch ends in a ragged lane.
The results must not depend on the grid, and must be the oracle's."""
import numpy as np
import pytest

from gpu_common import L, bits, make_pair, random_actions, ulp_mismatch

pytestmark = pytest.mark.gpu

CASES = [
    ("mcc", L.ENV_MOUNTAINCAR_CONT, 0, L.MODE_EXACT),
    ("pendulum", L.ENV_PENDULUM, 0, L.MODE_EXACT),
    ("acrobot", L.ENV_ACROBOT, 0, L.MODE_EXACT),
    ("mcc-fp64", L.ENV_MOUNTAINCAR_CONT, 0, L.MODE_FP64),
    ("acrobot-rk4", L.ENV_ACROBOT, L.ACROBOT_RK4, L.MODE_EXACT),
]


def _near_the_end(rng, env_id, o, n):
    """states from which episodes end within the run, so that the auto-reset inside a trip is exercised"""
    r, i = o.get_state()
    r = r.copy(); i = i.copy()
    if env_id == L.ENV_ACROBOT:
        i[0] = rng.integers(470, 500, n)
    elif env_id == L.ENV_MOUNTAINCAR_CONT:
        r[0] = rng.uniform(0.3, 0.45, n).astype(r.dtype); r[1] = 0.06
    return r, i


@pytest.mark.parametrize("written", [True, False], ids=["written-state", "reset-state"])
@pytest.mark.parametrize("name,env_id,extra,mode", CASES, ids=[c[0] for c in CASES])
def test_trips_requested_ahead_do_not_depend_on_the_grid(oracle, monkeypatch, name, env_id, extra, mode, written):
    n, steps = 5003, 40
    flags = L.AUTORESET | L.RECORD_STATS | extra
    monkeypatch.setenv("GYMX_GRID_BLOCKS", "3")       # 20 workgroups' worth of lanes on 3: seven trips each
    few, o = make_pair(oracle, env_id, n, seed=41, flags=flags, mode=mode)
    monkeypatch.delenv("GYMX_GRID_BLOCKS")
    many, _ = make_pair(oracle, env_id, n, seed=41, flags=flags, mode=mode)  # one trip per lane
    for b in (few, many, o):
        b.reset()
    rng = np.random.default_rng(7)
    if written:  # (a written state also makes the envs with derived returns read and write their return row)
        r, i = _near_the_end(rng, env_id, o, n)
        for b in (few, many, o):
            b.set_state(r, i)
    exact_reals = mode == L.MODE_EXACT and not extra
    bad = 0
    for t in range(steps):
        a = random_actions(rng, env_id, n)
        _, rew_f, done_f, _ = few.step_host(a)
        _, rew_m, done_m, _ = many.step_host(a)
        orew, odone, _ = o.step(a)
        fr, fi = few.get_state(); mr, mi = many.get_state(); orr, oi = o.get_state()
        assert np.array_equal(bits(fr), bits(mr)) and np.array_equal(fi, mi), t
        assert np.array_equal(bits(rew_f), bits(rew_m)) and np.array_equal(done_f, done_m), t
        assert np.array_equal(fi, oi), t
        if exact_reals:
            k, rel = ulp_mismatch(fr, orr)
            assert rel <= 1e-5, (t, rel)
            bad += k
            if k == 0:
                assert np.array_equal(done_f, odone), t
            else:
                few.set_state(orr, oi); many.set_state(orr, oi)
        else:
            err = np.abs(fr - orr) / np.maximum(np.abs(orr), 1e-3)
            assert err.max() <= 1e-7, (t, err.max())
            assert np.array_equal(done_f, odone), t
    assert bad <= 2, bad
    sf, sm, so = few.stats(), many.stats(), o.stats()
    assert (sf.episodes, sf.sum_length) == (sm.episodes, sm.sum_length) == (so.episodes, so.sum_length)
    assert so.episodes > 0 or env_id == L.ENV_PENDULUM or not written
    assert abs(sf.sum_return - so.sum_return) <= 1e-9 * max(1.0, abs(so.sum_return))


def test_blackjack_upstream_rules_requested_ahead(oracle, monkeypatch):
    n, steps = 20011, 60  # three workgroups of 256 lanes, four envs per lane: seven trips, the last one ragged
    flags = L.AUTORESET | L.RECORD_STATS | L.BLACKJACK_FULL
    monkeypatch.setenv("GYMX_GRID_BLOCKS", "3")
    few, o = make_pair(oracle, L.ENV_BLACKJACK, n, seed=43, flags=flags)
    monkeypatch.delenv("GYMX_GRID_BLOCKS")
    for b in (few, o):
        b.reset()
    rng = np.random.default_rng(9)
    for t in range(steps):
        a = random_actions(rng, L.ENV_BLACKJACK, n)
        _, rew, done, _ = few.step_host(a)
        orew, odone, _ = o.step(a)
        _, fi = few.get_state(); _, oi = o.get_state()
        assert np.array_equal(fi, oi), t
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone), t
    sf, so = few.stats(), o.stats()
    assert so.episodes > n and (sf.episodes, sf.sum_length, sf.sum_return) == (so.episodes, so.sum_length, so.sum_return)
