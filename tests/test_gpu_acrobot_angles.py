"""Acrobot in EXACT mode keeps the two angles in rows of their own behind the state image, so that a step does
not take them back out of the stored cosines and sines with atan2 (Envs.Acrobot.cs:36-37 does exactly that).
The six-row form of the same kernels (GYMX_ACROBOT_ANGLES=0) and the CPU checker both take the arc tangents:
the kept angles must never show in the state image, the rewards, the flags or the statistics."""
import numpy as np
import pytest

from gym_amd import _lib as L
from gpu_common import bits, make_pair, random_actions, ulp_mismatch

pytestmark = pytest.mark.gpu

FUSED = L.AUTORESET | L.RECORD_STATS


def _six_row(monkeypatch, oracle, n, seed, flags):
    monkeypatch.setenv("GYMX_ACROBOT_ANGLES", "0")
    g, _ = make_pair(oracle, L.ENV_ACROBOT, n, seed=seed, flags=flags)
    monkeypatch.delenv("GYMX_ACROBOT_ANGLES")
    return g


def test_long_free_run_equals_the_six_row_form(oracle, monkeypatch):
    """1100 steps with auto-reset: every env meets the 500-step limit twice, so the angles of freshly reset envs
    (six small uniforms, not points of the unit circle) come from the in-kernel reset as well"""
    n = 4099
    a, _ = make_pair(oracle, L.ENV_ACROBOT, n, seed=5, flags=FUSED)
    b = _six_row(monkeypatch, oracle, n, 5, FUSED)
    a.reset(); b.reset()
    for t in range(1100):
        a.step_random(); b.step_random()
        if t % 100 == 99 or t in (498, 499, 500, 501):
            a.ctx.synchronize()
            ar, ai = a.get_state(); br, bi = b.get_state()
            cnt, rel = ulp_mismatch(ar, br)
            assert cnt == 0 and np.array_equal(ai, bi), "step %d: %d elements differ (rel %g)" % (t, cnt, rel)
    sa, sb = a.stats(), b.stats()
    assert sa.episodes == sb.episodes >= 2 * n and sa.sum_return == sb.sum_return and sa.sum_length == sb.sum_length


def test_rollout_equals_single_steps_and_the_six_row_form(oracle, monkeypatch):
    n, k = 2051, 40
    a, _ = make_pair(oracle, L.ENV_ACROBOT, n, seed=21, flags=FUSED)
    b, _ = make_pair(oracle, L.ENV_ACROBOT, n, seed=21, flags=FUSED)
    c = _six_row(monkeypatch, oracle, n, 21, FUSED)
    for g in (a, b, c):
        g.reset()
    a.rollout(k); c.rollout(k)
    for _ in range(k):
        b.step_random()
    a.ctx.synchronize()
    ar, ai = a.get_state(); br, bi = b.get_state(); cr, ci = c.get_state()
    assert np.array_equal(bits(ar), bits(br)) and np.array_equal(ai, bi)
    assert np.array_equal(bits(ar), bits(cr)) and np.array_equal(ai, ci)


@pytest.mark.parametrize("flags", [0, FUSED])
def test_written_states_then_free_steps_against_the_checker(oracle, flags):
    """a state image written from outside (any finite values: neither pair need lie on the unit circle) is followed by
    the arc tangents; the steps after it run on the kept angles"""
    n = 8192
    g, o = make_pair(oracle, L.ENV_ACROBOT, n, seed=8, flags=flags)
    g.reset(); o.reset()
    rng = np.random.default_rng(2024)
    s = (rng.standard_normal((6, n)) * np.array([3, 3, 50, 20, 20, 20])[:, None]).astype(np.float32)
    s[:, :8] = 0.0                                  # atan2(0, 0)
    s[0, 8:16] = -1.0; s[1, 8:16] = 0.0             # the negative real axis: +pi
    s[0, 16:24] = -1.0; s[1, 16:24] = -0.0          # and -pi
    st = rng.integers(0, 480, (1, n)).astype(np.int32)
    g.set_state(s, st); o.set_state(s, st)
    bad = 0
    for t in range(30):
        act = random_actions(rng, L.ENV_ACROBOT, n)
        _, rew, done, _ = g.step_host(act)
        orew, odone, _ = o.step(act)
        gr, gi = g.get_state(); orr, oi = o.get_state()
        assert np.array_equal(gi, oi), "step counters differ at step %d" % t
        cnt, rel = ulp_mismatch(gr, orr)
        assert rel <= 1e-5
        if cnt == 0:
            assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone)
        else:  # a last-place event (about one per 2^28 trigonometric calls): note it and continue from the checker's state
            bad += cnt
            g.set_state(orr, oi)
    assert bad <= 2, "%d elements differed over 30 steps of %d envs" % (bad, n)


def test_state_image_has_six_rows(oracle):
    g, _ = make_pair(oracle, L.ENV_ACROBOT, 100, seed=1)
    assert g.info.n_real == 6 and g.info.state_bytes == 100 * (6 * 4 + 4)
