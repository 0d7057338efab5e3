"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact for integer/byte work (tabular envs, Pong game
state and frames, counters, flags); for continuous dynamics the bar is
relative error <= 1e-5 per step with the device state re-synchronised to the
oracle's after any step that differs (teacher forcing), and in practice the
results are bit-identical except where the device's binary64 sin/cos/atan2
and glibc's differ in the last place AND that flips the float rounding.
"""
import numpy as np
import pytest

from gpu_common import L, bits, make_pair, random_actions, ulp_mismatch

pytestmark = pytest.mark.gpu

CLASSIC = [L.ENV_CARTPOLE, L.ENV_MOUNTAINCAR, L.ENV_MOUNTAINCAR_CONT, L.ENV_PENDULUM, L.ENV_ACROBOT]
TABULAR = [L.ENV_FROZENLAKE, L.ENV_TAXI, L.ENV_CLIFFWALKING, L.ENV_DUMMY]
REL_TOL = 1e-5           # stated tolerance for continuous dynamics, binary32
MAX_MISMATCH_FRAC = 1e-5  # fraction of elements allowed to differ in the last place


def compare_step(g, o, a, t, exact):
    """one host-path step on both engines; returns number of state elements that differ"""
    obs, rew, done, _ = g.step_host(a)
    orew, odone, _ = o.step(a)
    gr, gi = g.get_state()
    orr, oi = o.get_state()
    assert np.array_equal(gi, oi), "int state differs at step %d" % t
    if exact:
        assert np.array_equal(bits(gr), bits(orr)), "state differs at step %d" % t
        assert np.array_equal(rew.astype(np.float64), orew), "reward differs at step %d" % t
        assert np.array_equal(done, odone), "done differs at step %d" % t
        return 0
    n, rel = ulp_mismatch(gr, orr)
    assert rel <= REL_TOL, "state relative error %g at step %d" % (rel, t)
    nr, relr = ulp_mismatch(rew.astype(np.float64), orew)
    assert relr <= REL_TOL
    if n == 0:
        assert np.array_equal(done, odone)
    else:
        g.set_state(orr, oi)  # teacher forcing
    return n


@pytest.mark.parametrize("env_id", CLASSIC + TABULAR + [L.ENV_PONG, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS])
def test_reset_matches_oracle(oracle, env_id):
    g, o = make_pair(oracle, env_id, 1000, seed=77, base=5)
    g.reset(); o.reset()
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
    g.reset(); o.reset()  # second reset uses a new tick, hence new draws
    gr2, _ = g.get_state(); orr2, _ = o.get_state()
    assert np.array_equal(bits(gr2), bits(orr2))
    if gr.size and env_id not in (L.ENV_PONG, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS):
        assert not np.array_equal(gr, gr2)
    assert np.array_equal(bits(g.get_obs()), bits(o.get_obs()))


@pytest.mark.parametrize("env_id", CLASSIC)
def test_classic_free_run(oracle, env_id):
    n, steps = 4099, 96  # ragged: not a multiple of the 4-env lane width
    g, o = make_pair(oracle, env_id, n, seed=11)
    g.reset(); o.reset()
    rng = np.random.default_rng(env_id)
    bad = 0
    for t in range(steps):
        bad += compare_step(g, o, random_actions(rng, env_id, n), t, exact=False)
    total = n * g.info.n_real * steps
    assert bad <= max(2, MAX_MISMATCH_FRAC * total), "%d of %d elements differ" % (bad, total)


@pytest.mark.parametrize("env_id", CLASSIC)
def test_classic_wide_states(oracle, env_id):
    """states far outside the reset range: large angles, saturated velocities, step counters near the limit"""
    n = 8192
    g, o = make_pair(oracle, env_id, n, seed=3)
    g.reset(); o.reset()
    rng = np.random.default_rng(100 + env_id)
    dim = g.info.n_real
    scale = np.array([3, 3, 50, 20, 20, 20][:dim])[:, None]
    s = (rng.standard_normal((dim, n)) * scale).astype(np.float32)
    st = rng.integers(0, 600, (1, n)).astype(np.int32)
    g.set_state(s, st); o.set_state(s, st)
    bad = compare_step(g, o, random_actions(rng, env_id, n), 0, exact=False)
    assert bad <= 2


@pytest.mark.parametrize("env_id", TABULAR)
def test_tabular_bit_exact(oracle, env_id):
    n, steps = 4099, 230
    g, o = make_pair(oracle, env_id, n, seed=5)
    g.reset(); o.reset()
    rng = np.random.default_rng(env_id)
    for t in range(steps):
        compare_step(g, o, random_actions(rng, env_id, n, wide=True), t, exact=True)
        assert np.array_equal(g.get_obs(), o.get_obs())


@pytest.mark.parametrize("rows", ["8x8", ["SFG", "HFF"], ["SFFFH", "FFHFF", "FFFFG"]])
def test_frozenlake_maps(oracle, rows):
    n = 1024
    g, o = make_pair(oracle, L.ENV_FROZENLAKE, n, map_rows=rows)
    g.reset(); o.reset()
    rng = np.random.default_rng(9)
    for t in range(120):
        compare_step(g, o, random_actions(rng, L.ENV_FROZENLAKE, n, wide=True), t, exact=True)


def test_frozenlake_slippery_and_taxi_full_extensions(oracle):
    """extensions (not in the reference): pinned by the oracle's own definition only"""
    n = 4096
    for env_id, flags, rows in [(L.ENV_FROZENLAKE, L.SLIPPERY, "8x8"), (L.ENV_TAXI, L.TAXI_FULL, None)]:
        g, o = make_pair(oracle, env_id, n, seed=21, flags=flags | L.AUTORESET, map_rows=rows)
        g.reset(); o.reset()
        rng = np.random.default_rng(2)
        for t in range(150):
            compare_step(g, o, random_actions(rng, env_id, n), t, exact=True)


def test_dummy_env_reference_kat_through_abi():
    """Gymnasium.Tests/DummyEnvTests.cs:6-21, through libgymx.so on the GPU."""
    from gym_amd.engine import DeviceBatch
    from gpu_common import ctx
    g = DeviceBatch(ctx(), L.ENV_DUMMY, 1)
    g.reset()
    assert g.get_obs()[0] == 0
    _, r, d, _ = g.step_host([1])
    assert g.get_obs()[0] == 1 and r[0] == 1.0 and not d[0]
    for _ in range(9):
        _, r, d, _ = g.step_host([1])
    assert d[0]


def test_hand_derived_paths_through_abi():
    from gym_amd.engine import DeviceBatch
    from gpu_common import ctx
    g = DeviceBatch(ctx(), L.ENV_FROZENLAKE, 1)
    g.reset()
    states, rewards, dones = [], [], []
    for a in [1, 1, 2, 1, 2, 2]:  # Envs.FrozenLake.cs:52-69 on the default map
        _, r, d, _ = g.step_host([a])
        states.append(int(g.get_obs()[0])); rewards.append(float(r[0])); dones.append(bool(d[0]))
    assert states == [4, 8, 9, 13, 14, 15] and rewards == [0, 0, 0, 0, 0, 1] and dones == [False] * 5 + [True]
    c = DeviceBatch(ctx(), L.ENV_CLIFFWALKING, 1)
    c.reset()
    out = []
    for a in [3] + [2] * 11 + [1]:  # Envs.CliffWalking.cs:36-54
        _, r, d, _ = c.step_host([a])
        out.append((int(c.get_obs()[0]), float(r[0]), bool(d[0])))
    assert [s for s, _, _ in out] == [24] + list(range(25, 36)) + [47]
    assert [r for _, r, _ in out] == [-1.0] * 12 + [0.0] and [d for _, _, d in out] == [False] * 12 + [True]
    c.reset()
    _, r, d, _ = c.step_host([2])
    assert c.get_obs()[0] == 37 and r[0] == -100.0 and d[0]
    t = DeviceBatch(ctx(), L.ENV_TAXI, 2, seed=4)
    t.reset()
    s0 = t.get_obs().copy()
    for k in range(200):  # Envs.Taxi.cs:25-33
        _, r, d, _ = t.step_host([k % 6, 5])
        assert (r == -1.0).all() and (t.get_obs() == s0).all() and d.all() == (k == 199)


def test_pong_state_and_frames_bit_exact(oracle):
    n, steps = 257, 600
    g, o = make_pair(oracle, L.ENV_PONG, n, seed=8)
    assert np.array_equal(g.get_obs(), o.get_obs())  # the constructor resets (Envs.Pong.cs:47-52)
    g.reset(); o.reset()
    assert np.array_equal(g.get_obs(), o.get_obs())
    rng = np.random.default_rng(4)
    scored = 0
    for t in range(steps):
        a = random_actions(rng, L.ENV_PONG, n, wide=True)
        obs, rew, done, _ = g.step_host(a, want_obs=(t % 50 == 0))
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone)
        scored += int((orew != 0).sum())
        if obs is not None:
            assert np.array_equal(obs, o.get_obs()), "frames differ at step %d" % t
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
    assert scored > 0
    assert np.array_equal(g.get_obs(), o.get_obs())


def test_pong_frames_edge_states(oracle):
    """ball in every corner, paddles at both ends, one-, two- and many-digit scores"""
    cases = np.array([
        [160.0, 208.0, 3, 1, 0.0, 195.0, 21, 9, 5], [0.0, 0.0, -3, -1, 195.0, 0.0, 10, 123, 7],
        [159.2, 104.9, 3, 0, 3.2, 190.8, 0, 21, 1], [79.5, 19.5, 3, 0, 97.5, 97.5, 7, 8, 2],
        [38.0, 21.0, -3, 2, 50.0, 60.0, 4, 1234567, 3], [118.0, 23.9, 3, 2, 0.4, 194.9, 2147483647, 99, 0],
    ])
    n = len(cases)
    g, o = make_pair(oracle, L.ENV_PONG, n)
    reals = cases[:, :6].T.copy()
    ints = cases[:, 6:].T.astype(np.int32).copy()
    g.set_state(reals, ints); o.set_state(reals, ints)
    assert np.array_equal(g.get_obs(), o.get_obs())


def _invaders_actions(rng, reals, ints, n):
    """odd envs act at random (with out-of-range actions); even envs move under the nearest occupied
    column and fire, which clears waves and draws alien fire onto the cannon"""
    a = rng.integers(-1, 8, n).astype(np.int32)
    bits_ = (ints[6].astype(np.int64) & 0xFFFFFFFF) | ((ints[7].astype(np.int64) & 0xFFFFFFFF) << 32)
    for i in range(0, n, 2):
        cols = [c for c in range(11) if any((int(bits_[i]) >> (r * 11 + c)) & 1 for r in range(5))]
        if not cols:
            a[i] = 1
            continue
        gun = reals[0, i] + 4.0
        target = min((reals[1, i] + c * 12 + 3.0 for c in cols), key=lambda x: abs(x - gun))
        a[i] = 4 if target > gun + 2 else (5 if target < gun - 2 else 1)
    return a


def test_invaders_state_and_frames_bit_exact(oracle):
    n, steps = 66, 2400
    g, o = make_pair(oracle, L.ENV_SPACEINVADERS, n, seed=8)
    assert np.array_equal(g.get_obs(), o.get_obs())  # the constructor resets (Envs.SpaceInvaders.cs:59-67)
    g.reset(); o.reset()
    assert np.array_equal(g.get_obs(), o.get_obs())
    rng = np.random.default_rng(4)
    total = np.zeros(n)
    waves = hit = fired = 0
    for t in range(steps):
        reals, ints = o.get_state()
        fired += int((ints[5] > 0).sum())
        a = _invaders_actions(rng, reals, ints, n)
        obs, rew, done, _ = g.step_host(a, want_obs=(t % 200 == 0 or t == 37))
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone), t
        waves += int((orew >= 100).sum()); hit += int((orew < 0).sum())
        if obs is not None:
            assert np.array_equal(obs, o.get_obs()), "frames differ at step %d" % t
        if t % 100 == 99:
            gr, gi = g.get_state(); orr, oi = o.get_state()
            assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi), t
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
    assert waves > 0 and hit > 0 and fired > 0  # cleared waves, alien fire and hits on the cannon all occurred
    assert np.array_equal(g.get_obs(), o.get_obs())


def test_invaders_edge_frames_and_injected_states(oracle):
    """sprites clipped at every border, bullets over sprites, full bullet lists, many-digit and negative scores,
    the formation at its lowest, one alien left"""
    #        playerX formX formY  pb0        pb1           pb2          ab0          ab1      | score lives frame right npb nab bits0 bits1 steps
    cases = [
        [0.0, 10.0, 40.0, 3.9, 0.0, 157.2, 208.9, 80.0, 100.0, 12.0, 42.0, 159.0, 170.5, 123456, 3, 7, 1, 3, 2, -1, 0x7FFFFF, 7],
        [152.0, 27.5, 120.0, 30.0, 121.0, 0, 0, 0, 0, 156.0, 172.0, 0, 0, -50, 1, 9, 0, 1, 1, 0x55555555, 0x2AAAAA, 9],
        [75.4, -3.0, 161.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 990, 0, 0, 1, 0, 0, -1, 0x7FFFFF, 0],
        [80.0, 150.0, 5.0, 11.0, 11.0, 121.0, 12.0, 100.0, 190.0, 50.0, 189.5, 20.0, 12.0, 2147483647, 13, 5, 1, 3, 2, 1 << 10, 1 << 22, 1],
        [76.0, 18.0, 128.0, 80.0, 5.9, 80.0, 140.0, 33.0, 133.0, 79.0, 168.0, 0, 0, 0, 3, 59, 1, 3, 1, 0, 1 << 18, 59],   # last alien, wave clear
        [76.0, 18.0, 122.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 10, 3, 9, 1, 0, 0, -1, 0x7FFFFF, 9999],                           # drop to the bottom
    ]
    cases = np.array(cases, np.float64)
    n = len(cases)
    g, o = make_pair(oracle, L.ENV_SPACEINVADERS, n, seed=5)
    reals = cases[:, :13].T.copy()
    ints = cases[:, 13:].T.astype(np.int64).astype(np.uint32).view(np.int32).copy()
    g.set_state(reals, ints); o.set_state(reals, ints)
    assert np.array_equal(g.get_obs(), o.get_obs())
    for t in range(4):
        a = np.array([4, 5, 1, 2, 1, 0], np.int32)
        obs, rew, done, _ = g.step_host(a, want_obs=True)
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone), t
        gr, gi = g.get_state(); orr, oi = o.get_state()
        assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi), t
        assert np.array_equal(obs, o.get_obs()), t


def _breakout_actions(rng, g, n):
    """half the envs act at random (with out-of-range actions), the others fire and follow the ball"""
    reals, ints = g.get_state()
    a = rng.integers(-1, 6, n).astype(np.int32)
    track = np.where(ints[2] != 0, 1, np.where(reals[0] + 1 > reals[4] + 10, 2, np.where(reals[0] + 1 < reals[4] + 6, 3, 0)))
    a[n // 2:] = track[n // 2:]
    return a


def test_breakout_state_and_frames_bit_exact(oracle):
    n, steps = 257, 1400
    g, o = make_pair(oracle, L.ENV_BREAKOUT, n, seed=8)
    assert np.array_equal(g.get_obs(), o.get_obs())  # the constructor resets (Envs.Breakout.cs:54-60)
    g.reset(); o.reset()
    assert np.array_equal(g.get_obs(), o.get_obs())
    rng = np.random.default_rng(4)
    total = np.zeros(n)
    for t in range(steps):
        a = _breakout_actions(rng, g, n)
        obs, rew, done, _ = g.step_host(a, want_obs=(t % 100 == 0))
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone), t
        total += orew
        if obs is not None:
            assert np.array_equal(obs, o.get_obs()), "frames differ at step %d" % t
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
    assert total.max() > 100 and (gi[1] <= 0).any() and (gi[1] > 0).any()  # long rallies and lost games both occurred
    assert np.array_equal(g.get_obs(), o.get_obs())


def test_breakout_win_and_edge_frames(oracle):
    """the last brick pays its row value + 100 and ends the game (Envs.Breakout.cs:182-199); frames at edge states"""
    full = [-1, -1, -1, 0xFFF]
    #        ballX  ballY velX velY paddleX | score lives stuck b0 b1 b2 b3 steps
    cases = [
        [43.0, 79.0, 0.5, -3.0, 72.0, 0, 5, 0, 0, 1 << 9, 0, 0, 3],          # about to enter the last brick (row 2, col 5)
        [43.0, 79.0, 0.5, -3.0, 72.0, 0, 5, 0, 1, 1 << 9, 0, 0, 3],          # same, another brick remains
        [0.0, 0.0, -3, -3, 0.0, 0, 5, 0] + full + [0], [158.9, 209.5, 3, 3, 144.0, 486, 1, 0, 0, 0, 0, 0, 9],
        [80.0, 180.0, 0, 0, 72.0, 7, 5, 1] + full + [1], [12.3, 11.7, 1, 1, 143.9, 1234567890, 9, 0, 0x12345678, -1698898192, 0x0F0F0F0F, 0xA5A, 5],
        [10.0, 25.0, 0, 0, 3.5, -42, 0, 1] + full + [2], [75.5, 250.0, 0, 3, 20.0, 99, -1, 0, 1, 0, 0, 0x800, 4],
        [3.0, 62.0, 1, -3, 0.0, 10, 30, 0] + full + [6], [100.0, 208.5, 0.2, 3.0, 90.0, 0, 1, 0] + full + [9999],
    ]
    cases = np.array(cases, np.float64)
    n = len(cases)
    g, o = make_pair(oracle, L.ENV_BREAKOUT, n)
    reals = cases[:, :5].T.copy()
    ints = cases[:, 5:].T.astype(np.int64).astype(np.int32).copy()
    g.set_state(reals, ints); o.set_state(reals, ints)
    assert np.array_equal(g.get_obs(), o.get_obs())
    a = np.zeros(n, np.int32)
    obs, rew, done, _ = g.step_host(a, want_obs=True)
    orew, odone, _ = o.step(a)
    assert rew[0] == 105.0 and done[0] == 1 and rew[1] == 5.0 and done[1] == 0
    assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone)
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
    assert np.array_equal(obs, o.get_obs())


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_PENDULUM, L.ENV_CLIFFWALKING, L.ENV_PONG, L.ENV_DUMMY, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS])
@pytest.mark.parametrize("flags,time_limit", [(L.AUTORESET | L.RECORD_STATS, 0), (L.RECORD_STATS, 17), (L.AUTORESET | L.RECORD_STATS, 9), (L.AUTORESET, 0)])
def test_fused_wrappers(oracle, env_id, flags, time_limit):
    """auto-reset, TimeLimit (Wrappers.TimeLimit.cs) and RecordEpisodeStatistics (EnvWrapper.cs:23-58) fused in the step"""
    n, steps = 1027, 70
    g, o = make_pair(oracle, env_id, n, seed=13, flags=flags, time_limit=time_limit, base=1 << 33)
    g.reset(); o.reset()
    rng = np.random.default_rng(6)
    exact = env_id in (L.ENV_CLIFFWALKING, L.ENV_PONG, L.ENV_DUMMY, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS)
    for t in range(steps):
        a = random_actions(rng, env_id, n)
        obs, rew, done, trunc = g.step_host(a, want_trunc=time_limit > 0)
        orew, odone, otrunc = o.step(a)
        gr, gi = g.get_state(); orr, oi = o.get_state()
        assert np.array_equal(gi, oi) and np.array_equal(done, odone)
        if time_limit > 0:
            assert np.array_equal(trunc, otrunc)
        if exact:
            assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(rew.astype(np.float64), orew)
        else:
            cnt, rel = ulp_mismatch(gr, orr)
            assert cnt == 0, "state mismatch (%d elements, rel %g) at step %d" % (cnt, rel, t)
        if t == 40:
            g.reset(); o.reset()  # EnvWrapper.cs:34-38: open episodes are recorded by Reset()
    if flags & L.RECORD_STATS:
        gs, os_ = g.stats(), o.stats()
        assert gs.episodes == os_.episodes and gs.sum_length == os_.sum_length
        assert gs.episodes > 0
        if env_id == L.ENV_PENDULUM:
            assert abs(gs.sum_return - os_.sum_return) <= 1e-9 * abs(os_.sum_return)
            assert abs(gs.min_return - os_.min_return) <= 1e-9 * abs(os_.min_return)
        else:
            assert gs.sum_return == os_.sum_return and gs.sum_return_sq == os_.sum_return_sq
            assert gs.min_return == os_.min_return and gs.max_return == os_.max_return


@pytest.mark.parametrize("env_id", CLASSIC + [L.ENV_FROZENLAKE, L.ENV_TAXI, L.ENV_PONG, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS])
def test_random_policy_and_rollout(oracle, env_id):
    """actions sampled on the device (Spaces.Discrete.cs:18 / Spaces.Box.cs:25-33) and K fused steps"""
    n, k = 2051, 24
    flags = L.AUTORESET | L.RECORD_STATS
    g, o = make_pair(oracle, env_id, n, seed=99, flags=flags)
    g2, _ = make_pair(oracle, env_id, n, seed=99, flags=flags)
    g.reset(); o.reset(); g2.reset()
    for t in range(k):
        g.step_random()
        o.step_random()
    g2.rollout(k)
    g.ctx.synchronize()
    gr, gi = g.get_state(); g2r, g2i = g2.get_state(); orr, oi = o.get_state()
    # K single steps and one K-step rollout are the same computation
    assert np.array_equal(bits(gr), bits(g2r)) and np.array_equal(gi, g2i)
    s1, s2, so = g.stats(), g2.stats(), o.stats()
    assert s1.episodes == s2.episodes and s1.sum_length == s2.sum_length
    if env_id in (L.ENV_FROZENLAKE, L.ENV_TAXI, L.ENV_PONG, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS):
        assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
        assert s1.episodes == so.episodes and s1.sum_return == so.sum_return and s1.sum_length == so.sum_length
    else:
        # free-running chaotic dynamics: compare only while no last-place event occurred
        cnt, rel = ulp_mismatch(gr, orr)
        assert cnt <= 8, "%d elements differ (rel %g)" % (cnt, rel)
        if cnt == 0:
            assert np.array_equal(gi, oi) and s1.episodes == so.episodes and s1.sum_length == so.sum_length


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_MOUNTAINCAR, L.ENV_ACROBOT, L.ENV_FROZENLAKE, L.ENV_CLIFFWALKING, L.ENV_TAXI])
def test_fused_statistics_survive_set_state(oracle, env_id):
    """auto-reset + statistics on a constant-reward env: the open-episode return is implied by the step
    counter until the caller replaces the state (arbitrary counters), then it is tracked explicitly"""
    n = 2051
    flags = L.AUTORESET | L.RECORD_STATS
    g, o = make_pair(oracle, env_id, n, seed=17, flags=flags)
    g.reset(); o.reset()
    rng = np.random.default_rng(3)
    for t in range(90):
        a = random_actions(rng, env_id, n)
        g.step_host(a); o.step(a)
        if t in (30, 60):
            reals, ints = o.get_state()
            ints = ints.copy()
            # any step counters, unrelated to the episodes' true lengths
            ints[-1] = rng.integers(0, 90 if env_id in TABULAR else 480, n)
            g.set_state(reals, ints); o.set_state(reals, ints)
        if t == 75:
            g.reset(); o.reset()
    gs, os_ = g.stats(), o.stats()
    assert gs.episodes == os_.episodes > 0 and gs.sum_length == os_.sum_length
    assert gs.sum_return == os_.sum_return and gs.sum_return_sq == os_.sum_return_sq
    assert gs.min_return == os_.min_return and gs.max_return == os_.max_return


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_MOUNTAINCAR])
def test_fast32_mode_with_autoreset(oracle, env_id):
    """FAST32 with auto-reset takes the bounded-argument binary32 routines; envs whose done flag ever differs from
    the oracle's (a state within 1e-7 of a threshold) are left out from then on, there may be only a handful.
    The run is free (set_state would switch the bounded routines off for a step), so the 1e-5 bounds the error
    accumulated since an env's last reset, over a dozen steps"""
    n = 8192
    g, o = make_pair(oracle, env_id, n, seed=33, mode=L.MODE_FAST32, flags=L.AUTORESET)
    g.reset(); o.reset()
    rng = np.random.default_rng(env_id)
    ok = np.ones(n, bool)
    for t in range(12):
        a = random_actions(rng, env_id, n)
        _, _, done, _ = g.step_host(a)
        _, odone, _ = o.step(a)
        ok &= done == odone
        gr, _ = g.get_state(); orr, _ = o.get_state()
        err = np.abs(gr.astype(np.float64) - orr) / np.maximum(np.abs(orr), 1e-1)
        assert err[:, ok].max() <= 1e-5, (t, err[:, ok].max())
    assert ok.sum() >= n - 8


def test_acrobot_invalid_action_is_an_error(oracle):
    """Envs.Acrobot.cs:45,51 throws IndexOutOfRangeException; the ABI returns GX_ERR_INVALID"""
    from gym_amd._lib import GymxError
    g, _ = make_pair(oracle, L.ENV_ACROBOT, 8)
    g.reset()
    with pytest.raises(GymxError) as e:
        g.step_host(np.array([0, 1, 2, 3, 0, 0, 0, 0], np.int32))
    assert e.value.code == L.GX_ERR_INVALID


def test_fast32_mode_within_tolerance(oracle):
    n = 4096
    for env_id in CLASSIC:
        g, o = make_pair(oracle, env_id, n, seed=31, mode=L.MODE_FAST32)
        g.reset(); o.reset()
        rng = np.random.default_rng(env_id)
        for t in range(30):
            a = random_actions(rng, env_id, n)
            g.step_host(a); o.step(a)
            gr, gi = g.get_state(); orr, oi = o.get_state()
            # relative to the value, floored at 0.1: a velocity that cancels to ~0, or cos(theta)
            # next to its zero crossing, has no meaningful relative error in binary32 (half an
            # ulp of theta ~ 1.57 is already 6e-8 absolute)
            err = np.abs(gr.astype(np.float64) - orr) / np.maximum(np.abs(orr), 1e-1)
            assert err.max() <= 1e-5, (env_id, t, err.max())
            g.set_state(orr, oi)  # teacher forcing per step


def test_acrobot_rk4_fp64_extension(oracle):
    """extension, pinned only by the oracle's own restatement: relative 1e-9 per step, teacher-forced"""
    n = 4096
    g, o = make_pair(oracle, L.ENV_ACROBOT, n, seed=17, flags=L.ACROBOT_RK4)
    g.reset(); o.reset()
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr))
    rng = np.random.default_rng(3)
    for t in range(40):
        a = random_actions(rng, L.ENV_ACROBOT, n)
        _, rew, done, _ = g.step_host(a)
        orew, odone, _ = o.step(a)
        gr, gi = g.get_state(); orr, oi = o.get_state()
        err = np.abs(gr - orr) / np.maximum(np.abs(orr), 1e-6)
        assert err.max() <= 1e-9, (t, err.max())
        assert np.array_equal(gi, oi)
        gobs, oobs = g.get_obs(), o.get_obs()
        assert np.abs(gobs - oobs).max() <= 1e-6
        g.set_state(orr, oi)


def test_rccl_stats_allreduce_single_rank(oracle):
    """gx_comm_* / gx_stats_allreduce over RCCL with a world of one: the reduced record equals the local one"""
    from gym_amd.engine import Comm
    g, o = make_pair(oracle, L.ENV_CARTPOLE, 4096, seed=3, flags=L.AUTORESET | L.RECORD_STATS)
    g.reset(); o.reset()
    for _ in range(40):
        g.step_random(); o.step_random()
    comm = Comm(g.ctx, Comm.unique_id(), 0, 1)
    s, local, so = comm.allreduce_stats(g), g.stats(), o.stats()
    comm.close()
    for f in ("episodes", "sum_return", "sum_length", "sum_return_sq", "min_return", "max_return"):
        assert getattr(s, f) == getattr(local, f) == getattr(so, f), f


FP64_TOL = 1e-9  # stated tolerance of the binary64 mode, relative per step (floored at 1e-6 in magnitude)


@pytest.mark.parametrize("env_id", CLASSIC)
def test_fp64_mode_teacher_forced(oracle, env_id):
    """extension, pinned only by the oracle's own binary64 restatement"""
    n = 4099  # ragged: not a multiple of the 2-env lane width
    g, o = make_pair(oracle, env_id, n, seed=23, mode=L.MODE_FP64, base=3)
    assert g.info.real_dtype == L.F64 and g.info.obs_dtype == L.F64 and g.info.obs_is_state == 1
    g.reset(); o.reset()
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert gr.dtype == np.float64 and np.array_equal(gi, oi)
    if env_id != L.ENV_PENDULUM:  # Pendulum's reset takes a cos and a sin
        assert np.array_equal(bits(gr), bits(orr))
    assert np.abs(gr - orr).max() <= 1e-15
    assert np.array_equal(bits(g.get_obs()), bits(gr))
    g.set_state(orr, oi)
    rng = np.random.default_rng(env_id)
    for t in range(40):
        a = random_actions(rng, env_id, n)
        obs, rew, done, _ = g.step_host(a, want_obs=True)
        orew, odone, _ = o.step(a)
        gr, gi = g.get_state(); orr, oi = o.get_state()
        err = np.abs(gr - orr) / np.maximum(np.abs(orr), 1e-6)
        assert err.max() <= FP64_TOL, (t, err.max())
        assert np.array_equal(gi, oi)
        assert np.abs(rew.astype(np.float64) - orew).max() <= FP64_TOL * max(1.0, np.abs(orew).max())
        assert np.array_equal(done, odone)
        assert obs.dtype == np.float64 and np.array_equal(bits(obs), bits(gr))
        g.set_state(orr, oi)


@pytest.mark.parametrize("env_id", CLASSIC)
def test_fp64_mode_fused_autoreset_statistics(oracle, env_id):
    """auto-reset + RecordEpisodeStatistics in the binary64 kernels (the BOUNDED variants), free-running:
    short enough that no env sits within rounding of a termination threshold"""
    n, steps = 2048, 60
    flags = L.AUTORESET | L.RECORD_STATS
    g, o = make_pair(oracle, env_id, n, seed=29, flags=flags, mode=L.MODE_FP64)
    g.reset(); o.reset()
    rng = np.random.default_rng(50 + env_id)
    if env_id in (L.ENV_MOUNTAINCAR, L.ENV_MOUNTAINCAR_CONT, L.ENV_ACROBOT):
        # bring episodes to an end within the run: counters close to the limits, positions near the goal
        gr, gi = o.get_state()
        gr = gr.copy(); gi = gi.copy()
        if env_id == L.ENV_ACROBOT:
            gi[0] = rng.integers(440, 500, n)
        else:
            gr[0] = rng.uniform(0.3, 0.5, n); gr[1] = 0.06
        g.set_state(gr, gi); o.set_state(gr, gi)
    for t in range(steps):
        a = random_actions(rng, env_id, n)
        _, rew, done, _ = g.step_host(a)
        orew, odone, _ = o.step(a)
        assert np.array_equal(done, odone), t
        gr, gi = g.get_state(); orr, oi = o.get_state()
        assert np.array_equal(gi, oi), t
        err = np.abs(gr - orr) / np.maximum(np.abs(orr), 1e-3)
        assert err.max() <= 1e-7, (t, err.max())  # free-running: rounding differences compound
    s, so = g.stats(), o.stats()
    assert s.episodes == so.episodes and s.sum_length == so.sum_length
    if env_id != L.ENV_PENDULUM:
        assert so.episodes > 0
    assert abs(s.sum_return - so.sum_return) <= 1e-9 * max(1.0, abs(so.sum_return))


def test_fp64_mode_rollout_matches_single_steps(oracle):
    """K fused steps == K single steps, bit for bit, with the in-kernel policy"""
    n = 3000
    for env_id in CLASSIC:
        flags = L.AUTORESET | L.RECORD_STATS
        a, _ = make_pair(oracle, env_id, n, seed=31, flags=flags, mode=L.MODE_FP64)
        b, _ = make_pair(oracle, env_id, n, seed=31, flags=flags, mode=L.MODE_FP64)
        a.reset(); b.reset()
        a.rollout(24)
        for _ in range(24):
            b.step_random()
        ar, ai = a.get_state(); br, bi = b.get_state()
        assert np.array_equal(bits(ar), bits(br)) and np.array_equal(ai, bi), env_id
        sa, sb = a.stats(), b.stats()
        assert sa.episodes == sb.episodes and sa.sum_return == sb.sum_return


@pytest.mark.parametrize("stage_limit", ["0", None], ids=["copy-engine", "staged"])
@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_PENDULUM, L.ENV_FROZENLAKE])
def test_host_path_both_transports(oracle, env_id, stage_limit, monkeypatch):
    """gx_vec_step_async / gx_vec_wait through the pinned staging (small batches) and through the copy
    engine (what large batches use), every output requested, a ragged batch size"""
    if stage_limit is None:
        monkeypatch.delenv("GYMX_STAGE_LIMIT", raising=False)
    else:
        monkeypatch.setenv("GYMX_STAGE_LIMIT", stage_limit)
    n = 1003
    g, o = make_pair(oracle, env_id, n, seed=41, flags=L.AUTORESET | L.RECORD_STATS, time_limit=7)
    g.reset(); o.reset()
    rng = np.random.default_rng(env_id)
    for t in range(20):
        a = random_actions(rng, env_id, n)
        obs, rew, done, trunc = g.step_host(a, want_obs=True, want_trunc=True)
        orew, odone, otrunc = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew), t
        assert np.array_equal(done, odone) and np.array_equal(trunc, otrunc), t
        assert np.array_equal(bits(obs), bits(o.get_obs())), t
    # two steps without a wait in between: the first one's results still reach its buffers
    import ctypes as C
    lib = L.load()
    a1, a2 = random_actions(rng, env_id, n), random_actions(rng, env_id, n)
    r1 = np.empty(n, g.reward_dtype); r2 = np.empty(n, g.reward_dtype)
    d1 = np.empty(n, np.uint8); d2 = np.empty(n, np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    L.check(lib.gx_vec_step_async(g._h, p(a1), None, p(r1), p(d1), None))
    L.check(lib.gx_vec_step_async(g._h, p(a2), None, p(r2), p(d2), None))
    L.check(lib.gx_vec_wait(g._h))
    o1, od1, _ = o.step(a1)
    o2, od2, _ = o.step(a2)
    assert np.array_equal(r1.astype(np.float64), o1) and np.array_equal(d1, od1)
    assert np.array_equal(r2.astype(np.float64), o2) and np.array_equal(d2, od2)


def test_abi_rejects_bad_arguments_on_the_device():
    """every entry point that can be misused returns a status and a message; nothing aborts and the batch stays usable"""
    import ctypes as C
    from gpu_common import ctx
    from gym_amd.engine import DeviceBatch
    lib = L.load()
    c = ctx()

    def create(**kw):
        d = L.GxDesc(struct_size=C.sizeof(L.GxDesc), env_id=L.ENV_CARTPOLE, num_envs=8, seed=1, env_index_base=0, mode=L.MODE_EXACT,
                   flags=0, time_limit=0, reserved=0, map_rows=None)
        for k, v in kw.items():
            setattr(d, k, v)
        h = C.c_void_p()
        return lib.gx_vec_create(c._h, C.byref(d), C.byref(h)), h

    for bad in (dict(struct_size=8), dict(env_id=99), dict(env_id=-1), dict(num_envs=0), dict(mode=7), dict(time_limit=-1),
                dict(flags=1 << 20), dict(num_envs=(1 << 31) + 1)):
        rc, h = create(**bad)
        assert rc in (L.GX_ERR_INVALID, L.GX_ERR_UNSUPPORTED) and not h.value, bad
        assert lib.gx_last_error()
    rc, h = create(env_id=L.ENV_FROZENLAKE, map_rows=b"SF/FFG")
    assert rc == L.GX_ERR_INVALID and b"map" in lib.gx_last_error()

    g = DeviceBatch(c, L.ENV_CARTPOLE, 10, seed=2)  # no statistics, no time limit
    g.reset()
    buf = np.zeros(7, np.uint8)
    p = buf.ctypes.data_as(C.c_void_p)
    assert lib.gx_vec_get_state(g._h, p, 7) == L.GX_ERR_INVALID
    assert lib.gx_vec_set_state(g._h, p, 7) == L.GX_ERR_INVALID
    assert lib.gx_vec_get_obs(g._h, p, 7) == L.GX_ERR_INVALID
    st = L.GxStats()
    assert lib.gx_vec_stats(g._h, C.byref(st)) == L.GX_ERR_INVALID  # created without GX_RECORD_STATS
    a = np.zeros(10, np.int32)
    tr = np.zeros(10, np.uint8)
    assert lib.gx_vec_step_async(g._h, a.ctypes.data_as(C.c_void_p), None, None, None, tr.ctypes.data_as(C.c_void_p)) == L.GX_ERR_INVALID
    assert lib.gx_vec_step_async(g._h, None, None, None, None, None) == L.GX_ERR_INVALID
    assert lib.gx_vec_step_device(g._h, None, None) == L.GX_ERR_INVALID
    assert lib.gx_vec_step_device(g._h, C.c_void_p(g.view().state_int + 4), None) == L.GX_ERR_INVALID  # not 16-byte aligned
    assert lib.gx_vec_rollout(g._h, 0, L.POLICY_RANDOM, None, None) == L.GX_ERR_INVALID
    assert lib.gx_vec_rollout(g._h, 4, 9, None, None) == L.GX_ERR_INVALID
    assert lib.gx_vec_rollout(g._h, 4, L.POLICY_DEVICE_ACTIONS, None, None) == L.GX_ERR_INVALID
    assert lib.gx_vec_sample_actions(g._h, 0, None, None) == L.GX_ERR_INVALID
    # still usable afterwards
    _, rew, done, _ = g.step_host(np.ones(10, np.int32))
    assert rew.shape == (10,) and not done.any()


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_PENDULUM, L.ENV_ACROBOT, L.ENV_FROZENLAKE, L.ENV_TAXI, L.ENV_PONG, L.ENV_BREAKOUT])
def test_rollout_with_device_action_rows(oracle, env_id):
    """gx_vec_rollout(K, DEVICE_ACTIONS, d_actions[K][stride]): K fused steps with the actions read from a
    device array equal K oracle steps with the same actions (also: a ragged batch, the padded stride)"""
    import torch
    n, k = (259 if env_id in (L.ENV_PONG, L.ENV_BREAKOUT) else 4099), 7
    flags = L.AUTORESET | L.RECORD_STATS
    g, o = make_pair(oracle, env_id, n, seed=51, flags=flags)
    g.reset(); o.reset()
    stride = int(g.view().stride)
    assert stride >= n and stride % 4 == 0
    rng = np.random.default_rng(env_id + 5)
    acts = np.stack([random_actions(rng, env_id, n) for _ in range(k)])
    host = np.zeros((k, stride), acts.dtype)
    host[:, :n] = acts
    dev = torch.from_numpy(host).cuda()
    for rounds in range(2):  # two rollouts back to back: the sweep direction alternates
        g.rollout(k, L.POLICY_DEVICE_ACTIONS, dev.data_ptr())
        g.ctx.synchronize()
        for t in range(k):
            o.step(acts[t])
        gr, gi = g.get_state(); orr, oi = o.get_state()
        assert np.array_equal(gi, oi), rounds
        if gr.size:
            if env_id <= L.ENV_ACROBOT:
                assert (bits(gr) != bits(orr)).sum() <= 2 and np.allclose(gr, orr, rtol=1e-4, atol=1e-5), rounds
                g.set_state(orr, oi)
            else:
                assert np.array_equal(bits(gr), bits(orr)), rounds
    s, so = g.stats(), o.stats()
    assert s.episodes == so.episodes and s.sum_length == so.sum_length
    if env_id in (L.ENV_PONG, L.ENV_BREAKOUT):
        assert np.array_equal(g.get_obs(), o.get_obs())


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_MOUNTAINCAR_CONT, L.ENV_TAXI, L.ENV_SPACEINVADERS])
def test_sampled_actions_replay_the_in_kernel_policy(oracle, env_id):
    """gx_vec_sample_actions gives the draws the in-kernel policy makes at a tick: they equal the oracle's,
    and stepping with them through the zero-copy path, on a caller's stream, equals gx_vec_step_random"""
    import torch
    n = 263 if env_id == L.ENV_SPACEINVADERS else 5003
    a, o = make_pair(oracle, env_id, n, seed=61, flags=L.AUTORESET, base=11)
    b, _ = make_pair(oracle, env_id, n, seed=61, flags=L.AUTORESET, base=11)
    a.reset(); b.reset(); o.reset()
    stride = int(a.view().stride)
    adt = torch.float32 if a.info.action_dtype == L.F32 else torch.int32
    dev = torch.zeros(stride, dtype=adt, device="cuda")
    stream = torch.cuda.Stream()
    for t in range(6):
        tick = a.tick
        a.sample_actions(tick, dev.data_ptr(), stream.cuda_stream)
        a.step_device(dev.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        b.step_random()
        b.ctx.synchronize()
        oa, _, _, _ = o.step_random()
        assert np.array_equal(bits(dev[:n].cpu().numpy()), bits(oa)), t
        ar, ai = a.get_state(); br, bi = b.get_state()
        assert np.array_equal(ai, bi) and np.array_equal(bits(ar), bits(br)), t


_FRAME_DIGEST = r"""
import hashlib, sys
sys.path.insert(0, %r)
from gym_amd import _lib as L
from gym_amd.engine import Context, DeviceBatch
ctx = Context(0)
for env_id in (L.ENV_PONG, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS):
    v = DeviceBatch(ctx, env_id, 101, seed=12, flags=L.AUTORESET)
    v.reset()
    h = hashlib.sha256(v.get_obs().tobytes())
    for _ in range(40):
        v.step_random()
    v.wait()
    h.update(v.get_obs().tobytes())
    print(h.hexdigest())
    v.close()
ctx.close()
"""


def test_frames_of_a_batch_rastered_in_several_launches():
    '''a batch beyond what one raster grid holds (524 280 frames) is rastered slice by slice; with the slice set to
    24 frames a batch of 101 takes five launches, the last one short, and must give the frames one launch gives'''
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def digests(slice_frames):
        env = dict(os.environ)
        env.pop("GYMX_RASTER_SLICE", None)
        if slice_frames:
            env["GYMX_RASTER_SLICE"] = str(slice_frames)
        r = subprocess.run([sys.executable, "-c", _FRAME_DIGEST % root], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.split()
    whole, sliced = digests(0), digests(24)
    assert len(whole) == 3 and whole == sliced


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_PENDULUM])
def test_rows_kept_off_power_of_two_spacing(oracle, env_id):
    """a classic-control batch whose rows would lie a multiple of 64 KiB apart gets a longer stride (and its row
    arrays staggered starts): every path that walks the rows -- steps, host copies, state images, views -- must
    follow the stride the view reports"""
    n = 1 << 16
    flags = L.AUTORESET | L.RECORD_STATS
    g, o = make_pair(oracle, env_id, n, seed=31, flags=flags)
    assert int(g.view().stride) == n + 256
    g.reset(); o.reset()
    rng = np.random.default_rng(9)
    for t in range(12):
        if t % 2:
            g.step_random(); g.ctx.synchronize()
            o.step_random()
        else:
            a = random_actions(rng, env_id, n)
            obs, rew, done, _ = g.step_host(a, want_obs=True)
            orew, odone, _ = o.step(a)
            assert np.array_equal(done, odone) and np.array_equal(bits(obs), bits(o.get_obs()))
            assert np.array_equal(rew.astype(np.float64), orew)
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(bits(gr), bits(orr)) and np.array_equal(gi, oi)
    # a state image written back lands in the same rows
    g.set_state(gr, gi)
    g2r, g2i = g.get_state()
    assert np.array_equal(bits(g2r), bits(gr)) and np.array_equal(g2i, gi)
    s, so = g.stats(), o.stats()
    assert s.episodes == so.episodes and s.sum_length == so.sum_length
