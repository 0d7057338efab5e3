"""The C oracle and the independent numpy restatement must agree bit-for-bit.

This is the compensation for the reference holding no numeric vectors on the
step path: two restatements written separately from the same source lines.
"""
import numpy as np
import pytest

from oracle import np_oracle as NP

N = 1500
STEPS = 40
SEED = 0xC0FFEE


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


CLASSIC = [
    ("CARTPOLE", NP.cartpole_reset, NP.cartpole_step, 2),
    ("MOUNTAINCAR", NP.mountaincar_reset, NP.mountaincar_step, 3),
    ("MOUNTAINCAR_CONT", NP.mountaincar_reset, NP.mountaincar_cont_step, None),
    ("PENDULUM", NP.pendulum_reset, NP.pendulum_step, None),
    ("ACROBOT", NP.acrobot_reset, NP.acrobot_step, 3),
]


@pytest.mark.parametrize("name,resetf,stepf,nact", CLASSIC, ids=[c[0] for c in CLASSIC])
def test_classic_control_bitwise(oracle, name, resetf, stepf, nact):
    eid = getattr(oracle, name)
    v = oracle.OracleVec(eid, N, seed=SEED, env_index_base=10)
    v.reset()
    reals, ints = v.get_state()
    assert np.array_equal(bits(reals), bits(resetf(SEED, np.arange(N, dtype=np.uint64) + 10, 0)))
    rng = np.random.default_rng(1)
    s, st = reals.copy(), ints[0].copy()
    for t in range(STEPS):
        if nact is None:
            a = (rng.random(N) * 6 - 3).astype(np.float32)  # exercises the clamp
        else:
            a = rng.integers(0, nact, N).astype(np.int32)
        rew, done, _ = v.step(a)
        s, st, rew2, done2 = stepf(s, st, a)
        reals, ints = v.get_state()
        assert np.array_equal(bits(reals), bits(s)), (name, t)
        assert np.array_equal(ints[0], st)
        assert np.array_equal(bits(rew), bits(np.asarray(rew2, np.float64))), (name, t)
        assert np.array_equal(done.astype(bool), done2)


def test_classic_control_wide_states(oracle):
    """Random states far outside the reset range: large angles, saturated velocities."""
    rng = np.random.default_rng(2)
    for name, _, stepf, nact in CLASSIC:
        eid = getattr(oracle, name)
        v = oracle.OracleVec(eid, N)
        v.reset()
        dim = v.n_real
        s = (rng.standard_normal((dim, N)) * np.array([3, 3, 50, 20, 20, 20][:dim])[:, None]).astype(np.float32)
        st = rng.integers(0, 600, N).astype(np.int32)
        v.set_state(s, st[None])
        a = (rng.random(N) * 6 - 3).astype(np.float32) if nact is None else rng.integers(0, nact, N).astype(np.int32)
        rew, done, _ = v.step(a)
        s2, st2, rew2, done2 = stepf(s, st, a)
        reals, ints = v.get_state()
        assert np.array_equal(bits(reals), bits(s2)), name
        assert np.array_equal(bits(rew), bits(np.asarray(rew2, np.float64))), name
        assert np.array_equal(done.astype(bool), done2), name


CLASSIC64 = [
    ("CARTPOLE", NP.cartpole64_reset, NP.cartpole64_step, 2),
    ("MOUNTAINCAR", NP.mountaincar64_reset, NP.mountaincar64_step, 3),
    ("MOUNTAINCAR_CONT", NP.mountaincar64_reset, NP.mountaincar_cont64_step, None),
    ("PENDULUM", NP.pendulum64_reset, NP.pendulum64_step, None),
    ("ACROBOT", NP.acrobot64_euler_reset, NP.acrobot64_euler_step, 3),
]


@pytest.mark.parametrize("name,resetf,stepf,nact", CLASSIC64, ids=[c[0] for c in CLASSIC64])
def test_classic_control_binary64_bitwise(oracle, name, resetf, stepf, nact):
    """the FP64 extension: both restatements run the same libm, so they agree to the bit"""
    eid = getattr(oracle, name)
    v = oracle.OracleVec(eid, N, seed=SEED, flags=oracle.FP64, env_index_base=10)
    assert v.real_is_f64 and v.obs_is_f64
    v.reset()
    reals, ints = v.get_state()
    assert reals.dtype == np.float64
    assert np.array_equal(bits(reals), bits(resetf(SEED, np.arange(N, dtype=np.uint64) + 10, 0)))
    assert np.array_equal(bits(v.get_obs()), bits(reals))
    rng = np.random.default_rng(1)
    s, st = reals.copy(), ints[0].copy()
    for t in range(STEPS):
        a = (rng.random(N) * 6 - 3).astype(np.float32) if nact is None else rng.integers(0, nact, N).astype(np.int32)
        rew, done, _ = v.step(a)
        s, st, rew2, done2 = stepf(s, st, a)
        reals, ints = v.get_state()
        assert np.array_equal(bits(reals), bits(s)), (name, t)
        assert np.array_equal(ints[0], st)
        assert np.array_equal(bits(rew), bits(np.asarray(rew2, np.float64))), (name, t)
        assert np.array_equal(done.astype(bool), done2)


def test_classic_control_binary64_tracks_binary32(oracle):
    """one step from the same state: the binary64 form stays within binary32 rounding of the reference form"""
    rng = np.random.default_rng(5)
    for name, _, _, nact in CLASSIC64:
        eid = getattr(oracle, name)
        v32 = oracle.OracleVec(eid, N, seed=3)
        v64 = oracle.OracleVec(eid, N, seed=3, flags=oracle.FP64)
        v32.reset(); v64.reset()
        for t in range(10):
            r32, i32 = v32.get_state()
            v64.set_state(r32.astype(np.float64), i32)
            a = (rng.random(N) * 4 - 2).astype(np.float32) if nact is None else rng.integers(0, nact, N).astype(np.int32)
            v32.step(a); v64.step(a)
            r32, _ = v32.get_state(); r64, _ = v64.get_state()
            err = np.abs(r32.astype(np.float64) - r64) / np.maximum(np.abs(r64), 1e-1)
            assert err.max() <= 1e-5, (name, t, err.max())


def test_grid_worlds_bitwise(oracle):
    rng = np.random.default_rng(3)
    maps = [None, ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"], ["SFG", "HFF"]]
    for rows in maps:
        v = oracle.OracleVec(oracle.FROZENLAKE, N, map_rows=rows)
        v.reset()
        st = np.zeros(N, np.int32); steps = np.zeros(N, np.int32)
        kw = {} if rows is None else {"rows": rows}
        for t in range(60):
            a = rng.integers(-1, 6, N).astype(np.int32)
            rew, done, _ = v.step(a)
            st, steps, rew2, done2 = NP.frozenlake_step(st, steps, a, **kw)
            assert np.array_equal(v.get_obs(), st) and np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2)
    v = oracle.OracleVec(oracle.CLIFFWALKING, N)
    v.reset()
    st = np.full(N, 36, np.int32); steps = np.zeros(N, np.int32)
    for t in range(120):
        a = rng.integers(0, 4, N).astype(np.int32)
        rew, done, _ = v.step(a)
        st, steps, rew2, done2 = NP.cliffwalking_step(st, steps, a)
        assert np.array_equal(v.get_obs(), st) and np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2)


def test_frozenlake_slippery_extension_bitwise(oracle):
    """is_slippery (an extension; the reference is deterministic): both restatements take the outcome from
    the env's word of the tick, the digit after the action digit"""
    rows8 = ["SFFFFFFF", "FFFFFFFF", "FFFHFFFF", "FFFFFHFF", "FFFHFFFF", "FHHFFFHF", "FHFFHFHF", "FFFHFFFG"]
    n = 4000
    rng = np.random.default_rng(8)
    v = oracle.OracleVec(oracle.FROZENLAKE, n, seed=SEED, flags=oracle.SLIPPERY, map_rows=rows8)
    v.reset()
    env = np.arange(n, dtype=np.uint64)
    st = np.zeros(n, np.int32); steps = np.zeros(n, np.int32)
    counts = np.zeros((4, 3), np.int64)
    for t in range(80):
        a = rng.integers(-1, 6, n).astype(np.int32)
        rew, done, _ = v.step(a)
        st, steps, rew2, done2 = NP.frozenlake_step(st, steps, a, rows=rows8, slippery=(SEED, env, t + 1))
        assert np.array_equal(v.get_obs(), st) and np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2)
        # the outcome is independent of the action the random policy would draw from the same word
        k = NP.slip_outcome(SEED, env, t + 1)
        pa = NP.rng_below(SEED, env, t + 1, 3, 0, 4)
        np.add.at(counts, (pa, k), 1)
    assert counts.min() > 0.9 * counts.sum() / 12 and counts.max() < 1.1 * counts.sum() / 12


def test_blackjack_reference_stub_bitwise(oracle):
    """Envs.Blackjack.cs:21-40: three reset draws, a state that never changes, reward 1 at step ten"""
    n = 500
    v = oracle.OracleVec(oracle.BLACKJACK, n, seed=SEED)
    v.reset()
    env = np.arange(n, dtype=np.uint64)
    ps, dc, ua = NP.blackjack_reset(SEED, env, 0)
    assert ps.min() >= 12 and ps.max() <= 21 and dc.min() >= 1 and dc.max() <= 10 and set(ua) == {0, 1}
    obs0 = ((ps * 11 + dc) * 2 + ua).astype(np.int32)
    assert np.array_equal(v.get_obs(), obs0) and obs0.max() < 32 * 11 * 2
    steps = np.zeros(n, np.int32)
    rng = np.random.default_rng(4)
    for t in range(12):
        rew, done, _ = v.step(rng.integers(-1, 3, n).astype(np.int32))
        steps, rew2, done2 = NP.blackjack_step(steps)
        assert np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2)
        assert np.array_equal(v.get_obs(), obs0)
        assert done.all() == (t >= 9) and rew.sum() == (n if t >= 9 else 0)
    _, ints = v.get_state()
    assert np.array_equal(ints[0], obs0) and not ints[1].any() and (ints[2] == 12).all()


def test_blackjack_full_rules_extension_bitwise(oracle):
    """upstream Blackjack-v1 rules (an extension): the C restatement keeps (sum, usable ace) and the
    face-down cards' sum, the numpy one keeps whole hands as lists; states, rewards and flags agree"""
    n = 3000
    v = oracle.OracleVec(oracle.BLACKJACK, n, seed=SEED, flags=oracle.BLACKJACK_FULL)
    env = np.arange(n, dtype=np.uint64)
    rng = np.random.default_rng(12)
    seen = set()
    for episode in range(6):
        v.reset()
        tick = v.tick - 1
        hands = NP.blackjack_full_deal(SEED, env, tick)
        steps = np.zeros(n, np.int32)
        _, ints = v.get_state()
        assert np.array_equal(ints, NP.blackjack_image(hands, steps))
        for t in range(5):  # keeps stepping finished games, as a caller of the reference's VectorEnv may
            a = rng.integers(0, 2, n).astype(np.int32) if episode else np.full(n, t < 1, np.int32)
            tick = v.tick
            rew, done, _ = v.step(a)
            rew2, done2 = NP.blackjack_full_step(hands, a, SEED, env, tick)
            steps += 1
            assert np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2), (episode, t)
            _, ints = v.get_state()
            assert np.array_equal(ints, NP.blackjack_image(hands, steps)), (episode, t)
            seen.update(zip(a[done2 > 0].tolist(), rew2[done2 > 0].tolist()))
    # busts, and sticks that lose, draw and win, all occurred
    assert seen >= {(1, -1.0), (0, -1.0), (0, 0.0), (0, 1.0)}


def test_blackjack_full_rules_known_answers(oracle):
    """hand-derived from upstream's rules: a player on 21 who sticks never loses; a dealer showing an
    ace over a face-down ten stands on 21 without drawing; hitting on a hard 21 always busts"""
    n = 256
    v = oracle.OracleVec(oracle.BLACKJACK, n, seed=3, flags=oracle.BLACKJACK_FULL)
    v.reset()
    ints = np.zeros((3, n), np.int32)
    ints[0] = (21 * 11 + 1) * 2 + 1   # player ace + ten, dealer shows an ace
    ints[1] = 10                      # ... over a ten: dealer 21
    v.set_state(np.zeros((0, n), np.float32), ints)
    rew, done, _ = v.step(np.zeros(n, np.int32))
    assert done.all() and (rew == 0.0).all()
    _, after = v.get_state()
    assert np.array_equal(after[1], ints[1])  # the dealer drew nothing
    ints[0] = (21 * 11 + 5) * 2 + 0   # hard 21
    ints[1] = 5
    v.set_state(np.zeros((0, n), np.float32), ints)
    rew, done, _ = v.step(np.ones(n, np.int32))
    assert done.all() and (rew == -1.0).all()
    ints[0] = (21 * 11 + 6) * 2 + 0
    ints[1] = 10                      # dealer 16: must draw; the player's 21 cannot lose
    v.set_state(np.zeros((0, n), np.float32), ints)
    rew, done, _ = v.step(np.zeros(n, np.int32))
    _, after = v.get_state()
    assert done.all() and (rew >= 0.0).all() and (rew == 1.0).any() and ((after[1] & 0xFF) > 10).all()
    # a soft 17 (ace + 6) that hits cannot bust, and a drawn ten turns it into a hard 17
    ints[0] = (17 * 11 + 2) * 2 + 1
    v.set_state(np.zeros((0, n), np.float32), ints)
    rew, done, _ = v.step(np.ones(n, np.int32))
    obs = v.get_obs()
    assert not done.any() and (rew == 0.0).all()
    total, usable = obs // 22, obs & 1
    assert ((total == 17) & (usable == 0)).any() and ((total >= 18) & (total <= 21) & (usable == 1)).any()
    assert ((usable == 1) | (total <= 17)).all() and (total >= 12).all()


def test_pong_bitwise(oracle):
    n = 300
    v = oracle.OracleVec(oracle.PONG, n, seed=SEED)
    v.reset()
    env = np.arange(n, dtype=np.uint64)
    bx, by, vx, vy = NP.pong_reset_ball(SEED, env, 0, 0)
    st = np.stack([bx, by, vx, vy, np.full(n, 97.5), np.full(n, 97.5)])
    ints = np.zeros((3, n), np.int32)
    reals, i0 = v.get_state()
    assert np.array_equal(bits(reals), bits(st)) and np.array_equal(i0, ints)
    rng = np.random.default_rng(4)
    for t in range(400):
        a = rng.integers(0, 4, n).astype(np.int32)
        rew, done, _ = v.step(a)
        st, ints, rew2, done2 = NP.pong_step(st, ints, a, SEED, env, t + 1)
        reals, i0 = v.get_state()
        assert np.array_equal(bits(reals), bits(st)), t
        assert np.array_equal(i0, ints) and np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2)
    assert ints[:2].sum() > 0  # some points were scored, so ResetBall was exercised
    frames = v.get_obs()
    for i in range(0, n, 37):
        exp = NP.pong_frame(st[0, i], st[1, i], st[4, i], st[5, i], ints[0, i], ints[1, i])
        assert np.array_equal(frames[i], exp)


def test_pong_frames_edge_cases(oracle):
    cases = [
        (160.0, 208.0, 0.0, 195.0, 21, 9), (0.0, 0.0, 195.0, 0.0, 10, 123), (159.2, 104.9, 3.2, 190.8, 0, 21),
        (79.5, 19.5, 97.5, 97.5, 7, 8), (38.0, 21.0, 50.0, 60.0, 4, 1234567),
    ]
    for c in cases:
        assert np.array_equal(oracle.pong_render(*c), NP.pong_frame(*c)), c


def _breakout_policy(rng, st, ints):
    """half the envs act at random, the others fire and then follow the ball (long rallies)"""
    n = st.shape[1]
    a = rng.integers(0, 4, n).astype(np.int32)
    track = np.where(ints[2] != 0, 1, np.where(st[0] + 1 > st[4] + 10, 2, np.where(st[0] + 1 < st[4] + 6, 3, 0)))
    a[n // 2:] = track[n // 2:]
    return a


def test_breakout_bitwise(oracle):
    n = 256
    v = oracle.OracleVec(oracle.BREAKOUT, n, seed=SEED)
    v.reset()
    env = np.arange(n, dtype=np.uint64)
    st = np.stack([np.full(n, 80.0), np.full(n, 180.0), np.zeros(n), np.zeros(n), np.full(n, 72.0)])
    ints = np.zeros((8, n), np.int32)
    ints[1] = 5; ints[2] = 1
    ints[3:7] = NP.breakout_pack(np.ones((n, 6, 18), bool))
    reals, i0 = v.get_state()
    assert np.array_equal(bits(reals), bits(st)) and np.array_equal(i0, ints)  # Envs.Breakout.cs:62-87
    rng = np.random.default_rng(4)
    total = np.zeros(n)
    for t in range(1500):
        a = _breakout_policy(rng, st, ints)
        rew, done, _ = v.step(a)
        st, ints, rew2, done2 = NP.breakout_step(st, ints, a, SEED, env, t + 1)
        reals, i0 = v.get_state()
        assert np.array_equal(bits(reals), bits(st)), t
        assert np.array_equal(i0, ints) and np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2), t
        total += rew
        if t % 250 == 249:
            frames = v.get_obs()
            for i in range(0, n, 31):
                exp = NP.breakout_frame(st[0, i], st[1, i], st[4, i], ints[2, i], ints[0, i], ints[1, i], ints[3:7, i])
                assert np.array_equal(frames[i], exp), (t, i)
    assert np.array_equal(total, ints[0])  # info["score"] accumulates the rewards (Envs.Breakout.cs:162-163)
    assert (ints[1] <= 0).any() and (ints[1] > 0).any() and ints[0].max() > 100


def test_breakout_win_and_edge_frames(oracle):
    """last brick: reward 100 + row value and done (Envs.Breakout.cs:182-199); frames at edge states"""
    v = oracle.OracleVec(oracle.BREAKOUT, 2, seed=1)
    v.reset()
    reals, ints = v.get_state()
    reals = reals.copy(); ints = ints.copy()
    one = np.zeros((2, 6, 18), bool)
    one[0, 2, 5] = True            # env 0: the ball is about to enter this brick
    one[1, 2, 5] = True; one[1, 0, 0] = True
    ints[3:7] = NP.breakout_pack(one)
    ints[2] = 0
    reals[0] = 5 * 8 + 3.0; reals[1] = 60 + 3 * 6 + 1.0; reals[2] = 0.5; reals[3] = -3.0
    v.set_state(reals, ints)
    rew, done, _ = v.step(np.zeros(2, np.int32))
    assert rew.tolist() == [105.0, 5.0] and done.tolist() == [1, 0]
    st2, i2, rew2, done2 = NP.breakout_step(reals, ints, np.zeros(2, np.int32), 1, np.arange(2, dtype=np.uint64), 1)
    r1, i1 = v.get_state()
    assert np.array_equal(bits(r1), bits(st2)) and np.array_equal(i1, i2) and np.array_equal(rew, rew2)
    full = [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFF]
    cases = [
        (0.0, 0.0, 0.0, 0, 0, 5, full), (158.9, 209.5, 144.0, 0, 486, 1, [0, 0, 0, 0]), (80.0, 180.0, 72.0, 1, 7, 5, full),
        (12.3, 11.7, 143.9, 0, 1234567890, 9, [0x12345678, 0x9ABCDEF0, 0x0F0F0F0F, 0xA5A]), (10.0, 25.0, 3.5, 1, -42, 0, full),
        (75.5, 250.0, 20.0, 0, 99, -1, [1, 0, 0, 0x800]), (3.0, 62.0, 0.0, 0, 10, 30, full),
    ]
    for c in cases:
        assert np.array_equal(oracle.breakout_render(*c), NP.breakout_frame(*c)), c


def _invaders_action(rng, game, i):
    """odd envs act at random; even envs hunt: move under the nearest occupied column and fire"""
    if i & 1:
        return int(rng.integers(0, 6))
    cols = [c for c in range(11) if any(game.aliens[r][c] for r in range(5))]
    if not cols:
        return 1
    gun = game.player_x + 4.0
    target = min((game.form_x + c * 12 + 3.0 for c in cols), key=lambda x: abs(x - gun))
    return 4 if target > gun + 2 else (5 if target < gun - 2 else 1)


def test_invaders_bitwise(oracle):
    n = 12
    v = oracle.OracleVec(oracle.SPACEINVADERS, n, seed=SEED)
    v.reset()
    games = [NP.InvadersGame() for _ in range(n)]
    rng = np.random.default_rng(7)
    shots = waves = 0
    for t in range(2600):
        a = np.array([_invaders_action(rng, g, i) for i, g in enumerate(games)], np.int32)
        rew, done, _ = v.step(a)
        reals, ints = v.get_state()
        for i, g in enumerate(games):
            r2, d2 = g.step(int(a[i]), SEED, i, t + 1)
            er, ei = g.image()
            assert np.array_equal(bits(reals[:, i]), bits(er)) and np.array_equal(ints[:, i], ei), (t, i)
            assert rew[i] == r2 and bool(done[i]) == d2, (t, i)
            shots += len(g.alien_bullets) > 0
            waves += r2 >= 100
        if t % 400 == 399 or t == 30:
            frames = v.get_obs()
            for i in range(0, n, 5):
                assert np.array_equal(frames[i], games[i].frame_bytes()), (t, i)
    assert shots > 0 and waves > 0            # alien fire and a cleared wave were both exercised
    assert any(g.lives < 3 for g in games)    # and a cannon was hit


def test_invaders_frames_edge_cases(oracle):
    """sprites clipped at the borders, bullets over sprites, many-digit and negative scores, no lives"""
    g = NP.InvadersGame()
    cases = [
        ([0.0, 10.0, 40.0, 3.9, 0.0, 157.2, 208.9, 80.0, 100.0, 12.0, 42.0, 159.0, 170.5],
         [123456, 3, 7, 1, 3, 2, -1, 0x7FFFFF, 7]),
        ([152.0, 27.5, 120.0, 30.0, 121.0, 0, 0, 0, 0, 156.0, 172.0, 0, 0], [-50, 1, 9, 0, 1, 1, 0x55555555, 0x2AAAAA, 9]),
        ([75.4, -3.0, 161.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [990, 0, 0, 1, 0, 0, -1, 0x7FFFFF, 0]),
        ([80.0, 150.0, 5.0, 11.0, 11.0, 121.0, 12.0, 100.0, 190.0, 50.0, 189.5, 20.0, 12.0], [2147483647, 13, 5, 1, 3, 2, 1 << 10, 1 << 22, 1]),
    ]
    for reals, ints in cases:
        ints = np.array(ints, np.int64).astype(np.uint32).view(np.int32)
        g.load(np.array(reals), ints)
        assert np.array_equal(oracle.invaders_render(reals, ints), g.frame_bytes())


def test_wrappers_match_numpy_restatement(oracle):
    """RecordEpisodeStatistics(TimeLimit(CartPole)) without autoreset, stepping past done."""
    n = 64
    v = oracle.OracleVec(oracle.CARTPOLE, n, seed=9, flags=oracle.RECORD_STATS, time_limit=25)
    v.reset()
    res = [NP.RecordEpisodeStatistics() for _ in range(n)]
    tls = [NP.TimeLimit(25) for _ in range(n)]
    s = NP.cartpole_reset(9, np.arange(n, dtype=np.uint64), 0); st = np.zeros(n, np.int32)
    rng = np.random.default_rng(5)
    for t in range(60):
        if t == 40:
            v.reset()
            for r_, tl in zip(res, tls):
                r_.reset(); tl.reset()
            s = NP.cartpole_reset(9, np.arange(n, dtype=np.uint64), v.tick - 1); st = np.zeros(n, np.int32)
        a = rng.integers(0, 2, n).astype(np.int32)
        rew, done, trunc = v.step(a)
        s, st, rew2, done2 = NP.cartpole_step(s, st, a)
        for i in range(n):
            d, tr = tls[i].step(done2[i])
            res[i].step(float(rew2[i]), d)
            assert d == bool(done[i]) and tr == bool(trunc[i])
    for i in range(n):
        assert v.episodes(i) == list(zip(res[i].episode_rewards, res[i].episode_lengths))


def test_table_env_bitwise_and_equal_to_the_envs_it_generalises(oracle):
    """the oracle's caller-supplied-table env: the C and numpy restatements agree, and FrozenLake / CliffWalking /
    the Taxi stub written out as P[s][a] step exactly like the oracle's envs of those ids"""
    import tables as T

    rng = np.random.default_rng(21)
    n = 2000
    env = np.arange(n, dtype=np.uint64)
    for n_out in (1, 3):
        table = T.random_table(rng, 300, 5, n_out)
        v = oracle.OracleVec(oracle.TABLE, n, seed=SEED, table=table, n_outcomes=n_out, start_state=-1, step_limit=40)
        v.reset()
        st = NP.rng_below(SEED, env, 0, 0, 0, 300).astype(np.int32)
        assert np.array_equal(v.get_obs(), st)
        steps = np.zeros(n, np.int32)
        for t in range(50):
            a = rng.integers(0, 5, n).astype(np.int32)
            rew, done, _ = v.step(a)
            st, steps, rew2, done2 = NP.table_step(st, steps, a, table, n_out, 40, rng=(SEED, env, t + 1))
            assert np.array_equal(v.get_obs(), st) and np.array_equal(rew, rew2) and np.array_equal(done.astype(bool), done2)
    cases = [(oracle.FROZENLAKE, 0, None, T.frozenlake(T.MAP4), dict(start_state=0, step_limit=100)),
             (oracle.FROZENLAKE, oracle.SLIPPERY, T.MAP8, T.frozenlake(T.MAP8, True), dict(start_state=0, step_limit=100, n_outcomes=3)),
             (oracle.CLIFFWALKING, 0, None, T.cliffwalking(), dict(start_state=36, step_limit=100)),
             (oracle.TAXI, 0, None, T.taxi_stub(), dict(start_state=-1, step_limit=200))]
    flags = oracle.AUTORESET | oracle.RECORD_STATS
    for env_id, xf, rows, table, kw in cases:
        a_n = table.shape[1]
        ref = oracle.OracleVec(env_id, n, seed=SEED, flags=flags | xf, map_rows=rows)
        tab = oracle.OracleVec(oracle.TABLE, n, seed=SEED, flags=flags, table=table, **kw)
        ref.reset(); tab.reset()
        for t in range(220):
            a = rng.integers(0, a_n, n).astype(np.int32)
            r1, d1, _ = ref.step(a)
            r2, d2, _ = tab.step(a)
            assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(ref.get_obs(), tab.get_obs())
        s1, s2 = ref.stats(), tab.stats()
        assert (s1.episodes, s1.sum_return, s1.sum_length) == (s2.episodes, s2.sum_return, s2.sum_length) and s1.episodes > 0
    # out-of-range action and arguments
    with pytest.raises(IndexError):
        tab.step(np.full(n, 6, np.int32))
    with pytest.raises(ValueError):
        oracle.OracleVec(oracle.TABLE, 4, table=T.taxi_stub(), start_state=500)
