"""Host-side mirror of the reference API: what runs without a GPU (spaces, registry, wrapper
composition, construction) and, marked gpu, the reference's own shape tests through the
device-backed classes (Gymnasium.Tests/*.cs)."""
import numpy as np
import pytest

import gym_amd as G


# ------------------------------------------------------------------ no GPU needed
def test_box_and_discrete_members():
    """Spaces.Box.cs / Spaces.Discrete.cs: the members agents read through duck typing"""
    b = G.Box([-1.0, -2.0], [1.0, 2.0])
    assert b.Dimension == 2 and b.Shape == [2] and b.Low.dtype == np.float32
    for _ in range(50):
        x = b.Sample()
        assert x.dtype == np.float32 and b.Contains(x)
    assert b.Contains([1.0, -2.0]) and not b.Contains([1.01, 0.0]) and not b.Contains([0.0])
    with pytest.raises(ValueError):
        G.Box([0.0], [1.0, 2.0])  # Spaces.Box.cs:18-19
    d = G.Discrete(3)
    assert d.N == 3 and d.Shape == [1]
    assert all(d.Contains(d.Sample()) for _ in range(50))
    assert not d.Contains(3) and not d.Contains(-1)


def test_composite_spaces():
    """Spaces.MultiBinary.cs, Spaces.MultiDiscrete.cs, Spaces.Tuple.cs, Spaces.Dict.cs"""
    mb = G.MultiBinary(5)
    assert mb.Shape == [5] and len(mb.Sample()) == 5 and mb.Contains([True] * 5) and not mb.Contains([True] * 4)
    md = G.MultiDiscrete([3, 5, 2])
    assert md.Shape == [3] and all(md.Contains(md.Sample()) for _ in range(50))
    assert not md.Contains([3, 0, 0]) and not md.Contains([0, 0]) and not md.Contains([0, -1, 0])
    tp = G.Tuple([G.Discrete(3), G.Box([0.0], [1.0])])
    assert tp.Shape == [2] and all(tp.Contains(tp.Sample()) for _ in range(20))
    assert not tp.Contains([3, [0.5]]) and not tp.Contains([1])
    dc = G.Dict({"a": G.Discrete(2), "b": md})
    assert dc.Shape == [2] and all(dc.Contains(dc.Sample()) for _ in range(20))
    assert not dc.Contains({"a": 1}) and not dc.Contains({"a": 2, "b": [0, 0, 0]})


def test_registry():
    """EnvRegistry.cs:13-34, GymnasiumRegistration.cs:10-28"""
    G.RegisterAll()
    ids = G.EnvRegistry.List()
    for i in ("CartPole-v1", "MountainCar-v0", "MountainCarContinuous-v0", "Acrobot-v1", "Pendulum-v1",
              "FrozenLake-v1", "Taxi-v3", "CliffWalking-v0", "Blackjack-v1", "Pong-v4"):
        assert i in ids
    env = G.EnvRegistry.Make("CartPole-v1")
    assert isinstance(env, G.CartPole)
    assert isinstance(G.EnvRegistry.Make("Pong-v4", expected_type=G.Env), G.Pong)
    with pytest.raises(ValueError, match="No environment registered with id 'Nope-v0'"):
        G.EnvRegistry.Make("Nope-v0")
    with pytest.raises(TypeError):
        G.EnvRegistry.Make("CartPole-v1", expected_type=G.Pong)
    G.EnvRegistry.Register("Mine-v0", lambda: G.DummyEnv())
    assert isinstance(G.EnvRegistry.Make("Mine-v0"), G.DummyEnv)


def test_space_declarations():
    """what agents read: Envs.*.cs ActionSpace / ObservationSpace"""
    assert G.CartPole().ActionSpace.N == 2 and G.CartPole().ObservationSpace.Dimension == 4
    assert G.MountainCar().ActionSpace.N == 3
    assert G.MountainCarContinuous().ActionSpace.Dimension == 1 and G.MountainCarContinuous().ActionSpace.High[0] == 1.0
    assert G.Pendulum().ActionSpace.High[0] == 2.0 and G.Pendulum().ObservationSpace.High[2] == 8.0
    assert G.Acrobot().ObservationSpace.High[4] == 6.0 and G.Acrobot().ActionSpace.N == 3
    assert G.FrozenLake().ObservationSpace.N == 16 and G.FrozenLake(["SFG", "HFF"]).ObservationSpace.N == 6
    assert G.Taxi().ActionSpace.N == 6 and G.Taxi().ObservationSpace.N == 500
    assert G.CliffWalking().ObservationSpace.N == 48
    assert G.Blackjack().ActionSpace.N == 2 and G.Blackjack().ObservationSpace.N == 32 * 11 * 2  # Envs.Blackjack.cs:12-13
    p = G.Pong()
    assert p.ActionSpace.N == 4 and p.ObservationSpace.Dimension == 210 * 160 * 3 and p.RewardRange == (-1, 1)


class _CountEnv(G.Env):
    """host-only env for wrapper-composition tests (DummyEnv.cs semantics)"""

    def __init__(self):
        self.s = 0

    def Reset(self):
        self.s = 0
        return self.s

    def Step(self, action):
        self.s += action
        return self.s, float(self.s), self.s >= 10, {}


def test_wrapper_semantics_host():
    """Wrappers.TimeLimit.cs:23-33 and EnvWrapper.cs:32-57 on a host env"""
    tl = G.TimeLimit(_CountEnv(), 3)
    tl.Reset()
    outs = [tl.Step(0) for _ in range(3)]
    assert [o[2] for o in outs] == [False, False, True] and outs[2][3] == {"TimeLimit.truncated": True}
    res = G.RecordEpisodeStatistics(_CountEnv())
    res.Reset()
    res.Step(5); res.Step(5)
    assert res.EpisodeRewards == [15.0] and res.EpisodeLengths == [2]
    res.Step(1)
    res.Reset()  # an open episode is recorded by Reset
    assert res.EpisodeRewards == [15.0, 11.0] and res.EpisodeLengths == [2, 1]


def test_vector_env_rejects_mixed_lists():
    with pytest.raises(ValueError):
        G.VectorEnv([G.CartPole(), G.MountainCar()])
    with pytest.raises(ValueError):
        G.VectorEnv([])
    with pytest.raises(NotImplementedError):
        G.VectorEnv([G.TimeLimit(G.RecordEpisodeStatistics(G.CartPole()), 5)])  # order the kernel does not fuse


# ------------------------------------------------------------------ reference shape tests on the device
@pytest.mark.gpu
def test_reference_shape_tests_on_device():
    env = G.CartPole()  # CartPoleTests.cs:8-19
    s = env.Reset()
    assert isinstance(s, tuple) and len(s) == 4
    for _ in range(10):
        s, r, d, info = env.Step(1)
        assert 0 <= r <= 1 and isinstance(info, dict)
    env = G.Acrobot()  # AcrobotTests.cs:7-18
    assert len(env.Reset()) == 6
    assert env.Step(1)[1] <= 0
    with pytest.raises(IndexError):
        env.Step(3)
    env = G.MountainCar()  # MountainCarTests.cs:7-19
    assert env.ObservationSpace.Contains(list(env.Reset()))
    assert env.Step(2)[1] <= 0
    env = G.MountainCarContinuous()  # MountainCarContinuousTests.cs:7-19
    assert env.ObservationSpace.Contains(list(env.Reset()))
    env.Step(0.5)
    env = G.Pendulum()  # PendulumTests.cs:7-18
    assert len(env.Reset()) == 3 and env.Step(0.0)[1] <= 0
    env = G.Pong()  # AtariEnvTests.cs:9-91
    frame = env.Reset()
    assert frame.dtype == np.uint8 and len(frame) == 210 * 160 * 3
    s, r, d, info = env.Step(2)
    assert -1 <= r <= 1 and set(info) == {"left_score", "right_score", "ball_x", "ball_y"}
    env = G.Breakout()  # AtariEnvTests.cs:37-62
    assert env.ActionSpace.N == 4
    frame = env.Reset()
    assert frame.dtype == np.uint8 and len(frame) == 210 * 160 * 3
    s, r, d, info = env.Step(0)
    assert len(s) == 210 * 160 * 3 and r >= 0
    assert info == {"score": 0, "lives": 5, "ball_x": 80.0, "ball_y": 180.0, "bricks_remaining": 108}
    env = G.SpaceInvaders()  # AtariEnvTests.cs:64-90
    assert env.ActionSpace.N == 6
    frame = env.Reset()
    assert frame.dtype == np.uint8 and len(frame) == 210 * 160 * 3
    s, r, d, info = env.Step(0)
    assert len(s) == 210 * 160 * 3 and r >= 0
    assert info == {"score": 0, "lives": 3, "aliens_remaining": 55, "player_x": 76.0}
    G.RegisterAll()  # GymnasiumRegistration.cs:22-24
    assert isinstance(G.EnvRegistry.Make("Breakout-v4"), G.Breakout) and isinstance(G.EnvRegistry.Make("SpaceInvaders-v4"), G.SpaceInvaders)
    env = G.Blackjack()  # Envs.Blackjack.cs:21-40: (sum, dealer card, usable ace); ten steps, then reward 1
    s0 = env.Reset()
    assert 12 <= s0[0] <= 21 and 1 <= s0[1] <= 10 and isinstance(s0[2], bool)
    outs = [env.Step(t & 1) for t in range(10)]
    assert all(o[0] == s0 for o in outs) and [o[2] for o in outs] == [False] * 9 + [True]
    assert [o[1] for o in outs] == [0.0] * 9 + [1.0]
    # envs built side by side without a seed are distinct, like the reference's unseeded Random per env
    firsts = {G.CartPole().Reset() for _ in range(4)}
    assert len(firsts) == 4
    assert G.CartPole(seed=5).Reset() == G.CartPole(seed=5).Reset()
    env = G.DummyEnv()  # DummyEnvTests.cs:6-21
    assert env.Reset() == 0
    s, r, d, _ = env.Step(1)
    assert (s, r, d) == (1, 1.0, False)
    for _ in range(9):
        s, r, d, _ = env.Step(1)
    assert d


@pytest.mark.gpu
def test_vector_env_drop_in(oracle):
    """VectorEnv.cs:10-35 built from a list of wrapped envs, stepped with a list of actions"""
    n = 64
    envs = [G.RecordEpisodeStatistics(G.TimeLimit(G.CartPole(seed=3), 12)) for _ in range(n)]
    v = G.VectorEnv(envs)
    assert v.NumEnvs == n
    o = oracle.OracleVec(oracle.CARTPOLE, n, seed=3, flags=oracle.RECORD_STATS, time_limit=12)
    states = v.Reset(); o.reset()
    assert len(states) == n and len(states[0]) == 4
    rng = np.random.default_rng(0)
    for t in range(30):
        a = [int(x) for x in rng.integers(0, 2, n)]
        res = v.Step(a)
        orew, odone, otrunc = o.step(a)
        assert [r[2] for r in res] == [bool(x) for x in odone]
        assert [r[1] for r in res] == list(orew)
        assert [("TimeLimit.truncated" in r[3]) for r in res] == [bool(x) for x in otrunc]
        if t == 20:
            v.Reset(); o.reset()
    for i in range(n):
        w = envs[i]
        assert list(zip(w.EpisodeRewards, w.EpisodeLengths)) == o.episodes(i)
    s, so = v.Batch.stats(), o.stats()
    assert s.episodes == so.episodes and s.sum_return == so.sum_return and s.sum_length == so.sum_length
    big = G.VectorEnv.Make(G.FrozenLake, 5000, map=["SFFF", "FHFH", "FFFH", "HFFG"], autoreset=True, record_stats=True)
    big.Reset()
    obs, rew, done, _ = big.StepArrays(np.ones(5000, np.int32))
    assert (obs == 4).all() and not done.any()


@pytest.mark.gpu
def test_fp64_mode_through_the_host_mirror(oracle):
    """the binary64 extension behind the same Env / VectorEnv surface: states come back as doubles"""
    from gym_amd import _lib as L
    env = G.Pendulum(seed=5, mode=L.MODE_FP64)
    o = oracle.OracleVec(oracle.PENDULUM, 1, seed=5, flags=oracle.FP64)
    s = env.Reset(); o.reset()
    assert s.dtype == np.float64 and s.shape == (3,)
    assert np.abs(s - o.get_obs()[:, 0]).max() <= 1e-15
    s2, r, d, _ = env.Step(1.5)
    orew, _, _ = o.step(np.array([1.5], np.float32))
    assert abs(r - orew[0]) <= 1e-12 and not d and s2.dtype == np.float64
    v = G.VectorEnv.Make(G.CartPole, 100, seed=9, mode=L.MODE_FP64)
    st = v.Reset()
    assert len(st) == 100 and all(isinstance(x, float) for x in st[0])
    ov = oracle.OracleVec(oracle.CARTPOLE, 100, seed=9, flags=oracle.FP64)
    ov.reset()
    assert st[7] == tuple(ov.get_obs()[:, 7])  # the reset formula has no transcendental: exact
