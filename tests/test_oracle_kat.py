"""Known-answer tests that pin the CPU oracle.

The only value-level test the reference holds on this path is the DummyEnv
one (Gymnasium.Tests/DummyEnvTests.cs:6-21); everything else here is derived
by hand from the reference source and stated next to the lines it follows.
"""
import math

import numpy as np
import pytest

f32 = np.float32


def test_philox_random123_vectors(oracle):
    # Philox4x32-10 known answers published with the Random123 library.
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_range_and_keying(oracle):
    us = [oracle.rng_uniform(7, e, t, s, d) for e in range(4) for t in range(3) for s in range(2) for d in range(4)]
    us += [oracle.rng_uniform(7, e, t, 2, 0) for e in range(4) for t in range(12)]  # the policy stream: one draw per tick
    assert all(0.0 <= u < 1.0 for u in us)
    assert len(set(us)) == len(us)
    # ticks 4m .. 4m+3 of the policy stream are the four words of one block: counter (env, 0, m, 2 << 24)
    assert [oracle.rng_u32(7, 5, 8 + i, 2, 0) for i in range(4)] == oracle.philox([5, 0, 2, 2 << 24], [7, 0])
    assert all(0 <= oracle.rng_below(7, e, 0, 2, 0, 3) < 3 for e in range(100))


def test_dummy_env_reference_kat(oracle):
    """Gymnasium.Tests/DummyEnvTests.cs:6-21."""
    v = oracle.OracleVec(oracle.DUMMY, 1)
    v.reset()
    assert v.get_obs()[0] == 0
    r, d, _ = v.step([1])
    assert v.get_obs()[0] == 1 and r[0] == 1.0 and not d[0]
    for _ in range(9):
        r, d, _ = v.step([1])
    assert d[0]


def test_folded_constants():
    # binary32 constant folding as the C# compiler does it
    assert float(f32(1.0) + f32(0.1)) == 1.100000023841858
    assert float(f32(0.1) * f32(0.5)) == 0.05000000074505806
    assert float(f32(4.0) / f32(3.0)) == 1.3333333730697632
    thr = f32(f32(12) * f32(math.pi)) / f32(180)
    assert float(thr) == 0.20943951606750488 and thr.view(np.uint32) == 0x3E567750
    assert float(f32(0.5) * f32(9.8)) == 4.900000095367432
    assert float(f32(1.5) * f32(9.8)) == 14.700000762939453  # one ulp above 14.7f
    assert float(f32(4) * f32(math.pi)) == 12.566370964050293
    assert float(f32(9) * f32(math.pi)) == 28.274333953857422


def test_frozenlake_paths(oracle):
    """Envs.FrozenLake.cs:52-69 on the default 4x4 map (:25-30)."""
    v = oracle.OracleVec(oracle.FROZENLAKE, 1)
    v.reset()
    states, rewards, dones = [], [], []
    for a in [1, 1, 2, 1, 2, 2]:
        r, d, _ = v.step([a])
        states.append(int(v.get_obs()[0])); rewards.append(r[0]); dones.append(bool(d[0]))
    assert states == [4, 8, 9, 13, 14, 15]
    assert rewards == [0, 0, 0, 0, 0, 1]
    assert dones == [False] * 5 + [True]
    v.reset()
    r, d, _ = v.step([2]); assert v.get_obs()[0] == 1 and not d[0]
    r, d, _ = v.step([1]); assert v.get_obs()[0] == 5 and d[0] and r[0] == 0.0  # hole
    for a in (0, 3, 7, -1):
        v.reset()
        v.step([a])
        assert v.get_obs()[0] == 0
    v.reset()
    for k in range(100):
        r, d, _ = v.step([0])
        assert bool(d[0]) == (k == 99) and r[0] == 0.0


def test_frozenlake_custom_map(oracle):
    v = oracle.OracleVec(oracle.FROZENLAKE, 1, map_rows=["SFG", "HFF"])
    v.reset()
    r, d, _ = v.step([1]); assert v.get_obs()[0] == 3 and d[0] and r[0] == 0
    v.reset()
    v.step([2]); r, d, _ = v.step([2]); assert v.get_obs()[0] == 2 and d[0] and r[0] == 1


def test_cliffwalking_paths(oracle):
    """Envs.CliffWalking.cs:36-54."""
    v = oracle.OracleVec(oracle.CLIFFWALKING, 1)
    v.reset()
    assert v.get_obs()[0] == 36
    seq = [3] + [2] * 11 + [1]
    states, rewards, dones = [], [], []
    for a in seq:
        r, d, _ = v.step([a])
        states.append(int(v.get_obs()[0])); rewards.append(r[0]); dones.append(bool(d[0]))
    assert states == [24] + list(range(25, 36)) + [47]
    assert rewards == [-1.0] * 12 + [0.0] and sum(rewards) == -12
    assert dones == [False] * 12 + [True]
    v.reset()
    r, d, _ = v.step([2]); assert v.get_obs()[0] == 37 and r[0] == -100.0 and d[0]
    for a in (0, 1):
        v.reset(); v.step([a]); assert v.get_obs()[0] == 36


def test_taxi_stub(oracle):
    """Envs.Taxi.cs:25-33."""
    v = oracle.OracleVec(oracle.TAXI, 3, seed=5)
    v.reset()
    s0 = v.get_obs().copy()
    assert ((0 <= s0) & (s0 < 500)).all()
    for k in range(200):
        r, d, _ = v.step([k % 6] * 3)
        assert (r == -1.0).all() and (v.get_obs() == s0).all()
        assert d.all() == (k == 199) and d.any() == (k == 199)


def test_cartpole_step_limit_and_terminal_reward(oracle):
    """Envs.CartPole.cs:69-72: steps>=500 forces done; reward is 0 on the terminal step."""
    v = oracle.OracleVec(oracle.CARTPOLE, 1)
    v.reset()
    v.set_state(np.zeros((4, 1), f32), np.array([[498]], np.int32))
    r, d, _ = v.step([1]); assert not d[0] and r[0] == 1.0
    v.set_state(np.zeros((4, 1), f32), np.array([[499]], np.int32))
    r, d, _ = v.step([1]); assert d[0] and r[0] == 0.0
    v.set_state(np.array([[2.39], [10.0], [0], [0]], f32), np.array([[0]], np.int32))
    r, d, _ = v.step([0]); assert d[0] and r[0] == 0.0


def test_mountaincar_wall_branch(oracle):
    """Envs.MountainCar.cs:39-43."""
    v = oracle.OracleVec(oracle.MOUNTAINCAR, 1)
    v.reset()
    v.set_state(np.array([[-1.2], [-0.07]], f32), np.array([[0]], np.int32))
    r, d, _ = v.step([0])
    s, _ = v.get_state()
    assert s[0, 0] == f32(-1.2) and s[1, 0] == 0.0 and r[0] == -1.0 and not d[0]


def test_mountaincar_cont_reward(oracle):
    """Envs.MountainCarContinuous.cs:41-44: velocity*0.1 in double, +100 at the goal."""
    v = oracle.OracleVec(oracle.MOUNTAINCAR_CONT, 1)
    v.reset()
    v.set_state(np.array([[0.44], [0.07]], f32), np.array([[0]], np.int32))
    r, d, _ = v.step(np.array([5.0], f32))
    s, _ = v.get_state()
    assert d[0] and r[0] == float(s[1, 0]) * 0.1 + 100.0


def test_pendulum_never_done_and_reward_uses_old_state(oracle):
    """Envs.Pendulum.cs:56-57."""
    v = oracle.OracleVec(oracle.PENDULUM, 1)
    v.reset()
    th = f32(0.5)
    v.set_state(np.array([[math.cos(th)], [math.sin(th)], [1.5]], f32), np.array([[0]], np.int32))
    s0, _ = v.get_state()
    r, d, _ = v.step(np.array([3.0], f32))  # clamped to 2
    th_old = f32(math.atan2(float(s0[1, 0]), float(s0[0, 0])))
    expect = -(float(th_old * th_old) + 0.1 * 1.5 * 1.5 + 0.001 * 2.0 * 2.0)
    assert r[0] == expect and not d[0]
    for _ in range(300):
        _, d, _ = v.step(np.array([0.0], f32))
        assert not d[0]


def test_acrobot_invalid_action_throws(oracle):
    """Envs.Acrobot.cs:45,51: torque[action] is an array index."""
    v = oracle.OracleVec(oracle.ACROBOT, 1)
    v.reset()
    with pytest.raises(IndexError):
        v.step([3])


def test_pong_reset_frame(oracle):
    """Envs.Pong.cs:54-70,167-190: the frame right after Reset()."""
    v = oracle.OracleVec(oracle.PONG, 1, seed=3)
    v.reset()
    frame = v.get_obs()[0].reshape(210, 160, 3)
    assert (frame[..., 0] == frame[..., 1]).all() and (frame[..., 1] == frame[..., 2]).all()
    m = frame[..., 0] == 255
    assert set(np.unique(frame)) == {0, 255}
    exp = np.zeros((210, 160), bool)
    exp[97:112, 0:2] = True      # left paddle, (int)97.5 = 97
    exp[97:112, 158:160] = True  # right paddle
    exp[105:107, 80:82] = True   # ball at (80,105)
    for y in range(0, 210, 8):   # 27 dashes
        exp[y:y + 4, 79:81] = True
    zero = np.array([[1, 1, 1], [1, 0, 1], [1, 0, 1], [1, 0, 1], [1, 1, 1]], bool)
    exp[20:25, 37:40] |= zero
    exp[20:25, 117:120] |= zero
    assert (m == exp).all()


def test_timelimit_over_dummy(oracle):
    """Wrappers.TimeLimit.cs:23-33 with maxSteps=3, action 0."""
    v = oracle.OracleVec(oracle.DUMMY, 1, time_limit=3)
    v.reset()
    out = [v.step([0]) for _ in range(3)]
    assert [bool(o[1][0]) for o in out] == [False, False, True]
    assert [bool(o[2][0]) for o in out] == [False, False, True]


def test_record_episode_statistics_over_dummy(oracle):
    """EnvWrapper.cs:44-57 with action 5: rewards 5, 10 -> episode (15.0, 2)."""
    v = oracle.OracleVec(oracle.DUMMY, 1, flags=oracle.RECORD_STATS)
    v.reset()
    v.step([5]); v.step([5])
    assert v.episodes(0) == [(15.0, 2)]
    # EnvWrapper.cs:34-38: Reset() with an open episode records it first
    v.step([1])
    v.reset()
    assert v.episodes(0) == [(15.0, 2), (11.0, 1)]
    s = v.stats()
    assert s.episodes == 2 and s.sum_return == 26.0 and s.sum_length == 3 and s.min_return == 11.0


def test_box_and_discrete_sample(oracle):
    """Spaces.Box.cs:30, Spaces.Discrete.cs:18."""
    L = oracle.lib()
    assert L.gxo_box_sample(0.0, -2.0, 2.0) == -2.0
    assert L.gxo_box_sample(0.5, -2.0, 2.0) == 0.0
    u = 0.123456789
    assert L.gxo_box_sample(u, -1.0, 1.0) == float(f32(u * float(f32(1.0) - f32(-1.0)) + float(f32(-1.0))))
