"""Caller-supplied transition tables (gx_desc.table, GX_ENV_TABLE) through the C ABI: bit-exact against the
oracle's table env, and the reference's own table-driven envs fed back through that door equal their built-in
batches."""
import ctypes as C

import numpy as np
import pytest

import tables as T
from gym_amd import _lib as L
from gym_amd.engine import DeviceBatch, torch_views
from gpu_common import ctx

pytestmark = pytest.mark.gpu

BOTH = L.AUTORESET | L.RECORD_STATS


def table_pair(oracle, table, n, seed=3, flags=0, time_limit=0, n_outcomes=1, start_state=0, step_limit=0, base=0):
    g = DeviceBatch(ctx(), L.ENV_TABLE, n, seed=seed, flags=flags, time_limit=time_limit, table=table, n_outcomes=n_outcomes,
                    start_state=start_state, step_limit=step_limit, env_index_base=base)
    o = oracle.OracleVec(oracle.TABLE, n, seed=seed, flags=flags, time_limit=time_limit, table=table, n_outcomes=n_outcomes,
                         start_state=start_state, step_limit=step_limit, env_index_base=base)
    return g, o


def same_stats(gs, os_):
    assert (gs.episodes, gs.sum_length, gs.sum_return, gs.sum_return_sq) == (os_.episodes, os_.sum_length, os_.sum_return, os_.sum_return_sq)
    if gs.episodes:
        assert (gs.min_return, gs.max_return) == (os_.min_return, os_.max_return)


@pytest.mark.parametrize("n_outcomes,start_state,flags", [(1, 7, BOTH), (1, -1, BOTH), (3, -1, BOTH), (3, 0, 0), (1, -1, L.RECORD_STATS)])
def test_random_table_host_path(oracle, n_outcomes, start_state, flags):
    """300 states x 5 actions, twelve distinct integer rewards, 5 % terminal entries, step limit 40"""
    rng = np.random.default_rng(11 + n_outcomes)
    table = T.random_table(rng, 300, 5, n_outcomes)
    n = 3001
    g, o = table_pair(oracle, table, n, flags=flags, n_outcomes=n_outcomes, start_state=start_state, step_limit=40, base=5)
    assert g.info.action_n == 5 and g.info.n_int == 2 and g.info.obs_dtype == L.I32
    g.reset(); o.reset()
    assert np.array_equal(g.get_state()[1], o.get_state()[1])
    for t in range(120):
        a = rng.integers(0, 5, n).astype(np.int32)
        obs, rew, done, _ = g.step_host(a, want_obs=True)
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone), t
        assert np.array_equal(obs, o.get_obs())
    assert np.array_equal(g.get_state()[1], o.get_state()[1])
    if flags & L.RECORD_STATS:
        same_stats(g.stats(), o.stats())
        assert g.stats().episodes > n or not (flags & L.AUTORESET)


def test_random_table_in_kernel_policy_rollouts_and_time_limit(oracle):
    rng = np.random.default_rng(5)
    table = T.random_table(rng, 300, 5, 3, p_done=0.01)
    n = 10_003
    g, o = table_pair(oracle, table, n, flags=BOTH, time_limit=25, n_outcomes=3, start_state=-1, step_limit=60, base=1 << 33)
    g.reset(); o.reset()
    for k in (1, 7, 1, 32):
        g.rollout(k) if k > 1 else g.step_random()
        for _ in range(k):
            _, orew, odone, otr = o.step_random()
        g.wait()
        tv = torch_views(g)
        assert np.array_equal(tv["reward"][:n].cpu().numpy().astype(np.float64), orew)
        assert np.array_equal(tv["done"][:n].cpu().numpy(), odone) and np.array_equal(tv["truncated"][:n].cpu().numpy(), otr)
        assert np.array_equal(g.get_state()[1], o.get_state()[1])
    same_stats(g.stats(), o.stats())
    # the draws of the random policy are what gx_vec_sample_actions hands to a replay harness
    import torch
    acts = torch.empty(int(g.view().stride), dtype=torch.int32, device="cuda")
    g.sample_actions(g.tick, acts.data_ptr())
    ctx().synchronize()
    a, *_ = o.step_random()
    assert np.array_equal(acts[:n].cpu().numpy(), a) and a.max() == 4 and a.min() == 0


def test_table_larger_than_lds_is_read_through_l2(oracle):
    """5 000 states x 6 actions: 35 000 packed words = 137 KiB, beyond the 64 KiB that are staged in LDS"""
    rng = np.random.default_rng(9)
    table = T.random_table(rng, 5000, 6, 1, rewards=np.arange(-100, 100), p_done=0.02)
    n = 20_001
    g, o = table_pair(oracle, table, n, flags=BOTH, start_state=-1, step_limit=30)
    g.reset(); o.reset()
    for t in range(40):
        a = rng.integers(0, 6, n).astype(np.int32)
        _, rew, done, _ = g.step_host(a)
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone)
    g.rollout(16)
    for _ in range(16):
        o.step_random()
    g.wait()
    assert np.array_equal(g.get_state()[1], o.get_state()[1])
    same_stats(g.stats(), o.stats())


CASES = [
    ("frozenlake 4x4", L.ENV_FROZENLAKE, 0, None, lambda: T.frozenlake(T.MAP4), dict(start_state=0, step_limit=100)),
    ("frozenlake 8x8", L.ENV_FROZENLAKE, 0, "8x8", lambda: T.frozenlake(T.MAP8), dict(start_state=0, step_limit=100)),
    ("frozenlake 8x8 slippery", L.ENV_FROZENLAKE, L.SLIPPERY, "8x8", lambda: T.frozenlake(T.MAP8, True), dict(start_state=0, step_limit=100, n_outcomes=3)),
    ("cliffwalking", L.ENV_CLIFFWALKING, 0, None, T.cliffwalking, dict(start_state=36, step_limit=100)),
    ("taxi stub", L.ENV_TAXI, 0, None, T.taxi_stub, dict(start_state=-1, step_limit=200)),
]


@pytest.mark.parametrize("name,env_id,xflags,map_rows,make,kw", CASES, ids=[c[0] for c in CASES])
def test_the_references_tables_through_the_new_door_equal_the_builtin_batches(oracle, name, env_id, xflags, map_rows, make, kw):
    """FrozenLake (Envs.FrozenLake.cs:23-69), CliffWalking and the Taxi stub written out as P[s][a] and fed through
    gx_desc.table: the same states, rewards, flags and statistics as the batch built from the env id (and the map),
    step for step, and as the oracle's env of that id"""
    n, seed = 6007, 17
    rng = np.random.default_rng(2)
    n_act = {L.ENV_FROZENLAKE: 4, L.ENV_CLIFFWALKING: 4, L.ENV_TAXI: 6}[env_id]
    builtin = DeviceBatch(ctx(), env_id, n, seed=seed, flags=BOTH | xflags, map_rows=map_rows)
    door = DeviceBatch(ctx(), L.ENV_TABLE, n, seed=seed, flags=BOTH, table=make(), **kw)
    o_rows = T.MAP8 if map_rows == "8x8" else None
    o = oracle.OracleVec(env_id, n, seed=seed, flags=BOTH | xflags, map_rows=o_rows)
    for b in (builtin, door, o):
        b.reset()
    assert np.array_equal(builtin.get_state()[1], door.get_state()[1])
    for t in range(230):
        a = rng.integers(0, n_act, n).astype(np.int32)
        _, r1, d1, _ = builtin.step_host(a)
        _, r2, d2, _ = door.step_host(a)
        orew, odone, _ = o.step(a)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2), (name, t)
        assert np.array_equal(r2.astype(np.float64), orew) and np.array_equal(d2, odone), (name, t)
    assert np.array_equal(builtin.get_state()[1], door.get_state()[1]) and np.array_equal(door.get_state()[1], o.get_state()[1])
    same_stats(door.stats(), builtin.stats())
    same_stats(door.stats(), o.stats())
    # and with the actions sampled in the kernel (same word of the tick for the action and the slippery outcome)
    builtin.rollout(24)
    door.rollout(24)
    builtin.wait(); door.wait()
    assert np.array_equal(builtin.get_state()[1], door.get_state()[1])
    same_stats(door.stats(), builtin.stats())


def test_fractional_rewards_without_statistics(oracle):
    rng = np.random.default_rng(4)
    table = T.random_table(rng, 64, 3, 1, rewards=[0.25, -1.5, 3.1415927, 1e-3, -0.0])
    n = 999
    g, o = table_pair(oracle, table, n, flags=L.AUTORESET, start_state=3, step_limit=20)
    g.reset(); o.reset()
    for t in range(50):
        a = rng.integers(0, 3, n).astype(np.int32)
        _, rew, done, _ = g.step_host(a)
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.view(np.uint32), orew.astype(np.float32).view(np.uint32)) and np.array_equal(done, odone)
    with pytest.raises(L.GymxError) as e:
        DeviceBatch(ctx(), L.ENV_TABLE, n, flags=BOTH, table=table)
    assert e.value.code == L.GX_ERR_UNSUPPORTED and "integers" in str(e.value)


def test_256_distinct_rewards_fit_and_257_do_not(oracle):
    rng = np.random.default_rng(6)
    table = T.random_table(rng, 400, 4, 1, rewards=[0])
    table["reward"].reshape(-1)[:256] = np.arange(256) - 128
    n = 4001
    g, o = table_pair(oracle, table, n, flags=BOTH, start_state=-1, step_limit=10)
    g.reset(); o.reset()
    for t in range(30):
        a = rng.integers(0, 4, n).astype(np.int32)
        _, rew, done, _ = g.step_host(a)
        orew, odone, _ = o.step(a)
        assert np.array_equal(rew.astype(np.float64), orew) and np.array_equal(done, odone)
    same_stats(g.stats(), o.stats())
    table["reward"].reshape(-1)[256] = 1000
    with pytest.raises(L.GymxError) as e:
        DeviceBatch(ctx(), L.ENV_TABLE, n, table=table)
    assert e.value.code == L.GX_ERR_UNSUPPORTED and "256" in str(e.value)


def test_what_create_rejects():
    t = T.random_table(np.random.default_rng(1), 10, 2)

    def bad(code, **kw):
        a = dict(table=t, start_state=0)
        a.update(kw)
        with pytest.raises(L.GymxError) as e:
            DeviceBatch(ctx(), L.ENV_TABLE, 16, **a)
        assert e.value.code == code, str(e.value)
        return str(e.value)

    t2 = t.copy(); t2["next_state"][3, 1, 0] = 10
    assert "state 10" in bad(L.GX_ERR_INVALID, table=t2)
    t2 = t.copy(); t2["next_state"][0, 0, 0] = -1
    bad(L.GX_ERR_INVALID, table=t2)
    t2 = t.copy(); t2["reward"][0, 0, 0] = np.inf
    bad(L.GX_ERR_INVALID, table=t2)
    bad(L.GX_ERR_INVALID, start_state=10)
    bad(L.GX_ERR_INVALID, start_state=-2)
    bad(L.GX_ERR_INVALID, step_limit=-1)
    bad(L.GX_ERR_UNSUPPORTED, n_outcomes=2)
    bad(L.GX_ERR_INVALID, n_outcomes=3)           # table_len is n_states * n_actions, not * 3
    bad(L.GX_ERR_INVALID, table=T.random_table(np.random.default_rng(1), 10, 256))
    lib = L.load()
    # no table at all, and a table with the first version of the struct
    d = L.GxDesc(C.sizeof(L.GxDesc), L.ENV_TABLE, 16, 0, 0, 0, 0, 0, 0, None)
    h = C.c_void_p()
    assert lib.gx_vec_create(ctx()._h, C.byref(d), C.byref(h)) == L.GX_ERR_INVALID and not h.value
    d.struct_size = L.DESC_SIZE_V1
    assert lib.gx_vec_create(ctx()._h, C.byref(d), C.byref(h)) == L.GX_ERR_INVALID and not h.value


def test_first_version_of_the_struct_still_creates_the_other_envs(oracle):
    lib = L.load()
    d = L.GxDesc(L.DESC_SIZE_V1, L.ENV_FROZENLAKE, 64, 5, 0, 0, 0, 0, 0, b"SFG/HFF")
    d.table, d.table_len, d.n_states = 0xDEAD0000, 99, 7   # beyond the declared size: never read
    h = C.c_void_p()
    assert lib.gx_vec_create(ctx()._h, C.byref(d), C.byref(h)) == L.GX_OK, lib.gx_last_error()
    info = L.GxInfo()
    assert lib.gx_vec_info(h, C.byref(info)) == L.GX_OK and info.action_n == 4
    assert lib.gx_vec_destroy(h) == L.GX_OK
    d.struct_size = 60
    assert lib.gx_vec_create(ctx()._h, C.byref(d), C.byref(h)) == L.GX_ERR_INVALID


def test_an_action_outside_the_space_is_an_error_on_both_paths(oracle):
    import torch

    table = T.random_table(np.random.default_rng(1), 10, 3)
    n = 515
    g, o = table_pair(oracle, table, n, start_state=0)
    g.reset(); o.reset()
    a = np.zeros(n, np.int32)
    a[77] = 3
    before = g.get_state()[1].copy()
    with pytest.raises(L.GymxError) as e:
        g.step_host(a)
    assert e.value.code == L.GX_ERR_INVALID and "env 77" in str(e.value)
    assert np.array_equal(g.get_state()[1], before)            # rejected before any env was stepped
    with pytest.raises(IndexError):
        o.step(a)
    # zero-copy path: the wait that follows reports it, once
    acts = torch.zeros(int(g.view().stride), dtype=torch.int32, device="cuda")
    acts[200] = -1
    acts[n:] = 99                                               # padding is never an error
    g.step_device(acts.data_ptr())
    with pytest.raises(L.GymxError) as e:
        g.wait()
    assert e.value.code == L.GX_ERR_INVALID
    g.wait()
    acts[200] = 2
    side = torch.cuda.Stream()                                  # a caller's own stream, destroyed before the wait
    with torch.cuda.stream(side):
        g.step_device(acts.data_ptr(), side.cuda_stream)
    side.synchronize()
    del side
    g.wait()


def test_table_env_through_the_host_mirror():
    """gym_amd.TableEnv / VectorEnv: the reference's CliffWalking known answers (Envs.CliffWalking.cs:36-54) from the
    table form, spaces as an agent reads them, IndexError for an action outside the space"""
    import gym_amd as G

    env = G.TableEnv(T.cliffwalking(), start_state=36, step_limit=100, seed=1)
    assert env.ActionSpace.N == 4 and env.ObservationSpace.N == 48
    assert env.Reset() == 36
    s, r, d, info = env.Step(3)
    assert (s, r, d) == (24, -1.0, False)
    total = r
    for a in [2] * 11 + [1]:
        s, r, d, info = env.Step(a)
        total += r
    assert (s, d, total) == (47, True, -12.0)
    assert env.Reset() == 36 and env.Step(2)[:3] == (37, -100.0, True)
    with pytest.raises(IndexError):
        env.Step(4)
    vec = G.VectorEnv([G.TableEnv(T.cliffwalking(), start_state=36, step_limit=100, seed=1) for _ in range(5)])
    assert vec.NumEnvs == 5 and vec.Reset() == [36] * 5
    res = vec.Step([3, 2, 0, 1, 3])
    assert [x[0] for x in res] == [24, 37, 36, 36, 24] and [x[2] for x in res] == [False, True, False, False, False]
    with pytest.raises(ValueError):
        G.VectorEnv([G.TableEnv(T.cliffwalking(), start_state=36), G.TableEnv(T.taxi_stub(), start_state=-1)])
