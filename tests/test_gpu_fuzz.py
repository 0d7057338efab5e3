"""Random sequences of ABI calls (reset, host-path step, in-kernel-policy step, fused rollout,
set_state, statistics) on the GPU engine and the same sequence on the oracle: the state after every
call and the statistics at the end must agree.  Catches what single-feature tests cannot: tick
accounting across call kinds, alternating sweep direction, derived returns meeting set_state,
results of the staged host path pending across other calls."""
import numpy as np
import pytest

from gpu_common import L, bits, make_pair, random_actions

pytestmark = pytest.mark.gpu

FUSED = L.AUTORESET | L.RECORD_STATS
CASES = [
    (L.ENV_CARTPOLE, FUSED, 0, None), (L.ENV_CARTPOLE, L.RECORD_STATS, 9, None), (L.ENV_MOUNTAINCAR, FUSED, 0, None),
    (L.ENV_PENDULUM, FUSED, 0, None), (L.ENV_ACROBOT, FUSED, 0, None), (L.ENV_FROZENLAKE, FUSED, 0, None),
    (L.ENV_FROZENLAKE, FUSED | L.SLIPPERY, 0, "8x8"), (L.ENV_TAXI, FUSED | L.TAXI_FULL, 0, None),
    (L.ENV_CLIFFWALKING, L.AUTORESET, 5, None), (L.ENV_PONG, FUSED, 0, None), (L.ENV_BREAKOUT, L.AUTORESET, 0, None),
]


def _same_state(g, o, env_id, what, fp64=False):
    gr, gi = g.get_state(); orr, oi = o.get_state()
    assert np.array_equal(gi, oi), what
    if gr.size == 0:
        return
    if fp64:
        # binary64 mode: held to a tolerance; resynchronise so that rounding differences do not compound
        assert (np.abs(gr - orr) / np.maximum(np.abs(orr), 1e-6)).max() <= 1e-8, what
        g.set_state(orr, oi)
        return
    if np.array_equal(bits(gr), bits(orr)):
        return
    # continuous control: a last-place trig difference may flip a float rounding; bound it and resynchronise
    assert env_id <= L.ENV_ACROBOT, what
    diff = bits(gr) != bits(orr)
    assert diff.sum() <= 2 and np.allclose(gr, orr, rtol=1e-5, atol=1e-6), what
    g.set_state(orr, oi)


@pytest.mark.parametrize("env_id", [L.ENV_CARTPOLE, L.ENV_MOUNTAINCAR_CONT, L.ENV_PENDULUM, L.ENV_ACROBOT])
def test_random_call_sequences_binary64(oracle, env_id):
    _run_sequence(oracle, env_id, FUSED, 0, None, L.MODE_FP64)


@pytest.mark.parametrize("env_id,flags,time_limit,rows", CASES)
def test_random_call_sequences(oracle, env_id, flags, time_limit, rows):
    _run_sequence(oracle, env_id, flags, time_limit, rows, L.MODE_EXACT)


def _run_sequence(oracle, env_id, flags, time_limit, rows, mode):
    fp64 = mode == L.MODE_FP64
    n = 257 if env_id in (L.ENV_PONG, L.ENV_BREAKOUT) else 1031
    g, o = make_pair(oracle, env_id, n, seed=1234 + env_id, flags=flags, time_limit=time_limit, map_rows=rows, base=7, mode=mode)
    rng = np.random.default_rng(99 + env_id + flags)
    g.reset(); o.reset()
    _same_state(g, o, env_id, "reset", fp64)
    for t in range(50):
        op = rng.choice(["host", "host", "random", "rollout", "set_state", "reset", "stats"], p=[0.3, 0.2, 0.2, 0.1, 0.1, 0.05, 0.05])
        if op == "host":
            a = random_actions(rng, env_id, n)
            _, rew, done, _ = g.step_host(a)
            orew, odone, _ = o.step(a)
            assert np.array_equal(done, odone), (t, op)
            if env_id > L.ENV_ACROBOT:  # integer and byte work: exact
                assert np.array_equal(rew.astype(np.float64), orew), (t, op)
            else:
                assert np.allclose(rew.astype(np.float64), orew, rtol=1e-5, atol=1e-6), (t, op)
        elif op == "random":
            g.step_random(); o.step_random()
        elif op == "rollout":
            k = 2 if fp64 else int(rng.integers(2, 6))  # binary64: short, so rounding differences stay far from thresholds
            g.rollout(k)
            for _ in range(k):
                o.step_random()
        elif op == "set_state":
            # shuffle the envs' states among themselves: valid states, arbitrary counters
            gr, gi = o.get_state()
            perm = rng.permutation(n)
            gr, gi = gr[:, perm].copy(), gi[:, perm].copy()
            g.set_state(gr, gi); o.set_state(gr, gi)
        elif op == "reset":
            g.reset(); o.reset()
        elif op == "stats" and (flags & L.RECORD_STATS):
            s, so = g.stats(), o.stats()
            assert s.episodes == so.episodes and s.sum_length == so.sum_length, (t, op)
        g.ctx.synchronize()
        _same_state(g, o, env_id, (t, op), fp64)
        assert g.tick == o.tick, (t, op)
    if flags & L.RECORD_STATS:
        s, so = g.stats(), o.stats()
        assert s.episodes == so.episodes and s.sum_length == so.sum_length
        assert abs(s.sum_return - so.sum_return) <= 1e-9 * max(1.0, abs(so.sum_return))
        if so.episodes:
            assert abs(s.min_return - so.min_return) <= 1e-9 * max(1.0, abs(so.min_return))
            assert abs(s.max_return - so.max_return) <= 1e-9 * max(1.0, abs(so.max_return))


def test_batches_sharing_a_context_do_not_interfere(oracle):
    """three batches on one context (one stream pair, one staging path each), stepped in turns"""
    specs = [(L.ENV_CARTPOLE, 1000, 3), (L.ENV_CARTPOLE, 37, 4), (L.ENV_FROZENLAKE, 513, 5), (L.ENV_PONG, 65, 6)]
    pairs = [make_pair(oracle, e, n, seed=s, flags=FUSED) for e, n, s in specs]
    for g, o in pairs:
        g.reset(); o.reset()
    rng = np.random.default_rng(8)
    for t in range(25):
        for idx in rng.permutation(len(pairs)):
            (g, o), (e, n, _) = pairs[idx], specs[idx]
            if rng.random() < 0.5:
                a = random_actions(rng, e, n)
                _, _, done, _ = g.step_host(a)
                _, odone, _ = o.step(a)
                assert np.array_equal(done, odone), (t, idx)
            else:
                g.step_random(); o.step_random()
    for (g, o), (e, n, _) in zip(pairs, specs):
        g.ctx.synchronize()
        _same_state(g, o, e, "end")
        s, so = g.stats(), o.stats()
        assert s.episodes == so.episodes and s.sum_length == so.sum_length


def test_distinct_handles_from_distinct_threads(oracle):
    """the ABI's threading rule: one handle per thread at a time, different handles concurrently"""
    import threading
    from gym_amd.engine import Context, DeviceBatch

    errors = []

    def worker(k):
        try:
            c = Context(0)
            n = 2000 + 17 * k
            g = DeviceBatch(c, L.ENV_CARTPOLE, n, seed=70 + k, flags=FUSED)
            o = oracle.OracleVec(oracle.CARTPOLE, n, seed=70 + k, flags=FUSED)
            g.reset(); o.reset()
            rng = np.random.default_rng(k)
            for t in range(40):
                a = rng.integers(0, 2, n).astype(np.int32)
                _, rew, done, _ = g.step_host(a)
                orew, odone, _ = o.step(a)
                assert np.array_equal(done, odone) and np.array_equal(rew.astype(np.float64), orew), (k, t)
            s, so = g.stats(), o.stats()
            assert s.episodes == so.episodes and s.sum_return == so.sum_return
            g.close(); c.close()
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
