"""gx_rows_step, the entry point that allocates nothing: K fused steps plus the statistics of the
episodes they finish, on device rows the caller owns, with the per-wave statistics records in the
library's fixed per-GPU array.  Checked through the C ABI against the existing entry points
(gx_vec_rollout + gx_vec_stats on a library-owned batch started from the same state): every row
and the statistics record bit for bit, on two calls in a row, the second on a smaller batch (the
shared array still holds the first call's records)."""
import ctypes as C

import numpy as np
import pytest

from gym_amd import _lib as L
from gym_amd.engine import DeviceBatch, rows_step, torch_views
from gpu_common import bits, ctx

pytestmark = pytest.mark.gpu

FLAGS = L.AUTORESET | L.RECORD_STATS
ENVS = [(L.ENV_CARTPOLE, 4), (L.ENV_MOUNTAINCAR, 2), (L.ENV_MOUNTAINCAR_CONT, 2), (L.ENV_PENDULUM, 3), (L.ENV_ACROBOT, 6)]


def _free_device_bytes(torch):
    """hipMemGetInfo's free bytes: sees every device allocation of the process, the library's own hipMalloc
    included (torch.cuda.memory_allocated() only counts torch's caching allocator)"""
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _one_call(torch, env_id, n_real, n, k, policy_random, warm_steps, seed, bounded=True):
    """-> nothing; asserts gx_rows_step == gx_vec_rollout + gx_vec_stats from the same start"""
    c = ctx()
    ref = DeviceBatch(c, env_id, n, seed=seed, flags=FLAGS)
    ref.reset()
    for _ in range(warm_steps):  # episodes under way, some envs close to their end
        ref.step_random()
    c.synchronize()
    s0 = ref.stats()  # the library-owned batch accumulates from its creation: subtract what it has so far
    tv = torch_views(ref)
    stride = int(ref.view().stride)
    f64 = env_id in (L.ENV_MOUNTAINCAR_CONT, L.ENV_PENDULUM)
    # the caller's rows: copies of the batch's, taken before either side steps
    state = tv["state_real"].clone()
    steps = tv["state_int"][ref.info.n_int - 1].clone()
    assert state.shape == (n_real, stride)
    ep_ret = None
    if f64:
        # the open episodes' returns are not part of the view: rebuild them as zero on both sides by
        # starting from a fresh Reset (no warm steps for these two envs)
        assert warm_steps == 0
        ep_ret = torch.zeros(stride, dtype=torch.float64, device="cuda")
    elif not bounded:
        # the kernel that tolerates states written from outside keeps the return row explicitly:
        # what the open episodes have collected so far is the reward per step times their length
        ep_ret = steps * (1 if env_id == L.ENV_CARTPOLE else -1)
    reward = torch.full((stride,), -7, dtype=torch.float64 if f64 else torch.float32, device="cuda")
    done = torch.full((stride,), 9, dtype=torch.uint8, device="cuda")
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    err = torch.zeros(4, dtype=torch.int32, device="cuda")
    tick = ref.tick
    actions = None
    if not policy_random:
        adt = torch.float32 if ref.info.action_dtype == L.F32 else torch.int32
        actions = torch.empty((k, stride), dtype=adt, device="cuda")
        for i in range(k):
            ref.sample_actions(1000 + i, actions[i].data_ptr())
        c.synchronize()
    torch.cuda.synchronize()

    def call(state, steps, reward, done, stats, ep_ret, err):
        rows_step(c, env_id, n, stride, seed, tick, state.data_ptr(), steps.data_ptr(), reward.data_ptr(), done.data_ptr(),
                  stats.data_ptr(), k_steps=k, actions=actions.data_ptr() if actions is not None else None,
                  ep_return=ep_ret.data_ptr() if ep_ret is not None else None, error=err.data_ptr(), state_bounded=bounded)
        c.synchronize()

    # one call of the same kind on copies of the rows first: whatever the runtime sets up once per kernel (code
    # objects, launch structures) exists before the measured call
    call(state.clone(), steps.clone(), reward.clone(), done.clone(), stats.clone(), None if ep_ret is None else ep_ret.clone(), err.clone())
    before = _free_device_bytes(torch)
    before_torch = torch.cuda.memory_allocated()
    call(state, steps, reward, done, stats, ep_ret, err)
    assert _free_device_bytes(torch) == before          # no hipMalloc / hipFree by anyone in the process
    assert torch.cuda.memory_allocated() == before_torch

    ref.rollout(k, L.POLICY_RANDOM if policy_random else L.POLICY_DEVICE_ACTIONS,
                actions.data_ptr() if actions is not None else None)
    ref.wait()
    s1 = ref.stats()
    assert ref.tick == tick + k

    assert np.array_equal(bits(state[:, :n].cpu().numpy()), bits(tv["state_real"][:, :n].cpu().numpy()))
    assert np.array_equal(steps[:n].cpu().numpy(), tv["state_int"][ref.info.n_int - 1][:n].cpu().numpy())
    assert np.array_equal(bits(reward[:n].cpu().numpy()), bits(tv["reward"][:n].cpu().numpy()))
    assert np.array_equal(done[:n].cpu().numpy(), tv["done"][:n].cpu().numpy())
    got = stats.cpu().numpy()
    episodes, sum_length = int(got[0:1].view(np.int64)[0]), int(got[2:3].view(np.int64)[0])
    assert episodes == s1.episodes - s0.episodes
    assert sum_length == s1.sum_length - s0.sum_length
    if f64:
        # doubles: the library-owned batch adds this call's episodes onto none (s0 is empty), in the same order
        assert s0.episodes == 0
        assert got[1] == s1.sum_return and got[3] == s1.sum_return_sq
    else:
        assert got[1] == s1.sum_return - s0.sum_return and got[3] == s1.sum_return_sq - s0.sum_return_sq  # integers in double
    if episodes and s0.episodes == 0:
        assert got[4] == s1.min_return and got[5] == s1.max_return
    assert int(err[0]) == 0
    ref.close()
    return episodes


@pytest.mark.parametrize("env_id,n_real", ENVS)
def test_two_calls_in_a_row_second_smaller(env_id, n_real):
    import torch

    f64 = env_id in (L.ENV_MOUNTAINCAR_CONT, L.ENV_PENDULUM)
    # ragged sizes (not multiples of 4 or 64); CartPole episodes under a random policy last ~22 steps
    warm = 0 if f64 else 30
    e1 = _one_call(torch, env_id, n_real, 300_007, 8, True, warm, seed=11)
    e2 = _one_call(torch, env_id, n_real, 4_099, 5, False, warm, seed=12)
    if env_id in (L.ENV_CARTPOLE, L.ENV_MOUNTAINCAR):
        # third call: states declared as written from outside (the general kernel, explicit int32 return row)
        _one_call(torch, env_id, n_real, 1_001, 4, True, warm, seed=13, bounded=False)
    if env_id == L.ENV_CARTPOLE:
        assert e1 > 1000 and e2 > 10  # the statistics path was exercised, and the smaller call did not see the larger one's records


def test_rejects_what_it_cannot_serve():
    import torch

    c = ctx()
    buf = torch.zeros(64, dtype=torch.float64, device="cuda")
    p = buf.data_ptr()

    def call(**kw):
        a = dict(env_id=L.ENV_CARTPOLE, num_envs=8, stride=8, seed=1, tick=1, state_real=p, steps=p, reward=p, done=p, stats=p)
        a.update(kw)
        with pytest.raises(L.GymxError) as e:
            rows_step(c, **a)
        return e.value

    assert call(num_envs=L.ROWS_MAX_ENVS + 1, stride=L.ROWS_MAX_ENVS + 4).code == L.GX_ERR_INVALID
    assert "scratch" in str(call(num_envs=L.ROWS_MAX_ENVS + 1, stride=L.ROWS_MAX_ENVS + 4))
    assert call(stride=6).code == L.GX_ERR_INVALID
    assert call(env_id=L.ENV_PONG).code == L.GX_ERR_UNSUPPORTED
    assert call(mode=L.MODE_FP64).code == L.GX_ERR_UNSUPPORTED
    assert call(env_id=L.ENV_PENDULUM).code == L.GX_ERR_INVALID  # needs its ep_return row
    assert call(state_bounded=False).code == L.GX_ERR_INVALID   # CartPole without the row, unbounded state
    assert call(state_real=p + 4).code == L.GX_ERR_INVALID
    assert call(k_steps=0).code == L.GX_ERR_INVALID


def test_runs_under_stream_capture():
    """the strictest form of "allocates nothing": the whole call is recorded into a HIP graph -- where an allocation, a
    free or a synchronisation ends the capture with an error -- and the replayed graph gives the existing entry
    points' results"""
    import torch

    c = ctx()
    env_id, n, k, seed = L.ENV_CARTPOLE, 70_001, 6, 21
    ref = DeviceBatch(c, env_id, n, seed=seed, flags=FLAGS)
    ref.reset()
    for _ in range(25):
        ref.step_random()
    c.synchronize()
    s0 = ref.stats()
    tv = torch_views(ref)
    stride = int(ref.view().stride)
    state = tv["state_real"].clone()
    steps = tv["state_int"][ref.info.n_int - 1].clone()
    reward = torch.zeros(stride, dtype=torch.float32, device="cuda")
    done = torch.zeros(stride, dtype=torch.uint8, device="cuda")
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    tick = ref.tick

    def call(stream):
        rows_step(c, env_id, n, stride, seed, tick, state.data_ptr(), steps.data_ptr(), reward.data_ptr(), done.data_ptr(),
                  stats.data_ptr(), k_steps=k, stream=stream)

    # a plain call on copies first, so that the kernels' code is resident before the capture
    keep = (state.clone(), steps.clone())
    side = torch.cuda.Stream()
    call(side.cuda_stream)
    side.synchronize()
    state.copy_(keep[0])
    steps.copy_(keep[1])
    torch.cuda.synchronize()

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="global"):
        call(torch.cuda.current_stream().cuda_stream)
    # nothing has run yet: the rows still hold the starting state
    torch.cuda.synchronize()
    assert np.array_equal(bits(state.cpu().numpy()), bits(keep[0].cpu().numpy()))
    g.replay()
    torch.cuda.synchronize()

    ref.rollout(k, L.POLICY_RANDOM, None)
    ref.wait()
    s1 = ref.stats()
    assert np.array_equal(bits(state[:, :n].cpu().numpy()), bits(tv["state_real"][:, :n].cpu().numpy()))
    assert np.array_equal(steps[:n].cpu().numpy(), tv["state_int"][ref.info.n_int - 1][:n].cpu().numpy())
    assert np.array_equal(bits(reward[:n].cpu().numpy()), bits(tv["reward"][:n].cpu().numpy()))
    assert np.array_equal(done[:n].cpu().numpy(), tv["done"][:n].cpu().numpy())
    got = stats.cpu().numpy()
    assert int(got[0:1].view(np.int64)[0]) == s1.episodes - s0.episodes > 0
    assert got[1] == s1.sum_return - s0.sum_return
    ref.close()


def test_largest_call_cartpole():
    """GX_ROWS_MAX_ENVS envs, one step from device actions: the scratch array holds a record for every wave of the
    grid gx_vec_create gives that batch"""
    import torch

    e = _one_call(torch, L.ENV_CARTPOLE, 4, L.ROWS_MAX_ENVS, 1, False, 40, seed=31)
    assert e > 100_000


def test_largest_call_sums_binary64_returns_in_the_batch_order():
    """MountainCarContinuous at GX_ROWS_MAX_ENVS envs, most of them one step from the goal with different velocities:
    their returns (velocity * 0.1 + 100, Envs.MountainCarContinuous.cs:42-44) are doubles whose sum depends on
    the order of the additions, i.e. on the grid and on which wave owns which record; gx_rows_step gives
    gx_vec_step_device + gx_vec_stats' sums bit for bit"""
    import torch

    c = ctx()
    n, seed = L.ROWS_MAX_ENVS, 41
    ref = DeviceBatch(c, L.ENV_MOUNTAINCAR_CONT, n, seed=seed, flags=FLAGS)
    ref.reset()
    rng = np.random.default_rng(5)
    reals = np.empty((2, n), np.float32)
    reals[0] = 0.40 + 0.06 * rng.random(n, dtype=np.float32)     # position: the goal is at 0.45
    reals[1] = 0.02 + 0.05 * rng.random(n, dtype=np.float32)     # velocity (clamped at 0.07)
    ints = rng.integers(1, 60, (1, n)).astype(np.int32)
    ref.set_state(reals, ints)
    tv = torch_views(ref)
    stride = int(ref.view().stride)
    state = tv["state_real"].clone()
    steps = tv["state_int"][0].clone()
    ep_ret = torch.zeros(stride, dtype=torch.float64, device="cuda")   # set_state keeps the returns collected so far: none
    reward = torch.zeros(stride, dtype=torch.float64, device="cuda")
    done = torch.zeros(stride, dtype=torch.uint8, device="cuda")
    stats = torch.zeros(6, dtype=torch.float64, device="cuda")
    actions = torch.empty(stride, dtype=torch.float32, device="cuda")
    ref.sample_actions(77, actions.data_ptr())
    c.synchronize()
    tick = ref.tick
    rows_step(c, L.ENV_MOUNTAINCAR_CONT, n, stride, seed, tick, state.data_ptr(), steps.data_ptr(), reward.data_ptr(),
              done.data_ptr(), stats.data_ptr(), k_steps=1, actions=actions.data_ptr(), ep_return=ep_ret.data_ptr(),
              state_bounded=False)
    c.synchronize()
    ref.step_device(actions.data_ptr())
    ref.wait()
    s1 = ref.stats()
    assert np.array_equal(bits(state[:, :n].cpu().numpy()), bits(tv["state_real"][:, :n].cpu().numpy()))
    assert np.array_equal(bits(reward[:n].cpu().numpy()), bits(tv["reward"][:n].cpu().numpy()))
    assert np.array_equal(done[:n].cpu().numpy(), tv["done"][:n].cpu().numpy())
    got = stats.cpu().numpy()
    episodes = int(got[0:1].view(np.int64)[0])
    assert episodes == s1.episodes and episodes > n // 4
    assert got[1] == s1.sum_return and got[3] == s1.sum_return_sq      # doubles, bit for bit
    assert got[4] == s1.min_return and got[5] == s1.max_return
    # and the sum does depend on the order: the same returns added in env order differ in the last places
    ret = reward[:n][done[:n] != 0].cpu().numpy()
    assert abs(float(np.sum(ret)) - got[1]) < 1e-6 * abs(got[1])
    ref.close()
