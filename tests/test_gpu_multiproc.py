"""The N > 1 path on the engine itself: two processes, each with its own HIP context and its shard
of one global batch in HBM (both on GPU 0 -- the test box has one GPU -- with gloo carrying the
statistics all-reduce, because RCCL refuses two ranks on one device).  The sharded run must equal
the unsharded one bit for bit: states, and the merged RecordEpisodeStatistics record.  This is the
libgymx twin of tests/test_distributed_cpu.py, and the same code path `bench.py --gpus N` runs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, env_id, total, steps, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from gym_amd import _lib as L
    from gym_amd.distributed import allreduce_stats, shard
    from gym_amd.engine import Context, DeviceBatch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, n = shard(total, rank, world)
    ctx = Context(0)
    v = DeviceBatch(ctx, env_id, n, seed=42, flags=L.AUTORESET | L.RECORD_STATS, env_index_base=base)
    v.reset()
    for _ in range(steps):
        v.step_random()
    v.rollout(7)  # and a fused rollout on top
    v.wait()
    merged = allreduce_stats(v.stats(), dist)
    np.save(os.path.join(out_dir, "state_%d.npy" % rank), v.get_state_raw())
    if rank == 0:
        json.dump(merged, open(os.path.join(out_dir, "stats.json"), "w"))
    dist.barrier()
    v.close()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("env_name", ["CARTPOLE", "FROZENLAKE"])
def test_two_gpu_processes_match_one(tmp_path, oracle, env_name):
    from gym_amd import _lib as L
    from gym_amd.engine import DeviceBatch
    from gpu_common import ctx

    env_id = getattr(L, "ENV_" + env_name)
    total, steps, world = 20_001, 60, 2
    mp.spawn(_worker, args=(world, _free_port(), env_id, total, steps, str(tmp_path)), nprocs=world, join=True)
    flags = L.AUTORESET | L.RECORD_STATS
    ref = DeviceBatch(ctx(), env_id, total, seed=42, flags=flags)
    ref.reset()
    for _ in range(steps):
        ref.step_random()
    ref.rollout(7)
    ref.wait()
    reals, ints = ref.get_state()
    parts = []
    for r in range(world):
        n_r = total // world + (1 if r < total % world else 0)
        raw = np.load(tmp_path / ("state_%d.npy" % r))
        nb = n_r * ref.info.n_real * 4
        parts.append((raw[:nb].view(np.float32).reshape(ref.info.n_real, n_r), raw[nb:].view(np.int32).reshape(ref.info.n_int, n_r)))
    assert np.array_equal(np.concatenate([p[0] for p in parts], axis=1).view(np.uint32), reals.view(np.uint32))
    assert np.array_equal(np.concatenate([p[1] for p in parts], axis=1), ints)
    s = ref.stats()
    got = json.load(open(tmp_path / "stats.json"))
    assert got["episodes"] == s.episodes > 0 and got["sum_length"] == s.sum_length
    assert got["sum_return"] == s.sum_return and got["sum_return_sq"] == s.sum_return_sq
    assert got["min_return"] == s.min_return and got["max_return"] == s.max_return
    # and the unsharded engine run is the oracle's
    o = oracle.OracleVec(env_id, total, seed=42, flags=flags)
    o.reset()
    for _ in range(steps + 7):
        o.step_random()
    orr, oi = o.get_state()
    assert np.array_equal(orr.view(np.uint32), reals.view(np.uint32)) and np.array_equal(oi, ints)
    os_ = o.stats()
    assert (os_.episodes, os_.sum_length, os_.sum_return) == (s.episodes, s.sum_length, s.sum_return)
    ref.close()


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-extra",
                        "--no-cpu-baseline"] + list(args), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks, one line, whole-job numbers"""
    two = _bench("--gpus", "2", "--share-device", "--dist-backend", "gloo", "--num-envs", "65536")
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert abs(two["value"] - 2 * 65536 * 6 / (two["ms_per_step"] * 6e-3)) / two["value"] < 1e-6
    # the same 131072 envs on one rank: the merged statistics are identical (RNG keyed by global env index)
    one = _bench("--gpus", "1", "--num-envs", "131072")
    assert one["n_gpus"] == 1
    assert two["episode_stats"] == one["episode_stats"] and two["episode_stats"]["episodes"] > 100
