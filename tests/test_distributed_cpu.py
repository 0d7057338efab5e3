"""The N>1 path on CPU: two gloo ranks each step their shard (the oracle stands in for the
engine here -- this test is about the sharding arithmetic and the statistics all-reduce, which
are the same code bench.py runs over RCCL) and must reproduce the unsharded run exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, steps, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from gym_amd.distributed import allreduce_stats, shard
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    base, n = shard(total, rank, world)
    v = O.OracleVec(O.CARTPOLE, n, seed=42, flags=O.AUTORESET | O.RECORD_STATS, env_index_base=base)
    v.reset()
    for _ in range(steps):
        v.step_random()
    merged = allreduce_stats(v.stats(), dist)
    np.save(os.path.join(out_dir, "state_%d.npy" % rank), v.get_state()[0])
    if rank == 0:
        np.save(os.path.join(out_dir, "stats.npy"), np.array([merged["episodes"], merged["sum_length"], merged["sum_return"],
                                                               merged["sum_return_sq"], merged["min_return"], merged["max_return"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_everything():
    from gym_amd.distributed import shard

    for total, world in [(10, 3), (16, 8), (7, 2), (5, 8), (1 << 24, 8)]:
        spans = [shard(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == total
        for (b0, n0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + n0 == b1


def test_two_ranks_match_one(tmp_path, oracle):
    total, steps, world = 1001, 60, 2
    mp.spawn(_worker, args=(world, _free_port(), total, steps, str(tmp_path)), nprocs=world, join=True)
    ref = oracle.OracleVec(oracle.CARTPOLE, total, seed=42, flags=oracle.AUTORESET | oracle.RECORD_STATS)
    ref.reset()
    for _ in range(steps):
        ref.step_random()
    state = np.concatenate([np.load(tmp_path / ("state_%d.npy" % r)) for r in range(world)], axis=1)
    assert np.array_equal(state.view(np.uint32), ref.get_state()[0].view(np.uint32))
    s = ref.stats()
    got = np.load(tmp_path / "stats.npy")
    assert got.tolist() == [s.episodes, s.sum_length, s.sum_return, s.sum_return_sq, s.min_return, s.max_return]
