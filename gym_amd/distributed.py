"""Multi-GPU sharding of an env batch: one process per GPU, no data-path collective.

Env i of the whole job lives on rank i // envs_per_rank; the counter RNG is keyed by the global
env index (gx_desc.env_index_base), so results do not depend on the number of ranks.  The only
exchange is the all-reduce of the episode-statistics record (sum for the counts and sums, min /
max for the extrema), over whatever backend torch.distributed was initialised with ("nccl" is
RCCL on ROCm; "gloo" in the CPU tests).
"""
import math


def shard(total_envs, rank, world):
    """-> (env_index_base, num_envs) of this rank; the first `total % world` ranks take one more"""
    q, r = divmod(int(total_envs), int(world))
    n = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, n


def stats_to_list(s):
    return [float(s.episodes), float(s.sum_length), float(s.sum_return), float(s.sum_return_sq),
            -float(s.min_return), float(s.max_return)]


def allreduce_stats(s, dist, device="cpu"):
    """All-reduce one statistics record (any object with the gx_stats fields) across ranks.
    Counts are exact in float64 below 2^53; min is reduced as the max of the negation.
    Returns a dict."""
    import torch

    v = stats_to_list(s)
    sums = torch.tensor(v[:4], dtype=torch.float64, device=device)
    ext = torch.tensor(v[4:], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(ext, op=dist.ReduceOp.MAX)
    sums, ext = sums.tolist(), ext.tolist()
    n = int(sums[0])
    return {"episodes": n, "sum_length": int(sums[1]), "sum_return": sums[2], "sum_return_sq": sums[3],
            "min_return": -ext[0] if n else math.inf, "max_return": ext[1] if n else -math.inf}
