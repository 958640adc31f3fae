"""Seed-sharding of independent games over the GPUs of one node.

Games never interact, so the path shards with NO data-path collective: rank r owns a
contiguous block of global game ids and passes `game_offset` to the kernels, which
key Philox by the global id -- results are identical for any number of GPUs.  The
only cross-rank step is the host-side weighted mean of the tiny [episodes, N] logs,
done here over a CPU (gloo) process group.
"""
import numpy as np


def shard_range(total_games, rank, world_size):
    """(game_offset, n_local) of the contiguous block owned by `rank`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d out of range for world_size %d" % (rank, world_size))
    base, extra = divmod(int(total_games), int(world_size))
    n_local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, n_local


def aggregate_logs(local_log, n_local, group=None):
    """Mean-over-all-games log from per-rank mean logs: sum_r(n_r * log_r) / sum_r(n_r).
    Uses torch.distributed (gloo, CPU tensors) when initialised, else returns local_log."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local_log, np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    buf = torch.from_numpy(np.concatenate([(local * float(n_local)).ravel(), [float(n_local)]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = buf.numpy()
    return (out[:-1] / out[-1]).reshape(local.shape)
