"""Sharding of the item-id range across ranks (one process per GPU) and the aggregation bench.py reports.

Every pair / call / read is independent, so the data path has NO collective: rank r owns the
contiguous id range [r * items, (r + 1) * items) (weak scaling: per-GPU work is fixed).  torch.distributed
(RCCL on GPUs, gloo in the CPU tests) is used only for the barrier around the timed region, the MAX of the
elapsed time and the SUM of the processed units.
"""
import os


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(rank, world, items_per_rank):
    """first id and count of this rank's shard"""
    assert 0 <= rank < world
    return rank * items_per_rank, items_per_rank


def aggregate(elapsed_s, units, dist=None, device=None):
    """whole-job figures: (max elapsed over ranks, sum of units over ranks)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
