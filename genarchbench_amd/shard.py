"""Sharding of the item-id range across ranks (one process per GPU) and the aggregation bench.py reports.

Every pair / call / read is independent, so the data path has NO collective.  Two modes:

* strong (the default of bench.py for N > 1; BASELINE.json configs[4]: "large inputs sharded across 8 x MI355X"): ONE fixed
  input of T items is split, rank r takes ids [r*T/N, (r+1)*T/N) -- what the reference does with its one input file under
  `omp for schedule(dynamic)` (bsw/src/main_banded.cpp:338-350, chain/src/host_kernel.cpp:98-105).  chain / fast-chain
  calls differ in size by three orders of magnitude, so they are sorted by descending anchor count and dealt longest first,
  each call to the rank with the least work so far ("one huge call can dominate, so sort calls by descending n
  first").
* weak: rank r owns [r*items, (r+1)*items) -- per-GPU work is fixed as N grows.

torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the barrier around the timed region, the MAX of
the elapsed time and the SUM of the processed units.
"""
import heapq
import os

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(rank, world, items_per_rank):
    """weak scaling: first id and count of this rank's shard"""
    assert 0 <= rank < world
    return rank * items_per_rank, items_per_rank


def shard_strong(rank, world, total):
    """strong scaling: first id and count of rank's share of the fixed id range [0, total)"""
    assert 0 <= rank < world and total >= 0
    first = rank * total // world
    return first, (rank + 1) * total // world - first


def deal_longest_first(sizes, world):
    """strong scaling of items of very different sizes (chain calls): ids sorted by descending size (ties: lower id
    first) and dealt one by one to the rank with the least work so far (ties: lower rank).  Returns one ascending id
    array per rank; together they tile range(len(sizes)).  Deterministic, so every rank computes the same deal."""
    sizes = np.asarray(sizes, dtype=np.int64)
    order = np.lexsort((np.arange(len(sizes)), -sizes))
    heap = [(0, r) for r in range(world)]
    mine = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        mine[r].append(int(i))
        heapq.heappush(heap, (load + int(sizes[i]), r))
    return [np.array(sorted(m), dtype=np.int64) for m in mine]


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
