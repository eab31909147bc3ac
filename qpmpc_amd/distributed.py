"""Multi-GPU: a batch of independent MPC problems split across ranks.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on
ROCm; ``gloo`` in CPU tests). Problems never interact, so the data path has NO
collective: each rank builds and solves a contiguous slice of the batch. RCCL is
used only (a) to gather results when the caller wants the whole batch on every
rank and (b) to reduce a handful of statistics. The reference has no
counterpart (it is single-process); the per-problem semantics are those of
qpmpc/solve_mpc.py:42-44.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

_PER_PROBLEM_KEYS = ("A", "B", "C", "D", "e", "x0", "goal", "targets")
_BLOCK_NDIM = {"A": 2, "B": 2, "C": 2, "D": 2, "e": 1}


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [start, stop) of ``total`` items for ``rank``:
    the first ``total % world`` ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_workload(w: Dict, rank: int, world: int) -> Dict:
    """Slice the per-problem arrays of a workload dict; shared operands are kept."""
    batch = int(np.asarray(w["x0"]).shape[0])
    lo, hi = shard_range(batch, rank, world)
    out = dict(w)
    for key in _PER_PROBLEM_KEYS:
        a = w.get(key)
        if a is None:
            continue
        a = np.asarray(a)
        if key in _BLOCK_NDIM:
            per_problem = a.ndim == _BLOCK_NDIM[key] + 2 and a.shape[0] == batch and batch > 1
        else:
            per_problem = a.ndim == 2 and a.shape[0] == batch
        out[key] = a[lo:hi] if per_problem else a
    return out


def _dist():
    import torch.distributed as dist

    return dist


def gather_batch(local, total: int, group=None):
    """all_gather of per-problem rows (e.g. U [B_local, n] or status [B_local]) into
    the full batch, in rank order. Shards may differ by one row; they are padded
    to the largest shard for the collective and trimmed afterwards."""
    import torch

    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def reduce_stats(status, iters, group=None) -> Dict[str, float]:
    """Whole-job counts from per-rank status/iters tensors: one all_reduce(SUM) of
    four scalars and one all_reduce(MAX)."""
    import torch

    dist = _dist()
    sums = torch.stack([
        (status == 0).sum(), (status == 2).sum(), (status != 0).sum(), iters.sum(),
    ]).to(torch.float64)
    mx = iters.max().to(torch.float64).reshape(1) if iters.numel() else torch.zeros(1, dtype=torch.float64, device=status.device)
    count = torch.tensor([float(status.numel())], dtype=torch.float64, device=status.device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    total = float(count.item())
    return {
        "problems": total,
        "solved": float(sums[0].item()),
        "infeasible": float(sums[1].item()),
        "failed": float(sums[2].item()),
        "mean_iters": float(sums[3].item()) / max(total, 1.0),
        "max_iters": float(mx.item()),
    }


def solve_workload_sharded(w: Dict, gather: bool = True, dtype=None, group=None,
                           rank: Optional[int] = None, world: Optional[int] = None):
    """Solve this rank's slice of a workload on its GPU and (optionally) all_gather
    the inputs U and statuses. Returns (U, status, stats)."""
    from .batch import solve_mpc_batch
    from .workloads import to_batch_problem

    dist = _dist()
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    total = int(np.asarray(w["x0"]).shape[0])
    plan = solve_mpc_batch(to_batch_problem(shard_workload(w, rank, world), dtype=dtype))
    stats = reduce_stats(plan.status, plan.iters, group)
    if gather:
        return gather_batch(plan.U, total, group), gather_batch(plan.status, total, group), stats
    return plan.U, plan.status, stats
