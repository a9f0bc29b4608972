"""Collectives with autograd semantics for the data-parallel path (torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has NO explicit collective (SURVEY: multi-GPU is Lightning DDP only), so its contrastive /
cross-modality heads only ever see the rank-local batch.  `all_gather_rows` is the new piece that lets
those heads see the global batch: forward = all-gather along dim 0, backward = sum over ranks of the
incoming gradient, sliced back to the local rows (reduce-scatter semantics).  With every rank evaluating
the same global loss, the later 1/world gradient averaging of the DP step then yields exactly the
gradient of that global loss for backbone and head parameters alike."""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        world = dist.get_world_size()
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x)
        ctx.n = x.shape[0]
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        r = dist.get_rank()
        return g[r * ctx.n:(r + 1) * ctx.n]


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """(n, ...) on every rank -> (world * n, ...), differentiable.  Identity when not distributed."""
    return x if world_size() == 1 else _AllGatherRows.apply(x)


def all_gather_meta(meta: List[dict]) -> List[dict]:
    """Rank-ordered concatenation of the per-sample meta dicts (ids may be strings)."""
    if world_size() == 1:
        return list(meta)
    out = [None] * world_size()
    dist.all_gather_object(out, list(meta))
    return [m for part in out for m in part]
