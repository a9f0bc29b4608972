"""Collectives with autograd semantics for the data-parallel path (torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference has NO explicit collective (SURVEY: multi-GPU is Lightning DDP only), so its contrastive /
cross-modality heads only ever see the rank-local batch.  `all_gather_rows` is the new piece that lets
those heads see the global batch: forward = all-gather along dim 0 (one collective into a single buffer), backward =
reduce-scatter of the incoming gradient: each rank receives the rank-summed gradient of ITS rows only — 1/world of the
bytes an all-reduce would move (32 KB per rank for the CM latents; 8 MB per rank for node-level NT-Xent inputs).  With every rank evaluating
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
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
        # the path is chosen from the backend, never by catching an exception around a collective: a genuine RCCL failure
        # on one rank must surface, not move that rank to a different collective than its peers are in
        if dist.get_backend() == "gloo":
            # gloo has no reduce_scatter: same result through an all-reduce + slice (CPU tests, two ranks on one GPU)
            full = g.clone()
            dist.all_reduce(full, op=dist.ReduceOp.SUM)
            n, r = ctx.shape[0], dist.get_rank()
            out = full[r * n:(r + 1) * n].clone()
        else:
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        return out


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """(n, ...) on every rank -> (world * n, ...), differentiable.  Identity when not distributed."""
    return x if world_size() == 1 else _AllGatherRows.apply(x)


def all_gather_meta(meta: List[dict]) -> List[dict]:
    """Rank-ordered concatenation of the per-sample meta dicts (ids may be strings).  A host pickle collective: NOT on the
    training step since round 4 (the global-batch CM head gathers integer id codes as a tensor, id_codes / all_gather_codes);
    kept for tools and tests that want the records themselves."""
    if world_size() == 1:
        return list(meta)
    out = [None] * world_size()
    dist.all_gather_object(out, list(meta))
    return [m for part in out for m in part]


def id_code(v) -> int:
    """A stable 63-bit integer for an entity id (integers of ANY integral type — int, numpy.int64, pandas scalars — map to
    themselves, so `5` on one rank and `np.int64(5)` on another are one entity, as they are for the reference's dict keys:
    hash(np.int64(5)) == hash(5); everything else through blake2b of str(v) — equal ids give equal codes on every rank and
    in every process, unlike Python's salted hash(), and independent of the numpy version's repr)."""
    import numbers
    import operator
    if isinstance(v, numbers.Integral) and not isinstance(v, bool):
        v = operator.index(v)
        if 0 <= v < (1 << 62):
            return v
    elif isinstance(v, numbers.Real) and not isinstance(v, bool):
        import math
        f = float(v)
        # non-finite ids (a pandas missing id is NaN; int(nan) / int(inf) raise opaque errors) take the hash path below:
        # every missing id is then ONE entity named 'nan', on every rank
        if math.isfinite(f) and f == int(f) and 0 <= int(f) < (1 << 62):
            return int(f)                   # (1.0 == 1 as a dict key too)
    import hashlib
    h = hashlib.blake2b((v if isinstance(v, str) else str(v)).encode(), digest_size=8).digest()
    return (int.from_bytes(h, "little") & ((1 << 62) - 1)) | (1 << 62)       # disjoint from the small-int range


def id_codes(meta: List[dict]):
    """(n, 3) int64 numpy array: prot code, drug code, label (0 / 1) per sample — what the global-batch CM head gathers."""
    import numpy as np
    out = np.empty((len(meta), 3), dtype=np.int64)
    for t, m in enumerate(meta):
        out[t, 0], out[t, 1], out[t, 2] = id_code(m["Prot_ID"]), id_code(m["Drug_ID"]), int(m["Y"])
    return out


def all_gather_codes(codes: torch.Tensor) -> torch.Tensor:
    """(n, 3) int64 on every rank -> (world * n, 3), rank-ordered: ONE tensor collective (capturable on RCCL)."""
    if world_size() == 1:
        return codes
    out = torch.empty((world_size() * codes.shape[0], codes.shape[1]), dtype=codes.dtype, device=codes.device)
    dist.all_gather_into_tensor(out, codes.contiguous())
    return out
