"""ProteinCNN on distinct rows: the host-side plan (round 4).

The reference tiles every protein — L residues, L + 2 ESM-2 token rows with CLS / SEP — to PROTEIN.SEQ_LEN = 2304 positions
with period P = L + 2 (`repeat_integer_label_protein`, utils.py:392-412; `repeat_pad`, utils.py:314-324) and leaves zeros
behind the last whole period.  ProteinCNN (model/basic_model.py:155-180: three Conv1d 'same' with k = 3 / 6 / 9 + ReLU +
BatchNorm1d) has a receptive field of 7 positions to the left and 8 to the right, so its output at position t equals its
output at t - P whenever both windows lie inside the periodic region, and is one constant deep inside the zero tail: a
sample has about P + 31 DISTINCT output rows instead of 2304 (tools/cnn_tiling_probe.py counted them on the reference's
arithmetic; tools/cnn_compact_proto.py validated the weighted form in fp64).

The plan keeps, per sample, up to three contiguous SEGMENTS of positions so that the convolutions stay what they are in
the full layout — one GEMM over overlapping rows — and every distinct output row has its whole receptive field inside its
segment:
    A = [0, P + 14]          representatives 0 .. P + 6       (left edge of A = the real left boundary: zero padding)
    B = [E - 15, E + 15]     representatives E - 8 .. E + 7   (E = reps * P, the end of the periodic region; E + 7 stands
                                                               for the whole deep tail E + 7 .. S - 9)
    C = [S - 15, S - 1]      representatives S - 8 .. S - 1   (right edge of C = the real right boundary)
(B and C merge into [E - 15, S - 1] when the tail is short; a sample whose plan would not be smaller than S keeps all S
positions as one segment.)  Every segment is surrounded by HALO zero rows: real boundaries need them as the convolutions'
zero padding, cut edges just become deterministic.  Rows near a cut edge that are not representatives are CONTEXT rows:
computed, never used as outputs, weight 0 in the BatchNorm statistics.

Per compact row the plan records
    src   flat input index b * S + position, or -1 for a halo / padding row
    w     -1 halo row (kept at zero), 0 context row, m >= 1 representative of m positions (BatchNorm multiplicity)
    (first, stride, count)  the positions a representative stands for: first + k * stride, k < count (flat b * S + pos)
and per position the compact row that represents it (`row_of`).  Identical computation to the full layout: for ANY
parameter values the compact network's representative rows equal the full network's rows at those positions, so outputs,
BatchNorm statistics (weighted) and all parameter gradients are those of the reference's computation.

The tables depend on (L, S) only.  Whether a batch really has this structure is verified ON THE DEVICE by the kernel
that builds the compact input (dl_embed_rows: ids and fill bits periodic with period P up to E, constant behind E); a
violation sets a sticky flag word that the trainer polls (Trainer.check_device_flags).

Round 5: on the GPU the tables are BUILT ON THE DEVICE (csrc/compact.hip, dl_protein_plan_build) from the batch's B residue
counts — what the training thread does per batch is `PlanSpec` (a vectorised row count, microseconds) and a B x 4 byte
copy; round 4 built ~170 k table rows in per-sample numpy calls (8-14 ms at batch 256) and copied 5.8 MB per
batch.  `ProteinPlan` below remains the host statement of the same tables: the CPU tests run the network through it, and the
GPU tests compare the device-built tables with it entry by entry.  Row capacities: eager steps round the needed rows up to
ROW_BUCKET; captured graphs use geometric size classes (`row_class`, ratio ~1.25) so that batches of varying lengths share
one graph instead of one per 2048-row bucket.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np

HALO = 4            # zero rows around every segment (max 'same' padding of k = 9); equals functional._CNN_HALO
RF_LEFT, RF_RIGHT = 7, 8          # receptive field of the three convolutions: 1 + 2 + 4 left, 1 + 3 + 4 right
ROW_BUCKET = 2048   # compact row counts are rounded up to a multiple of this (few distinct shapes for captured graphs)

_template_cache: Dict[Tuple[int, int], tuple] = {}


def sample_rows(lengths, S: int) -> np.ndarray:
    """Compact rows (halo rows included) of every sample, vectorised — the same case split as `_segments`."""
    L = np.asarray(lengths, dtype=np.int64).reshape(-1)
    P = L + 2
    reps = np.where(P > 0, S // np.maximum(P, 1), 0)
    E = reps * P
    plain = (reps < 2) | (P + 2 * RF_LEFT + 2 * RF_RIGHT + 40 >= S) | (P + RF_LEFT - 1 + RF_RIGHT >= E - RF_RIGHT - RF_LEFT)
    sep = (S - E) > 2 * (RF_LEFT + RF_RIGHT) + 2
    return np.where(plain, S + 2 * HALO, np.where(sep, P + 85, P + S - E + 46))


def row_class(rows: int, unit: int = ROW_BUCKET) -> int:
    """Row capacity of a captured graph's tables: the next value of a geometric ladder of `unit` multiples (ratio ~1.25 once the
    integer step exceeds one: 2048 x {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 15, 18, 22, 27, 33, 41, ...} — identical to the plain
    2048-row bucket up to 16 k rows).  Batches whose random lengths differ by a few buckets land in
    one class; a graph captured for a class serves every batch that needs at most that many rows."""
    k = -(-max(int(rows), 1) // unit)
    c = 1
    while c < k:
        c = max(c + 1, int(c * 1.25))
    return c * unit


class PlanSpec:
    """What the training thread knows about a batch's ProteinCNN layout without building any table: the residue counts, the
    rows the compact layout needs (bucket padding excluded) and whether it pays."""
    __slots__ = ("lengths", "S", "B", "need", "n")

    def __init__(self, lengths: Sequence[int], S: int):
        self.lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int32).reshape(-1))
        self.S, self.B = int(S), int(self.lengths.shape[0])
        self.need = int(sample_rows(self.lengths, self.S).sum())
        self.n = self.B * self.S

    def rows(self, bucket: int = ROW_BUCKET) -> int:
        return -(-max(self.need, 1) // bucket) * bucket

    def pays(self, rows: int = 0) -> bool:
        """Whether the compact layout (at `rows` capacity) is meaningfully smaller than the padded full one."""
        return (rows or self.rows()) * 4 <= self.B * (self.S + 2 * HALO) * 3


def _segments(P: int, S: int):
    """[(first position, last position, [(rep_first, rep_last)])] of one sample; None when compaction does not pay."""
    reps = S // P if P > 0 else 0
    E = reps * P
    if reps < 2 or P + 2 * RF_LEFT + 2 * RF_RIGHT + 40 >= S:
        return None
    a_hi = P + RF_LEFT - 1                      # last representative of A: positions [P + 7, E - 9] are copies
    if a_hi + RF_RIGHT >= E - RF_RIGHT - RF_LEFT:     # A would run into B: no room for copies
        return None
    segs = [(0, a_hi + RF_RIGHT, 0, a_hi)]
    b_lo = E - RF_RIGHT                          # first position whose window leaves the periodic region
    if S - E > 2 * (RF_LEFT + RF_RIGHT) + 2:
        segs.append((b_lo - RF_LEFT, E + RF_LEFT + RF_RIGHT, b_lo, E + RF_LEFT))        # B: E + 7 = the deep-tail rep
        segs.append((S - RF_RIGHT - RF_LEFT, S - 1, S - RF_RIGHT, S - 1))               # C
    else:
        segs.append((b_lo - RF_LEFT, S - 1, b_lo, S - 1))                                # B and C merged
    return segs


def sample_template(L: int, S: int):
    """Tables of ONE sample with L residues (period P = L + 2) in a sequence of S positions, local indices:
    (src [R] position or -1, w [R], first [R], stride [R], count [R], row_of [S])."""
    key = (int(L), int(S))
    t = _template_cache.get(key)
    if t is not None:
        return t
    P = L + 2
    segs = _segments(P, S)
    if segs is None:                                           # plain layout: every position is its own representative
        R = S + 2 * HALO
        src = np.full(R, -1, np.int32)
        src[HALO:HALO + S] = np.arange(S, dtype=np.int32)
        w = np.full(R, -1.0, np.float32)
        w[HALO:HALO + S] = 1.0
        first = np.where(src >= 0, src, 0).astype(np.int32)
        t = (src, w, first, np.ones(R, np.int32), (src >= 0).astype(np.int32), (np.arange(S, dtype=np.int32) + HALO))
        _template_cache[key] = t
        return t
    reps = S // P
    E = reps * P
    src_l, w_l, first_l, stride_l, count_l = [], [], [], [], []
    row_of = np.full(S, -1, np.int64)
    row = 0
    for (lo, hi, r_lo, r_hi) in segs:
        n = hi - lo + 1
        pos = np.arange(lo, hi + 1, dtype=np.int64)
        src = np.concatenate([np.full(HALO, -1), pos, np.full(HALO, -1)])
        w = np.zeros(n, np.float64)
        first = pos.copy()
        stride = np.ones(n, np.int64)
        count = np.zeros(n, np.int64)
        is_rep = (pos >= r_lo) & (pos <= r_hi)
        count[is_rep] = 1
        if lo == 0:                                            # segment A: representative t >= 7 also stands for t + k P <= E - 9
            per = is_rep & (pos >= RF_LEFT)
            k = (E - RF_RIGHT - 1 - pos[per]) // P + 1          # number of k >= 0 with t + k P <= E - 9
            count[per] = np.maximum(k, 1)
            stride[per] = P
        else:
            deep = pos == E + RF_LEFT                           # the deep-tail representative (segment B only)
            if (hi < S - 1) and deep.any():
                count[deep] = max(S - RF_RIGHT - (E + RF_LEFT), 1)       # positions E + 7 .. S - 9
        w[:] = count
        base = row + HALO
        for j in np.nonzero(is_rep)[0]:
            cnt, st, f = int(count[j]), int(stride[j]), int(first[j])
            row_of[f:f + cnt * st:st] = base + j
        src_l.append(src)
        w_l.append(np.concatenate([np.full(HALO, -1.0), w, np.full(HALO, -1.0)]))
        first_l.append(np.concatenate([np.zeros(HALO, np.int64), first, np.zeros(HALO, np.int64)]))
        stride_l.append(np.concatenate([np.ones(HALO, np.int64), stride, np.ones(HALO, np.int64)]))
        count_l.append(np.concatenate([np.zeros(HALO, np.int64), count, np.zeros(HALO, np.int64)]))
        row += n + 2 * HALO
    if (row_of < 0).any():
        raise AssertionError("protein_plan: positions without a representative (L = %d, S = %d)" % (L, S))
    t = (np.concatenate(src_l).astype(np.int32), np.concatenate(w_l).astype(np.float32), np.concatenate(first_l).astype(np.int32),
         np.concatenate(stride_l).astype(np.int32), np.concatenate(count_l).astype(np.int32), row_of.astype(np.int32))
    _template_cache[key] = t
    if len(_template_cache) > 8192:
        _template_cache.clear()
    return t


class ProteinPlan:
    """Batch tables (numpy, host).  rows: compact rows incl. halo and bucket padding; n: the BatchNorm row count B * S."""

    __slots__ = ("lengths", "S", "B", "rows", "src", "w", "rep", "row_of", "period", "n", "key")

    def __init__(self, lengths: Sequence[int], S: int, bucket: int = ROW_BUCKET):
        self.lengths = tuple(int(v) for v in lengths)
        self.S, self.B = int(S), len(self.lengths)
        srcs, ws, reps, maps, off = [], [], [], [], 0
        for b, L in enumerate(self.lengths):
            src, w, first, stride, count, row_of = sample_template(L, S)
            base = b * S
            srcs.append(np.where(src >= 0, src + base, -1))
            ws.append(w)
            reps.append(np.stack([first + base, stride, count], axis=1))
            maps.append(row_of + off)
            off += src.shape[0]
        rows = -(-max(off, 1) // bucket) * bucket
        pad = rows - off
        self.rows = rows
        self.src = np.concatenate(srcs + [np.full(pad, -1, np.int32)]).astype(np.int32)
        self.w = np.concatenate(ws + [np.full(pad, -1.0, np.float32)]).astype(np.float32)
        rep = np.concatenate(reps + [np.zeros((pad, 3), np.int64)]).astype(np.int32)
        rep[:, 1] = np.maximum(rep[:, 1], 1)
        self.rep = np.ascontiguousarray(rep)
        self.row_of = np.concatenate(maps).astype(np.int32)
        # period 0 = "plain layout, no periodic claim" (the device guard then checks nothing for the sample: a protein whose
        # period exceeds the sequence, reps = 0, is legal and keeps every position — ADVICE r4)
        self.period = np.array([0 if _segments(L + 2, self.S) is None else L + 2 for L in self.lengths], dtype=np.int32)
        self.n = self.B * self.S
        self.key = (self.rows, self.B, self.S)

    def pays(self) -> bool:
        """Whether the compact layout is meaningfully smaller than the padded full one."""
        return self.rows * 4 <= self.B * (self.S + 2 * HALO) * 3


class PlanDev:
    """Row tables on the device: ONE buffer, five views (src, w, rep, row_of, period) + the residue counts, with a fixed row
    capacity.  `fill(spec)` makes them the tables of a batch: on the GPU a B x 4 byte copy from a pinned ring and one
    dl_protein_plan_build launch on the current stream (a captured graph keeps pointing at the same tensors; the refill is
    stream-ordered behind the previous step's kernels); on the CPU (tests) the host tables of `ProteinPlan`.
    key = (rows, B, S) is what shapes — and captured graphs — depend on."""

    def __init__(self, plan, device, rows: int = 0):
        """plan: a PlanSpec (rows = capacity, default: the spec's rows rounded up to ROW_BUCKET) or a ProteinPlan (its rows)."""
        import torch
        if isinstance(plan, ProteinPlan):
            rows = rows or plan.rows
            plan = PlanSpec(plan.lengths, plan.S)
        R, B, S = int(rows or plan.rows()), plan.B, plan.S
        if plan.need > R:
            raise ValueError("PlanDev: %d rows do not hold a batch that needs %d" % (R, plan.need))
        self.key = (R, B, S)
        self.rows, self.B, self.S, self.n = R, B, S, plan.n
        o_w, o_rep, o_map, o_per = 4 * R, 8 * R, 20 * R, 20 * R + 4 * B * S
        o_len = o_per + 4 * B
        self.nbytes = (o_len + 4 * B + 15) // 16 * 16
        self._off = (o_w, o_rep, o_map, o_per, o_len)
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.src = self.buf[:o_w].view(torch.int32)
        self.w = self.buf[o_w:o_rep].view(torch.float32)
        self.rep = self.buf[o_rep:o_map].view(torch.int32).view(R, 3)
        self.row_of = self.buf[o_map:o_per].view(torch.int32)
        self.period = self.buf[o_per:o_len].view(torch.int32)
        self.len_dev = self.buf[o_len:o_len + 4 * B].view(torch.int32)
        self.on_gpu = torch.device(device).type == "cuda"
        self._pins = [torch.zeros(B, dtype=torch.int32).pin_memory() for _ in range(4)] if self.on_gpu else []
        self._events = [None] * len(self._pins)
        self._slot = 0
        self.lengths = None
        self.fill(plan)

    def fill(self, plan):
        import torch
        if isinstance(plan, ProteinPlan):
            plan = PlanSpec(plan.lengths, plan.S)
        if (plan.B, plan.S) != (self.B, self.S) or plan.need > self.rows:
            raise ValueError("PlanDev: a batch of shape (%d, %d) needing %d rows does not fit tables of shape %s"
                             % (plan.B, plan.S, plan.need, self.key))
        if self.lengths is not None and np.array_equal(plan.lengths, self.lengths):
            return self
        if not self.on_gpu:
            host = ProteinPlan(plan.lengths, self.S, bucket=1)
            pad = self.rows - host.rows
            self.src.copy_(torch.from_numpy(np.concatenate([host.src, np.full(pad, -1, np.int32)])))
            self.w.copy_(torch.from_numpy(np.concatenate([host.w, np.full(pad, -1.0, np.float32)])))
            prep = np.zeros((pad, 3), np.int32)
            prep[:, 1] = 1
            self.rep.copy_(torch.from_numpy(np.concatenate([host.rep, prep])))
            self.row_of.copy_(torch.from_numpy(host.row_of))
            self.period.copy_(torch.from_numpy(host.period))
            self.len_dev.copy_(torch.from_numpy(plan.lengths))
        else:
            from . import ops
            k = self._slot
            self._slot = (k + 1) % len(self._pins)
            if self._events[k] is not None:
                self._events[k].synchronize()
            self._pins[k].numpy()[:] = plan.lengths
            self.len_dev.copy_(self._pins[k], non_blocking=True)
            self._events[k] = torch.cuda.Event()
            self._events[k].record()
            ops.protein_plan_build(self)
        self.lengths = plan.lengths.copy()
        return self


def plan_of(lengths: Sequence[int], S: int):
    """The PlanSpec of a batch, or None when the compact layout would not be meaningfully smaller."""
    p = PlanSpec(lengths, S)
    return p if p.pays() else None


class BatchHints:
    """Host-side knowledge about a batch that its tensors do not carry, handed to the model's forward as an explicit argument
    (`model(vd, vp, xd, xp, hints=...)`; round 3 kept it in a module global):
      drug_tokens   a block size (multiple of 128) that covers every molecule's ChemBERTa token count — the rows beyond it
                    are identical zero rows, computed once (basic_model._llm_adaptors)
      protein_plan  PlanDev tables of the ProteinCNN compact layout (from the proteins' residue counts)
      branch_streams  see __init__
    The first two are verified on the device (ops.guard_flags)."""
    __slots__ = ("drug_tokens", "protein_plan", "branch_streams", "raw_attention")

    def __init__(self, drug_tokens: int = 0, protein_plan=None, branch_streams: bool = False, raw_attention: bool = True):
        self.raw_attention = bool(raw_attention)      # False: the PGCA raw-logit maps (A_v_gca / A_x_gca) are not produced
        self.drug_tokens = int(drug_tokens or 0)
        self.protein_plan = protein_plan
        # run the forward's independent branches on side HIP streams (model/DrugLAMP.py): the trainer asks for it on the
        # steps where it was measured to pay (cls steps: 13.8 -> 13.3 ms at batch 256, 4.0 -> 3.6 at 32; neutral on SSL
        # epochs; a LOSS on steps whose only backward is the cross-modality head's: 8.9 -> 10.7 ms)
        self.branch_streams = bool(branch_streams)

    def key(self) -> tuple:
        """What a captured graph is keyed by (by-value knowledge of the capture)."""
        return (self.drug_tokens, None if self.protein_plan is None else self.protein_plan.key)


def lengths_from_codes(vp) -> list:
    """Residue counts recovered from tiled residue codes (B, S) — position 0 is the CLS slot (0), the L codes follow, the
    SEP slot (0) ends the first period (utils.py:392-412).  For callers without `Prot_Len` records; exact when no residue
    inside the sequence was coded 0 (an unknown letter) — and the device-side guard rejects a wrong period anyway."""
    a = np.asarray(vp.detach().cpu().numpy() if hasattr(vp, "detach") else vp)
    out = []
    for row in a:
        z = np.nonzero(row[1:] == 0)[0]
        out.append(int(z[0]) if z.size else int(row.shape[0]) - 2)
    return out
