"""Data-parallel training loop (reference: trainer.py:39-292 + main.py:155-168), one process per GPU.

What is restated from the reference, in its order (trainer.py:179-231):
    forward -> opt.zero_grad -> cls backward -> [opt_ssl.zero_grad -> ssl forward/backward]
            -> [opt_cm.zero_grad -> cm forward (+ one-time weight auto-scale) -> backward]
            -> opt.step -> [opt_ssl.step] -> [opt_cm.step]
All three AdamW instances own the SAME parameter list (main.py:158-160), so every zero_grad wipes the
previous loss's gradients and each optimiser steps on whatever gradient is present last — kept as is.
Parameters whose gradient is None are skipped by torch.optim.AdamW; here too (per-parameter step
counts, contiguous "runs" of parameters with gradients).

What is new (MI355X-native):
  * parameters live in ONE flat fp32 arena; gradients are packed into a flat buffer by a single
    multi-tensor copy; each optimiser step is a handful of fused dl_adamw_step launches over
    contiguous runs instead of ~250 per-tensor updates;
  * gradient all-reduce = RCCL all-reduce (sum; the 1/world factor is folded into the AdamW kernel's
    grad_scale) on the flat buffer after backward, or (DL_GRAD_OVERLAP=1) on 8 MB buckets started from
    inside backward as soon as a bucket's gradients exist (GradOverlap), instead of Lightning's DDP wrapper.  Unlike the reference's DDP
    (which only reduces the cls backward that ran under the wrapper), the gradients the optimisers
    consume are reduced whichever loss produced them, so replicas never drift;
  * a backward pass whose gradients the next zero_grad wipes before any optimiser steps (cls on SSL / CM
    steps, ssl on CM steps) is not run: the parameters after the step are the same, the step is shorter.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import functional as Fn
from . import ops
from .model.basic_model import binary_cross_entropy, cross_entropy_logits


class CosineAnnealingWarmupRestarts:
    """scheduler/cosine_annealing_warmup.py:5-88 as plain arithmetic (cycle_mult = 1, gamma = 1): lr starts
    at min_lr; the k-th step() gives linear warm-up for k < warmup_steps, then half-cosine to min_lr."""

    def __init__(self, first_cycle_steps: int, max_lr: float, min_lr: float = 1e-8, warmup_steps: int = 0):
        assert warmup_steps < first_cycle_steps
        self.first_cycle_steps, self.max_lr, self.min_lr, self.warmup_steps = first_cycle_steps, max_lr, min_lr, warmup_steps
        self.step_in_cycle = 0
        self.cycle = 0
        self.lr = min_lr

    def step(self) -> float:
        self.step_in_cycle += 1
        if self.step_in_cycle >= self.first_cycle_steps:
            self.cycle += 1
            self.step_in_cycle -= self.first_cycle_steps
        s = self.step_in_cycle
        if s < self.warmup_steps:
            self.lr = (self.max_lr - self.min_lr) * s / self.warmup_steps + self.min_lr
        else:
            self.lr = self.min_lr + (self.max_lr - self.min_lr) * (
                1 + math.cos(math.pi * (s - self.warmup_steps) / (self.first_cycle_steps - self.warmup_steps))) / 2
        return self.lr


class FlatParams:
    """All optimised parameters as views into one fp32 arena (+ a same-layout gradient buffer)."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = list(params)
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.numel = n
        self.arena = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.arena[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self.grad_views = [self.grads[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)]

    def pack_grads(self) -> List[int]:
        """Copy the present .grad tensors into the flat buffer; returns the indices that had one."""
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if idx:
            torch._foreach_copy_([self.grad_views[i] for i in idx], [self.params[i].grad for i in idx])
        return idx

    def runs(self, idx: List[int], key) -> List[tuple]:
        """Merge adjacent parameter indices with equal key(i) into (start_elem, end_elem, first_index)."""
        out = []
        for i in idx:
            s, e = self.offsets[i], self.offsets[i] + (self.params[i].numel() + 3) // 4 * 4
            if out and out[-1][1] == s and key(out[-1][2]) == key(i) and out[-1][3] + 1 == i:
                out[-1] = (out[-1][0], e, out[-1][2], i)
            else:
                out.append((s, e, i, i))
        return [(s, e, first) for s, e, first, _ in out]


class GradOverlap:
    """Gradient all-reduce overlapped with the backward pass that produces the gradients (world > 1).

    The flat gradient buffer is cut into buckets of consecutive parameters, last parameter first (the
    order backward roughly finishes them in).  A post-accumulate hook on every parameter counts its
    bucket down; a complete bucket is packed into the flat buffer (one multi-tensor copy) and an
    asynchronous all-reduce (sum) is started on each of its contiguous runs.  Buckets go out in strict
    index order (a complete bucket waits for its predecessors), and `finish()` sends the rest after
    backward, so every rank issues the same collectives in the same order whatever the timing.

    Which parameters a backward pass of a `kind` ("cls" / "ssl" / "cm") reaches is agreed on by all
    ranks during the first pass of that kind (one small all-reduce of a presence mask, the only host
    sync; that pass reduces after backward).  Later passes reduce exactly the agreed set: a parameter
    of the set without a local gradient contributes zeros, a gradient outside the set is an error."""

    def __init__(self, flat: FlatParams, bucket_bytes: int = 8 << 20):
        self.flat = flat
        self.bucket_elems = max(bucket_bytes // 4, 1)
        self.expected: Dict[str, frozenset] = {}
        self.armed: Optional[str] = None
        self.bucket_of: Dict[int, int] = {}
        self.buckets: List[List[int]] = []
        self.pending: List[int] = []
        self.ready: set = set()
        self.next_bucket = 0
        self.works: list = []
        self.reduced: set = set()          # parameter indices whose all-reduce has been started
        self.index = {id(p): i for i, p in enumerate(flat.params)}
        self.handles = [p.register_post_accumulate_grad_hook(self._hook) for p in flat.params if p.requires_grad]

    def _plan(self, kind: str):
        exp = sorted(self.expected[kind], reverse=True)
        self.buckets, self.bucket_of, cur, n = [], {}, [], 0
        for i in exp:
            cur.append(i)
            n += self.flat.params[i].numel()
            if n >= self.bucket_elems:
                self.buckets.append(sorted(cur))
                cur, n = [], 0
        if cur:
            self.buckets.append(sorted(cur))
        for b, idx in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self.pending = [len(b) for b in self.buckets]

    def arm(self, kind: str):
        """Call right before the backward pass whose gradients the optimisers will consume."""
        self.works, self.reduced, self.ready, self.next_bucket, self.armed = [], set(), set(), 0, kind
        if kind in self.expected:
            self._plan(kind)
        else:
            self.buckets, self.bucket_of, self.pending = [], {}, []

    def _hook(self, p):
        if self.armed is None:
            return
        b = self.bucket_of.get(self.index[id(p)])
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self.ready.add(b)
            while self.next_bucket in self.ready:
                self._launch(self.buckets[self.next_bucket])
                self.next_bucket += 1

    def _launch(self, idx: List[int]):
        fl = self.flat
        local = [i for i in idx if fl.params[i].grad is not None]
        absent = [i for i in idx if fl.params[i].grad is None]
        if local:
            torch._foreach_copy_([fl.grad_views[i] for i in local], [fl.params[i].grad for i in local])
        if absent:
            torch._foreach_zero_([fl.grad_views[i] for i in absent])
        for s, e, _ in fl.runs(idx, lambda i: 0):
            self.works.append(dist.all_reduce(fl.grads[s:e], op=dist.ReduceOp.SUM, async_op=True))
        self.reduced.update(idx)

    def _agree(self, have: List[int]) -> frozenset:
        """Union over ranks of the parameters that received a gradient."""
        mask = torch.zeros(len(self.flat.params), dtype=torch.int32)
        mask[have] = 1
        mask = mask.to(self.flat.grads.device)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX)
        return frozenset(mask.cpu().nonzero().flatten().tolist())

    def finish(self) -> List[int]:
        """After backward: send the buckets the hooks did not, wait for everything; returns the indices the
        optimisers step (the agreed set)."""
        kind, self.armed = self.armed, None
        have = [i for i, p in enumerate(self.flat.params) if p.grad is not None]
        if kind not in self.expected:
            self.expected[kind] = self._agree(have)
            self._plan(kind)
            self.next_bucket = 0
        else:
            extra = set(have) - self.expected[kind]
            if extra:
                raise RuntimeError("GradOverlap: %d parameters outside the set agreed for '%s' backward passes received a "
                                   "gradient (first: index %d); the graph changed between steps" % (len(extra), kind, min(extra)))
        for b in range(self.next_bucket, len(self.buckets)):
            self._launch(self.buckets[b])
        self.next_bucket = len(self.buckets)
        for w in self.works:
            w.wait()
        self.works = []
        return sorted(self.expected[kind])


class FusedAdamW:
    """torch.optim.AdamW semantics (lr, betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2) on a FlatParams."""

    def __init__(self, flat: FlatParams, lr: float, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.arena)
        self.exp_avg_sq = torch.zeros_like(flat.arena)
        self.steps = [0] * len(flat.params)

    def step(self, idx: List[int], grad_scale: float = 1.0):
        for s, e, first in self.flat.runs(idx, lambda i: self.steps[i]):
            ops.adamw_step(self.flat.arena[s:e], self.flat.grads[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e],
                           lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, weight_decay=self.wd,
                           step=self.steps[first] + 1, grad_scale=grad_scale)
        for i in idx:
            self.steps[i] += 1


class GraphedStep:
    """The cls-only part of a training step — forward, BCE, backward, gradient packing — captured ONCE per batch shape
    into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on a side stream; every libdruglamp_hip launch
    goes to torch's current stream, so the whole launch sequence — ~260 library launches + ~100 torch launches — lands
    in the graph) and replayed with one host call per step.  At 32-64 pairs per GPU the eager step is bound by the
    7-9 ms the host needs to enqueue it; the replay costs the host < 1 ms.

    What stays outside the graph, on purpose: the gradient all-reduce (RCCL) and the fused AdamW launches (their step
    counts and learning rates are by-value arguments that change from step to step).
    What makes a replay a NEW step although every launch argument is frozen:
      * inputs are copied into the graph's static input tensors (skipped when the caller hands over those very
        tensors, Trainer.static_batch — a loader can assemble the next batch in place);
      * dropout masks are keyed by (site seed + *device offset*): the first node of the graph bumps the offset
        (ops.seed_offset_tensor), forward and backward of one replay read the same value;
      * the weight images are refreshed from the fp32 masters by the dl_weight_prep launch at the head of the graph;
      * BatchNorm running statistics are updated by the in-graph dl_bn_finalize launches."""

    def __init__(self, trainer: "Trainer", batch, kind: str = "cls", meta=None, caps=None):
        """Capture only records (nothing executes): the caller has already run eager steps of this shape AND kind, so lazy
        allocations (the SimSiam projectors of the first SSL forward), weight-image tables and workspaces exist.
        kind "ssl" (round 3): a step of an SSL epoch without the CM head — forward, BCE (logged only: its backward is dead,
        the next zero_grad wipes it, trainer.py:196-212 of the reference), SSL heads, their backward, gradient packing.  The
        MLM mask draw (rand / topk / scatter on the device) is captured with the step: torch's device generator is
        graph-safe (its Philox offset is advanced per replay), so every replay draws fresh masks.
        kinds "cm" / "sslcm" (round 3): a step with the cross-modality head (and, on SSL epochs, the SSL forward whose loss
        is logged and whose backward is dead).  The head is shape-static (model/cross_modality.py: unique-row blocks padded
        to the batch, masked BatchNorm statistics, (B, B) label matrix with ignored padding); the batch's label matrix is
        built on the host and copied into this graph's CMLabels before every replay.  The margin of the triplet loss and
        Trainer.cm_weight are by-value arguments of the capture: the trainer keys its graphs by them."""
        self.kind = kind
        self.tr = trainer
        dev = trainer.device
        self.static = self._clone(batch)
        # this graph's own device tables (refilled per replay) at the CAPACITIES the capture is made for: caps = (drug-token
        # block, ProteinCNN table rows), by-value knowledge of the capture — any batch that needs at most that much replays it
        self.hints = trainer.hints_of(meta, batch, own_tables=True, kind=kind, caps=caps)
        self.block = self.hints.drug_tokens                    # 0 = every drug-token row computed (serves any batch)
        self.rows_cap = None if self.hints.protein_plan is None else self.hints.protein_plan.rows
        self.base = None                                        # (set by the trainer: kind + batch signature + by-value scalars)
        self.byval = ()
        self.labels = None
        if "cm" in kind:
            from .model.cross_modality import CMCodes, CMLabels
            if trainer.model.cm_model.global_batch and trainer.world > 1:
                # global-batch form: the step gathers integer id codes and builds the label matrix on the device
                self.labels = CMCodes(int(batch[2].shape[0]), dev).fill(meta)
            else:
                self.labels = CMLabels(int(batch[2].shape[0]), dev).fill(meta, trainer.model.cm_model.use_cm)
        # The weight-image refresh (dl_weight_prep) must be a node of the graph whatever ran last: an eager forward with no
        # optimiser step behind it (evaluate() between the warm-up steps and this capture, a validate-every-N loop) leaves
        # the images current, lowp() would skip the refresh during capture, and every replay would then compute with the
        # images frozen at capture time while AdamW keeps moving the fp32 masters.  Making the epoch stale puts the
        # refresh (and the derived-layout rebuilds) at the head of the captured forward; counted below.
        # (the item tables of the refresh are rebuilt — a host-to-device copy, illegal during capture — whenever the set of
        #  live images changed, e.g. an earlier model's images were garbage-collected: do that eagerly first)
        Fn.bump_param_epoch()
        Fn.refresh_images()
        Fn.bump_param_epoch()
        before = ops.weight_prep_launches()
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        from . import _lib as _L
        if _L.AUDIT:
            _L.lib()
            _L.audit_begin()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out, self.idx = self._body()
        if _L.AUDIT:
            self.audit = self._audit_pointers(_L.audit_end())
        if ops.weight_prep_launches() == before and Fn.planned_image_count() > 0:
            raise RuntimeError("GraphedStep: no dl_weight_prep launch was captured; replays would use stale weight images")
        self.replays = 0

    def _pinned_ranges(self):
        """(address, bytes) of every buffer that lives as long as the trainer / process: what a captured launch may point at
        besides memory of the graph's own pool."""
        tr = self.tr
        ts = [tr.flat.arena, tr.flat.grads] + [t for o in (tr.opt, tr.opt_ssl, tr.opt_cm) if o is not None for t in (o.exp_avg, o.exp_avg_sq)]
        ts += [b for b in tr.model.buffers()] + [p_ for p_ in tr.model.parameters()]
        ts += [t for b in self.static for t in (b if isinstance(b, (tuple, list)) else (b,))]
        if self.labels is not None:
            ts += [self.labels.buf if hasattr(self.labels, "buf") else self.labels.dev]
        if self.hints.protein_plan is not None:
            ts += [self.hints.protein_plan.buf]
        ts += list(ops._guard_flags.values())
        ts += list(ops._seed_offsets.values()) + list(ops._tickets.values())
        ts += [e.image for e in Fn._lowp_cache.values() if torch.is_tensor(e.image)]
        for tab in list(Fn._lowp_tables.values()) + [t for t in Fn._lowp_retired if isinstance(t, tuple) and len(t) == 4]:
            ts += [tab[1], tab[2]]
        out = []
        for t in ts:
            if torch.is_tensor(t) and t.is_cuda:
                st = t.untyped_storage()
                out.append((st.data_ptr(), st.nbytes()))
        return out

    def _audit_pointers(self, log):
        """DL_GRAPH_PTR_AUDIT=1: every device pointer a captured library launch received lies in the graph's private pool or
        in a pinned buffer; anything else could be freed and re-used under the graph (the two use-after-free classes of
        round 2 were exactly that).  Raises with the offending entry points; returns (pointers checked, in pool, pinned)."""
        segs = [(s["address"], s["address"] + s["total_size"], tuple(s.get("segment_pool_id", (0, 0)))) for s in torch.cuda.memory_snapshot()]
        pinned = self._pinned_ranges()
        n_pool = n_pin = 0
        bad = []
        for fn, what, ptr in log:
            seg = next((s for s in segs if s[0] <= ptr < s[1]), None)
            if seg is not None and seg[2] != (0, 0):
                n_pool += 1
            elif any(a <= ptr < a + n for a, n in pinned):
                n_pin += 1
            elif seg is None:
                pass        # not device memory of the caching allocator (e.g. a module-scope constant of the library)
            else:
                bad.append((fn, what, hex(ptr)))
        if bad:
            raise RuntimeError("GraphedStep: %d pointers of captured launches lie in neither the graph's pool nor a pinned buffer "
                               "(first: %s)" % (len(bad), bad[:6]))
        return len(log), n_pool, n_pin

    @staticmethod
    def _clone(batch):
        return tuple(tuple(t.clone() for t in b) if isinstance(b, (tuple, list)) else b.clone() for b in batch)

    def _is_static(self, batch) -> bool:
        flat = lambda bt: [t for b in bt for t in (b if isinstance(b, (tuple, list)) else (b,))]   # noqa: E731
        return all(a is b for a, b in zip(flat(batch), flat(self.static)))

    @staticmethod
    def signature(batch):
        return tuple(tuple((tuple(t.shape), t.dtype) for t in b) if isinstance(b, (tuple, list)) else (tuple(b.shape), b.dtype)
                     for b in batch)

    def _body(self):
        tr, m = self.tr, self.tr.model
        ops.seed_offset_tensor(tr.device).add_(1)
        feat_d, feat_p, labels, llm_d, llm_p = self.static
        # (hints: by-value knowledge of the capture — the block sizes are part of the graph key, the table CONTENTS are refilled)
        _, _, ssl_input, cm_input, score = m(feat_d, feat_p, llm_d, llm_p, hints=self.hints)
        tr._zero_grad()
        _, cls_loss = binary_cross_entropy(score, labels) if tr.n_class == 1 else cross_entropy_logits(score, labels)
        out = {"cls": cls_loss.detach()}
        if "ssl" in self.kind:
            with m._glue(), Fn.deferred_bn_ticks():     # (the heads' BatchNorm step counters advance in one launch)
                d = m.ssl_model(**ssl_input)
            ssl_loss = (d["prot_ssl"] + d["drug_ssl"]) * 0.1
            if self.kind == "ssl":               # with the CM head behind it this backward is dead too
                ssl_loss.backward()
            out["ssl"] = ssl_loss.detach()
        if "cm" in self.kind:
            with m._glue():
                cm_loss = m.cm_model(**cm_input, labels=self.labels)
            cm_loss = cm_loss * tr.cm_weight
            cm_loss.backward()
            out["cm"] = cm_loss.detach()
        elif self.kind == "cls":
            cls_loss.backward()
        idx = tr.flat.pack_grads()
        self.reduced = False
        if tr.graph_allreduce and (tr.world > 1 or tr.graph_allreduce == "force"):
            # the gradient all-reduce as nodes of the graph (RCCL collectives can be captured: tools/graph_allreduce_probe.py):
            # no host launch between backward and the collective.  The agreed parameter set must be cached already (the
            # eager warm-up steps of this shape did that): agreeing needs a host sync, which capture forbids.
            if tr.world > 1:
                if self.last not in (tr._agreed_sets or {}):
                    raise RuntimeError("GraphedStep: the agreed gradient set of %s steps must exist before capture" % self.last)
                idx = tr._agreed(self.last, idx)
            tr._all_reduce_runs(idx)
            self.reduced = True
        return out, idx

    @property
    def last(self) -> str:
        """The backward pass the optimisers consume (the key of the agreed gradient set)."""
        return "cm" if "cm" in self.kind else self.kind

    def run(self, batch, meta=None):
        if self.hints.protein_plan is not None:    # this batch's residue counts -> the graph's static tables (built on the device)
            spec = self.tr.protein_plan_of(meta, batch, must_pay=False)
            if spec is None or spec.need > self.hints.protein_plan.rows:
                raise RuntimeError("GraphedStep: the batch's ProteinCNN row tables do not fit the captured capacity")
            self.hints.protein_plan.fill(spec)
        if self.labels is not None:                # host label matrix / id codes -> the graph's static tensors
            if hasattr(self.labels, "buf"):
                self.labels.fill(meta, self.tr.model.cm_model.use_cm)
            else:
                self.labels.fill(meta)
        if not self._is_static(batch):                  # copy into the static inputs (device-to-device)
            for dst, src in zip(self.static, batch):
                if isinstance(dst, tuple):
                    for d, s_ in zip(dst, src):
                        d.copy_(s_)
                else:
                    dst.copy_(src)
        self.graph.replay()
        self.replays += 1
        return self.out, self.idx


class Trainer:
    """ExpModule restated (trainer.py:39-292).  `cfg` is the merged config tree.
    graph_steps=True: cls, SSL-epoch and CM steps run as hipGraph replays (GraphedStep); the first steps of every new
    (batch shape, step kind), the epoch the CM head starts in and the global-batch CM form stay eager.
    graph_steps="auto" (round 5): a replay where it was measured to pay (`wants_graph`): every kind at per-GPU batches <= 128
    (the eager step is host-enqueue bound there) and, at any batch, the kinds with the cross-modality head (600+ launches per
    step, many of them 3 us: eager 9.2-10.4 / 12.5-14.8 ms against 8.76 / 11.5 ms replayed at batch 256, same box); the cls and
    SSL-epoch steps at batch 256 stay eager (12.4 / 10.6 ms against 12.6 / 11.0 replayed)."""
    overlap = None
    # (drug-token block, ProteinCNN table rows) used for EVERY step, eager or captured, instead of per-batch capacities.
    # Reductions over rows (BatchNorm statistics, split-K weight gradients) associate by the row count, so an eager trainer
    # matches a graph-replaying one BIT FOR BIT only at equal capacities: the graph-vs-eager tests pin this on both trainers
    # when their batches differ (`fixed_caps_for`).  Eager steps otherwise round the needed rows up to 2048, captures use the
    # geometric classes of protein_plan.row_class (identical up to 16 k rows).
    fixed_caps = None
    _agreed_sets = None
    grad_bf16 = False
    graph_allreduce = False

    def __init__(self, model, cfg, device=None, compute_dtype=torch.float32, graph_steps: bool = False):
        self.model = model
        self.cfg = cfg
        self.device = device or next(model.parameters()).device
        self.n_class = cfg["DECODER"]["BINARY"]
        self.epochs = cfg["SOLVER"]["MAX_EPOCH"]
        self.use_ssl = bool(cfg["RS"]["SSL"])
        self.use_cm = bool(cfg["RS"]["CM"])
        self.ssl_epoch_step = cfg["RS"]["EPOCH_STEP"]
        self.cm_init_epoch = cfg["RS"]["INIT_EPOCH"]
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        model.set_compute_dtype(compute_dtype)
        # main.py:158-160: three AdamW over model.parameters() built BEFORE any SSL forward (the lazily
        # created SimSiam projectors are therefore in none of them)
        self.flat = FlatParams([p for p in model.parameters()])
        if self.world > 1:
            # what the reference's DDP wrapper does at wrap time (trainer.py:143-148 -> torch DDP): every replica starts
            # from rank 0's parameters and buffers (BatchNorm running statistics, step counters), whatever each rank's
            # own initialisation drew.  Without it replicas that were seeded differently drift silently: the all-reduce
            # averages gradients, never weights.
            self.sync_replicas()
        Fn.bump_param_epoch()
        warm = int(self.epochs * 0.2)
        self.opt = FusedAdamW(self.flat, cfg["SOLVER"]["LR"])
        self.opt_ssl = FusedAdamW(self.flat, cfg["SOLVER"]["SSL_LR"]) if self.use_ssl else None
        self.opt_cm = FusedAdamW(self.flat, cfg["SOLVER"]["CM_LR"]) if self.use_cm else None
        self.schd = CosineAnnealingWarmupRestarts(self.epochs, cfg["SOLVER"]["LR"], 1e-8, warm)
        self.schd_ssl = CosineAnnealingWarmupRestarts(self.epochs, cfg["SOLVER"]["SSL_LR"], 1e-8, warm) if self.use_ssl else None
        self.schd_cm = CosineAnnealingWarmupRestarts(self.epochs, cfg["SOLVER"]["CM_LR"], 1e-8, warm) if self.use_cm else None
        self.opt.lr = self.schd.lr
        if self.opt_ssl:
            self.opt_ssl.lr = self.schd_ssl.lr
        if self.opt_cm:
            self.opt_cm.lr = self.schd_cm.lr
        self.cm_weight = 1.0
        # the reference wipes the cls (and ssl) gradients with the next zero_grad before any optimiser steps
        # (header); a backward pass whose gradients nobody consumes is skipped unless asked for
        self.run_dead_backward = os.environ.get("DL_DEAD_BACKWARD", "0") == "1"
        # Gradient all-reduce at world > 1.  "1": bucketed all-reduce from inside backward (GradOverlap) for
        # eagerly run steps; the persistent large-tile GEMMs then hand their tiles out dynamically (ops.dynamic_tiles):
        # a CU that an RCCL channel workgroup occupies for a while takes fewer tiles instead of forcing a second round —
        # the reason overlap was off by default in round 1.  "0" (DEFAULT until a multi-GPU RCCL run of the overlapped form
        # is on record under profiles/): one reduction after backward.  "force": overlap also at
        # world size 1 (RCCL sanity runs).  hipGraph-replayed steps (graph_steps) always reduce after the replay: hooks
        # cannot launch collectives from inside a replay, and at the small per-GPU batches where graphs are used the
        # 5 ms saved on launches outweigh the <= 0.6 ms all-reduce they leave exposed.  UNMEASURED on RCCL with N > 1
        # (no multi-GPU box was available; two ranks sharing one GPU over gloo: tests/test_grad_overlap_gpu.py).
        self.grad_bf16 = os.environ.get("DL_GRAD_BF16", "0") == "1"     # bf16 gradient all-reduce (see _all_reduce_runs)
        # DL_GRAPH_ALLREDUCE=1: graph-replayed steps capture their gradient all-reduce (RCCL) instead of issuing it after
        # the replay; "force": also at world size 1 (single-GPU check of the capture path).  Off by default until a
        # multi-GPU run of it is on record.
        ga = os.environ.get("DL_GRAPH_ALLREDUCE", "0")
        self.graph_allreduce = "force" if ga == "force" else (ga == "1")
        ov = os.environ.get("DL_GRAD_OVERLAP", "0")
        grouped = dist.is_available() and dist.is_initialized()
        want_overlap = ((self.world > 1 and ov not in ("0", "")) or (ov == "force" and grouped)) and not graph_steps
        self.overlap = GradOverlap(self.flat) if want_overlap else None
        if self.overlap is not None:
            ops.dynamic_tiles(True)      # persistent GEMMs hand their tiles out dynamically while collectives share the CUs
            ops.reset_tickets()
            # (bucket all-reduces are issued from backward hooks: hints_of keeps the forward on one stream then)
        # hook-driven collectives cannot be launched from inside a graph replay: graphed steps reduce after the replay
        self.graph_steps = ("auto" if graph_steps == "auto" else bool(graph_steps)) if self.overlap is None else False
        self._graphs: Dict[tuple, GraphedStep] = {}
        self._agreed_sets: Dict[str, frozenset] = {}
        self._eager_seen: Dict[tuple, int] = {}
        self._graph_caps: Dict[tuple, tuple] = {}   # per (kind, batch signature, by-value scalars): the largest capacities asked for
        self.graph_captures = 0                     # captures made so far (tests / bench: bounded under varying lengths)
        self.graph_warmup = 2            # eager (real) steps of a batch shape before its graph is captured
        self.graph_cache_max = max(1, int(os.environ.get("DL_GRAPH_CACHE_MAX", "6")))     # live GraphedSteps (LRU)
        # A device synchronisation every `graph_sync_every` replays (0 = never).  Round 5 saw ONE hardware exception after
        # ~300 replays that no host call separated (an experimental head, since removed); every shipped step
        # kind soaks clean for 2000+ unsynchronised replays (tools/graph_nosync_soak.py), but the mechanism was never found,
        # so the replay queue is bounded: one stream synchronise per 256 steps costs < 0.1 % of a training loop that does not
        # read its losses (one that does synchronises every step anyway).
        self.graph_sync_every = max(0, int(os.environ.get("DL_GRAPH_SYNC_EVERY", "256")))
        self._replays_since_sync = 0
        self._plan_devs: Dict[tuple, object] = {}           # ProteinCNN row tables of eager steps, one set per shape
        # DL_FIXED_CAPS="<drug-token block>,<ProteinCNN table rows | none>": one capacity for every step of this trainer (the
        # deterministic mode: row reductions associate by capacity, so bits otherwise depend on which graph / bucket serves a
        # batch — INTEGRATION.md "Reproducibility"); the attribute `fixed_caps` is the programmatic form.
        fc = os.environ.get("DL_FIXED_CAPS", "")
        if fc:
            a, b = (fc.split(",") + ["none"])[:2]
            self.fixed_caps = (int(a), None if b.strip().lower() in ("", "none") else int(b))
        on_gpu = torch.device(self.device).type == "cuda"
        self._guard_pin = torch.zeros(1, dtype=torch.int32).pin_memory() if on_gpu else None
        self._guard_event = None
        if self.graph_steps:
            ops.use_seed_offset(True)    # one dropout-seed regime for eager and replayed steps

    # -- helpers ------------------------------------------------------------------------------------
    def sync_replicas(self, src: int = 0):
        """Broadcast the parameter arena and every module buffer from rank `src` (no-op at world size 1)."""
        if self.world <= 1:
            return
        dist.broadcast(self.flat.arena, src=src)
        for b in self.model.buffers():
            dist.broadcast(b, src=src)
        Fn.bump_param_epoch()

    def replicas_in_sync(self) -> bool:
        """True iff every rank holds bit-identical parameters (a cheap checksum all-gather; used by tests / on demand)."""
        if self.world <= 1:
            return True
        a = self.flat.arena
        chk = torch.stack((a.double().sum(), a.double().abs().sum(), (a.double() * torch.arange(1, a.numel() + 1, device=a.device,
                                                                                                   dtype=torch.float64) % 9973).sum()))
        allc = [torch.empty_like(chk) for _ in range(self.world)]
        dist.all_gather(allc, chk)
        return all(torch.equal(allc[0], c) for c in allc[1:])

    def _zero_grad(self):
        for p in self.flat.params:
            p.grad = None

    def _arm(self, last: str, kind: str):
        if self.overlap is not None and last == kind:
            self.overlap.arm(kind)

    def _reduce_and_pack(self, kind: str = "cls") -> List[int]:
        if self.overlap is not None:
            return self.overlap.finish()
        idx = self.flat.pack_grads()
        if self.world > 1:
            idx = self._agreed(kind, idx)
            self._all_reduce_runs(idx)
        return idx

    def _all_reduce_runs(self, idx: List[int]):
        """Sum the flat gradient buffer over the ranks, one collective per contiguous run of parameters (usually one).
        grad_bf16 (DL_GRAD_BF16=1, off by default: the reference's DDP reduces fp32 and bit-parity tests need it): the
        run is cast to bf16, reduced (half the bytes on the xGMI links: 28 MB instead of 56 MB) and cast back into the
        fp32 buffer the optimiser reads — fp32 master weights and moments are untouched; every rank applies the same
        rounded sum, so replicas stay bit-identical to each other."""
        for s, e, _ in self.flat.runs(idx, lambda i: 0):
            g = self.flat.grads[s:e]
            if self.grad_bf16:
                h = ops.cast(g, torch.bfloat16)
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                g.copy_(h)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)

    def _agreed(self, kind: str, have: List[int]) -> List[int]:
        """The parameter set every rank reduces for a backward pass of `kind`: the union over ranks of the parameters that
        received a gradient, agreed on ONCE per kind (one small MAX all-reduce in the first pass of that kind — the only
        host synchronisation; later steps issue no extra collective).  A parameter of the set without a local gradient
        contributes zeros (what DDP's find_unused_parameters does for the reference, trainer.py:147); a gradient OUTSIDE
        the agreed set means the graph changed between steps and raises (ranks must not silently issue different
        collectives)."""
        n = len(self.flat.params)
        if self._agreed_sets is None:
            self._agreed_sets = {}
        cached = self._agreed_sets.get(kind)
        if cached is None:
            mask = torch.zeros(n, dtype=torch.int32)
            mask[have] = 1
            mask = mask.to(self.flat.grads.device)
            dist.all_reduce(mask, op=dist.ReduceOp.MAX)
            cached = self._agreed_sets[kind] = frozenset(mask.cpu().nonzero().flatten().tolist())
        extra = set(have) - cached
        if extra:
            raise RuntimeError("Trainer: %d parameters outside the set agreed for '%s' backward passes received a gradient "
                               "(first: index %d); the graph changed between steps" % (len(extra), kind, min(extra)))
        missing = sorted(cached - set(have))
        if missing:
            torch._foreach_zero_([self.flat.grad_views[i] for i in missing])
        return sorted(cached)

    def set_lrs(self, lr=None, ssl_lr=None, cm_lr=None):
        if lr is not None:
            self.opt.lr = lr
        if ssl_lr is not None and self.opt_ssl:
            self.opt_ssl.lr = ssl_lr
        if cm_lr is not None and self.opt_cm:
            self.opt_cm.lr = cm_lr

    def model_has_global_ntxent(self) -> bool:
        ssl = getattr(self.model, "ssl_model", None)
        return bool(ssl is not None and getattr(ssl, "global_batch", False) and getattr(ssl, "drug_ssl_type", "") == "simclr")

    GRAPH_AUTO_MAX_BATCH = 128

    def wants_graph(self, batch_size: int, compute_cm: bool) -> bool:
        """graph_steps as a per-step decision (True / False / "auto": see the class docstring)."""
        if self.graph_steps == "auto":
            return batch_size <= self.GRAPH_AUTO_MAX_BATCH or bool(compute_cm)
        return bool(self.graph_steps)

    def graphed_kind_ok(self, compute_ssl: bool, compute_cm: bool) -> bool:
        """Whether a step kind may replay a graph at this world size (bench.py reports `hip_graph` from it)."""
        coll_ok = self.world == 1 or (bool(self.graph_allreduce) and dist.get_backend() == "nccl")
        cm = getattr(self.model, "cm_model", None)
        return ((not compute_ssl) or coll_ok or not self.model_has_global_ntxent()) and \
            ((not compute_cm) or coll_ok or not (cm is not None and cm.global_batch))

    def protein_plan_of(self, meta, batch, must_pay: bool = True):
        """The ProteinCNN distinct-row spec (protein_plan.PlanSpec: residue counts + the rows the compact layout needs; no
        tables — those are built on the device) of a batch from the collate's `Prot_Len` records (residue counts after the
        reference's truncation, handler/dataset.py:36,139), or None (no records / switched off / nothing to save)."""
        if not meta or not getattr(self.model, "compact_cnn", False) or any("Prot_Len" not in m_ for m_ in meta):
            return None
        vp = batch[1]
        if not (torch.is_tensor(vp) and vp.dim() == 2 and vp.shape[0] == len(meta)):
            return None
        from .protein_plan import PlanSpec
        spec = PlanSpec([m_["Prot_Len"] for m_ in meta], int(vp.shape[1]))
        return spec if (spec.pays() or not must_pay) else None

    def hints_of(self, meta, batch, own_tables: bool = False, kind: str = "cls", caps=None):
        """BatchHints for model(..., hints=...): the drug-token block (padding_hints_of) and the ProteinCNN plan's device
        tables.  Eager steps share one table set per shape (refilled in place, stream-ordered); own_tables=True gives the
        caller its own (a captured graph keeps pointing at them).  caps = (drug-token block, table rows or None): capacities
        to use instead of the tightest ones for this batch (the steps around a graph capture run at the capture's sizes)."""
        from .protein_plan import BatchHints, PlanDev
        if caps is None:
            caps = self.fixed_caps
        plan = self.protein_plan_of(meta, batch, must_pay=caps is None)
        blk = self.padding_hints_of(meta, batch).get("drug_tokens", 0)
        rows = 0
        if caps is not None:
            blk, rows = caps[0], caps[1]
            if rows is None:
                plan = None
        pd = None
        if plan is not None:
            rows = rows or plan.rows()
            key = (rows, plan.B, plan.S)
            if own_tables:
                pd = PlanDev(plan, self.device, rows=rows)
            else:
                pd = self._plan_devs.pop(key, None)
                if pd is None:
                    while len(self._plan_devs) >= 4:
                        del self._plan_devs[next(iter(self._plan_devs))]
                    pd = PlanDev(plan, self.device, rows=rows)
                else:
                    pd.fill(plan)
                self._plan_devs[key] = pd           # most recently used last
        # side streams for the forward's independent branches on cls steps only (measured; BatchHints.__init__), and never
        # while gradient buckets are reduced from backward hooks (one stream for the collectives)
        return BatchHints(blk, pd,
                          branch_streams=(kind == "cls" and self.overlap is None),
                          raw_attention=False)       # (a training step never reads the PGCA raw-logit maps: reference trainer.py:179-231)

    # -- device-side padding guards (ops.guard_flags): polled without a host sync, checked with one on demand -------------
    def _poll_guard(self):
        """Raise if a guard tripped in an earlier step (reads the pinned copy the previous steps posted; never waits)."""
        if self._guard_pin is None:
            return
        ev = self._guard_event
        if ev is not None and ev.query():
            bits = int(self._guard_pin[0])
            self._guard_event = None
            if bits:
                ops.guard_flags(self.device).zero_()
                raise RuntimeError("druglamp_amd: padding guard tripped in an earlier step — " + ops.guard_text(bits))

    def _post_guard(self):
        if self._guard_pin is None or self._guard_event is not None:
            return                                        # one copy in flight at a time (the word is sticky)
        self._guard_pin.copy_(ops.guard_flags(self.device), non_blocking=True)
        self._guard_event = torch.cuda.Event()
        self._guard_event.record()

    def check_device_flags(self):
        """Synchronising check of the padding guards (end of an epoch, of an evaluation, of a benchmark)."""
        if torch.device(self.device).type == "cuda":
            ops.check_guard_flags(self.device)

    @staticmethod
    def padding_hints_of(meta, batch) -> dict:
        """Host-side knowledge about the batch for the model (functional.padding_hints).  drug_tokens: a block size (multiple
        of 64, so that captured graphs are keyed by few values) that covers the token count of every molecule of the batch,
        from the collate's `Drug_Tokens` records; absent records or a block that saves nothing: no hint."""
        if not meta or any("Drug_Tokens" not in m_ for m_ in meta):
            return {}
        n_rows = int(batch[3].shape[1]) if torch.is_tensor(batch[3]) and batch[3].dim() == 3 else 0
        blk = (max(int(m_["Drug_Tokens"]) for m_ in meta) + 127) // 128 * 128      # 128 / 256 / 384: few graph keys
        return {"drug_tokens": blk} if 0 < blk <= n_rows - 128 else {}

    # -- the step ---------------------------------------------------------------------------------------
    def training_step(self, batch, meta=None, cur_epoch: int = 1, ssl_masks=None) -> Dict[str, float]:
        """batch = (feat_d, feat_p, labels, llm_d, llm_p) as the reference's collate yields them.
        cur_epoch is 1-based (trainer.py:180).  Returns the losses as detached DEVICE scalars ({'cls': ..., 'ssl': ..., 'cm': ...}): the
        step itself never synchronises with the host (reading a value does), so steps are enqueued back to back."""
        m = self.model
        self._poll_guard()
        if not m.training:
            m.train()                  # (walks every submodule: 0.7 ms per call)
        compute_ssl = self.use_ssl and (cur_epoch % self.ssl_epoch_step == 0)
        compute_cm = self.use_cm and (cur_epoch >= self.cm_init_epoch)
        eager_caps = None
        # steps with the CM head replay a graph too (round 3) except: in the epoch the head starts (the cm_weight
        # auto-scale below reads losses on the host), without id records, and in the global-batch form (object collectives)
        # Heads that see the GLOBAL batch issue collectives in their forward and backward (NT-Xent rows, CM token means and
        # id codes: tensor collectives only since round 4).  A step with them is captured only where captured collectives
        # are switched on (DL_GRAPH_ALLREDUCE) AND the backend is RCCL; on gloo (every two-rank test) it stays eager.
        coll_ok = self.world == 1 or (bool(self.graph_allreduce) and dist.get_backend() == "nccl")
        cm_ok = (not compute_cm) or (cur_epoch > self.cm_init_epoch and meta is not None and
                                     (coll_ok or not m.cm_model.global_batch))
        ssl_ok = (not compute_ssl) or coll_ok or not self.model_has_global_ntxent()
        if self.wants_graph(int(batch[2].shape[0]), compute_cm) and cm_ok and ssl_ok and not self.run_dead_backward and \
                (not compute_ssl or ssl_masks is None):
            kind = (("ssl" if compute_ssl else "") + ("cm" if compute_cm else "")) or "cls"
            byval = (float(m.cm_model.m_sch_loss_fn.margin), float(self.cm_weight)) if compute_cm else ()
            base = (kind,) + GraphedStep.signature(batch) + byval
            # (must_pay=False for the LOOKUP: a batch whose compact layout would not pay on its own still fits — and replays —
            #  a graph that holds tables of its size; GraphedStep.run builds its spec the same way)
            spec_any = self.protein_plan_of(meta, batch, must_pay=False)
            blk = self.padding_hints_of(meta, batch).get("drug_tokens", 0)
            g = self._find_graph(base, blk, spec_any)
            if g is not None:
                return self._graphed_step(g, batch, meta)
            spec = spec_any if (spec_any is not None and spec_any.pays()) else None
            # No captured graph serves this batch.  Capacities of the next capture: size CLASSES (row_class: a geometric
            # ladder; drug-token blocks of 128), never below what an earlier batch of this shape asked for — so the captures
            # of a shape form a chain of growing capacities and batches of varying lengths converge on one or two graphs
            # (round 4 keyed graphs by the exact 2048-row bucket: ~10 keys at batch 256, capture / eviction thrash).
            # The eager warm-up steps of a capacity run at that capacity (same shapes as the capture).
            eager_caps = self._capture_caps(base, blk, spec)
            key = base + eager_caps
            if self._eager_seen.get(key, 0) >= self.graph_warmup:
                return self._graphed_step(self._capture(key, base, byval, batch, kind, meta, eager_caps), batch, meta)
            if len(self._eager_seen) > 256:            # bounded bookkeeping (signatures of shapes seen once and never again)
                self._eager_seen.clear()
            self._eager_seen[key] = self._eager_seen.get(key, 0) + 1
        feat_d, feat_p, labels, llm_d, llm_p = batch
        kind_now = "cm" if compute_cm else "ssl" if compute_ssl else "cls"
        _, _, ssl_input, cm_input, score = m(feat_d, feat_p, llm_d, llm_p, hints=self.hints_of(meta, batch, kind=kind_now, caps=eager_caps))
        self._zero_grad()
        _, cls_loss = binary_cross_entropy(score, labels) if self.n_class == 1 else cross_entropy_logits(score, labels)
        last = "cm" if compute_cm else "ssl" if compute_ssl else "cls"     # the backward the optimisers consume
        if last == "cls" or self.run_dead_backward:
            self._arm(last, "cls")
            cls_loss.backward(retain_graph=compute_ssl or compute_cm)
        out = {"cls": cls_loss.detach()}
        if compute_ssl:
            self._zero_grad()
            kw = dict(ssl_input)
            if ssl_masks is not None:
                kw.update(mask=ssl_masks[0], replace=ssl_masks[1])
            with m._glue(), Fn.deferred_bn_ticks():     # bf16 compute dtype: the heads' torch layers run under bf16 autocast; the
                d = m.ssl_model(**kw)                   # BatchNorm step counters of the heads advance in one launch
            ssl_loss = (d["prot_ssl"] + d["drug_ssl"]) * 0.1
            if last == "ssl" or self.run_dead_backward:
                self._arm(last, "ssl")
                ssl_loss.backward(retain_graph=compute_cm)
            out["ssl"] = ssl_loss.detach()
        if compute_cm:
            self._zero_grad()
            with m._glue():
                cm_loss = m.cm_model(**cm_input, meta=meta)
            if cur_epoch == self.cm_init_epoch:
                c, l = float(cm_loss.detach()), float(cls_loss.detach())
                if c > 0:
                    while c * self.cm_weight / 10 > l:
                        self.cm_weight /= 10
                    while c * self.cm_weight * 10 < l:
                        self.cm_weight *= 10
            cm_loss = cm_loss * self.cm_weight
            self._arm(last, "cm")
            cm_loss.backward()
            out["cm"] = cm_loss.detach()
        idx = self._reduce_and_pack(last)
        scale = 1.0 / self.world
        self.opt.step(idx, scale)
        if compute_ssl:
            self.opt_ssl.step(idx, scale)
        if compute_cm:
            self.opt_cm.step(idx, scale)
        Fn.bump_param_epoch()
        self._post_guard()
        return out

    # -- captured graphs: lookup by capacity, capture at size classes -------------------------------------------------------
    _INF = 1 << 60

    def _find_graph(self, base, blk: int, spec):
        """The tightest captured graph of this (kind, batch signature, by-value scalars) whose capacities hold the batch: a
        drug-token block >= the batch's (0 = every row computed: holds anything) and ProteinCNN tables with >= the rows the
        batch needs (no tables = every position computed: holds anything)."""
        best, best_cost = None, None
        for g in self._graphs.values():
            if g.base != base:
                continue
            if g.block != 0 and (blk == 0 or g.block < blk):
                continue
            if g.rows_cap is not None and (spec is None or spec.need > g.rows_cap):
                continue
            cost = (g.rows_cap if g.rows_cap is not None else self._INF, g.block or self._INF)
            if best is None or cost < best_cost:
                best, best_cost = g, cost
        return best

    def fixed_caps_for(self, batches) -> tuple:
        """Capacities that hold every (batch, meta) of `batches` (for `fixed_caps`)."""
        from .protein_plan import row_class
        blks = [self.padding_hints_of(mt, b).get("drug_tokens", 0) for b, mt in batches]
        specs = [self.protein_plan_of(mt, b) for b, mt in batches]
        rows = None if any(sp is None for sp in specs) else row_class(max(sp.need for sp in specs))
        return (0 if 0 in blks else max(blks), rows)

    def _capture_caps(self, base, blk: int, spec) -> tuple:
        from .protein_plan import row_class
        if self.fixed_caps is not None:
            return tuple(self.fixed_caps)
        rows = None if spec is None else row_class(spec.need)
        if spec is not None and not spec.pays(rows):
            rows = None
        # The chain of growing capacities per shape.  A batch WITHOUT a usable plan (no Prot_Len records, or a compact layout
        # that does not pay) is served by the full layout for that batch only: it must not erase what earlier batches asked for
        # (round 5 made rows=None sticky: one such batch and every later capture of the shape used ~3.5x the rows).
        old = self._graph_caps.get(base)
        keep_blk, keep_rows = blk, rows
        if old is not None:
            if blk != 0 and old[0] != 0:
                blk = keep_blk = max(blk, old[0])
            elif blk == 0:
                keep_blk = old[0]               # (0 = every drug row computed: this batch has no token records)
            if rows is not None and old[1] is not None:
                rows = keep_rows = max(rows, old[1])
            elif rows is None:
                keep_rows = old[1]
        if len(self._graph_caps) > 256:
            self._graph_caps.clear()
        self._graph_caps[base] = (keep_blk, keep_rows)
        return (blk, rows)

    def _capture(self, key, base, byval, batch, kind, meta, caps) -> "GraphedStep":
        if "cm" in kind:
            # margin / cm_weight are by-value arguments of a capture: EVERY graph with the CM head ("cm" and "sslcm"
            # kinds, any shape or hint) captured under other values will not be replayed again (the margin moves once
            # per epoch) — release their pools first
            for old in [k for k, v in self._graphs.items() if "cm" in v.kind and v.byval != byval]:
                del self._graphs[old]
        # Every GraphedStep owns a private pool with a whole step's activations (GBs at batch 64-128): keep at most
        # graph_cache_max of them, least recently replayed first out
        while len(self._graphs) >= self.graph_cache_max:
            del self._graphs[next(iter(self._graphs))]
        g = self._graphs[key] = GraphedStep(self, batch, kind, meta, caps)     # records only; the replay that follows is the step
        g.base, g.byval, g.key = base, byval, key
        self.graph_captures += 1
        return g

    def _graphed_step(self, g: "GraphedStep", batch, meta=None) -> Dict[str, float]:
        kind = g.kind
        self._graphs[g.key] = self._graphs.pop(g.key)                        # most recently used last
        out, idx = g.run(batch, meta)
        if self.world > 1 and not g.reduced:
            idx = self._agreed(g.last, idx)
            self._all_reduce_runs(idx)
        self.opt.step(idx, 1.0 / self.world)
        if "ssl" in kind:
            self.opt_ssl.step(idx, 1.0 / self.world)
        if "cm" in kind:
            self.opt_cm.step(idx, 1.0 / self.world)
        Fn.bump_param_epoch()
        self._post_guard()
        self._replays_since_sync += 1
        if self.graph_sync_every and self._replays_since_sync >= self.graph_sync_every:
            torch.cuda.current_stream(self.device).synchronize()
            self._replays_since_sync = 0
        return out

    def static_batch(self, batch):
        """The captured graph's own input tensors for batches of this shape (or `batch` itself while no graph exists):
        a producer that fills them in place — and passes them back — saves the per-step input copy."""
        shape = GraphedStep.signature(batch)
        g = next((v for k, v in self._graphs.items() if k[1:1 + len(shape)] == shape), None)
        return batch if g is None else g.static

    def on_train_epoch_end(self, cur_epoch: int):
        compute_ssl = self.use_ssl and (cur_epoch % self.ssl_epoch_step == 0)
        compute_cm = self.use_cm and (cur_epoch >= self.cm_init_epoch)
        self.opt.lr = self.schd.step()
        if compute_ssl:
            self.opt_ssl.lr = self.schd_ssl.step()
        if compute_cm:
            self.opt_cm.lr = self.schd_cm.step()
            self.model.cm_model.step()
        self.check_device_flags()                      # (an epoch boundary: a host sync costs nothing here)

    # -- evaluation (trainer.py:256-292; torchmetrics replaced by sklearn on the gathered predictions) -----
    @torch.no_grad()
    def predict(self, batches):
        """Eval-mode forward over `batches` (any batch size: the reference validates / tests at batch 1, main.py:146-153;
        the per-sample results do not depend on the batch because BatchNorm uses its running statistics).  Returns this
        rank's (probabilities, labels, summed BCE, sample count)."""
        self.model.eval()
        preds, labs = [], []
        loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        for batch in batches:
            feat_d, feat_p, labels, llm_d, llm_p = batch
            _, _, _, _, score = self.model(feat_d, feat_p, llm_d, llm_p)
            n, loss = binary_cross_entropy(score, labels) if self.n_class == 1 else cross_entropy_logits(score, labels)
            preds.append(n.float().reshape(-1))
            labs.append(labels.float().reshape(-1))
            loss_sum += loss.double() * labels.numel()              # per-batch means -> a sum over samples
        p = torch.cat(preds) if preds else torch.zeros(0, device=self.device)
        y = torch.cat(labs) if labs else torch.zeros(0, device=self.device)
        self.check_device_flags()
        return p, y, loss_sum, p.numel()

    def evaluate(self, batches) -> Dict[str, float]:
        """Metrics over the union of all ranks' samples (the reference's torchmetrics objects gather their states at
        epoch end, trainer.py:262-292; its logged loss is `sync_dist=True`): ranks may hold different numbers of samples."""
        p, y, loss_sum, n = self.predict(batches)
        p, y, loss_sum, n = gather_predictions(p, y, loss_sum, n, self.world)
        out = binary_metrics(p.cpu().numpy(), y.cpu().numpy())
        out["loss"] = float(loss_sum) / max(n, 1)
        return out

    def _epoch_losses(self, sums, steps) -> Dict[str, float]:
        return _epoch_loss_means(sums, steps, self.world, self.device)

    def fit(self, train_batches, val_batches, epochs: Optional[int] = None, on_epoch=None) -> Dict[str, object]:
        """The reference's run_experiment loop (trainer.py:131-135,150-163): one pass over `train_batches()` per epoch
        (a callable yielding (batch, meta) pairs), on_train_epoch_end, validation; the parameters with the best
        `val_ausum` (AUROC + AUPRC: ModelCheckpoint(monitor='val_ausum', mode='max')) are kept, training stops after
        patience = epochs / 4 epochs without improvement (EarlyStopping, min_delta 0), and the best parameters are
        reloaded before returning (trainer.py:134) — ready for evaluate(test_batches)."""
        epochs = int(epochs or self.epochs)
        patience = max(int(epochs / 4), 1)
        best = {"ausum": -float("inf"), "epoch": 0, "arena": None, "buffers": None}
        history, bad = [], 0
        for ep in range(1, epochs + 1):
            sums: Dict[str, torch.Tensor] = {}
            steps = 0
            for batch, meta in train_batches():
                out = self.training_step(batch, meta=meta, cur_epoch=ep)
                for k, v in out.items():                 # device-side running sums: no host synchronisation per step
                    sums[k] = sums[k] + v.detach().float() if k in sums else v.detach().float().clone()
                steps += 1
            self.on_train_epoch_end(ep)
            val = self.evaluate(val_batches() if callable(val_batches) else val_batches)
            val.update(self._epoch_losses(sums, steps))
            history.append(val)
            if on_epoch is not None:
                on_epoch(ep, val)
            score = val["ausum"]
            if score == score and score > best["ausum"]:
                best.update(ausum=score, epoch=ep, arena=self.flat.arena.clone(),
                            buffers=[b.detach().clone() for b in self.model.buffers()])
                bad = 0
            else:
                bad += 1
                if bad >= patience:
                    break
        if best["arena"] is not None:                                    # reload the best checkpoint (trainer.py:134)
            with torch.no_grad():
                self.flat.arena.copy_(best["arena"])
                for b, src in zip(self.model.buffers(), best["buffers"]):
                    b.copy_(src)
            Fn.bump_param_epoch()
        return {"best_epoch": best["epoch"], "best_val_ausum": best["ausum"], "epochs_run": len(history), "history": history}


def _epoch_loss_means(sums: Dict[str, torch.Tensor], steps: int, world: int, device) -> Dict[str, float]:
    """Epoch means of the step losses over steps AND ranks — what the reference logs with `on_epoch=True, sync_dist=True`
    (trainer.py:201,208,222,231): train_loss (cls), ssl_loss, cm_loss (weighted, as added to the total) and all_loss (their
    sum on the steps that had them).  One all-reduce of a 4-vector of sums + a step count per epoch."""
    keys = ("cls", "ssl", "cm")
    vec = torch.zeros(len(keys) + 1, dtype=torch.float64, device=device)
    for i, k in enumerate(keys):
        if k in sums:
            vec[i] = sums[k].double()
    vec[len(keys)] = float(steps)
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    v = vec.tolist()
    n = max(v[len(keys)], 1.0)
    out = {"train_loss": v[0] / n, "ssl_loss": v[1] / n, "cm_loss": v[2] / n}
    out["all_loss"] = out["train_loss"] + out["ssl_loss"] + out["cm_loss"]
    return out


def gather_predictions(p: torch.Tensor, y: torch.Tensor, loss_sum: torch.Tensor, n: int, world: int):
    """All ranks' predictions / labels / loss sums, for shards of DIFFERENT sizes: counts are gathered first, every
    shard is padded to the largest and trimmed after the gather (equal-size all_gather would hang or mis-slice)."""
    if world <= 1:
        return p, y, loss_sum, n
    dev = p.device
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev))
    counts = [int(c) for c in counts]
    m = max(max(counts), 1)
    buf = torch.zeros((2, m), dtype=torch.float32, device=dev)
    buf[0, :n], buf[1, :n] = p.float(), y.float()
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    ls = loss_sum.detach().double().reshape(1).clone()
    dist.all_reduce(ls, op=dist.ReduceOp.SUM)
    p = torch.cat([b[0, :c] for b, c in zip(bufs, counts)])
    y = torch.cat([b[1, :c] for b, c in zip(bufs, counts)])
    return p, y, ls[0], sum(counts)


def binary_metrics(p, y) -> Dict[str, float]:
    """AUROC / AUPRC (sklearn = torchmetrics' exact, threshold-free definitions), their sum (the reference's BinaryAUSum,
    trainer.py:17-37) and the thresholded metrics at 0.5 (trainer.py:109-119)."""
    from sklearn.metrics import average_precision_score, roc_auc_score
    auroc = float(roc_auc_score(y, p)) if len(set(y.tolist())) > 1 else float("nan")
    auprc = float(average_precision_score(y, p)) if y.sum() > 0 else float("nan")
    yhat = (p >= 0.5).astype("float32")
    tp, tn = float(((yhat == 1) & (y == 1)).sum()), float(((yhat == 0) & (y == 0)).sum())
    fp, fn = float(((yhat == 1) & (y == 0)).sum()), float(((yhat == 0) & (y == 1)).sum())
    prec = tp / max(tp + fp, 1.0)
    rec = tp / max(tp + fn, 1.0)
    return {"auroc": auroc, "auprc": auprc, "ausum": auroc + auprc, "acc": (tp + tn) / max(len(y), 1), "sn": rec,
            "sp": tn / max(tn + fp, 1.0), "pr": prec, "f1": 2 * prec * rec / max(prec + rec, 1e-12)}
