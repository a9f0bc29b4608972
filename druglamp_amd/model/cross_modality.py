"""Cross-modality 2C2P head (reference: model/cross_modality.py).  Label-matrix construction stays on
the host (python dicts in the reference, one numpy matrix here); the O(n_p * n_d) python triplet loop
of ccpp_p_tri_loss is replaced by the all-pairs sigmoid-cosine + masked triplet HIP kernel.

Round 3: the device side is SHAPE-STATIC.  A batch of B pairs has n_p <= B unique proteins and n_d <= B unique
drugs; the reference runs its BatchNorms / Linears / triplet loop over (n_p, ...) and (n_d, ...) tensors whose
size changes from batch to batch.  Here every tensor has B rows: the unique-row gathers are padded with row 0, the
four BatchNorm1d layers take their statistics over the first n_p / n_d rows only (row mask + device-side row count),
the label matrix is (B, B) with the padding marked "ignored" (-1: no triplet is formed with it, and a padded
row's gradient is exactly zero).  Values are those of the unpadded computation (the padding adds exact zeros to
every sum); the shapes no longer depend on the batch's ids, so a step with this head can be captured in a hipGraph
(trainer.GraphedStep kind "cm"): per step only the CMLabels tensors are refreshed from the host."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dist_ops
from .. import functional as Fn


def tanh_decay(m_ori, n_re, step):
    return m_ori * (1 - np.tanh(2 * (1 - step / n_re)))


class MarginSchedule:
    """MarginScheduledLossFunction (cross_modality.py:49-102): margin starts at m_ori; the k-th step()
    sets tanh_decay(k); at k == n_re the counter resets to 0."""

    def __init__(self, m_ori=0.25, n_epoch=100, n_re=-1):
        self.m_ori = m_ori
        self.n_re = int(n_epoch * 0.2) if n_re == -1 else n_re
        self._step = 0
        self.m_cur = m_ori

    @property
    def margin(self):
        return self.m_cur

    def step(self):
        self._step += 1
        if self._step == self.n_re:
            self._step = 0
        self.m_cur = float(tanh_decay(self.m_ori, self.n_re, self._step))


def label_matrix(meta, use_cm=True):
    """Unique ids in first-seen order with the LAST occurrence index per id; unobserved pairs get
    `0 if use_cm else -1` (cross_modality.py:138-144)."""
    pid2t, did2t = {}, {}
    for t, m in enumerate(meta):
        pid2t[m["Prot_ID"]] = t
        did2t[m["Drug_ID"]] = t
    pi = {p: i for i, p in enumerate(pid2t)}
    di = {d: i for i, d in enumerate(did2t)}
    gt = np.full((len(pi), len(di)), 0 if use_cm else -1, dtype=np.int8)
    for m in meta:
        gt[pi[m["Prot_ID"]], di[m["Drug_ID"]]] = int(m["Y"])
    return list(pid2t.values()), list(did2t.values()), gt


def Mean2Embed(hidden=128):
    return nn.Sequential(nn.BatchNorm1d(hidden), nn.ReLU(inplace=True), nn.Linear(hidden, hidden))


class CMLabels:
    """Device-side form of label_matrix() for batches of `rows` pairs, padded to static shapes: idx[0] / idx[1] = sample index
    of each unique protein / drug (padding: 0), mask[0] / mask[1] = 1.0 on real rows, n = (n_p, n_d) as floats,
    gt (rows, rows) int8 with -1 outside [n_p) x [n_d).  fill() refreshes the SAME tensors (a captured graph points at
    them)."""

    def __init__(self, rows: int, device):
        self.rows = B = rows
        # one device buffer, four views: a batch's labels arrive with ONE host-to-device copy from pinned memory that
        # nothing waits for (a pageable copy would make the host wait for every launch queued before it)
        o_mask, o_n, o_gt = 16 * B, 16 * B + 8 * B, 16 * B + 8 * B + 16
        self.nbytes = o_gt + B * B
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.idx = self.buf[:o_mask].view(torch.int64).view(2, B)
        self.mask = self.buf[o_mask:o_n].view(torch.float32).view(2, B, 1)
        self.n = self.buf[o_n:o_n + 8].view(torch.float32)
        self.gt = self.buf[o_gt:].view(torch.int8).view(B, B)
        self._off = (o_mask, o_n, o_gt)
        self.n.fill_(1.0)
        self.gt.fill_(-1)
        # pinned staging ring (allocated once: a pinned allocation per step costs milliseconds and synchronises the device);
        # a slot is rewritten only after the copy that read it has run (event per slot)
        on_gpu = torch.device(device).type == "cuda"
        self._pins = [torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory() if on_gpu else torch.zeros(self.nbytes, dtype=torch.uint8)
                      for _ in range(4)]
        self._events = [None] * len(self._pins)
        self._slot = 0

    def fill(self, meta, use_cm=True):
        pidx, didx, gt = label_matrix(meta, use_cm)
        B = self.rows
        if len(meta) != B:
            raise ValueError("CMLabels: %d meta rows for a %d-row label block" % (len(meta), B))
        n_p, n_d = len(pidx), len(didx)
        o_mask, o_n, o_gt = self._off
        k = self._slot
        self._slot = (k + 1) % len(self._pins)
        if self._events[k] is not None:
            self._events[k].synchronize()
        host = self._pins[k].numpy()
        host[:] = 0
        idx = host[:o_mask].view(np.int64).reshape(2, B)
        idx[0, :n_p], idx[1, :n_d] = pidx, didx
        mask = host[o_mask:o_n].view(np.float32).reshape(2, B)
        mask[0, :n_p], mask[1, :n_d] = 1.0, 1.0
        host[o_n:o_n + 8].view(np.float32)[:] = (float(n_p), float(n_d))
        g = host[o_gt:].view(np.int8).reshape(B, B)
        g[:] = -1
        g[:n_p, :n_d] = gt
        self.buf.copy_(self._pins[k], non_blocking=True)
        if self.buf.is_cuda:
            self._events[k] = torch.cuda.Event()
            self._events[k].record()
        return self


class CMCodes:
    """A batch's (prot code, drug code, label) rows on the device: (rows, 3) int64, refilled in place from a pinned ring
    (one non-blocking copy per step; a captured graph points at `dev`).  Input of the device-built label matrix."""

    def __init__(self, rows: int, device):
        self.rows = rows
        self.dev = torch.zeros((rows, 3), dtype=torch.int64, device=device)
        on_gpu = torch.device(device).type == "cuda"
        self._pins = [torch.zeros((rows, 3), dtype=torch.int64).pin_memory() if on_gpu else torch.zeros((rows, 3), dtype=torch.int64)
                      for _ in range(4)]
        self._events = [None] * len(self._pins)
        self._slot = 0

    def fill(self, meta):
        if len(meta) != self.rows:
            raise ValueError("CMCodes: %d meta rows for a %d-row block" % (len(meta), self.rows))
        k = self._slot
        self._slot = (k + 1) % len(self._pins)
        if self._events[k] is not None:
            self._events[k].synchronize()
        self._pins[k].numpy()[:] = dist_ops.id_codes(meta)
        self.dev.copy_(self._pins[k], non_blocking=True)
        if self.dev.is_cuda:
            self._events[k] = torch.cuda.Event()
            self._events[k].record()
        return self


class DeviceLabels:
    """label_matrix() built ON THE DEVICE from (G, 3) integer codes with static shapes — the same fields as CMLabels
    (idx, mask, n, gt), so the head below does not care which one it gets.  Reference semantics (cross_modality.py:138-150):
    unique ids in first-seen order, each represented by the LAST sample that carries it; gt[p][d] = the label of the LAST
    sample of that (protein, drug) pair; unobserved pairs 0 (use_cm) or -1; everything outside [n_p) x [n_d) is -1
    (ignored).  G x G comparison matrices (G = global batch <= a few thousand); no host round trip, capturable."""

    def __init__(self, codes: torch.Tensor, use_cm=True):
        G = codes.shape[0]
        dev = codes.device
        ar = torch.arange(G, device=dev)

        def uniq(c):
            eq = c.unsqueeze(1) == c.unsqueeze(0)                                   # (G, G)
            first = torch.where(eq, ar.unsqueeze(0), G).min(dim=1).values           # first sample with this id
            last = torch.where(eq, ar.unsqueeze(0), -1).max(dim=1).values           # last sample with this id
            is_first = first == ar
            pos = torch.cumsum(is_first.to(torch.int64), 0) - 1                     # position in the unique list (at firsts)
            n = is_first.sum()
            upos = pos.gather(0, first)                                             # unique position of EVERY sample
            idx = torch.zeros(G + 1, dtype=torch.int64, device=dev)
            idx.scatter_(0, torch.where(is_first, pos, torch.full_like(pos, G)), last)   # non-firsts write the spare slot
            return eq, upos, idx[:G], n

        eqp, up, pidx, n_p = uniq(codes[:, 0])
        eqd, ud, didx, n_d = uniq(codes[:, 1])
        last_pair = torch.where(eqp & eqd, ar.unsqueeze(0), -1).max(dim=1).values == ar     # one writer per (p, d) cell
        inside = (ar.unsqueeze(1) < n_p) & (ar.unsqueeze(0) < n_d)
        gt = torch.where(inside, torch.full((), 0 if use_cm else -1, dtype=torch.int8, device=dev),
                         torch.full((), -1, dtype=torch.int8, device=dev)).reshape(-1)
        gt = torch.cat([gt, gt.new_zeros(1)])
        cell = torch.where(last_pair, up * G + ud, torch.full_like(up, G * G))
        gt.scatter_(0, cell, codes[:, 2].to(torch.int8))
        self.gt = gt[:G * G].view(G, G).contiguous()
        self.idx = (pidx, didx)
        self.mask = ((ar < n_p).float().unsqueeze(1), (ar < n_d).float().unsqueeze(1))
        self.n = (n_p.float(), n_d.float())


def _bn_rows(bn: nn.BatchNorm1d, x, mask, n):
    """nn.BatchNorm1d over the rows with mask == 1 (n of them, a device scalar) of a padded (rows, hidden) block:
    batch statistics (biased variance) for the normalisation, running statistics updated with the unbiased variance
    and the module's momentum — what torch's layer does on the unpadded rows."""
    if bn.training:
        mean = (x * mask).sum(0) / n
        xc = (x - mean) * mask
        var = (xc * xc).sum(0) / n
        with torch.no_grad():
            mom = bn.momentum
            bn.running_mean.mul_(1.0 - mom).add_(mean * mom)
            bn.running_var.mul_(1.0 - mom).add_(var * (n / (n - 1.0).clamp(min=1.0)) * mom)
            bn.num_batches_tracked.add_(1)
    else:
        mean, var = bn.running_mean, bn.running_var
    return (x - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias


def _mean2embed(seq: nn.Sequential, x, mask, n):
    return seq[2](F.relu(_bn_rows(seq[0], x, mask, n)))


class CrossModality(nn.Module):
    def __init__(self, *, use_cm=True, hidden_size=128, max_margin=0.5, n_re=100, **kwargs):
        super().__init__()
        self.use_cm = use_cm
        self.prot2latent = Mean2Embed(hidden_size)
        self.aug_prot2latent = Mean2Embed(hidden_size)
        self.drug2latent = Mean2Embed(hidden_size)
        self.aug_drug2latent = Mean2Embed(hidden_size)
        self.to_prot_latent = nn.Linear(hidden_size * 2, hidden_size * 2, bias=False)
        self.to_drug_latent = nn.Linear(hidden_size * 2, hidden_size * 2, bias=False)
        self.m_sch_loss_fn = MarginSchedule(m_ori=max_margin, n_re=n_re)
        self.global_batch = bool(kwargs.get("global_batch", False))
        self._label_blocks = {}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_label_blocks"] = {}            # device / pinned staging state (CUDA events): rebuilt on demand, never copied or pickled
        return d

    def step(self):
        self.m_sch_loss_fn.step()

    def latents_from_means(self, pm, apm, dm, adm, labels: CMLabels):
        """pm/apm/dm/adm: per-sample token means (n, hidden).  Mean2Embed x4 -> concat -> Linear -> l2norm, on the padded
        unique-row blocks of `labels`."""
        pi, di = labels.idx[0], labels.idx[1]
        (pk, dk), (n_p, n_d) = labels.mask, labels.n
        pe = torch.cat([_mean2embed(self.prot2latent, pm.index_select(0, pi), pk, n_p),
                        _mean2embed(self.aug_prot2latent, apm.index_select(0, pi), pk, n_p)], dim=-1)
        de = torch.cat([_mean2embed(self.drug2latent, dm.index_select(0, di), dk, n_d),
                        _mean2embed(self.aug_drug2latent, adm.index_select(0, di), dk, n_d)], dim=-1)
        return F.normalize(self.to_prot_latent(pe), dim=-1), F.normalize(self.to_drug_latent(de), dim=-1)

    def forward(self, prot, aug_prot, drug, aug_drug, meta=None, labels=None):
        """meta: the batch's id / label records (the reference's argument); labels: the same thing already on the device
        (CMLabels.fill, or CMCodes.fill in the global-batch form — what a captured step hands over).  One of the two."""
        means = [t.float().mean(dim=1) for t in (prot, aug_prot, drug, aug_drug)]
        if self.global_batch and dist_ops.world_size() > 1:
            # NEW vs the reference (off by default): the label matrix and the triplets span the GLOBAL batch.
            # Only the (n, hidden) token means and three integers per pair travel (4 x 128 floats + 24 bytes per pair over
            # xGMI), all as TENSOR collectives: the ids go as 63-bit codes (dist_ops.id_codes), the label matrix of the
            # gathered batch is built on the device (DeviceLabels) — no host pickle collective on the step (round 4).
            if isinstance(labels, CMCodes):
                codes = labels
            else:
                if meta is None:
                    raise RuntimeError("CrossModality: the global-batch form needs the id records (meta) or a filled CMCodes block")
                key = ("codes", len(meta), means[0].device)
                codes = self._label_blocks.get(key)
                if codes is None:
                    codes = self._label_blocks[key] = CMCodes(len(meta), means[0].device)
                codes.fill(meta)
            means = [dist_ops.all_gather_rows(m) for m in means]
            labels = DeviceLabels(dist_ops.all_gather_codes(codes.dev), self.use_cm)
        if labels is None:
            # one label block per batch size, refilled per step (stream order keeps the previous step's backward ahead of
            # the refill; a captured step brings its own block)
            key = (len(meta), means[0].device)
            labels = self._label_blocks.get(key)
            if labels is None:
                labels = self._label_blocks[key] = CMLabels(*key)
            labels.fill(meta, self.use_cm)
        p_lats, d_lats = self.latents_from_means(*means, labels)
        return Fn.TripletSigCosFn.apply(p_lats.float(), d_lats.float(), labels.gt, float(self.m_sch_loss_fn.margin))
