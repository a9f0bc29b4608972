"""Cross-modality 2C2P head (reference: model/cross_modality.py).  Label-matrix construction stays on
the host (python dicts in the reference, one numpy matrix here); the O(n_p * n_d) python triplet loop
of ccpp_p_tri_loss is replaced by the all-pairs sigmoid-cosine + masked triplet HIP kernel."""
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

    def step(self):
        self.m_sch_loss_fn.step()

    def latents_from_means(self, pm, apm, dm, adm, pidx, didx):
        """pm/apm/dm/adm: per-sample token means (n, hidden).  Mean2Embed x4 -> concat -> Linear -> l2norm."""
        pe = torch.cat([self.prot2latent(pm[pidx]), self.aug_prot2latent(apm[pidx])], dim=-1)
        de = torch.cat([self.drug2latent(dm[didx]), self.aug_drug2latent(adm[didx])], dim=-1)
        return F.normalize(self.to_prot_latent(pe), dim=-1), F.normalize(self.to_drug_latent(de), dim=-1)

    def forward(self, prot, aug_prot, drug, aug_drug, meta):
        means = [t.float().mean(dim=1) for t in (prot, aug_prot, drug, aug_drug)]
        if self.global_batch and dist_ops.world_size() > 1:
            # NEW vs the reference (off by default): the label matrix and the triplets span the GLOBAL batch.
            # Only the (n, hidden) token means and the ids travel: 4 x 128 floats per pair over xGMI.
            means = [dist_ops.all_gather_rows(m) for m in means]
            meta = dist_ops.all_gather_meta(meta)
        pidx, didx, gt = label_matrix(meta, self.use_cm)
        p_lats, d_lats = self.latents_from_means(*means, pidx, didx)
        gt_dev = torch.from_numpy(gt).to(p_lats.device)
        return Fn.TripletSigCosFn.apply(p_lats.float(), d_lats.float(), gt_dev, float(self.m_sch_loss_fn.margin))
