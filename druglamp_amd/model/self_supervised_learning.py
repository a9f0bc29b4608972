"""SSL head (reference: model/self_supervised_learning.py).  Same constructor / forward surface and
state_dict keys.  The protein masked-LM re-runs the shared ProteinCNN (torch glue); the SimSiam row
loss and the (normally unreachable) NT-Xent loss run as HIP kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import functional as Fn


def mask_with_tokens(t, token_ids):
    m = torch.zeros_like(t, dtype=torch.bool)
    for tid in token_ids:
        m |= (t == tid)
    return m


def get_mask_subset_with_prob(mask, prob):
    """Exactly ceil(prob * n_tokens) random positions per row among mask==True (utils.py:537-551)."""
    batch, seq_len = mask.shape
    max_masked = math.ceil(prob * seq_len)
    num_tokens = mask.sum(dim=-1, keepdim=True)
    excess = (mask.cumsum(dim=-1) > (num_tokens * prob).ceil())[:, :max_masked]
    rand = torch.rand((batch, seq_len), device=mask.device).masked_fill(~mask, -1e9)
    _, idx = rand.topk(max_masked, dim=-1)
    idx = (idx + 1).masked_fill_(excess, 0)
    new_mask = torch.zeros((batch, seq_len + 1), device=mask.device)
    new_mask.scatter_(-1, idx, 1)
    return new_mask[:, 1:].bool()


def prob_mask_like(t, prob):
    return torch.zeros_like(t).float().uniform_(0, 1) < prob


def _on_hip(x, what):
    """The heads run on the HIP path only (no torch fallback: a CPU tensor or an exotic dtype is an error, not a detour)."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)):
        raise RuntimeError("%s: a device tensor in fp32 or bf16 is required (got %s on %s); druglamp_amd has no CPU path"
                           % (what, x.dtype, x.device))
    return x


def _simsiam_mlp(dim, proj_out, hidden=512):
    return nn.Sequential(nn.Linear(dim, hidden, bias=False), nn.BatchNorm1d(hidden), nn.ReLU(inplace=True),
                         nn.Linear(hidden, hidden, bias=False), nn.BatchNorm1d(hidden), nn.ReLU(inplace=True),
                         nn.Linear(hidden, proj_out, bias=False), nn.BatchNorm1d(proj_out, affine=False))


class SimProj(nn.Module):
    """Projector built lazily on first use, as in the reference (:126-143): it therefore is in no
    optimiser that was created before the first SSL forward."""

    def __init__(self, projection_out, projection_hidden_size=512):
        super().__init__()
        self.projector = None
        self.projection_out = projection_out
        self.projection_hidden_size = projection_hidden_size

    def forward(self, x, in_dim=None, tail=None):
        """x may carry zero padding columns beyond in_dim (the fill-bit-augmented LLM features are 385 -> 392 wide).
        tail: the row multiplicities of the compact drug layout (functional.run_mlp)."""
        if self.projector is None:
            self.projector = _simsiam_mlp(in_dim or x.shape[1], self.projection_out, self.projection_hidden_size).to(x.device)
        return Fn.run_mlp(self.projector, _on_hip(x, "SimProj"), tail)


class SSL(nn.Module):
    def __init__(self, prot_extractor, n_prot_feature, *, drug_ssl_type="simsiam", n_hidden=128, **kwargs):
        super().__init__()
        self.extractor = prot_extractor
        self.to_logits = nn.Linear(128, 26 + 1)
        self.llm_to_logits = nn.Linear(n_prot_feature + 1, 26 + 1)
        if drug_ssl_type not in ("simsiam", "simclr"):
            raise ValueError("drug_ssl_type must be 'simsiam' or 'simclr' (got %r)" % (drug_ssl_type,))
        self.drug_ssl_type = drug_ssl_type
        # NEW (RS.GLOBAL_BATCH): at world > 1 the NT-Xent denominator of drug_simclr runs over the all-gathered batch
        self.global_batch = bool(kwargs.get("global_batch", False))
        self.net = SimProj(n_hidden)
        self.llm_net = SimProj(n_hidden)
        if drug_ssl_type == "simsiam":
            self.predictor = nn.Sequential(nn.Linear(n_hidden, n_hidden * 4), nn.BatchNorm1d(n_hidden * 4),
                                           nn.ReLU(inplace=True), nn.Linear(n_hidden * 4, n_hidden))
        else:
            self.temperature = 0.1

    def build_projectors(self, vd_dim: int, xd_dim: int, device=None):
        """Create the lazily-built SimSiam projectors ahead of time (e.g. before load_state_dict of a
        checkpoint that contains them).  The reference creates them on the first SSL forward."""
        for proj, dim in ((self.net, vd_dim), (self.llm_net, xd_dim)):
            if proj.projector is None:
                proj.projector = _simsiam_mlp(dim, proj.projection_out, proj.projection_hidden_size).to(device)

    compute_dtype = torch.float32

    def _rows(self, vd, xd):
        """(vd rows, xd rows, xd feature width).  xd is either the reference's (B, L, 385) tensor or the
        (padded features (B, L, 392), 385) pair the DrugLAMP forward hands over."""
        xd_dim = None
        if isinstance(xd, (tuple, list)) and isinstance(xd[1], int):
            xd, xd_dim = xd
        one, two = vd.reshape(-1, vd.shape[-1]), xd.reshape(-1, xd.shape[-1])
        one, two = Fn.cast(_on_hip(one, "SSL"), self.compute_dtype), Fn.cast(_on_hip(two, "SSL"), self.compute_dtype)
        return one, two, xd_dim

    def _predict(self, x, tail=None):
        return Fn.run_mlp(self.predictor, _on_hip(x, "SSL.predictor"), tail)

    def drug_simclr(self, vd, xd):
        one, two, xd_dim = self._rows(vd, xd)
        q = self.net(one)
        k = self.llm_net(two, xd_dim)
        # rows stay in the compute dtype: bf16 rows take the bf16 matrix pipe, log-sum-exp / loss / gradients are fp32
        return Fn.NTXentFn.apply(q, k, self.temperature, self.global_batch)

    def drug_simsiam(self, vd, xd, drug_rows=None):
        """drug_rows = lead (round 5, from the model's forward when both drug tensors come from compact padding forms): rows
        lead .. 511 of every molecule are identical in vd AND in xd (virtual GCN nodes / zero token rows).  The SimSiam MLPs
        and the row loss then run on rows 0 .. lead + 7 only, the last 8 standing for (512 - lead) / 8 rows each: BatchNorm
        statistics and the loss mean carry that multiplicity (functional.run_mlp tail; same values as over all 512 rows up
        to summation order), 136 instead of 512 rows per molecule with 128-row blocks."""
        tail, wrow = None, None
        if drug_rows is not None and self.training and isinstance(xd, (tuple, list)) and vd.dim() == 3:
            N, lead, T = vd.shape[1], int(drug_rows), 8
            if 0 < lead and lead + 2 * T <= N and (N - lead) % T == 0 and xd[0].shape[1] == N:
                w = (N - lead) // T
                tail = (lead + T, lead, w)
                vd = vd[:, :lead + T]
                xd = (xd[0][:, :lead + T], xd[1])
                wrow = torch.ones(lead + T, dtype=torch.float32, device=vd.device)
                wrow[lead:] = float(w)
        one, two, xd_dim = self._rows(vd, xd)
        pred_one = self._predict(self.net(one, tail=tail), tail)
        pred_two = self._predict(self.llm_net(two, xd_dim, tail), tail)
        with torch.no_grad():
            t_one = self.net(one, tail=tail)
            t_two = self.llm_net(two, xd_dim, tail)
        rows = Fn.CosRowLossFn.apply(pred_one.float(), t_two.float()) + Fn.CosRowLossFn.apply(pred_two.float(), t_one.float())
        if wrow is None:
            return rows.mean()
        B = rows.numel() // wrow.numel()
        return (rows.view(B, -1) * wrow).sum() / float(B * N)

    def prot_mlm(self, seq, extractor, xp, fill_bit, mode, mask_ignore_token_ids=(0,), mask_prob=0.15,
                 replace_prob=0.9, pad_token_id=0, mask_token_id=26, mask=None, replace=None):
        if mask is None:
            mask = get_mask_subset_with_prob(~mask_with_tokens(seq, mask_ignore_token_ids), mask_prob)
        if replace is None:
            replace = prob_mask_like(seq, replace_prob)
        labels = seq.masked_fill(~mask, pad_token_id).long()
        masked_seq = seq.clone().detach().masked_fill(mask & replace, mask_token_id)
        loss = 0.0
        n_cls = self.to_logits.out_features

        def head(lin, h):
            # nn.Linear on the HIP GEMM path (DenseFn: output padded to 32 columns, input may carry zero padding)
            return Fn.dense(_on_hip(h, "SSL.prot_mlm"), lin.weight, lin.bias)

        def ce(lg):
            # the reference's F.cross_entropy(logits.transpose(1, 2), labels, ignore_index=0) (:93-99) = mean over the
            # non-ignored tokens of the batch, over the flattened (B*L, 32) rows of the padded head output (27 classes):
            # dl_ce_rows_* (round 5; torch's log_softmax + nll_loss took 0.9 ms per head at batch 256 — nll_loss reduces the
            # 590 k rows in one workgroup — plus the slice / fp32 copies in front of it), fixed summation order
            return Fn.CrossEntropyRowsFn.apply(lg.reshape(-1, lg.shape[-1]), labels, n_cls, pad_token_id)

        if mode != "xp":
            loss = loss + ce(head(self.to_logits, extractor(masked_seq, fill_bit)))
        if mode != "vp":
            loss = loss + ce(head(self.llm_to_logits, xp))
        return loss / 2 if mode == "double" else loss

    def forward(self, vp, xp, fill_bit_p, vd, xd, p_mode="double", mask=None, replace=None, drug_rows=None):
        if isinstance(xp, (tuple, list)):          # (embeddings (B,S,640), fill bit (B,S)) -> (B,S,641)
            # one pass: fill-bit-augmented features, zero-padded to 648 columns, compute dtype
            from .. import ops
            xp = ops.fill_pool(_on_hip(xp[0], "SSL"), 1, self.compute_dtype)[1]
        if isinstance(xd, (tuple, list)) and not isinstance(xd[1], int):
            xd = torch.cat((xd[0], xd[1].unsqueeze(-1).to(xd[0].dtype)), dim=-1)
        prot = self.prot_mlm(vp, self.extractor, xp, fill_bit_p, p_mode, mask=mask, replace=replace)
        if vd is None or xd is None:
            drug = 0
        elif self.drug_ssl_type == "simsiam":
            drug = self.drug_simsiam(vd, xd, drug_rows)
        else:
            drug = self.drug_simclr(vd, xd)
        return {"prot_ssl": prot, "drug_ssl": drug}
