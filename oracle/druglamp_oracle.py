"""CPU restatement of the DrugLAMP hot path — TEST INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (druglamp_amd/) never does and fails loudly without its HIP library.

Plain fp32 torch-CPU functions over a flat state_dict (reference key names), written from the
reference's algorithm; each function cites the reference file:line it follows.  Pinned against
the committed golden fixtures (tests/golden/*.npz) that tests/golden/make_golden.py produced by
running the REAL reference (/root/reference, imported in the build container) on inputs and weights
from oracle/detgen.py — see tests/test_oracle_golden.py.  Parts of the reference that cannot run
in the build container (DGL graph conv, Lightning trainer, dataset featurisation) are "parity
unpinned" and are not restated here, except the optimiser-step ordering of trainer.py:179-231 which
is restated in `training_step` and pinned against the same sequence driven with the reference model.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _bn(sd: SD, p: str, x: torch.Tensor, training: bool, affine: bool = True) -> torch.Tensor:
    """BatchNorm1d over dim 1 (x is (N,C) or (N,C,L)); training=True uses batch statistics."""
    w = sd[p + ".weight"] if affine else None
    b = sd[p + ".bias"] if affine else None
    return F.batch_norm(x, sd[p + ".running_mean"].detach().clone(), sd[p + ".running_var"].detach().clone(), w, b,
                        training, 0.1, 1e-5)


# ------------------------------------------------------------------------------------------
# PMMA  (model/PMMA/*.py)
# ------------------------------------------------------------------------------------------
def _heads(x: torch.Tensor, H: int) -> torch.Tensor:
    B, L, D = x.shape                                  # attention.py:38-42 transpose_for_scores
    return x.view(B, L, H, D // H).permute(0, 2, 1, 3)


def _sdpa(q, k, v):
    """softmax(q k^T / sqrt(hd)) v, heads merged back — attention.py:58-68 / 109-118."""
    hd = q.shape[-1]
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = torch.matmul(a, v).permute(0, 2, 1, 3)
    return o.reshape(o.shape[0], o.shape[1], -1), a


def pmma_attention_paired(sd: SD, p: str, prot: torch.Tensor, mol: torch.Tensor, H: int):
    """Attention.forward with mol given (attention.py:90-127) + paired_attention (44-88)."""
    q, k, v = (_heads(_lin(sd, p + "." + n, prot), H) for n in ("query", "key", "value"))
    qm, km, vm = (_heads(_lin(sd, p + "." + n, mol), H) for n in ("query_mol", "key_mol", "value_mol"))
    a, w = _sdpa(q, k, v)            # prot stream, own queries
    a_p, gw = _sdpa(qm, k, v)        # prot K/V queried by the mol stream ("guided")
    out_prot = _lin(sd, p + ".out", _lin(sd, p + ".fc", torch.cat((a, a_p), dim=-1)))
    b, _ = _sdpa(qm, km, vm)
    b_p, _ = _sdpa(q, km, vm)
    out_mol = _lin(sd, p + ".out_mol", _lin(sd, p + ".fc_mol", torch.cat((b, b_p), dim=-1)))
    return out_prot, out_mol, w, gw


def pmma_attention_self(sd: SD, p: str, x: torch.Tensor, H: int):
    q, k, v = (_heads(_lin(sd, p + "." + n, x), H) for n in ("query", "key", "value"))
    a, w = _sdpa(q, k, v)
    return _lin(sd, p + ".out", a), w


def _mlp(sd: SD, p: str, x: torch.Tensor, m1: Optional[torch.Tensor] = None, m2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mlp.py:44-50: fc1 -> GELU (erf) -> dropout -> fc2 -> dropout.  m1 / m2: the two dropout draws as multiplicative keep
    masks (0 or 1 / (1 - p)); None = eval mode / p = 0."""
    h = F.gelu(_lin(sd, p + ".fc1", x))
    if m1 is not None:
        h = h * m1
    y = _lin(sd, p + ".fc2", h)
    return y if m2 is None else y * m2


def pmma_forward(sd: SD, prot: torch.Tensor, mol: torch.Tensor, prefix: str = "", num_heads: int = 4,
                 n_layers: int = 4, return_maps: bool = False, dropout_masks: Optional[Dict[str, torch.Tensor]] = None):
    """PairedMultimodelAttention.forward (paired_multi_model_attention_model.py:22-29,
    embed.py:38-54, encoder.py:41-56, block.py:33-62).  The `embedding(prot)` Linear of embed.py:50
    is computed and discarded by the reference and is skipped here.
    dropout_masks=None: eval mode.  Training mode takes the Bernoulli draws of every dropout site as DATA — multiplicative
    keep masks keyed emb_mol / emb_prot (embed.py:42,52: after the positional add) and l{i}.s{s}.fc1 / .fc2 (mlp.py:47,49;
    s = 0: prot / the single stream, 1: mol); pinned by tests/golden/pmma_drop.npz (the reference in train mode with its
    nn.Dropout modules fed the same masks)."""
    dm = dropout_masks or {}
    e = prefix + "embeddings."
    mol = _lin(sd, e + "mol_embeddings", mol) + sd[e + "pe_mol"]
    prot = prot + sd[e + "pe_prot"]
    if "emb_mol" in dm:
        mol = mol * dm["emb_mol"]
    if "emb_prot" in dm:
        prot = prot * dm["emb_prot"]
    maps = []
    x = None
    for i in range(n_layers):
        p = "%sencoder.layer_with_mol.%d" % (prefix, i)
        if i < 2:
            h, hm = prot, mol
            ap, am, w, gw = pmma_attention_paired(sd, p + ".attn", _ln(sd, p + ".attention_norm", prot, 1e-6),
                                                  _ln(sd, p + ".att_norm_mol", mol, 1e-6), num_heads)
            prot, mol = ap + h, am + hm
            prot = _mlp(sd, p + ".ffn", _ln(sd, p + ".ffn_norm", prot, 1e-6), dm.get("l%d.s0.fc1" % i), dm.get("l%d.s0.fc2" % i)) + prot
            mol = _mlp(sd, p + ".ffn_mol", _ln(sd, p + ".ffn_norm_mol", mol, 1e-6), dm.get("l%d.s1.fc1" % i), dm.get("l%d.s1.fc2" % i)) + mol
            maps.append((w, gw))
        else:
            if i == 2:
                x = torch.cat((prot, mol), dim=-1)
            a, w = pmma_attention_self(sd, p + ".attn", _ln(sd, p + ".attention_norm", x, 1e-6), num_heads)
            x = a + x
            x = _mlp(sd, p + ".ffn", _ln(sd, p + ".ffn_norm", x, 1e-6), dm.get("l%d.s0.fc1" % i), dm.get("l%d.s0.fc2" % i)) + x
            maps.append((w, None))
    out = _ln(sd, prefix + "encoder.encoder_norm", x, 1e-6)
    return (out, maps) if return_maps else out


# ------------------------------------------------------------------------------------------
# MHLA (model/PMMA/encoder.py:88-140)
# ------------------------------------------------------------------------------------------
def mhla_forward(sd: SD, p: str, v: torch.Tensor, nhead: int = 8) -> torch.Tensor:
    g = _lin(sd, p + ".lin2", F.gelu(_lin(sd, p + ".lin1", v)))       # (B, L, H)
    g = torch.softmax(g, dim=1).transpose(1, 2).contiguous()            # softmax over L -> (B, H, L)
    B, L, D = v.shape
    # the reference does v.view(B*H, L, D/H) on the CONTIGUOUS (B, L, D) buffer: a flat reinterpretation
    out = g.view(B * nhead, L, 1) * v.contiguous().view(B * nhead, L, D // nhead)
    return out.view(B, L, D)


# ------------------------------------------------------------------------------------------
# PGCA (model/PGCA/guided_cross_attention_model.py:15-329, branch "key is value", 1 head)
# ------------------------------------------------------------------------------------------
def pgca_forward(sd: SD, p: str, query: torch.Tensor, key: torch.Tensor, num_heads: int = 1):
    """query (Lq, B, E), key == value (Lk, B, E) -> (out (Lq, B, E), raw logits (B, H, Lq, Lk))."""
    Lq, B, E = query.shape
    Lk = key.shape[0]
    hd = E // num_heads
    W, bias = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(query, W[:E], bias[:E]) * (float(hd) ** -0.5)           # :146-147, :212
    kv = F.linear(key, W[E:], bias[E:])                                   # :155-161
    k, v = kv.chunk(2, dim=-1)
    q = q.contiguous().view(Lq, B * num_heads, hd).transpose(0, 1)
    k = k.contiguous().view(Lk, B * num_heads, hd).transpose(0, 1)
    v = v.contiguous().view(Lk, B * num_heads, hd).transpose(0, 1)
    raw = torch.bmm(q, k.transpose(1, 2))                                 # :290, saved raw at :307
    o = torch.bmm(torch.softmax(raw, dim=-1), v)
    o = o.transpose(0, 1).contiguous().view(Lq, B, E)
    return _lin(sd, p + ".out_proj", o), raw.view(B, num_heads, Lq, Lk)


# ------------------------------------------------------------------------------------------
# model glue (model/basic_model.py, model/DrugLAMP*.py)
# ------------------------------------------------------------------------------------------
def protein_cnn(sd: SD, p: str, ids: torch.Tensor, fill: torch.Tensor, bn_training: bool) -> torch.Tensor:
    """ProteinCNN.forward (basic_model.py:172-180), including the view-not-transpose at :179."""
    v = F.embedding(ids.long(), sd[p + ".embedding.weight"], padding_idx=0)
    v = torch.cat((v, fill.unsqueeze(-1).to(v.dtype)), dim=-1).transpose(2, 1)
    for i in (1, 2, 3):
        v = F.conv1d(v, sd["%s.conv%d.weight" % (p, i)], sd["%s.conv%d.bias" % (p, i)], padding="same")
        v = _bn(sd, "%s.bn%d" % (p, i), F.relu(v), bn_training)
    return v.reshape(v.size(0), v.size(2), -1)


def molecular_gcn(sd: SD, p: str, h: torch.Tensor, adj: torch.Tensor, bn_training: bool) -> torch.Tensor:
    """MolecularGCN.forward (basic_model.py:137-153) -> GCN / GCNLayer (:342-436) -> GraphConv norm='both' (:545-638) on a
    batch of DENSE graphs.  h: (B, N, 75) node features (N = 512: real atoms first, then the virtual padding nodes of
    handler/dataset.py:211-222); adj: (B, n, n) edge counts adj[b][u][v] = number of edges u -> v among the first n nodes,
    self loops included (real atoms carry TWO: smiles_to_bigraph(add_self_loop=True), then add_self_loop() again); nodes
    >= n have exactly one edge, their self loop.  GraphConv (:594-631):
        rst[v] = in_deg[v]^-1/2 * sum_{u -> v} out_deg[u]^-1/2 * feat[u]  (W applied after the sum: in_feats == out_feats)
    with degrees clamped at 1; dgl's `update_all(copy_u, sum)` is the sum over incoming edges, written here as a dense
    product with the edge-count matrix.  Pinned by tests/golden/gcn.npz: the reference's own GCN classes driven through
    a scipy.sparse stand-in for the DGL graph (tests/golden/make_golden.py gen_gcn)."""
    B, N, _ = h.shape
    n = adj.shape[-1]
    A = torch.zeros(B, N, N, dtype=h.dtype)
    A[:, :n, :n] = adj
    idx = torch.arange(n, N)
    A[:, idx, idx] = 1.0
    dout = A.sum(-1).clamp(min=1).pow(-0.5).unsqueeze(-1)             # out-degree of the source
    din = A.sum(-2).clamp(min=1).pow(-0.5).unsqueeze(-1)              # in-degree of the destination
    x = h @ sd[p + ".init_transform.weight"].t()
    i = 0
    while "%s.gnn.gnn_layers.%d.graph_conv.weight" % (p, i) in sd:
        q = "%s.gnn.gnn_layers.%d" % (p, i)
        rst = torch.bmm(A.transpose(1, 2), x * dout) @ sd[q + ".graph_conv.weight"]
        conv = F.relu(rst * din + sd[q + ".graph_conv.bias"])                         # activation=F.relu inside GraphConv
        new = conv + F.relu(_lin(sd, q + ".res_connection", x))                      # GCNLayer: residual, dropout 0
        x = _bn(sd, q + ".bn_layer", new.reshape(B * N, -1), bn_training).reshape(B, N, -1)
        i += 1
    return x


def _fill_bit(x: torch.Tensor) -> torch.Tensor:
    return (x.sum(dim=-1) == 0).to(x.dtype)                               # DrugLAMP.py:11-19


def classifier(sd: SD, p: str, f: torch.Tensor, bn_training: bool) -> torch.Tensor:
    x = _bn(sd, p + ".bn1", F.gelu(_lin(sd, p + ".fc1", f)), bn_training)   # basic_model.py:209-214
    x = _bn(sd, p + ".bn2", F.gelu(_lin(sd, p + ".fc2", x)), bn_training)
    x = _bn(sd, p + ".bn3", F.gelu(_lin(sd, p + ".fc3", x)), bn_training)
    return _lin(sd, p + ".fc4", x)


def model_forward(sd: SD, kind: str, vd: torch.Tensor, vp: torch.Tensor, xd: Optional[torch.Tensor],
                  xp: torch.Tensor, bn_training: bool = False, site_len: int = 9, seq_len: int = 2304):
    """DrugLAMP / DrugLAMP2C2P / DrugLAMPwoLLM forward (DrugLAMP.py:8-78, DrugLAMP2C2P.py:8-89,
    DrugLAMPwoLLM.py:8-51) with the drug GCN bypassed: `vd` is the (B, 512, 128) node-feature
    tensor the MolecularGCN would return (DGL is absent in the build container).
    Returns dict(score, vd, vp, ssl, cm, A_v, A_x)."""
    fill_p = _fill_bit(xp)
    ssl = {"vp": vp, "fill_bit_p": fill_p, "vd": vd}
    if kind != "DrugLAMPwoLLM":
        xp = torch.cat((xp, fill_p.unsqueeze(-1)), dim=-1)
        xd = torch.cat((xd, _fill_bit(xd).unsqueeze(-1)), dim=-1)
        ssl.update(xp=xp, xd=xd)
    else:
        ssl.update(xp=None, xd=None, p_mode="vp")
    vpf = protein_cnn(sd, "protein_extractor", vp, fill_p, bn_training)
    n_site = seq_len // site_len
    vpf = vpf.view(-1, site_len, n_site, vpf.shape[-1]).mean(dim=1)       # DrugLAMP.py:35-37
    out = {"ssl": ssl, "cm": None, "A_x": None}
    mv, A_v = pgca_forward(sd, "v_gca", vpf.permute(1, 0, 2), vd.permute(1, 0, 2))
    mv = torch.cat((vpf, mv.permute(1, 0, 2)), 2)
    mv = _ln(sd, "v_gca_norm", mhla_forward(sd, "v_mhla", mv) + mv, 1e-5)
    if kind == "DrugLAMPwoLLM":
        f = pmma_forward(sd, mv, mv, prefix="pmma.")                      # DrugLAMPwoLLM.py:46
    else:
        xps = xp.view(-1, site_len, n_site, xp.shape[-1]).mean(dim=1)
        a = "p_adaptor_wo_skip_connect"
        t = _lin(sd, a + ".lin2", _ln(sd, a + ".norm", F.gelu(_lin(sd, a + ".lin1", xps)), 1e-5)) + xps
        xpf = _lin(sd, "lin_p2", _ln(sd, "p_norm", F.gelu(_lin(sd, "lin_p1", t)), 1e-5))
        xdf = _lin(sd, "lin_d2", _ln(sd, "d_norm", F.gelu(_lin(sd, "lin_d1", xd)), 1e-5))
        if kind == "DrugLAMP2C2P":
            out["cm"] = {"prot": vpf, "aug_prot": xpf, "drug": vd, "aug_drug": xdf}
        mx, A_x = pgca_forward(sd, "x_gca", xpf.permute(1, 0, 2), xdf.permute(1, 0, 2))
        mx = torch.cat((xpf, mx.permute(1, 0, 2)), 2)
        mx = _ln(sd, "x_gca_norm", mhla_forward(sd, "x_mhla", mx) + mx, 1e-5)
        out["A_x"] = A_x
        f = pmma_forward(sd, mx, mv, prefix="pmma.")                      # DrugLAMP.py:73 (prot=mx, mol=mv)
    score = classifier(sd, "mlp_classifier", f.mean(dim=1), bn_training)
    out.update(score=score, vd=vd, vp=vpf, A_v=A_v)
    return out


def bce_loss(score: torch.Tensor, labels: torch.Tensor):
    n = torch.sigmoid(score).squeeze(1)                                   # basic_model.py:17-22
    return n, F.binary_cross_entropy(n, labels.to(n.dtype))


# ------------------------------------------------------------------------------------------
# SSL (model/self_supervised_learning.py)
# ------------------------------------------------------------------------------------------
def nt_xent(q: torch.Tensor, k: torch.Tensor, temperature: float = 0.1) -> torch.Tensor:
    """nt_xent_loss (self_supervised_learning.py:168-182): no l2-norm, diagonal removed, sum/2n."""
    n = q.shape[0]
    P = torch.cat((q, k))
    logits = (P @ P.t()) / temperature
    logits = logits.masked_fill(torch.eye(2 * n, dtype=torch.bool), float("-inf"))
    pos = torch.cat((torch.arange(n) + n, torch.arange(n)))
    return (torch.logsumexp(logits, dim=1) - logits[torch.arange(2 * n), pos]).sum() / (2 * n)


def cos_rowloss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return 2 - 2 * (F.normalize(x, dim=-1) * F.normalize(y, dim=-1)).sum(dim=-1)     # :184-187


def _simsiam_mlp(sd: SD, p: str, x: torch.Tensor, bn_training: bool) -> torch.Tensor:
    """SimSiamMLP (:155-166): Linear-BN-ReLU x2, Linear, BN(affine=False); all Linear bias-free."""
    x = F.relu(_bn(sd, p + ".1", F.linear(x, sd[p + ".0.weight"]), bn_training))
    x = F.relu(_bn(sd, p + ".4", F.linear(x, sd[p + ".3.weight"]), bn_training))
    return _bn(sd, p + ".7", F.linear(x, sd[p + ".6.weight"]), bn_training, affine=False)


def _predictor(sd: SD, p: str, x: torch.Tensor, bn_training: bool) -> torch.Tensor:
    return _lin(sd, p + ".3", F.relu(_bn(sd, p + ".1", _lin(sd, p + ".0", x), bn_training)))   # :145-153


def simsiam_loss(sd: SD, p: str, vd: torch.Tensor, xd: torch.Tensor, bn_training: bool = True) -> torch.Tensor:
    """SSL.drug_simsiam (:43-65). Targets are the same projectors under no_grad."""
    one, two = vd.reshape(-1, vd.shape[-1]), xd.reshape(-1, xd.shape[-1])
    proj1 = _simsiam_mlp(sd, p + ".net.projector", one, bn_training)
    proj2 = _simsiam_mlp(sd, p + ".llm_net.projector", two, bn_training)
    pred1, pred2 = _predictor(sd, p + ".predictor", proj1, bn_training), _predictor(sd, p + ".predictor", proj2, bn_training)
    with torch.no_grad():
        t1 = _simsiam_mlp(sd, p + ".net.projector", one, bn_training)
        t2 = _simsiam_mlp(sd, p + ".llm_net.projector", two, bn_training)
    return (cos_rowloss(pred1, t2) + cos_rowloss(pred2, t1)).mean()


def prot_mlm_loss(sd: SD, p: str, seq: torch.Tensor, xp: Optional[torch.Tensor], fill_bit: torch.Tensor, mode: str,
                  mask: torch.Tensor, replace: torch.Tensor, bn_training: bool = True) -> torch.Tensor:
    """SSL.prot_mlm (:67-101) with the random draws (`mask` from get_mask_subset_with_prob, `replace`
    from prob_mask_like) passed in so that the result is a pure function of its inputs."""
    labels = seq.masked_fill(~mask, 0).long()
    masked = seq.clone().masked_fill(mask & replace, 26)
    loss = 0.0
    if mode != "xp":
        emb = protein_cnn(sd, p + ".extractor", masked, fill_bit, bn_training)
        loss = loss + F.cross_entropy(_lin(sd, p + ".to_logits", emb).transpose(1, 2), labels, ignore_index=0)
    if mode != "vp":
        loss = loss + F.cross_entropy(_lin(sd, p + ".llm_to_logits", xp).transpose(1, 2), labels, ignore_index=0)
    return loss / 2 if mode == "double" else loss


# ------------------------------------------------------------------------------------------
# Cross-modality (model/cross_modality.py, utils.py:559-574)
# ------------------------------------------------------------------------------------------
def tanh_decay_margin(m_ori: float, n_re: int, step: int) -> float:
    return float(m_ori * (1 - np.tanh(2 * (1 - step / n_re))))            # utils.py:559-560


class MarginSchedule:
    """MarginScheduledLossFunction (cross_modality.py:49-102): starts at m_ori (NOT tanh_decay(0)),
    k-th step() -> tanh_decay(k), reset to tanh_decay(0) when k == n_re."""

    def __init__(self, m_ori: float = 0.5, n_re: int = 100):
        self.m_ori, self.n_re, self._step, self.margin = m_ori, n_re, 0, m_ori

    def step(self):
        self._step += 1
        if self._step == self.n_re:
            self._step = 0
        self.margin = tanh_decay_margin(self.m_ori, self.n_re, self._step)


def cm_label_matrix(meta: List[dict]):
    """cross_modality.py:138-150: unique ids in first-seen order, LAST occurrence index per id;
    unobserved (prot, drug) pairs count as negatives (default cell 0)."""
    pid2t = {m["Prot_ID"]: t for t, m in enumerate(meta)}
    did2t = {m["Drug_ID"]: t for t, m in enumerate(meta)}
    pids, dids = list(pid2t), list(did2t)
    gt = np.zeros((len(pids), len(dids)), dtype=np.int8)
    pi = {p: i for i, p in enumerate(pids)}
    di = {d: i for i, d in enumerate(dids)}
    for m in meta:
        gt[pi[m["Prot_ID"]], di[m["Drug_ID"]]] = int(m["Y"])
    return list(pid2t.values()), list(did2t.values()), gt


def triplet_sigcos(p_lats: torch.Tensor, d_lats: torch.Tensor, gt: np.ndarray, margin: float) -> torch.Tensor:
    """ccpp_p_tri_loss (cross_modality.py:15-47) with D = 1 - sigmoid(cos) (utils.py:571-574)."""
    def dist(a, b):
        return 1 - torch.sigmoid(F.cosine_similarity(a, b))
    total, n_tri = torch.zeros(()), 0
    for i in range(gt.shape[0]):
        pos = [j for j in range(gt.shape[1]) if gt[i, j] == 1]
        neg = [j for j in range(gt.shape[1]) if gt[i, j] == 0]
        if pos and neg:
            pj = [a for a in pos for _ in neg]
            nj = [b for _ in pos for b in neg]
            a = p_lats[[i] * len(pj)]
            total = total + F.relu(dist(a, d_lats[pj]) - dist(a, d_lats[nj]) + margin).sum()
            n_tri += len(pj)
        elif neg:
            a = p_lats[[i] * len(neg)]
            total = total + F.relu(dist(a, a) - dist(a, d_lats[neg]) + margin).sum()
            n_tri += len(neg)
    return total / max(n_tri, 1)


def cm_forward(sd: SD, p: str, prot, aug_prot, drug, aug_drug, meta: List[dict], margin: float,
               bn_training: bool = True, return_latents: bool = False):
    """CrossModality.forward (cross_modality.py:129-164); Mean2Embed = BN -> ReLU -> Linear (:166-171)."""
    pidx, didx, gt = cm_label_matrix(meta)

    def m2e(name, x):
        return _lin(sd, "%s.%s.2" % (p, name), F.relu(_bn(sd, "%s.%s.0" % (p, name), x, bn_training)))
    pe = torch.cat([m2e("prot2latent", prot[pidx].mean(dim=1)), m2e("aug_prot2latent", aug_prot[pidx].mean(dim=1))], -1)
    de = torch.cat([m2e("drug2latent", drug[didx].mean(dim=1)), m2e("aug_drug2latent", aug_drug[didx].mean(dim=1))], -1)
    pl = F.normalize(F.linear(pe, sd[p + ".to_prot_latent.weight"]), dim=-1)
    dl = F.normalize(F.linear(de, sd[p + ".to_drug_latent.weight"]), dim=-1)
    loss = triplet_sigcos(pl, dl, gt, margin)
    return (loss, pl, dl, gt) if return_latents else loss


# ------------------------------------------------------------------------------------------
# training step (trainer.py:179-231) — also the CPU baseline timed by bench.py
# ------------------------------------------------------------------------------------------
class OracleTrainer:
    """The reference's manual-optimisation step on the functional oracle model: three torch AdamW over
    the SAME parameter list (main.py:158-160), zero_grad before each loss's backward, cm-weight
    auto-scale at cur_epoch == INIT_EPOCH.  Lazily created SimSiam projectors are in no optimiser."""

    def __init__(self, sd: SD, kind: str, lr=1e-4, ssl_lr=3e-5, cm_lr=3e-5, use_ssl=True, use_cm=True,
                 epoch_step=5, init_epoch=5, max_margin=0.5, n_re=100):
        self.sd, self.kind = sd, kind
        self.params = []
        for k, v in sd.items():
            if v.is_floating_point() and "running_" not in k and ".projector." not in k:
                v.requires_grad_(True)
                self.params.append(v)
        # `protein_extractor.*` and `ssl_model.extractor.*` are the same tensors: optimise them once
        uniq, seen = [], set()
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.opt = torch.optim.AdamW(self.params, lr=lr)
        self.opt_ssl = torch.optim.AdamW(self.params, lr=ssl_lr) if use_ssl else None
        self.opt_cm = torch.optim.AdamW(self.params, lr=cm_lr) if use_cm else None
        self.use_ssl, self.use_cm, self.epoch_step, self.init_epoch = use_ssl, use_cm, epoch_step, init_epoch
        self.margin = MarginSchedule(max_margin, n_re)
        self.cm_weight = 1.0

    def step(self, vd, vp, xd, xp, y, meta=None, cur_epoch=1, mask=None, replace=None):
        compute_ssl = self.use_ssl and cur_epoch % self.epoch_step == 0
        compute_cm = self.use_cm and cur_epoch >= self.init_epoch
        out = model_forward(self.sd, self.kind, vd, vp, xd, xp, bn_training=True)
        self.opt.zero_grad()
        _, cls = bce_loss(out["score"], y)
        cls.backward(retain_graph=compute_ssl or compute_cm)
        rec = {"cls": float(cls), "ssl": 0.0, "cm": 0.0}
        if compute_ssl:
            self.opt_ssl.zero_grad()
            s = out["ssl"]
            mode = s.get("p_mode", "double")
            prot = prot_mlm_loss(self.sd, "ssl_model", s["vp"], s["xp"], s["fill_bit_p"], mode, mask, replace)
            drug = simsiam_loss(self.sd, "ssl_model", s["vd"], s["xd"]) if s["xd"] is not None else 0.0
            ssl = (prot + drug) * 0.1
            ssl.backward(retain_graph=compute_cm)
            rec["ssl"] = float(ssl)
        if compute_cm:
            self.opt_cm.zero_grad()
            cm = cm_forward(self.sd, "cm_model", **out["cm"], meta=meta, margin=self.margin.margin)
            if cur_epoch == self.init_epoch and float(cm) > 0:
                while float(cm) * self.cm_weight / 10 > float(cls):
                    self.cm_weight /= 10
                while float(cm) * self.cm_weight * 10 < float(cls):
                    self.cm_weight *= 10
            cm = cm * self.cm_weight
            cm.backward()
            rec["cm"] = float(cm)
        self.opt.step()
        if compute_ssl:
            self.opt_ssl.step()
        if compute_cm:
            self.opt_cm.step()
        rec["cm_weight"] = self.cm_weight
        return rec
