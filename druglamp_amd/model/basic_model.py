"""DrugLAMPBase and the glue modules around the hot path (reference: model/basic_model.py).

Hot path (HIP): v_gca / x_gca (PGCA), v_mhla / x_mhla (MHLA), pmma (PMMA), the SSL / CM loss kernels.
Glue (torch-ROCm ops, run under bf16 autocast when the model computes in bf16): ProteinCNN, the LLM
adaptors, the BN-MLP classifier, and a DGL-free dense restatement of MolecularGCN.
"""
from __future__ import annotations

import contextlib

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn
from .. import ops
from ..configs import get_model_defaults
from .cross_modality import CrossModality
from .PGCA import GuidedCrossAttention
from .PMMA import MultiHeadLinearAttention, PairedMultimodelAttention
from .self_supervised_learning import SSL

from .._lib import FLAG_DRUG_TOKEN_PAD as _FLAG_TOK, FLAG_GCN_NODE_PAD as _FLAG_GCN   # noqa: E402

_TAIL_ROWS = 8      # rows that stand for the identical padding rows in the compact forms of the drug branch (MolecularGCN, drug LLM adaptor)

CONFIGS = {"LAMP": get_model_defaults}


def binary_cross_entropy(pred_output, labels):
    n = torch.sigmoid(pred_output.float()).squeeze(1)
    return n, F.binary_cross_entropy(n, labels.float())


def cross_entropy_logits(linear_output, label, weights=None):
    logp = F.log_softmax(linear_output.float(), dim=1)
    n = logp.exp()[:, 1]
    tgt = label.long().view(label.size(0))
    if weights is None:
        return n, F.nll_loss(logp, tgt)
    losses = F.nll_loss(logp, tgt, reduction="none")
    return n, torch.sum(weights * losses) / torch.sum(weights)


class _GraphConvDense(nn.Module):
    """weight (in, out) + bias, symmetric normalisation D^-1/2 A D^-1/2 (reference GraphConv 'both',
    basic_model.py:545-638) on a dense per-sample adjacency."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        self.bias = nn.Parameter(torch.zeros(out_feats))
        nn.init.xavier_uniform_(self.weight)

    @staticmethod
    def normalised_adjacency(adj, dtype):
        """ahat[b][i][j] = din[i] * adj[b][j][i] * dout[j] (D^-1/2 A^T D^-1/2 with clamped degrees) — computed ONCE per
        batch and shared by every layer (the adjacency carries no gradient)."""
        if not (adj.is_cuda and dtype in (torch.float32, torch.bfloat16)):
            raise RuntimeError("normalised_adjacency: a device adjacency and an fp32 / bf16 compute dtype are required "
                               "(the HIP path has no torch fallback)")
        return ops.norm_adjacency(adj if adj.dtype == torch.float32 else adj.float(), dtype)   # one launch (any graph size)

    def forward(self, ahat, feat):
        """ahat: normalised adjacency (B, Nr, Nr) with Nr <= N nodes.  Nodes >= Nr are virtual padding nodes whose
        only edge is their self loop (degree 1): for them the normalised aggregation is the identity, so only
        the real-atom block goes through the batched product."""
        agg = Fn.GraphAggregateFn.apply(ahat, feat)
        # feature transform + ReLU on the HIP GEMM path; the (in, out) parameter itself is handed over (weight_t): a
        # `.t()` view would be a new tensor object per call and miss the weight-image cache every step
        return Fn.dense(agg, self.weight, self.bias, act="relu", weight_t=True)


class _GCNLayerDense(nn.Module):
    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.graph_conv = _GraphConvDense(in_feats, out_feats)
        self.res_connection = nn.Linear(in_feats, out_feats)
        self.bn_layer = nn.BatchNorm1d(out_feats)

    def forward(self, ahat, feats, tail_weight: int = 1):
        """tail_weight > 1: compact layout — rows ahat.shape[1].. of every molecule stand for tail_weight virtual nodes each."""
        new = self.graph_conv(ahat, feats) + Fn.dense(feats, self.res_connection.weight, self.res_connection.bias, act="relu")
        B, N, C = new.shape
        if tail_weight > 1:
            return Fn.batch_norm_rows_weighted_tail(self.bn_layer, new.reshape(B * N, C), N, ahat.shape[1], tail_weight).reshape(B, N, C)
        return Fn.batch_norm_rows(self.bn_layer, new.reshape(B * N, C)).reshape(B, N, C)


class _GCNDense(nn.Module):
    def __init__(self, in_feats, hidden_feats):
        super().__init__()
        self.gnn_layers = nn.ModuleList()
        for h in hidden_feats:
            self.gnn_layers.append(_GCNLayerDense(in_feats, h))
            in_feats = h

    def forward(self, adj, feats, tail_weight: int = 1):
        ahat = _GraphConvDense.normalised_adjacency(adj, feats.dtype)
        for layer in self.gnn_layers:
            feats = layer(ahat, feats, tail_weight)
        return feats


class MolecularGCN(nn.Module):
    """MolecularGCN (basic_model.py:137-153) without DGL.  Accepts either
      * a (node_feats (B, N, in_feats), adjacency (B, Nr, Nr)) pair — batched dense graphs; the adjacency
        covers the first Nr <= N nodes (all real atoms, self loops included), the remaining nodes are the
        reference's virtual padding nodes whose only edge is their self loop — or
      * an already extracted (B, N, dim_embedding) node-feature tensor, returned unchanged
        (pre-extracted features; DGL/dgllife featurisation is out of scope).
    state_dict keys equal the reference's (init_transform, gnn.gnn_layers.i.{graph_conv,res_connection,bn_layer})."""

    def __init__(self, in_feats, dim_embedding=128, padding=True, hidden_feats=None, activation=None):
        super().__init__()
        self.init_transform = nn.Linear(in_feats, dim_embedding, bias=False)
        if padding:
            with torch.no_grad():
                self.init_transform.weight[-1].fill_(0)
        self.gnn = _GCNDense(dim_embedding, hidden_feats)
        self.output_feats = hidden_feats[-1]
        self.in_feats = in_feats
        self.compute_dtype = torch.float32
        self.compact_padding = os.environ.get("DL_GCN_COMPACT", "1") != "0"    # A/B switch for tools
        self.check_padding = os.environ.get("DL_GCN_CHECK", "0") == "1"         # debug: verify the padding rows on the host (sync)
        self.guard_padding = os.environ.get("DL_PAD_GUARD", "1") != "0"         # device-side check of the padding rows (no sync)
        self.compact_min_rows = int(os.environ.get("DL_GCN_COMPACT_MIN_ROWS", "4096"))   # padding rows saved per batch

    def forward(self, batch_graph):
        if torch.is_tensor(batch_graph):
            if batch_graph.shape[-1] != self.output_feats:
                raise ValueError("MolecularGCN: a bare tensor must be pre-extracted node features of width %d"
                                 % self.output_feats)
            return batch_graph
        node_feats, adj = batch_graph
        cdt = self.compute_dtype
        B, N, _ = node_feats.shape
        Nr = adj.shape[1]
        # Compact form (round 3).  Every node beyond the adjacency block is one of the reference's virtual padding nodes
        # (handler/dataset.py:216-221: zero features + the indicator bit, one self loop): they all carry the SAME feature
        # vector in every layer — the aggregation is the identity for them, the dense layers act per row, BatchNorm per
        # column — so the 512 - Nr of them per molecule are computed as 8 rows standing for (512 - Nr) / 8 nodes each, with
        # that weight in the BatchNorm statistics and gradients (functional.BatchNormWeightedTailFn), and expanded at the
        # end.  Same values as the 512-row computation (the BatchNorm sums associate differently); with 128-atom blocks the
        # GCN touches 136 rows per molecule instead of 512.
        TAIL = _TAIL_ROWS
        w = (N - Nr) // TAIL
        # (measured down to 32 molecules per batch: 4.16 -> 4.10 ms; tiny batches keep the plain form)
        if (self.compact_padding and N - Nr >= 2 * TAIL and (N - Nr) % TAIL == 0 and node_feats.is_cuda and
                B * (N - Nr - TAIL) >= self.compact_min_rows):
            if self.guard_padding:
                # device-side guard (round 4): one pass over the padding nodes; a node that differs sets a sticky flag
                # word that the trainer polls (ops.check_guard_flags) — a malformed batch is an error, not silent garbage
                ops.rows_equal_check(node_feats, Nr, _FLAG_GCN)
            if self.check_padding:
                pad = node_feats[:, Nr:]
                if not bool((pad == pad[:1, :1]).all()):
                    raise ValueError("MolecularGCN: the nodes beyond the adjacency block are not identical virtual padding nodes")
            h = Fn.cast(F.pad(node_feats[:, :Nr + TAIL].float(), (0, (-node_feats.shape[-1]) % 8)), cdt)
            y = self.gnn(adj, Fn.dense(h, self.init_transform.weight), tail_weight=w)          # (B, Nr + 8, C)
            out = Fn.ExpandTailFn.apply(y, Nr, w)                                              # row Nr + j <- tail row j % 8
            out._dl_tail = (y, w)                           # the distinct rows, for consumers that take multiplicities (PGCA)
            return out
        h = Fn.cast(F.pad(node_feats.float(), (0, (-node_feats.shape[-1]) % 8)), cdt)   # 75 -> 80 columns
        return self.gnn(adj, Fn.dense(h, self.init_transform.weight))


class ProteinCNN(nn.Module):
    """basic_model.py:155-180 (torch glue; conv via MIOpen).  Keeps the `view`-not-transpose output."""

    def __init__(self, embedding_dim, num_filters, kernel_size, padding=True):
        super().__init__()
        self.embedding = nn.Embedding(26 + 1, embedding_dim - 1, padding_idx=0 if padding else None)
        in_ch = [embedding_dim] + num_filters
        self.in_ch = in_ch[-1]
        self.conv1 = nn.Conv1d(in_ch[0], in_ch[1], kernel_size[0], padding="same")
        self.bn1 = nn.BatchNorm1d(in_ch[1])
        self.conv2 = nn.Conv1d(in_ch[1], in_ch[2], kernel_size[1], padding="same")
        self.bn2 = nn.BatchNorm1d(in_ch[2])
        self.conv3 = nn.Conv1d(in_ch[2], in_ch[3], kernel_size[2], padding="same")
        self.bn3 = nn.BatchNorm1d(in_ch[3])
        # A/B switch (DL_POOL_THROUGH_MAP=0): site pooling of the distinct-row output by expansion + the dense kernels
        self.pool_through_map = os.environ.get("DL_POOL_THROUGH_MAP", "1") != "0"

    compute_dtype = torch.float32

    def forward(self, v, fill_mask, site_pool=0, plan=None):
        """Embedding lookup + fill bit are torch glue; the three Conv1d + ReLU + BatchNorm1d stages run as
        channel-last implicit GEMMs + BatchNorm kernels (functional.ProteinCNNFn).  The reference's final
        `.view(B, L, C)` of the channel-first (B, C, L) buffer is reproduced exactly.  site_pool = site_len (> 0)
        additionally applies the caller's site pooling (DrugLAMP.py:39-40) inside the same kernel and returns
        (B, L // site_len, C).
        plan (protein_plan.PlanDev, round 4): the batch's distinct-row tables — the sequences are tiled with period L + 2
        (utils.py:392-412), so only ~L + 31 of the 2304 positions of a sample have distinct outputs; the network runs on
        those rows (BatchNorm weighted by the multiplicities) and the result is expanded: same values, ~3.5x fewer rows.
        The tiling is verified on the device (ops.guard_flags)."""
        from ..functional import EmbedPadFn, EmbedRowsFn, ExpandRowsFn, ProteinCNNFn, SitePoolFn, SitePoolRowsFn, cast
        ids = v.long()
        w = self.embedding.weight
        wc = cast(w, self.compute_dtype) if w.requires_grad else w.detach().to(self.compute_dtype)
        B, L = ids.shape
        params = []
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)):
            params += [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        momenta = (self.bn1.momentum, self.bn2.momentum, self.bn3.momentum) if self.training else None
        if plan is not None:
            if (plan.B, plan.S) != (B, L):
                raise ValueError("ProteinCNN: the plan is for a (%d, %d) batch, the ids are (%d, %d)" % (plan.B, plan.S, B, L))
            x = EmbedRowsFn.apply(ids, wc, fill_mask, self.embedding.padding_idx, plan.src, plan.period)     # (R, C) compact rows
            C = x.shape[-1]
            outs = ProteinCNNFn.apply(x, self.training, self.bn1.eps, 0, momenta, True, plan.w, plan.n, *params)
            if self.training:
                for bn in (self.bn1, self.bn2, self.bn3):
                    Fn.bn_tick(bn.num_batches_tracked)
            if site_pool and self.pool_through_map and ops.cnn_sitepool_rows_supported(L, int(site_pool), outs[0].shape[1], outs[0].dtype):
                # the reference's view reinterpretation + site pooling read through the row map: no (B, 2304, C) round trip
                return SitePoolRowsFn.apply(outs[0], plan.row_of, plan.rep, B, L, int(site_pool))
            z = ExpandRowsFn.apply(outs[0], plan.row_of, plan.rep).view(B, L, C)      # channel-last, every position
            if site_pool and z.dtype == torch.bfloat16 and L % int(site_pool) == 0:
                return SitePoolFn.apply(z, int(site_pool))
            z = z.transpose(1, 2).contiguous().view(B, L, C)
            if site_pool:
                z = z.view(B, int(site_pool), L // int(site_pool), C).mean(dim=1)
            return z
        x = EmbedPadFn.apply(ids, wc, fill_mask, self.embedding.padding_idx)        # (B, L + 2*HALO, C) channel-last
        C = x.shape[-1]
        # the reference's `.view(B, L, C)` of the channel-first buffer (+ the caller's site pooling) inside the tail kernel; without
        # site pooling (the masked-LM pass of the SSL epochs) that is site_len = 1: the reinterpretation alone, one launch each
        # way (round 5; two 150 MB torch copies per direction before: 0.5 ms of an SSL-epoch step at batch 256)
        fused_pool = (int(site_pool) or 1) if (x.dtype == torch.bfloat16 and L % (int(site_pool) or 1) == 0) else 0
        outs = ProteinCNNFn.apply(x, self.training, self.bn1.eps, fused_pool, momenta, True, None, 0, *params)
        z = outs[0]
        if self.training:
            for bn in (self.bn1, self.bn2, self.bn3):
                Fn.bn_tick(bn.num_batches_tracked)    # running mean / var were updated inside (dl_bn_finalize)
        if fused_pool:
            return z
        z = z.transpose(1, 2).contiguous().view(B, L, C)                            # (B, C, L) reinterpreted, like the reference
        if site_pool:
            z = z.view(B, int(site_pool), L // int(site_pool), C).mean(dim=1)
        return z


class FeedForwardLayer(nn.Module):
    def __init__(self, d_in, d_h):
        super().__init__()
        self.lin1 = nn.Linear(d_in, d_h)
        self.lin2 = nn.Linear(d_h, d_in)
        self.act = nn.GELU()
        self.norm = nn.LayerNorm(d_h)

    def forward(self, x):
        return self.lin2(self.norm(self.act(self.lin1(x))))


class MLP(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, binary=1):
        super().__init__()
        self.fc1 = nn.Linear(in_dim, hidden_dim)
        self.bn1 = nn.BatchNorm1d(hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.bn2 = nn.BatchNorm1d(hidden_dim)
        self.fc3 = nn.Linear(hidden_dim, out_dim)
        self.bn3 = nn.BatchNorm1d(out_dim)
        self.fc4 = nn.Linear(out_dim, binary)

    compute_dtype = torch.float32

    def forward(self, x):
        # HIP path only (ops._need_gpu rejects host tensors): GELU in the GEMM epilogue, BatchNorm1d by the dl_bn_*
        # kernels (fp32 statistics, running statistics updated in the finalize kernel).  On (256, <=1024) inputs the
        # torch layers are ~40 small launches of 8-20 us each; these are ~30 of 3-5 us.
        x = Fn.cast(x, self.compute_dtype)
        for fc, bn in ((self.fc1, self.bn1), (self.fc2, self.bn2), (self.fc3, self.bn3)):
            x = Fn.batch_norm_rows(bn, Fn.dense(x, fc.weight, fc.bias, act=True))
        return Fn.dense(x, self.fc4.weight, self.fc4.bias)[:, :self.fc4.out_features]


class DrugLAMPBase(nn.Module):
    def __init__(self, n_drug_feature, n_prot_feature, n_hidden=128, **cfg):
        super().__init__()
        self.site_len = cfg["PROTEIN"]["SITE_LEN"]
        self.seq_len_q = cfg["PROTEIN"]["SEQ_LEN"]
        dec = cfg["DECODER"]
        self.compact_padding = os.environ.get("DL_PAD_COMPACT", "1") != "0"      # A/B switch: compact padding rows of the drug LLM adaptor
        self.check_padding = os.environ.get("DL_PAD_CHECK", "0") == "1"         # debug: verify the padding rows on the host (sync)
        self.guard_padding = os.environ.get("DL_PAD_GUARD", "1") != "0"         # device-side check of the padding rows (no sync)
        self.compact_cnn = os.environ.get("DL_CNN_COMPACT", "1") != "0"         # A/B switch: ProteinCNN on distinct rows
        self.compact_keys = os.environ.get("DL_KEY_COMPACT", "1") != "0"        # A/B switch: PGCA over the distinct drug rows (round 5)
        # Independent branches of the forward (MolecularGCN, ProteinCNN, the two LLM adaptors, the v cross-attention branch)
        # on side HIP streams — and, through autograd, their backward passes — when the caller's hints ask for it
        # (BatchHints.branch_streams: the trainer does on cls steps).  DL_BRANCH_STREAMS=0: never.
        self.branch_streams = os.environ.get("DL_BRANCH_STREAMS", "1") != "0"
        self._streams = None
        self.drug_extractor = MolecularGCN(in_feats=cfg["DRUG"]["NODE_IN_FEATS"], dim_embedding=n_hidden,
                                           padding=cfg["DRUG"]["PADDING"], hidden_feats=[n_hidden] * 3)
        self.protein_extractor = ProteinCNN(n_hidden, [n_hidden] * 3, cfg["PROTEIN"]["KERNEL_SIZE"],
                                            cfg["PROTEIN"]["PADDING"])
        self.ssl_model = SSL(prot_extractor=self.protein_extractor, n_prot_feature=n_prot_feature,
                             drug_ssl_type=str(cfg["RS"].get("DRUG_SSL_TYPE", "simsiam")), n_hidden=n_hidden,
                             global_batch=bool(cfg["RS"].get("GLOBAL_BATCH", False)))
        self.cm_model = CrossModality(use_cm=True, hidden_size=n_hidden, max_margin=cfg["RS"]["MAX_MARGIN"],
                                      n_re=cfg["RS"]["RESET_EPOCH"], global_batch=bool(cfg["RS"].get("GLOBAL_BATCH", False)))
        if self.seq_len_q % self.site_len:
            raise ValueError("PROTEIN.SEQ_LEN (%d) must be a multiple of PROTEIN.SITE_LEN (%d)" % (self.seq_len_q, self.site_len))
        # the reference hard-codes 256 sites in get_lamp_config (default_config.py:83) although every forward derives
        # the site count from SEQ_LEN // SITE_LEN (DrugLAMP.py:35): here the PMMA tables follow the config
        model_cfg = CONFIGS["LAMP"](n_hidden, feat_len=self.seq_len_q // self.site_len)

        self.lin_d1 = nn.Linear(n_drug_feature + 1, 2 * n_hidden)
        self.act_d = nn.GELU()
        self.d_norm = nn.LayerNorm(2 * n_hidden)
        self.lin_d2 = nn.Linear(2 * n_hidden, n_hidden)

        self.p_adaptor_wo_skip_connect = FeedForwardLayer(n_prot_feature + 1, n_hidden)
        self.lin_p1 = nn.Linear(n_prot_feature + 1, 2 * n_hidden)
        self.act_p = nn.GELU()
        self.p_norm = nn.LayerNorm(2 * n_hidden)
        self.lin_p2 = nn.Linear(2 * n_hidden, n_hidden)

        self.v_gca = GuidedCrossAttention(embed_dim=n_hidden, num_heads=1)
        self.v_mhla = MultiHeadLinearAttention(d_model=n_hidden * 2, d_diff=n_hidden * 8, nhead=8,
                                               dropout=model_cfg.mlha_dropout, activation="gelu")
        self.v_gca_norm = nn.LayerNorm(n_hidden * 2)
        self.x_gca = GuidedCrossAttention(embed_dim=n_hidden, num_heads=1)
        self.x_mhla = MultiHeadLinearAttention(d_model=n_hidden * 2, d_diff=n_hidden * 8, nhead=8,
                                               dropout=model_cfg.mlha_dropout, activation="gelu")
        self.x_gca_norm = nn.LayerNorm(n_hidden * 2)

        self.pmma = PairedMultimodelAttention(config=model_cfg, vis=False)
        self.pmma.keep_compute_dtype = True            # only pooled here (Fn.TokenMeanFn): no fp32 copy of (B, L, 512)
        self.mlp_classifier = MLP(dec["IN_DIM"] * 2, dec["HIDDEN_DIM"] * 2, dec["OUT_DIM"] * 2, binary=dec["BINARY"])
        self.A_v_gca = None
        self.A_x_gca = None
        self.attn, self.guide_attn = [], []
        self.keep_raw_attention = True
        self.compute_dtype = torch.float32

    # ---- precision ---------------------------------------------------------------------------
    def set_compute_dtype(self, dtype: torch.dtype):
        """float32: exact-fp32 MFMA kernels (parity mode).  bfloat16: bf16 MFMA kernels with fp32
        accumulation / statistics, and bf16 autocast for the torch glue."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, (GuidedCrossAttention, MultiHeadLinearAttention, PairedMultimodelAttention, ProteinCNN, MolecularGCN, SSL, MLP)):
                m.compute_dtype = dtype
        return self

    def _glue(self):
        if self.compute_dtype == torch.bfloat16:
            return torch.autocast(device_type="cuda", dtype=torch.bfloat16)
        return contextlib.nullcontext()

    # ---- shared pieces of the three forwards ------------------------------------------------------
    @staticmethod
    def _fill_bit(x):
        return (x.sum(dim=-1) == 0).to(x.dtype)

    def _site_pool(self, t):
        n_site = self.seq_len_q // self.site_len
        return t.view(-1, self.site_len, n_site, t.size(-1)).mean(dim=1)

    def _gca_branch(self, gca, mhla, norm, prot_sites, drug_nodes, raw=True, tail=None):
        """PGCA -> concat -> MHLA + residual -> LayerNorm (DrugLAMP.py:55-71).  Returns (m, raw logits).
        raw=False (BatchHints.raw_attention, the trainer's steps): the (B, 1, 256, 512) fp32 pre-softmax logits the reference
        keeps on self.A_*_gca for get_cross_attn_mat (basic_model.py:123-129) are not written — 134 MB per branch and step at
        batch 256 that nothing in a training step reads."""
        want_raw = self.keep_raw_attention and raw
        if tail is not None and not want_raw and self.compact_keys:
            # Round 5: the drug side arrives from a compact padding form (MolecularGCN / the drug LLM adaptor: a block of
            # real rows + 8 rows that each stand for w identical padding rows).  The cross-attention runs over those distinct
            # rows with the multiplicity as a logit bias — the same softmax as over all 512 rows — so its in-projection,
            # core and gradients touch block + 8 key rows per molecule instead of 512 (`tail` = (rows (B, block + 8, C), w)).
            keys, w = tail
            if keys.dtype != drug_nodes.dtype:
                keys = Fn.cast(keys, drug_nodes.dtype)
            kt = keys.permute(1, 0, 2)
            m, raw = gca(prot_sites.permute(1, 0, 2), kt, kt, need_weights=False, need_raw=True, key_tail=(_TAIL_ROWS, w))
        else:
            m, raw = gca(prot_sites.permute(1, 0, 2), drug_nodes.permute(1, 0, 2), drug_nodes.permute(1, 0, 2),
                         need_weights=want_raw, need_raw=True)
        g = m.permute(1, 0, 2)
        if prot_sites.dtype != g.dtype:
            prot_sites = Fn.cast(prot_sites, g.dtype)
        m = Fn.Concat2Fn.apply(prot_sites, g)          # dl_concat2 (row widths are multiples of 16 bytes: n_hidden = 128)
        m = mhla(m, add_residual=True)                                  # mhla(h) + h in one launch set
        # inputs arrive in the compute dtype and stay in it (the LayerNorm kernel keeps fp32 statistics either way):
        # no fp32 round trips between PGCA, MHLA, LayerNorm and PMMA
        m = Fn.layer_norm(m, norm.weight, norm.bias, norm.eps)
        return m, raw

    def _ssl_drug_rows(self, vtail, xtail):
        """The block size behind which BOTH drug tensors of the SSL head (MolecularGCN output, fill-augmented LLM features) are
        identical padding rows — known when both came through the compact padding forms — or None."""
        if vtail is None or xtail is None or not self.compact_keys:
            return None
        return max(vtail[0].shape[1], xtail[0].shape[1]) - _TAIL_ROWS

    def _llm_adaptors(self, xp_cat, xd_cat, drug_tokens: int = 0):
        """xp_cat: site-pooled protein LLM features + fill bit, zero-padded (B, 256, 648); xd_cat: drug LLM
        features + fill bit, zero-padded (B, 512, 392) — both straight from ops.fill_pool, compute dtype.
        Protein / drug LLM adaptors (DrugLAMP.py:39-52) on the HIP GEMM path: 641- and 385-wide features
        are zero-padded to 648 / 392 so that every product is an aligned MFMA GEMM."""
        return self._prot_adaptor(xp_cat), self._drug_adaptor(xd_cat, drug_tokens)     # compute dtype

    def _prot_adaptor(self, xp_cat):
        xps = xp_cat                                   # site-pooled, fill-augmented, padded (B, 256, 648)
        a = self.p_adaptor_wo_skip_connect
        h = Fn.dense(xps, a.lin1.weight, a.lin1.bias, act=True)
        h = Fn.layer_norm(h, a.norm.weight, a.norm.bias, a.norm.eps)
        t = Fn.dense(h, a.lin2.weight, a.lin2.bias, residual=xps)                   # lin2(...) + xps, 648 wide
        h = Fn.dense(t, self.lin_p1.weight, self.lin_p1.bias, act=True)
        h = Fn.layer_norm(h, self.p_norm.weight, self.p_norm.bias, self.p_norm.eps)
        return Fn.dense(h, self.lin_p2.weight, self.lin_p2.bias)

    def _drug_adaptor(self, xd_cat, drug_tokens: int = 0):
        xd = xd_cat                                    # fill-augmented, padded (B, 512, 392)
        # Compact form (round 3): the token rows beyond a molecule's tokens are zero (+ the fill bit) — identical rows, and the
        # drug adaptor is row-wise (Linear, GELU, LayerNorm, Linear).  With the collate's hint `drug_tokens` (a block size that
        # covers every molecule of the batch) the rows beyond it are computed as 8 rows standing for (512 - block) / 8 rows each
        # and expanded; the expansion's backward sums the copies' gradients, which is all a row-wise layer needs.
        blk, N, TAIL = int(drug_tokens or 0), xd.shape[1], _TAIL_ROWS
        if blk and self.compact_padding and N - blk >= 2 * TAIL and (N - blk) % TAIL == 0 and blk % 8 == 0:
            if self.guard_padding:
                ops.rows_equal_check(xd, blk, _FLAG_TOK)       # device-side guard: a wrong token count is an error, not garbage
            if self.check_padding and not bool((xd[:, blk:] == xd[:1, blk:blk + 1]).all()):
                raise ValueError("drug LLM adaptor: the token rows beyond the hinted block of %d are not identical padding rows" % blk)
            xd = xd[:, :blk + TAIL].contiguous()
        h = Fn.dense(xd, self.lin_d1.weight, self.lin_d1.bias, act=True)
        h = Fn.layer_norm(h, self.d_norm.weight, self.d_norm.bias, self.d_norm.eps)
        xdf = Fn.dense(h, self.lin_d2.weight, self.lin_d2.bias)
        if xdf.shape[1] != N:
            comp = xdf
            xdf = Fn.ExpandTailFn.apply(comp, blk, (N - blk) // TAIL)
            xdf._dl_tail = (comp, (N - blk) // TAIL)        # the distinct rows, for consumers that take multiplicities (PGCA)
        return xdf

    def get_cross_attn_mat(self, modality="v"):
        if modality == "v":
            self.A_v_gca = self.A_v_gca.cpu()
            return self.A_v_gca
        self.A_x_gca = self.A_x_gca.cpu()
        return self.A_x_gca

    def get_inter_attn_mat(self):
        return self.attn, self.guide_attn

    def forward(self, vd, vp, xd, xp, mode="train", hints=None):
        pass

    def _protein_plan(self, hints, vp):
        """The compact-layout tables for this batch's ProteinCNN pass (None: every position is computed)."""
        if hints is None or hints.protein_plan is None or not self.compact_cnn or not vp.is_cuda:
            return None
        return hints.protein_plan
