"""PMMA — paired multi-modal attention, and the MHLA token gate (reference: model/PMMA/*.py).

Same class names, constructor arguments, forward signatures and state_dict keys as the reference;
the modules below only HOLD parameters — all arithmetic is done by the block-level HIP Functions in
druglamp_amd/functional.py (one autograd node per transformer block).
"""
from __future__ import annotations

import copy

import torch
import torch.nn as nn

from .. import functional as Fn


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) and name in cfg else getattr(cfg, name)


class Attention(nn.Module):
    """Parameter holder for attention.py:6-36 (query/key/value[/_mol], fc[/_mol], out[/_mol])."""

    def __init__(self, config, vis, mm=True):
        super().__init__()
        self.vis = vis
        hs = config.hidden_size
        self.num_attention_heads = _cfg_get(config.transformer, "num_heads")
        self.attn_head_size = int(hs / self.num_attention_heads)
        self.query, self.key, self.value = nn.Linear(hs, hs), nn.Linear(hs, hs), nn.Linear(hs, hs)
        if mm:
            self.query_mol, self.key_mol, self.value_mol = nn.Linear(hs, hs), nn.Linear(hs, hs), nn.Linear(hs, hs)
            self.out_mol = nn.Linear(hs, hs)
            self.fc = nn.Linear(hs * 2, hs)
            self.fc_mol = nn.Linear(hs * 2, hs)
        self.out = nn.Linear(hs, hs)


class Mlp(nn.Module):
    """Parameter holder for mlp.py:29-42 (xavier-uniform weights, N(0, 1e-6) biases)."""

    def __init__(self, config):
        super().__init__()
        self.fc1 = nn.Linear(config.hidden_size, config.hidden_size * 4)
        self.fc2 = nn.Linear(config.hidden_size * 4, config.hidden_size)
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)
        nn.init.normal_(self.fc1.bias, std=1e-6)
        nn.init.normal_(self.fc2.bias, std=1e-6)


class PMMABlock(nn.Module):
    """block.py:19-31.  `stream_params(s)` lists one stream's tensors in the order
    functional.TransformerBlockFn expects."""

    def __init__(self, config, vis, mm=False):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.mm = mm
        self.attention_norm = nn.LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn_norm = nn.LayerNorm(config.hidden_size, eps=1e-6)
        if mm:
            self.att_norm_mol = nn.LayerNorm(config.hidden_size, eps=1e-6)
            self.ffn_norm_mol = nn.LayerNorm(config.hidden_size, eps=1e-6)
            self.ffn_mol = Mlp(config)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis, mm)

    def stream_params(self, s: int):
        a = self.attn
        if s == 0:
            ln1, ln2, ffn = self.attention_norm, self.ffn_norm, self.ffn
            q, k, v, out = a.query, a.key, a.value, a.out
            fc = a.fc if self.mm else None
        else:
            ln1, ln2, ffn = self.att_norm_mol, self.ffn_norm_mol, self.ffn_mol
            q, k, v, out, fc = a.query_mol, a.key_mol, a.value_mol, a.out_mol, a.fc_mol
        ps = [ln1.weight, ln1.bias, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias]
        if fc is not None:
            ps += [fc.weight, fc.bias]
        ps += [out.weight, out.bias, ln2.weight, ln2.bias, ffn.fc1.weight, ffn.fc1.bias, ffn.fc2.weight, ffn.fc2.bias]
        return ps


class Embeddings(nn.Module):
    """embed.py:20-54.  `embedding` is dead in the reference (its output is overwritten at :51) and is
    kept only so that state_dicts stay interchangeable; it never runs here."""

    def __init__(self, config, mol_len):
        super().__init__()
        self.embedding = nn.Linear(config.hidden_size, config.hidden_size)
        self.mol_embeddings = nn.Linear(config.hidden_size, config.hidden_size)
        self.pe_prot = nn.Parameter(torch.zeros(1, config.feat_len, config.hidden_size))
        self.pe_mol = nn.Parameter(torch.zeros(1, mol_len, config.hidden_size))
        self.p_drop = _cfg_get(config.transformer, "dropout_rate")


class Encoder(nn.Module):
    """encoder.py:26-56: layers 0,1 paired at width d, then channel-concat, remaining layers plain self
    blocks at width 2d; final LayerNorm(2d, eps 1e-6).  As in the reference, `config.hidden_size` is
    doubled IN PLACE when layer 2 is built (encoder.py:36-37)."""

    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.layer_with_mol = nn.ModuleList()
        self.encoder_norm = nn.LayerNorm(config.hidden_size * 2, eps=1e-6)
        for i in range(_cfg_get(config.transformer, "num_p_plus_s_layers")):
            if i < 2:
                layer = PMMABlock(config, vis, mm=True)
            else:
                if i == 2:
                    config.hidden_size = config.hidden_size * 2
                layer = PMMABlock(config, vis)
            self.layer_with_mol.append(copy.deepcopy(layer))


class PairedMultimodelAttention(nn.Module):
    """paired_multi_model_attention_model.py:15-29.  forward(prot, mol) -> (encoded, [], []).

    Attention maps (`vis=True` in the reference) are never materialised by the fused kernels; with
    vis=True the two lists are filled by a separate, non-differentiable probability computation."""

    def __init__(self, config, vis=True):
        super().__init__()
        self.num_heads = _cfg_get(config.transformer, "num_heads")
        self.p_drop = _cfg_get(config.transformer, "dropout_rate")
        self.embeddings = Embeddings(config, mol_len=config.mol_len)
        self.encoder = Encoder(config, vis)
        self.vis = vis
        self.compute_dtype = torch.float32
        self.keep_compute_dtype = False

    def forward(self, prot, mol=None):
        if mol is None:
            raise ValueError("PairedMultimodelAttention needs both streams (the reference's mol=None path "
                             "fails at encoder.py:50 as well)")
        if prot.shape != mol.shape:
            raise RuntimeError("PMMA paired attention needs equal-shaped streams, got %s and %s"
                               % (tuple(prot.shape), tuple(mol.shape)))
        cdt = self.compute_dtype
        emb, enc = self.embeddings, self.encoder
        H, p, tr = self.num_heads, self.p_drop, self.training
        prot = Fn.cast(prot, cdt)
        mol = Fn.cast(mol, cdt)
        mol = Fn.LinearFn.apply(mol, emb.mol_embeddings.weight, emb.mol_embeddings.bias, emb.pe_mol, p, tr)
        prot = Fn.AddPeDropoutFn.apply(prot, emb.pe_prot, p, tr)
        x = torch.stack((prot, mol), dim=0)                      # [2, B, L, d]
        attn_maps, guided_maps = [], []
        for i, blk in enumerate(enc.layer_with_mol):
            if i < 2:
                if self.vis:
                    w, gw = Fn.attention_maps(x, blk, H, paired=True)
                    attn_maps.append(w)
                    guided_maps.append(gw)
                x = Fn.transformer_block(x, True, H, p, tr, 1e-6, blk.stream_params(0) + blk.stream_params(1))
            else:
                if i == 2:
                    x = _concat_streams(x)                             # [1, B, L, 2d]
                if self.vis:
                    w, _ = Fn.attention_maps(x, blk, H, paired=False)
                    attn_maps.append(w)
                    guided_maps.append(None)
                x = Fn.transformer_block(x, False, H, p, tr, 1e-6, blk.stream_params(0))
        if x.shape[0] == 2:                                       # fewer than 3 layers configured
            x = _concat_streams(x)
        encoded = Fn.layer_norm(x.reshape(x.shape[1:]), enc.encoder_norm.weight, enc.encoder_norm.bias, 1e-6)
        # fp32 out like the reference; the DrugLAMP models, which only pool `encoded`, ask for the compute dtype
        return (encoded if self.keep_compute_dtype else Fn.cast(encoded, torch.float32)), attn_maps, guided_maps


class _ConcatStreamsFn(torch.autograd.Function):
    """[2, B, L, d] -> [1, B, L, 2d] = cat((x[0], x[1]), -1) (encoder.py:50): one dl_interleave_streams launch each way
    (torch's strided copies of the same 67 MB ran at 1.2-2 TB/s; indexing x[0] / x[1] additionally costs two zero
    fills and an add in backward)."""

    @staticmethod
    def forward(ctx, x):
        from .. import ops
        return ops.interleave_streams(x).unsqueeze(0)

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        return ops.interleave_streams(g[0], inverse=True)


def _concat_streams(x):
    S, B, L, d = x.shape
    if S == 2 and x.is_cuda and (d * x.element_size()) % 16 == 0:
        return _ConcatStreamsFn.apply(x)
    return x.permute(1, 2, 0, 3).reshape(1, B, L, S * d)


class MultiHeadLinearAttention(nn.Module):
    """encoder.py:88-140: per-token gates softmax_L(lin2(act(lin1 v))) applied through the reference's
    flat `view(B*H, L, hd)` reinterpretation.  Only activation='gelu' with dropout 0 (what every
    DrugLAMP model uses, basic_model.py:114,117) is implemented in HIP."""

    def __init__(self, d_model, nhead, d_diff=32, dropout=0.1, activation="tanh"):
        super().__init__()
        if activation != "gelu":
            raise NotImplementedError("MultiHeadLinearAttention: only activation='gelu' is implemented")
        if dropout not in (0, 0.0):
            raise NotImplementedError("MultiHeadLinearAttention: dropout must be 0 (mlha_dropout = 0)")
        self.nhead = nhead
        self.lin1 = nn.Linear(d_model, d_diff)
        self.lin2 = nn.Linear(d_diff, nhead)
        self.compute_dtype = torch.float32

    def forward(self, v, add_residual: bool = False):
        x = Fn.cast(v, self.compute_dtype)
        out = Fn.TokenGateFn.apply(x, self.lin1.weight, self.lin1.bias, self.lin2.weight, self.lin2.bias, self.nhead,
                                   add_residual)
        return Fn.cast(out, v.dtype)
