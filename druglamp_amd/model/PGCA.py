"""PGCA — guided cross attention (reference: model/PGCA/guided_cross_attention_model.py).

GuidedCrossAttention keeps the reference's constructor, seq-first forward signature and state_dict
keys (in_proj_weight / in_proj_bias / out_proj.*).  The path the DrugLAMP models take — key is value,
no masks, dropout 0, need_weights & need_raw — runs as in-proj GEMMs + one fused attention launch +
out-proj GEMM; the raw (pre-softmax) logits the reference returns as its second output are written by
the same attention launch.  Other argument combinations of the torch-1.x MultiheadAttention fork
(masks, bias_kv, zero-attn, separate kdim/vdim) are not on the path and raise.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as Fn


class GuidedCrossAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None):
        super().__init__()
        if add_bias_kv or add_zero_attn or (kdim not in (None, embed_dim)) or (vdim not in (None, embed_dim)):
            raise NotImplementedError("GuidedCrossAttention: bias_kv / zero_attn / kdim / vdim are not implemented")
        if dropout not in (0, 0.0):
            raise NotImplementedError("GuidedCrossAttention: dropout must be 0")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim, "embed_dim must be divisible by num_heads"
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter("in_proj_bias", None)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        if bias:
            nn.init.constant_(self.in_proj_bias, 0.)
            nn.init.constant_(self.out_proj.bias, 0.)
        self.compute_dtype = torch.float32

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, need_raw=True, attn_mask=None, key_tail=None):
        """key_tail = (rows, weight) (not in the reference's signature; default None = the reference's call): `key` holds the
        distinct key rows only, its last `rows` rows each standing for `weight` identical rows of the full key set
        (functional.GuidedCrossAttentionFn).  The raw logits are those of the full key set and are not produced then."""
        if key_padding_mask is not None or attn_mask is not None:
            raise NotImplementedError("GuidedCrossAttention: masks are not on the DrugLAMP path")
        if not (key is value or (key.data_ptr() == value.data_ptr() and key.shape == value.shape
                                 and key.stride() == value.stride())):
            raise NotImplementedError("GuidedCrossAttention: only the key-is-value (encoder-decoder) branch is "
                                      "implemented (guided_cross_attention_model.py:138-162)")
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim
        assert key.size(1) == bsz and key.size(2) == embed_dim
        want_raw = bool(need_weights and need_raw)
        if key_tail is not None and want_raw:
            raise ValueError("GuidedCrossAttention: raw logits are not available with key_tail (pass need_weights=False)")
        if need_weights and not need_raw:
            raise NotImplementedError("GuidedCrossAttention: head-averaged softmax weights are not implemented")
        q = Fn.cast(query, self.compute_dtype)
        k = Fn.cast(key, self.compute_dtype)
        out, raw = Fn.GuidedCrossAttentionFn.apply(q, k, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight,
                                                   self.out_proj.bias, self.num_heads, want_raw, key_tail)
        return Fn.cast(out, query.dtype), raw
