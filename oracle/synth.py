"""Synthetic DrugLAMP inputs (test infrastructure): numpy only, deterministic.

`model_inputs` mimics what the reference's collate hands the model (utils.py:304-334): post-GCN drug
node features zero-padded to 512 nodes, ChemBERTa token embeddings zero-padded to 512, the ESM-2
block of an (Lp+2)-token protein tiled to 2304 positions (repeat_pad) with zero tail, and the integer
residue codes tiled with the same period (stored as float64, as the reference does).
"""
from __future__ import annotations

import numpy as np

from . import detgen


def model_inputs(tag: str, B: int, salt: int = 0, seq_len: int = 2304, lp_range=(100, 600)):
    """Synthetic pre-extracted inputs with the padding structure of the real collate (utils.py:326-334).
    seq_len = PROTEIN.SEQ_LEN (the reference hard-codes 9 * 256 in its collate, utils.py:333)."""
    rs = np.random.RandomState(1234 + salt)
    vd = detgen.normalish(tag + ".vd", (B, 512, 128), salt).copy()
    xd = detgen.normalish(tag + ".xd", (B, 512, 384), salt).copy()
    xp = np.zeros((B, seq_len, 640), dtype=np.float32)
    vp = np.zeros((B, seq_len), dtype=np.float64)
    for b in range(B):
        n_atom = int(rs.randint(10, 80))
        vd[b, n_atom:] = 0
        n_tok = int(rs.randint(12, 128))
        xd[b, n_tok:] = 0
        Lp = int(rs.randint(*lp_range))
        seq = rs.randint(1, 26, size=Lp).astype(np.float64)
        blk = detgen.normalish("%s.xp%d" % (tag, b), (Lp + 2, 640), salt)
        reps = seq_len // (Lp + 2)
        for r in range(reps):
            xp[b, r * (Lp + 2):(r + 1) * (Lp + 2)] = blk
            vp[b, r * (Lp + 2) + 1: r * (Lp + 2) + 1 + Lp] = seq
    y = (detgen.uniform(tag + ".y", (B,), salt=salt) > 0).astype(np.float32)
    return vd, vp, xd, xp, y




def pmma_dropout_masks(tag: str, B: int, L: int, d: int, p: float, n_layers: int = 4):
    """Deterministic keep masks (1 / (1 - p) where kept, 0 where dropped; float32 numpy) for every dropout site of a
    training-mode PMMA forward, keyed like druglamp_oracle.pmma_forward(dropout_masks=...): emb_mol / emb_prot (embed.py:42,52),
    l{i}.s{s}.fc1 / .fc2 (mlp.py:47,49; s = 0 prot / the single stream, 1 mol).  The golden generator feeds the same masks to
    the reference's nn.Dropout modules (tests/golden/make_golden.py::gen_pmma_dropout)."""
    out = {}

    def mk(name, shape):
        u = detgen.uniform(tag + ".mask." + name, shape, 0.0, 1.0)
        out[name] = (u >= p).astype(np.float32) / np.float32(1.0 - p)
    mk("emb_mol", (B, L, d))
    mk("emb_prot", (B, L, d))
    for i in range(n_layers):
        w = d if i < 2 else 2 * d
        for s_ in range(2 if i < 2 else 1):
            mk("l%d.s%d.fc1" % (i, s_), (B, L, 4 * w))
            mk("l%d.s%d.fc2" % (i, s_), (B, L, w))
    return out
