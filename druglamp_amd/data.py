"""Host side of the input pipeline for row-wise interaction tables (BASELINE config 1).

The reference builds every sample on the fly in `MultiModalityDataset.__getitem__` (handler/dataset.py:173-226: SMILES ->
DGL graph, protein string -> `repeat_integer_label_protein`, two `torch.load`s of cached LLM embeddings) and pads in
`multimodality_collate_func` (utils.py:326-334).  Here every UNIQUE drug / protein is prepared once and kept on the
device; a batch is an index lookup (`torch.index_select` for the small per-entity tensors, one `dl_gather_pad` launch per
LLM modality through EmbeddingStore).

What is real and what is synthetic for tests/golden/human_random_rows.npz (config 1: datasets/human, random split):
real = the (drug, protein, label) rows, the Drug_ID / Prot_ID numbering, the protein residue codes and their tiling;
synthetic = drug graphs and LLM embeddings (rdkit / dgllife / ESM-2 / ChemBERTa are not available offline), generated
deterministically per entity id so that a drug or protein looks the same in every row it occurs in.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch


def repeat_integer_label(codes: np.ndarray, max_length: int = 9 * 256) -> np.ndarray:
    """The tiling of `repeat_integer_label_protein` (reference utils.py:392-412) for an already coded sequence: the L
    codes are written floor(max_length / (L + 2)) times with period L + 2 starting at offset 1 (the CLS / SEP slots of
    the matching ESM-2 token row stay 0); float64 like the reference's `np.zeros(max_length)`."""
    out = np.zeros(max_length)
    L = int(codes.shape[0])
    for i in range(max_length // (L + 2)):
        st = i * (L + 2) + 1
        out[st:st + L] = codes
    return out


def shard_order(n_rows: int, shuffle_seed: Optional[int], rank: int = 0, world: int = 1) -> torch.Tensor:
    """Row order of one rank's shard.  shuffle_seed given (training): a seeded permutation, padded by wrap-around to
    ceil(n / world) * world before striding — torch's DistributedSampler, which Lightning's DDP hands the reference
    (main.py:138-141) — so every rank yields the same number and shapes of batches: ranks issue one collective per step
    and must agree on the step count.  Evaluation shards (no shuffle) stay unpadded: Trainer.evaluate gathers unequal
    shards, and a duplicated sample would bias the metrics."""
    order = torch.arange(n_rows)
    if shuffle_seed is not None:
        order = torch.randperm(n_rows, generator=torch.Generator().manual_seed(shuffle_seed))
        if world > 1 and n_rows > 0:
            total = -(-n_rows // world) * world
            if total > n_rows:
                pad = total - n_rows
                order = torch.cat([order, order.repeat(-(-pad // n_rows))[:pad]])
    return order[rank::world]


class RowTable:
    """(drug, protein, label) rows + per-entity inputs, device resident.  `batches(split, B)` yields what the reference's
    collate hands the model: ((node_feats, adjacency), residue codes (B, seq_len) float64, labels, llm_d, llm_p), meta."""

    def __init__(self, npz_path: str, device, llm_dtype: torch.dtype = torch.bfloat16, seq_len: int = 9 * 256,
                 adj_nodes: int = 128, with_llm: bool = True, seed: int = 0):
        from .embedding_store import EmbeddingStore
        g = np.load(npz_path, allow_pickle=False)
        self.rows = {k: torch.from_numpy(g[k].astype(np.int64)) for k in ("train", "val", "test")}
        self.device, self.seq_len = torch.device(device), seq_len
        offs, codes = g["prot_offsets"], g["prot_codes"]
        n_prot, n_drug = len(offs) - 1, len(g["drug_atoms"])
        vp = np.zeros((n_prot, seq_len))
        self.prot_len = np.diff(offs)
        for p in range(n_prot):
            vp[p] = repeat_integer_label(codes[offs[p]:offs[p + 1]], seq_len)
        self.vp = torch.from_numpy(vp).to(self.device)                                  # (n_prot, seq_len) float64
        # synthetic drug graphs, one per unique drug: one-hot-ish atom features, a chain plus a few ring-closing bonds
        gen = torch.Generator().manual_seed(seed)
        n_atom = np.minimum(np.maximum(g["drug_atoms"].astype(np.int64), 2), adj_nodes)
        h = torch.zeros(n_drug, 512, 75)
        adj = torch.zeros(n_drug, adj_nodes, adj_nodes)
        for d in range(n_drug):
            n = int(n_atom[d])
            h[d, :n, :74] = (torch.rand(n, 74, generator=gen) < 0.1).float()
            h[d, n:, 74] = 1.0                                                          # virtual-node indicator bit
            idx = torch.arange(n - 1)
            adj[d, idx, idx + 1] = 1
            adj[d, idx + 1, idx] = 1
            extra = torch.randint(0, n, (max(n // 5, 1), 2), generator=gen)
            adj[d, extra[:, 0], extra[:, 1]] = 1
            adj[d, extra[:, 1], extra[:, 0]] = 1
            adj[d].fill_diagonal_(1)
            adj[d, torch.arange(n), torch.arange(n)] = 2             # real atoms carry two self loops in the reference's graphs
        self.h, self.adj = h.to(self.device), adj.to(self.device)
        # synthetic "pre-extracted" LLM embeddings in a device-resident store (rows: Lp + 2 protein tokens, <= 128 drug tokens)
        self.prot_store = EmbeddingStore(640, dtype=llm_dtype, device=self.device)
        self.drug_store = EmbeddingStore(384, dtype=llm_dtype, device=self.device)
        for p in range(n_prot):
            self.prot_store.add(p, torch.randn(int(self.prot_len[p]) + 2, 640 if with_llm else 640, generator=gen))
        for d in range(n_drug):
            self.drug_store.add(d, torch.randn(int(min(max(n_atom[d], 12), 128)), 384, generator=gen))
        self.prot_store.finalize()
        self.drug_store.finalize()

    def n_rows(self, split: str) -> int:
        return int(self.rows[split].shape[0])

    def batches(self, split: str, batch_size: int, shuffle_seed: Optional[int] = None, drop_last: bool = False,
                rank: int = 0, world: int = 1) -> Iterator[Tuple[tuple, List[Dict]]]:
        """shuffle_seed given: a seeded permutation (the reference's DataLoader(shuffle=True, drop_last=True) for training,
        main.py:138-141); rank / world: the strided shard a DistributedSampler would hand this rank (training = shuffled:
        padded by wrap-around to equal length on every rank; evaluation: unpadded)."""
        rows = self.rows[split]
        order = shard_order(rows.shape[0], shuffle_seed, rank, world)
        for s in range(0, order.numel(), batch_size):
            sel = order[s:s + batch_size]
            if drop_last and sel.numel() < batch_size:
                break
            r = rows[sel]
            d_idx, p_idx = r[:, 0].to(self.device), r[:, 1].to(self.device)
            feat_d = (self.h.index_select(0, d_idx), self.adj.index_select(0, d_idx))
            vp = self.vp.index_select(0, p_idx)
            y = r[:, 2].to(self.device, dtype=torch.float32)
            llm_d = self.drug_store.batch(r[:, 0].tolist(), 512, repeat=False)
            llm_p = self.prot_store.batch(r[:, 1].tolist(), self.seq_len, repeat=True)
            meta = [{"Drug_ID": int(a), "Prot_ID": int(b), "Y": float(c), "Drug_Tokens": self.drug_store.length(int(a)),
                     "Prot_Len": int(self.prot_len[int(b)])} for a, b, c in r.tolist()]
            yield (feat_d, vp, y, llm_d, llm_p), meta

    def batches_only(self, split: str, batch_size: int, **kw):
        """batches() without the meta dicts (what Trainer.evaluate iterates over)."""
        for batch, _ in self.batches(split, batch_size, **kw):
            yield batch
