"""Synthetic DrugLAMP batches with the padding structure of the reference's collate (utils.py:304-334,
373-412): used by bench.py and smoke().  Generated on the host with a seeded generator, then moved to
the device once (inputs are resident in HBM before any timed region starts)."""
from __future__ import annotations

import torch


def make_batch(B: int, device, seed: int = 0, with_graph: bool = True, llm_dtype=torch.float32, adj_nodes: int = 128,
               seq_len: int = 2304, max_prot_len: int = 1022):
    """Returns (feat_d, feat_p, labels, llm_d, llm_p), meta.
    feat_d: (node_feats (B,512,75), adjacency (B,adj_nodes,adj_nodes)) dense batched graphs (real-atom block;
            nodes beyond it are self-looped virtual padding nodes) when with_graph, else pre-extracted GCN features (B,512,128);
    feat_p: (B,seq_len) float64 residue codes tiled like repeat_integer_label_protein (seq_len = PROTEIN.SEQ_LEN, 2304;
            long-protein configurations use 9216 with max_prot_len 4094);
    llm_d : (B,512,384) ChemBERTa-shaped token embeddings, zero after the last token;
    llm_p : (B,2304,640) ESM-2-shaped embeddings of an (Lp+2)-token protein tiled to 2304, zero tail."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    n_atom = torch.randint(10, 81, (B,), generator=g)
    n_tok = torch.randint(12, 129, (B,), generator=g)
    Lp = torch.randint(100, min(max_prot_len, seq_len - 2) + 1, (B,), generator=g)
    xd = torch.randn(B, 512, 384, generator=g)
    xp = torch.zeros(B, seq_len, 640)
    vp = torch.zeros(B, seq_len, dtype=torch.float64)
    for b in range(B):
        xd[b, int(n_tok[b]):] = 0
        L = int(Lp[b])
        blk = torch.randn(L + 2, 640, generator=g)
        seq = torch.randint(1, 26, (L,), generator=g).double()
        reps = seq_len // (L + 2)
        for r in range(reps):
            xp[b, r * (L + 2):(r + 1) * (L + 2)] = blk
            vp[b, r * (L + 2) + 1:r * (L + 2) + 1 + L] = seq
    y = (torch.rand(B, generator=g) < 0.5).float()
    if with_graph:
        h = torch.zeros(B, 512, 75)
        adj = torch.zeros(B, adj_nodes, adj_nodes)
        for b in range(B):
            n = int(n_atom[b])
            h[b, :n, :74] = (torch.rand(n, 74, generator=g) < 0.1).float()
            h[b, n:, 74] = 1.0                                    # virtual-node indicator bit
            # chain + a few random bonds, symmetric, self loops everywhere
            idx = torch.arange(n - 1)
            adj[b, idx, idx + 1] = 1
            adj[b, idx + 1, idx] = 1
            extra = torch.randint(0, n, (max(n // 5, 1), 2), generator=g)
            adj[b, extra[:, 0], extra[:, 1]] = 1
            adj[b, extra[:, 1], extra[:, 0]] = 1
            adj[b].fill_diagonal_(1)                                # nodes n..adj_nodes-1 of the block: virtual, one self loop
            adj[b, torch.arange(n), torch.arange(n)] = 2             # real atoms: TWO self loops, as handler/dataset.py:211-222 builds them
                                                                     # (smiles_to_bigraph(add_self_loop=True), then add_self_loop() again)
        feat_d = (h.to(device), adj.to(device))
    else:
        vd = torch.randn(B, 512, 128, generator=g)
        for b in range(B):
            vd[b, int(n_atom[b]):] = 0
        feat_d = vd.to(device)
    n_prot = max(int(0.6 * B), 1)
    n_drug = max(int(0.9 * B), 1)
    pid = torch.randint(0, n_prot, (B,), generator=g).tolist()
    did = torch.randint(0, n_drug, (B,), generator=g).tolist()
    # (Drug_Tokens: the molecule's token count, which a collate knows — Trainer turns the batch maximum into a padding hint)
    # (Prot_Len: the protein's residue count, likewise collate knowledge — the ProteinCNN distinct-row plan is built from it)
    meta = [{"Prot_ID": pid[t], "Drug_ID": did[t], "Y": float(y[t]), "Drug_Tokens": int(n_tok[t]), "Prot_Len": int(Lp[t])} for t in range(B)]
    batch = (feat_d, vp.to(device), y.to(device), xd.to(device=device, dtype=llm_dtype),
             xp.to(device=device, dtype=llm_dtype))
    return batch, meta
