"""Pins the oracle (oracle/druglamp_oracle.py) against golden outputs of the REAL reference
(tests/golden/*.npz, produced by tests/golden/make_golden.py in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import druglamp_oracle as O
from tests.helpers import T, check_sub, det_state_dict, elemerr, gradnorms, load, model_inputs, pmma_dropout_masks, relerr

TOL = 2e-5


def _grads(sd, keys):
    for k in keys:
        sd[k].requires_grad_(True)


def test_pmma_mid_forward_backward():
    g = load("pmma_mid")
    sd = det_state_dict(g)
    for v in sd.values():
        v.requires_grad_(True)
    prot = T("pmma_mid.prot", (2, 64, 256)).requires_grad_(True)
    mol = T("pmma_mid.mol", (2, 64, 256)).requires_grad_(True)
    enc, maps = O.pmma_forward(sd, prot, mol, return_maps=True)
    assert relerr(enc, g["encoded"]) <= TOL
    assert relerr(maps[0][0][:, :, :4, :8], g["w0"]) <= TOL
    assert relerr(maps[0][1][:, :, :4, :8], g["gw0"]) <= TOL
    assert relerr(maps[3][0][:, :, :4, :8], g["w3"]) <= TOL
    (enc * T("pmma_mid.G", tuple(enc.shape))).sum().backward()
    assert relerr(prot.grad, g["dprot"]) <= TOL
    assert relerr(mol.grad, g["dmol"]) <= TOL
    assert relerr(sd["encoder.layer_with_mol.0.attn.query.weight"].grad[:8, :16], g["dW_l0_query"]) <= TOL
    assert relerr(sd["encoder.layer_with_mol.3.ffn.fc2.weight"].grad[:8, :16], g["dW_l3_fc2"]) <= TOL
    assert relerr(sd["encoder.layer_with_mol.1.attn.fc_mol.bias"].grad, g["db_l1_fc_mol"]) <= TOL
    assert relerr(sd["embeddings.pe_mol"].grad[0, :4, :16], g["dpe_mol"]) <= TOL
    for k, n in gradnorms(g).items():
        assert abs(float(sd[k].grad.double().norm()) - n) <= 1e-4 * max(n, 1e-6), k
    # the reference never produces a gradient for the dead `embeddings.embedding` Linear
    assert "embeddings.embedding.weight" not in gradnorms(g)


def test_pmma_training_mode_with_recorded_dropout_masks():
    """The oracle's training-mode PMMA (dropout draws as data) against the reference in train mode with its nn.Dropout modules
    fed the same masks (tests/golden/pmma_drop.npz): output, input gradients, weight-gradient samples on both sides of every
    dropout site, every parameter's gradient norm — element-wise as well as max-norm."""
    g = load("pmma_drop")
    sd = det_state_dict(g)
    for v in sd.values():
        v.requires_grad_(True)
    B, L = 2, 64
    prot = T("pmma_drop.prot", (B, L, 256)).requires_grad_(True)
    mol = T("pmma_drop.mol", (B, L, 256)).requires_grad_(True)
    masks = pmma_dropout_masks("pmma_drop", B, L, 256, float(g["p"]))
    assert 0.08 < float((masks["l0.s0.fc1"] == 0).float().mean()) < 0.12
    enc = O.pmma_forward(sd, prot, mol, dropout_masks=masks)
    assert relerr(enc, g["encoded"]) <= TOL and elemerr(enc, g["encoded"]) <= 10 * TOL
    assert relerr(O.pmma_forward(sd, prot, mol), g["encoded"]) > 1e-2        # (the masks matter: eval mode is far away)
    (enc * T("pmma_drop.G", tuple(enc.shape))).sum().backward()
    for got, key in ((prot.grad, "dprot"), (mol.grad, "dmol"),
                     (sd["encoder.layer_with_mol.0.ffn.fc1.weight"].grad[:8, :16], "dW_l0_fc1"),
                     (sd["encoder.layer_with_mol.1.ffn_mol.fc2.weight"].grad[:8, :16], "dW_l1_fc2_mol"),
                     (sd["encoder.layer_with_mol.3.ffn.fc2.weight"].grad[:8, :16], "dW_l3_fc2"),
                     (sd["embeddings.pe_prot"].grad[0, :4, :16], "dpe_prot")):
        assert relerr(got, g[key]) <= TOL and elemerr(got, g[key]) <= 10 * TOL, key
    for k, n in gradnorms(g).items():
        assert abs(float(sd[k].grad.double().norm()) - n) <= 1e-4 * max(n, 1e-6), k


def test_pmma_full_forward():
    g = load("pmma_full")
    sd = det_state_dict(g)
    prot, mol = T("pmma_full.prot", (2, 256, 256)), T("pmma_full.mol", (2, 256, 256))
    enc = O.pmma_forward(sd, prot, mol)
    check_sub(enc, g, "encoded", TOL)


@pytest.mark.parametrize("tag,shape", [("pgca_small", (48, 80, 3)), ("pgca_full", (256, 512, 2))])
def test_pgca(tag, shape):
    Lq, Lk, B = shape
    g = load(tag)
    sd = det_state_dict(g)
    q = T(tag + ".q", (Lq, B, 128)).requires_grad_(True)
    kv = T(tag + ".kv", (Lk, B, 128)).requires_grad_(True)
    out, raw = O.pgca_forward({"m." + k: v for k, v in sd.items()}, "m", q, kv)
    assert relerr(out, g["out"]) <= TOL
    assert relerr(raw[:, :, :8, :16], g["raw"]) <= TOL
    assert relerr(raw.double().norm(dim=-1), g["rawnorm"]) <= TOL
    (out * T(tag + ".G", tuple(out.shape))).sum().backward()
    assert relerr(q.grad, g["dq"]) <= TOL
    assert relerr(kv.grad, g["dkv"]) <= TOL


@pytest.mark.parametrize("tag,shape", [("mhla_toy", (32, 2, 5)), ("mhla_full", (256, 2, 256))])
def test_mhla(tag, shape):
    d, B, L = shape
    g = load(tag)
    sd = {"m." + k: v for k, v in det_state_dict(g).items()}
    v = T(tag + ".v", (B, L, d)).requires_grad_(True)
    out = O.mhla_forward(sd, "m", v)
    assert relerr(out, g["out"]) <= TOL
    (out * T(tag + ".G", tuple(out.shape))).sum().backward()
    assert relerr(v.grad, g["dv"]) <= TOL


def test_losses():
    g = load("losses")
    for tag, (n, d) in (("ntx_small", (24, 64)), ("ntx_big", (512, 128))):
        q = T(tag + ".q", (n, d), 0.3).requires_grad_(True)
        k = T(tag + ".k", (n, d), 0.3).requires_grad_(True)
        loss = O.nt_xent(q, k, 0.1)
        assert abs(float(loss) - float(g[tag + "/loss"])) <= 1e-5 * abs(float(g[tag + "/loss"]))
        loss.backward()
        assert relerr(q.grad[:8], g[tag + "/dq"]) <= 1e-4
        assert relerr(k.grad[:8], g[tag + "/dk"]) <= 1e-4
    x = T("cos.x", (300, 128)).requires_grad_(True)
    y = T("cos.y", (300, 128))
    rows = O.cos_rowloss(x, y)
    assert relerr(rows, g["cos/rows"]) <= TOL
    rows.mean().backward()
    assert relerr(x.grad, g["cos/dx"]) <= TOL
    pl = torch.nn.functional.normalize(T("tri.p", (12, 256)), dim=-1).requires_grad_(True)
    dl = torch.nn.functional.normalize(T("tri.d", (17, 256)), dim=-1).requires_grad_(True)
    tl = O.triplet_sigcos(pl, dl, g["tri/gt"], 0.3)
    assert abs(float(tl) - float(g["tri/loss"])) <= 1e-6
    tl.backward()
    assert relerr(pl.grad, g["tri/dp"]) <= 1e-4
    assert relerr(dl.grad, g["tri/dd"]) <= 1e-4
    sch = O.MarginSchedule(0.5, 100)
    margins = [sch.margin]
    for _ in range(205):
        sch.step()
        margins.append(sch.margin)
    assert np.allclose(margins, g["margins"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", ["DrugLAMP", "DrugLAMP2C2P", "DrugLAMPwoLLM"])
def test_model_forward(kind):
    g = load("model_" + kind)
    sd = det_state_dict(g)
    vd, vp, xd, xp, y = model_inputs("model." + kind, 2)
    with torch.no_grad():
        out = O.model_forward(sd, kind, vd, vp, xd, xp, bn_training=False)
    assert relerr(out["score"], g["score"]) <= 1e-4
    check_sub(out["vp"], g, "vp", 1e-4)
    assert relerr(out["A_v"][:, :, :4, :8], g["A_v"]) <= 1e-4
    if kind != "DrugLAMPwoLLM":
        assert relerr(out["A_x"][:, :, :4, :8], g["A_x"]) <= 1e-4
    if kind == "DrugLAMP2C2P":
        check_sub(out["cm"]["aug_prot"], g, "cm_aug_prot", 1e-4)
        check_sub(out["cm"]["aug_drug"], g, "cm_aug_drug", 1e-4)
    # train-mode BN + BCE backward
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    vd, vp, xd, xp, y = model_inputs("modeltrain." + kind, 8)
    out = O.model_forward(sd, kind, vd, vp, xd, xp, bn_training=True)
    # B=8 batch statistics in the classifier's BatchNorm amplify fp32 rounding differences
    assert relerr(out["score"], g["score_train"]) <= 1e-3
    n, loss = O.bce_loss(out["score"], y)
    assert abs(float(loss) - float(g["cls_loss"])) <= 1e-4
    loss.backward()
    gn = gradnorms(g)
    worst = 0.0
    for k, n_ref in gn.items():
        got = float(sd[k].grad.double().norm()) if sd[k].grad is not None else 0.0
        # key-projection / gate biases have analytically ZERO gradient (softmax shift invariance): their
        # norms are fp32 rounding noise (~1e-8), hence the absolute floor
        worst = max(worst, abs(got - n_ref) / max(n_ref, 1e-5))
    assert worst <= 2e-3, worst
    # the fixed sample of the reference's whole gradient vector (direction, not just norms): keys in the reference's
    # state_dict order, the shared ProteinCNN's alias keys once
    from tests.helpers import sd_spec
    parts, seen = [], set()
    for k, _, _ in sd_spec(g):
        if k.startswith("ssl_model.extractor."):
            continue
        v = sd.get(k)
        if v is not None and v.grad is not None and id(v) not in seen:
            seen.add(id(v))
            parts.append(v.grad.detach().flatten().double())
    gv = torch.cat(parts)
    assert gv.numel() == int(g["gtotal"]), (gv.numel(), int(g["gtotal"]))
    got = gv[torch.from_numpy(g["gidx"])]
    assert float((got - torch.from_numpy(g["gsample"]).double()).abs().max()) <= 2e-3 * float(g["gmax"])


def _unpack(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(shape).astype(bool))


def test_ssl_cm():
    g = load("ssl_cm")
    sd = det_state_dict(g)
    B = 6
    vd, vp, xd, xp, y = model_inputs("sslcm", B)
    out = O.model_forward(sd, "DrugLAMP2C2P", vd, vp, xd, xp, bn_training=True)
    mask = _unpack(g["mask"], (B, 2304))
    replace = _unpack(g["replace"], (B, 2304))
    assert int(mask.sum()) == int(g["n_masked"])
    ssl = out["ssl"]
    prot = O.prot_mlm_loss(sd, "ssl_model", ssl["vp"], ssl["xp"], ssl["fill_bit_p"], "double", mask, replace)
    drug = O.simsiam_loss(sd, "ssl_model", ssl["vd"], ssl["xd"])
    assert abs(float(prot) - float(g["prot_ssl"])) <= 2e-5 * abs(float(g["prot_ssl"]))
    assert abs(float(drug) - float(g["drug_ssl"])) <= 2e-5 * abs(float(g["drug_ssl"]))
    meta = [{"Prot_ID": int(p), "Drug_ID": int(d), "Y": float(y[t])} for t, (p, d) in
            enumerate(zip(g["meta_pid"], g["meta_did"]))]
    cm = O.cm_forward(sd, "cm_model", **out["cm"], meta=meta, margin=0.5)
    assert abs(float(cm) - float(g["cm_loss"])) <= 2e-5 * max(abs(float(g["cm_loss"])), 1e-3)


def test_collate_padding_restatement_matches_reference_functions():
    """oracle/collate.py vs the outputs of the reference's own tail_pad / repeat_pad (tests/golden/collate_pad.npz,
    generated by tests/golden/make_collate_golden.py): bit-exact, incl. a sequence longer than maxsize (all zeros)."""
    import os
    import numpy as np
    from oracle import collate
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate_pad.npz"))
    ms, feat = int(g["maxsize"]), int(g["feat"])
    tail_in = collate.ragged_inputs("collate.tail", [int(n) for n in g["tail_lens"]], feat)
    rep_in = collate.ragged_inputs("collate.rep", [int(n) for n in g["rep_lens"]], feat)
    assert np.array_equal(collate.tail_pad(tail_in, ms), g["tail_out"])
    assert np.array_equal(collate.repeat_pad(rep_in, ms), g["rep_out"])
    assert not g["rep_out"][list(g["rep_lens"]).index(60)].any()


# ---- BASELINE config 5: long proteins (PROTEIN.SEQ_LEN = 9216 -> 1024 sites) ---------------------------------------
def test_pmma_long_forward_backward():
    """PMMA at feat_len = mol_len = 1024 (reference embed.py:32-33 accepts any length) — the oracle pinned at the length
    where K/V no longer fit LDS and the HIP path has to stream key tiles."""
    g = load("pmma_L1024")
    sd = det_state_dict(g)
    for v in sd.values():
        v.requires_grad_(True)
    prot = T("pmma_L1024.prot", (2, 1024, 256)).requires_grad_(True)
    mol = T("pmma_L1024.mol", (2, 1024, 256)).requires_grad_(True)
    enc, maps = O.pmma_forward(sd, prot, mol, return_maps=True)
    check_sub(enc, g, "encoded", TOL)
    assert relerr(maps[0][0][:, :, :4, :8], g["w0"]) <= TOL
    assert relerr(maps[0][1][:, :, :4, :8], g["gw0"]) <= TOL
    assert relerr(maps[1][1][:, :, -4:, -8:], g["gw1_tail"]) <= TOL
    assert relerr(maps[3][0][:, :, :4, :8], g["w3"]) <= TOL
    (enc * T("pmma_L1024.G", tuple(enc.shape))).sum().backward()
    check_sub(prot.grad, g, "dprot", TOL)
    check_sub(mol.grad, g, "dmol", TOL)
    assert relerr(sd["encoder.layer_with_mol.0.attn.query.weight"].grad[:8, :16], g["dW_l0_query"]) <= TOL
    assert relerr(sd["encoder.layer_with_mol.3.ffn.fc2.weight"].grad[:8, :16], g["dW_l3_fc2"]) <= TOL
    assert relerr(sd["embeddings.pe_mol"].grad[0, -4:, :16], g["dpe_mol"]) <= TOL
    for k, n in gradnorms(g).items():
        assert abs(float(sd[k].grad.double().norm()) - n) <= 1e-4 * max(n, 1e-6), k


def test_model_long_forward_and_gradients():
    """Whole DrugLAMP at SEQ_LEN = 9216 / 1024 sites against the reference run with the same config."""
    g = load("model_L1024")
    sd = det_state_dict(g)
    S = 9216
    vd, vp, xd, xp, y = model_inputs("modelL.DrugLAMP", 2, seq_len=S, lp_range=(1000, 4000))
    with torch.no_grad():
        out = O.model_forward(sd, "DrugLAMP", vd, vp, xd, xp, bn_training=False, seq_len=S)
    assert relerr(out["score"], g["score"]) <= 1e-4
    check_sub(out["vp"], g, "vp", 1e-4)
    assert relerr(out["A_v"][:, :, :4, :8], g["A_v"]) <= 1e-4
    assert relerr(out["A_x"][:, :, -4:, -8:], g["A_x"]) <= 1e-4
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    vd, vp, xd, xp, y = model_inputs("modelLtrain.DrugLAMP", 4, seq_len=S, lp_range=(1000, 4000))
    out = O.model_forward(sd, "DrugLAMP", vd, vp, xd, xp, bn_training=True, seq_len=S)
    assert relerr(out["score"], g["score_train"]) <= 1e-3
    n, loss = O.bce_loss(out["score"], y)
    assert abs(float(loss) - float(g["cls_loss"])) <= 1e-4
    loss.backward()
    worst = 0.0
    for k, n_ref in gradnorms(g).items():
        got = float(sd[k].grad.double().norm()) if sd[k].grad is not None else 0.0
        worst = max(worst, abs(got - n_ref) / max(n_ref, 1e-5))
    assert worst <= 2e-3, worst


def test_molecular_gcn_vs_reference_gcn_classes():
    """tests/golden/gcn.npz: the reference's OWN MolecularGCN / GCN / GCNLayer / GraphConv code
    (/root/reference/model/basic_model.py:137-153,342-638) run on a scipy.sparse stand-in for the batched DGL graph
    (make_golden.py gen_gcn: DGL is absent; its update_all(copy_u, sum) / in_degrees / out_degrees follow the published
    semantics).  Graphs of 23 / 77 / 150 atoms, double self loops on real atoms as handler/dataset.py:211-222 builds them."""
    g = load("gcn")
    sd = det_state_dict(g, salt=41)
    sd["init_transform.weight"][-1].fill_(0)                               # basic_model.py:141-143 (padding=True)
    for v in sd.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    sd = {"gcn." + k: v for k, v in sd.items()}
    h, adj = torch.from_numpy(g["h"]), torch.from_numpy(g["adj"])
    out = O.molecular_gcn(sd, "gcn", h, adj, bn_training=True)
    assert relerr(out, g["out"]) <= TOL
    (out * T("gcn.cot", tuple(out.shape))).sum().backward()
    assert relerr(sd["gcn.init_transform.weight"].grad, g["g_init"]) <= 1e-4
    assert relerr(sd["gcn.gnn.gnn_layers.1.graph_conv.weight"].grad, g["g_conv1"]) <= 1e-4
    assert relerr(sd["gcn.gnn.gnn_layers.2.res_connection.weight"].grad, g["g_res2"]) <= 1e-4
    assert relerr(sd["gcn.gnn.gnn_layers.0.bn_layer.weight"].grad, g["g_bn0"]) <= 1e-4
