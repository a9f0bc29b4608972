"""Generate the committed golden fixtures by running the REAL reference in the build container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Weights and inputs come from oracle/detgen.py (regenerated bit-identically by the tests), so the
.npz files hold only the reference's OUTPUTS (full tensors where small, sub-samples + row norms
where large).  Nothing from /root/reference is copied: the fixtures are data.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_harness  # noqa: E402
from oracle import detgen, synth  # noqa: E402

ref_harness.import_reference()
torch.set_num_threads(8)


def fill_module(mod: torch.nn.Module, salt: int = 0):
    sd = mod.state_dict()
    vals = detgen.fill_state_dict([(k, tuple(v.shape), str(v.dtype)) for k, v in sd.items()], salt)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return mod


def T(name, shape, scale=1.0, salt=0):
    return torch.from_numpy(detgen.normalish(name, shape, salt) * np.float32(scale))


def sub(x: torch.Tensor):
    """sub-sample of a big (B, L, D) tensor: first/last 4 rows, per-row norms, checksum."""
    x = x.detach().double()
    return {"head": x[:, :4].float().numpy(), "tail": x[:, -4:].float().numpy(),
            "rownorm": x.norm(dim=-1).float().numpy(), "sum": np.float64(x.sum().item())}


def sd_spec(mod):
    import json
    return np.asarray(json.dumps([(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in mod.state_dict().items()]))


def save(name, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat["%s/%s" % (k, kk)] = np.asarray(vv)
        else:
            flat[k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def pmma_config(L):
    from configs import get_model_defaults
    cfg = get_model_defaults(128)
    cfg.feat_len = L
    cfg.mol_len = L
    cfg.transformer.dropout_rate = 0.0
    return cfg


def grad_norms(mod):
    return {k: np.float64(p.grad.double().norm().item()) for k, p in mod.named_parameters() if p.grad is not None}


def gen_pmma():
    from model.PMMA import PairedMultimodelAttention
    for tag, L in (("pmma_mid", 64), ("pmma_full", 256)):
        m = fill_module(PairedMultimodelAttention(pmma_config(L), vis=True)).eval()
        prot = T(tag + ".prot", (2, L, 256)).requires_grad_(True)
        mol = T(tag + ".mol", (2, L, 256)).requires_grad_(True)
        enc, w, gw = m(prot, mol)
        G = T(tag + ".G", tuple(enc.shape))
        (enc * G).sum().backward()
        gn = grad_norms(m)
        out = dict(sd=sd_spec(m), gradnorm=gn, w0=w[0][:, :, :4, :8], gw0=gw[0][:, :, :4, :8], w3=w[3][:, :, :4, :8])
        if L == 64:
            out.update(encoded=enc, dprot=prot.grad, dmol=mol.grad,
                       dW_l0_query=m.encoder.layer_with_mol[0].attn.query.weight.grad[:8, :16],
                       dW_l3_fc2=m.encoder.layer_with_mol[3].ffn.fc2.weight.grad[:8, :16],
                       db_l1_fc_mol=m.encoder.layer_with_mol[1].attn.fc_mol.bias.grad,
                       dpe_mol=m.embeddings.pe_mol.grad[0, :4, :16])
        else:
            out.update(encoded=sub(enc), dprot=sub(prot.grad), dmol=sub(mol.grad))
        save(tag, **out)


class _MaskDropout(torch.nn.Module):
    """Stand-in for ONE nn.Dropout instance of the reference: multiplies by the given masks, one per call in call order —
    F.dropout's arithmetic (x * keep / (1 - p)) with the Bernoulli draw replaced by recorded data."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.calls = masks, 0

    def forward(self, x):
        m = self.masks[self.calls % len(self.masks)]
        self.calls += 1
        return x * m


def gen_pmma_dropout():
    """The reference's PMMA in TRAINING mode (dropout_rate 0.1, the shipped value: default_config.py:67-89) with its Dropout
    modules fed recorded masks: pins WHERE dropout acts (after the positional adds, after GELU and after fc2) and the
    gradients through it.  The GPU test replays the HIP kernels' own masks through the oracle pinned by this fixture."""
    from model.PMMA import PairedMultimodelAttention
    tag, L, B, pdrop = "pmma_drop", 64, 2, 0.1
    cfg = pmma_config(L)
    cfg.transformer.dropout_rate = pdrop
    m = fill_module(PairedMultimodelAttention(cfg, vis=False)).train()
    masks = {k: torch.from_numpy(v) for k, v in synth.pmma_dropout_masks(tag, B, L, 256, pdrop).items()}
    m.embeddings.dropout_mol = _MaskDropout([masks["emb_mol"]])
    m.embeddings.dropout = _MaskDropout([masks["emb_prot"]])
    for i, blk in enumerate(m.encoder.layer_with_mol):
        blk.ffn.dropout = _MaskDropout([masks["l%d.s0.fc1" % i], masks["l%d.s0.fc2" % i]])
        if i < 2:
            blk.ffn_mol.dropout = _MaskDropout([masks["l%d.s1.fc1" % i], masks["l%d.s1.fc2" % i]])
    prot = T(tag + ".prot", (B, L, 256)).requires_grad_(True)
    mol = T(tag + ".mol", (B, L, 256)).requires_grad_(True)
    enc, _, _ = m(prot, mol)
    G = T(tag + ".G", tuple(enc.shape))
    (enc * G).sum().backward()
    save(tag, sd=sd_spec(m), p=np.float32(pdrop), encoded=enc, dprot=prot.grad, dmol=mol.grad, gradnorm=grad_norms(m),
         dW_l0_fc1=m.encoder.layer_with_mol[0].ffn.fc1.weight.grad[:8, :16],
         dW_l1_fc2_mol=m.encoder.layer_with_mol[1].ffn_mol.fc2.weight.grad[:8, :16],
         dW_l3_fc2=m.encoder.layer_with_mol[3].ffn.fc2.weight.grad[:8, :16],
         dpe_prot=m.embeddings.pe_prot.grad[0, :4, :16])


def gen_pmma_long():
    """BASELINE config 5 (long proteins: 1024 sites).  The reference's PMMA accepts any feat_len (embed.py:32-33)."""
    from model.PMMA import PairedMultimodelAttention
    tag, L = "pmma_L1024", 1024
    m = fill_module(PairedMultimodelAttention(pmma_config(L), vis=True)).eval()
    prot = T(tag + ".prot", (2, L, 256)).requires_grad_(True)
    mol = T(tag + ".mol", (2, L, 256)).requires_grad_(True)
    enc, w, gw = m(prot, mol)
    G = T(tag + ".G", tuple(enc.shape))
    (enc * G).sum().backward()
    save(tag, sd=sd_spec(m), gradnorm=grad_norms(m), w0=w[0][:, :, :4, :8], gw0=gw[0][:, :, :4, :8], w3=w[3][:, :, :4, :8],
         gw1_tail=gw[1][:, :, -4:, -8:], encoded=sub(enc), dprot=sub(prot.grad), dmol=sub(mol.grad),
         dW_l0_query=m.encoder.layer_with_mol[0].attn.query.weight.grad[:8, :16],
         dW_l3_fc2=m.encoder.layer_with_mol[3].ffn.fc2.weight.grad[:8, :16],
         dpe_mol=m.embeddings.pe_mol.grad[0, -4:, :16])


def gen_model_long():
    """Whole DrugLAMP at PROTEIN.SEQ_LEN = 9216 (1024 sites): eval score + train-mode BCE gradient norms."""
    from model.basic_model import binary_cross_entropy
    kind, S = "DrugLAMP", 9216
    torch.manual_seed(0)
    m = build_model(kind, seq_len=S)
    fill_module(m)
    vd, vp, xd, xp, y = model_inputs("modelL." + kind, 2, seq_len=S, lp_range=(1000, 4000))
    m.eval()
    with torch.no_grad():
        vd_o, vp_o, ssl, cm, score = m(vd, vp, xd, xp)
    out = dict(sd=sd_spec(m), score=score, vp=sub(vp_o), A_v=m.A_v_gca[:, :, :4, :8], A_x=m.A_x_gca[:, :, -4:, -8:])
    vd, vp, xd, xp, y = model_inputs("modelLtrain." + kind, 4, seq_len=S, lp_range=(1000, 4000))
    m.train()
    m.zero_grad()
    vd_o, vp_o, ssl, cm, score_t = m(vd, vp, xd, xp)
    n, loss = binary_cross_entropy(score_t, y)
    loss.backward()
    out.update(score_train=score_t, cls_loss=loss, gradnorm=grad_norms(m))
    save("model_L1024", **out)


def gen_pgca():
    from model.PGCA.guided_cross_attention_model import GuidedCrossAttention
    m = fill_module(GuidedCrossAttention(embed_dim=128, num_heads=1)).eval()
    for tag, (Lq, Lk, B) in (("pgca_small", (48, 80, 3)), ("pgca_full", (256, 512, 2))):
        q = T(tag + ".q", (Lq, B, 128)).requires_grad_(True)
        kv = T(tag + ".kv", (Lk, B, 128)).requires_grad_(True)
        m.zero_grad()
        out, raw = m(q, kv, kv)
        G = T(tag + ".G", tuple(out.shape))
        (out * G).sum().backward()
        save(tag, sd=sd_spec(m), out=out, raw=raw[:, :, :8, :16], rawnorm=raw.double().norm(dim=-1).float(), dq=q.grad, dkv=kv.grad,
             gradnorm=grad_norms(m))


def gen_mhla():
    from model.PMMA import MultiHeadLinearAttention
    for tag, (d, dd, B, L) in (("mhla_toy", (32, 64, 2, 5)), ("mhla_full", (256, 1024, 2, 256))):
        m = fill_module(MultiHeadLinearAttention(d_model=d, d_diff=dd, nhead=8, dropout=0, activation="gelu")).eval()
        v = T(tag + ".v", (B, L, d)).requires_grad_(True)
        out = m(v)
        G = T(tag + ".G", tuple(out.shape))
        (out * G).sum().backward()
        save(tag, sd=sd_spec(m), out=out, dv=v.grad, gradnorm=grad_norms(m))


def gen_losses():
    import model.self_supervised_learning as S
    import model.cross_modality as CMm
    from utils import sigmoid_cosine_distance_p
    out = {}
    for tag, (n, d) in (("ntx_small", (24, 64)), ("ntx_big", (512, 128))):
        q = T(tag + ".q", (n, d), 0.3).requires_grad_(True)
        k = T(tag + ".k", (n, d), 0.3).requires_grad_(True)
        loss = S.nt_xent_loss(q, k, temperature=0.1)
        loss.backward()
        out[tag + "/loss"] = loss
        out[tag + "/dq"] = q.grad[:8]
        out[tag + "/dk"] = k.grad[:8]
    x = T("cos.x", (300, 128)).requires_grad_(True)
    y = T("cos.y", (300, 128))
    l = S.loss_fn(x, y)
    l.mean().backward()
    out["cos/rows"] = l
    out["cos/dx"] = x.grad
    # triplet loss over an explicit label dict (ccpp_p_tri_loss with the margin-scheduled distance loss)
    n_p, n_d = 12, 17
    pl = torch.nn.functional.normalize(T("tri.p", (n_p, 256)), dim=-1).requires_grad_(True)
    dl = torch.nn.functional.normalize(T("tri.d", (n_d, 256)), dim=-1).requires_grad_(True)
    gtm = (detgen.uniform("tri.gt", (n_p, n_d)) > 0.3).astype(np.int8)
    gtm[0] = 0
    gtm[1] = 1
    pid2t = {"p%d" % i: i for i in range(n_p)}
    did2t = {"d%d" % j: j for j in range(n_d)}
    gt = {"p%d" % i: {"d%d" % j: int(gtm[i, j]) for j in range(n_d)} for i in range(n_p)}
    lf = torch.nn.TripletMarginWithDistanceLoss(distance_function=sigmoid_cosine_distance_p, margin=0.3, reduction="sum")
    tl = CMm.ccpp_p_tri_loss(lf, gt, pid2t, did2t, pl, dl)
    tl.backward()
    out.update({"tri/loss": tl, "tri/dp": pl.grad, "tri/dd": dl.grad, "tri/gt": gtm})
    # margin schedule
    sch = CMm.MarginScheduledLossFunction(CMm.ccpp_p_tri_loss, m_ori=0.5, n_re=100)
    margins = [sch.margin]
    for _ in range(205):
        sch.step()
        margins.append(sch.margin)
    out["margins"] = np.asarray(margins, dtype=np.float64)
    save("losses", **out)


# ---------------------------------------------------------------------------------------------------
# whole models
# ---------------------------------------------------------------------------------------------------
class _PassThrough(torch.nn.Module):
    def forward(self, x):
        return x


def model_inputs(tag, B, salt=0, **kw):
    return tuple(torch.from_numpy(a) for a in synth.model_inputs(tag, B, salt, **kw))


def build_model(kind, seq_len=2304):
    import importlib
    cfg = ref_harness.default_cfg()
    Model = getattr(importlib.import_module("model." + kind), kind)
    if seq_len != 2304:
        # long-protein configuration (BASELINE config 5): every forward of the reference already derives the site count
        # from PROTEIN.SEQ_LEN // SITE_LEN (DrugLAMP.py:35); only the PMMA table length is hard-coded
        # (default_config.py:83 feat_len = 256).  The harness overrides that one number through the reference's own
        # CONFIGS hook (basic_model.py:13-15,98) — the reference's modules run unmodified.
        import model.basic_model as BM
        from configs import get_model_defaults
        cfg.PROTEIN.SEQ_LEN = seq_len

        def lamp(hidden):
            c = get_model_defaults(hidden)
            c.feat_len = c.mol_len = seq_len // cfg.PROTEIN.SITE_LEN
            return c
        old = BM.CONFIGS["LAMP"]
        BM.CONFIGS["LAMP"] = lamp
        try:
            m = Model(n_drug_feature=384, n_prot_feature=640, n_hidden=128, **cfg)
        finally:
            BM.CONFIGS["LAMP"] = old
    else:
        m = Model(n_drug_feature=384, n_prot_feature=640, n_hidden=128, **cfg)
    m.drug_extractor = _PassThrough()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def gen_models():
    for kind in ("DrugLAMP", "DrugLAMP2C2P", "DrugLAMPwoLLM"):
        torch.manual_seed(0)
        m = build_model(kind)
        vd, vp, xd, xp, y = model_inputs("model." + kind, 2)
        fill_module(m)
        m.eval()
        with torch.no_grad():
            vd_o, vp_o, ssl, cm, score = m(vd, vp, xd, xp)
        out = dict(sd=sd_spec(m), score=score, vp=sub(vp_o), A_v=m.A_v_gca[:, :, :4, :8])
        if kind != "DrugLAMPwoLLM":
            out["A_x"] = m.A_x_gca[:, :, :4, :8]
        if cm is not None:
            out["cm_aug_prot"] = sub(cm["aug_prot"])
            out["cm_aug_drug"] = sub(cm["aug_drug"])
        # train-mode BN forward + cls-loss backward (dropout 0); B=8 so that the classifier's BatchNorm
        # batch statistics are well conditioned
        vd, vp, xd, xp, y = model_inputs("modeltrain." + kind, 8)
        m.train()
        m.zero_grad()
        vd_o, vp_o, ssl, cm, score_t = m(vd, vp, xd, xp)
        from model.basic_model import binary_cross_entropy
        n, loss = binary_cross_entropy(score_t, y)
        loss.backward()
        out.update(score_train=score_t, cls_loss=loss, gradnorm=grad_norms(m))
        # a fixed sample of the WHOLE gradient vector (parameters with a gradient, named_parameters order; aliases of the
        # shared ProteinCNN counted once): norms alone pass with a wrong direction
        seen, parts = set(), []
        for k, p_ in m.named_parameters():
            if p_.grad is not None and id(p_) not in seen:
                seen.add(id(p_))
                parts.append(p_.grad.detach().flatten())
        gvec = torch.cat(parts).double()
        gi = torch.from_numpy(np.random.RandomState(123).randint(0, gvec.numel(), 8192))
        out.update(gsample=gvec[gi].float(), gidx=gi, gtotal=np.int64(gvec.numel()), gmax=np.float64(gvec.abs().max().item()))
        save("model_" + kind, **out)


def gen_ssl_cm():
    """SSL (prot MLM with captured masks + SimSiam) and CM losses on a DrugLAMP2C2P instance."""
    import model.self_supervised_learning as S
    torch.manual_seed(0)
    m = build_model("DrugLAMP2C2P")
    B = 6
    vd, vp, xd, xp, y = model_inputs("sslcm", B)
    m.train()
    vd_o, vp_o, ssl, cm, score = m(vd, vp, xd, xp)
    m.ssl_model(**ssl)                      # creates the lazy SimSiam projectors
    fill_module(m)
    # capture the random draws of prot_mlm
    cap = {}
    real_subset, real_like = S.get_mask_subset_with_prob, S.prob_mask_like

    def subset(mask, prob):
        cap["mask"] = real_subset(mask, prob)
        return cap["mask"]

    def like(t, prob):
        cap["replace"] = real_like(t, prob)
        return cap["replace"]

    S.get_mask_subset_with_prob, S.prob_mask_like = subset, like
    torch.manual_seed(7)
    m.zero_grad()
    vd_o, vp_o, ssl, cm, score = m(vd, vp, xd, xp)
    d = m.ssl_model(**ssl)
    S.get_mask_subset_with_prob, S.prob_mask_like = real_subset, real_like
    ((d["prot_ssl"] + d["drug_ssl"]) * 0.1).backward(retain_graph=True)
    gn_ssl = grad_norms(m)
    meta = []
    pid = [0, 1, 0, 2, 3, 1]
    did = [5, 5, 6, 7, 5, 8]
    for t in range(B):
        meta.append({"Prot_ID": pid[t], "Drug_ID": did[t], "Y": float(y[t])})
    m.zero_grad()
    cm_loss = m.cm_model(**cm, meta=meta)
    cm_loss.backward()
    gn_cm = grad_norms(m)
    n_tok = int((vp != 0).sum())
    save("ssl_cm", sd=sd_spec(m), prot_ssl=d["prot_ssl"], drug_ssl=d["drug_ssl"], cm_loss=cm_loss,
         mask=np.packbits(cap["mask"].numpy()), replace=np.packbits(cap["replace"].numpy()),
         n_masked=np.int64(cap["mask"].sum().item()), n_tok=np.int64(n_tok),
         meta_pid=np.asarray(pid), meta_did=np.asarray(did), gradnorm_ssl=gn_ssl, gradnorm_cm=gn_cm)


def gen_train_steps():
    """trainer.py:179-231 driven by hand (Lightning is absent) with the reference model and three
    torch.optim.AdamW over the SAME parameter list (main.py:158-160)."""
    import model.self_supervised_learning as S
    from model.basic_model import binary_cross_entropy
    torch.manual_seed(0)
    m = build_model("DrugLAMP2C2P")
    B = 8
    vd, vp, xd, xp, y = model_inputs("train", B)
    m.train()
    _, _, ssl, cm, _ = m(vd, vp, xd, xp)
    m.ssl_model(**ssl)
    fill_module(m)
    params = list(m.parameters())               # NOTE: taken before... the lazy projectors are excluded
    # reference: optimisers are built in main.py BEFORE the first SSL forward, so the lazily created
    # projectors are in no optimiser.  Reproduce by filtering them out.
    lazy = {id(p) for n, p in m.named_parameters() if ".projector." in n}
    params = [p for p in params if id(p) not in lazy]
    opt = torch.optim.AdamW(params, lr=1e-4)
    opt_ssl = torch.optim.AdamW(params, lr=3e-5)
    opt_cm = torch.optim.AdamW(params, lr=3e-5)
    meta = [{"Prot_ID": [0, 1, 0, 2, 3, 1, 4, 0][t], "Drug_ID": [5, 5, 6, 7, 5, 8, 9, 7][t], "Y": float(y[t])} for t in range(B)]
    masks = {"mask": [], "replace": []}
    real_subset, real_like = S.get_mask_subset_with_prob, S.prob_mask_like

    def subset(mask, prob):
        r = real_subset(mask, prob)
        masks["mask"].append(np.packbits(r.numpy()))
        return r

    def like(t, prob):
        r = real_like(t, prob)
        masks["replace"].append(np.packbits(r.numpy()))
        return r

    S.get_mask_subset_with_prob, S.prob_mask_like = subset, like
    torch.manual_seed(11)
    rec = {"cls": [], "ssl": [], "cm": [], "cm_weight": [], "delta": [], "pnorm": [], "dsample": []}
    cm_weight = 1.0
    n_total = sum(p.numel() for p in params)
    # fixed sub-sample of the flattened parameter vector: the per-step UPDATE of these elements is stored so that a
    # wrong update direction / wrong optimiser ordering is visible (a norm of the update is not: with AdamW from zero
    # moments it is ~ lr * sqrt(N) whatever the gradient)
    didx = (torch.arange(4096, dtype=torch.int64) * (n_total // 4096) + (torch.arange(4096, dtype=torch.int64) * 2654435761 % 97))
    names = [n for n, p in m.named_parameters()]
    for step, cur_epoch in enumerate([1, 5, 5, 6]):
        compute_ssl = cur_epoch % 5 == 0
        compute_cm = cur_epoch >= 5
        before = torch.cat([p.detach().flatten() for p in params]).double()
        _, _, ssl, cm, score = m(vd, vp, xd, xp)
        opt.zero_grad()
        _, cls_loss = binary_cross_entropy(score, y)
        cls_loss.backward(retain_graph=compute_ssl or compute_cm)
        ssl_v, cm_v = 0.0, 0.0
        if compute_ssl:
            opt_ssl.zero_grad()
            d = m.ssl_model(**ssl)
            ssl_loss = (d["prot_ssl"] + d["drug_ssl"]) * 0.1
            ssl_loss.backward(retain_graph=compute_cm)
            ssl_v = ssl_loss.item()
        if compute_cm:
            opt_cm.zero_grad()
            cm_loss = m.cm_model(**cm, meta=meta)
            if cur_epoch == 5 and cm_loss.item() > 0:
                while cm_loss.item() * cm_weight / 10 > cls_loss.item():
                    cm_weight /= 10
                while cm_loss.item() * cm_weight * 10 < cls_loss.item():
                    cm_weight *= 10
            cm_loss = cm_loss * cm_weight
            cm_loss.backward()
            cm_v = cm_loss.item()
        opt.step()
        if compute_ssl:
            opt_ssl.step()
        if compute_cm:
            opt_cm.step()
        after = torch.cat([p.detach().flatten() for p in params]).double()
        rec["cls"].append(cls_loss.item()); rec["ssl"].append(ssl_v); rec["cm"].append(cm_v)
        rec["cm_weight"].append(cm_weight)
        rec["delta"].append((after - before).norm().item()); rec["pnorm"].append(after.norm().item())
        rec["dsample"].append((after - before)[didx].numpy())
    S.get_mask_subset_with_prob, S.prob_mask_like = real_subset, real_like
    save("train_steps", sd=sd_spec(m), cls=np.asarray(rec["cls"]), ssl=np.asarray(rec["ssl"]), cm=np.asarray(rec["cm"]),
         cm_weight=np.asarray(rec["cm_weight"]), delta=np.asarray(rec["delta"]), pnorm=np.asarray(rec["pnorm"]),
         masks=np.stack(masks["mask"]), replaces=np.stack(masks["replace"]),
         didx=didx.numpy(), dsample=np.stack(rec["dsample"]), n_total=np.int64(n_total))


# ---- MolecularGCN: the reference's OWN GCN / GCNLayer / GraphConv code (basic_model.py:137-153, 342-638) on a batched graph
# object that stands in for DGL's batched DGLGraph.  DGL (1.0.2 in the reference's environment) is absent here; what the
# reference takes from it on this path is exactly:  graph.in_degrees() / out_degrees(),  graph.update_all(copy_u('h','m'),
# sum('m','h'))  (dgl message passing: every edge u -> v carries the source's feature, a destination sums what arrives),
# local_scope(), srcdata / dstdata, ndata, batch_size, is_block.  The stand-in implements those with scipy.sparse from an
# edge list (multi-edges kept: DGL counts them).  So this fixture pins everything the reference's source says about the
# GCN; DGL's own SpMM is replaced by the published semantics, not executed ("partially pinned").
def gen_gcn():
    import contextlib
    import scipy.sparse as sp
    import dgl.function as dfn                      # the harness' empty stand-in module: give it the two descriptors
    dfn.copy_u = lambda u, out: ("copy_u", u, out)
    dfn.sum = lambda msg, out: ("sum", msg, out)
    from model.basic_model import MolecularGCN

    class _SpMM(torch.autograd.Function):
        @staticmethod
        def forward(ctx, mat, x):
            ctx.mat = mat
            return torch.from_numpy(np.asarray(mat @ x.detach().double().numpy())).to(x.dtype)

        @staticmethod
        def backward(ctx, g):
            return None, torch.from_numpy(np.asarray(ctx.mat.T @ g.double().numpy())).to(g.dtype)

    class BatchedGraph:
        is_block = False

        def __init__(self, src, dst, n_nodes, batch_size, h):
            self.src, self.dst, self.n, self.batch_size = src, dst, n_nodes, batch_size
            self.ndata = {"h": h}
            self.srcdata = self.dstdata = {}
            # destination-by-source incidence counts: rst[v] = sum over edges u -> v of feat[u]
            self.m = sp.coo_matrix((np.ones(len(src)), (dst, src)), shape=(n_nodes, n_nodes)).tocsr()

        @contextlib.contextmanager
        def local_scope(self):
            saved = dict(self.srcdata)
            try:
                yield
            finally:
                self.srcdata.clear()
                self.srcdata.update(saved)

        def out_degrees(self):
            return torch.from_numpy(np.bincount(self.src, minlength=self.n))

        def in_degrees(self):
            return torch.from_numpy(np.bincount(self.dst, minlength=self.n))

        def update_all(self, message, reduce):
            assert message[0] == "copy_u" and reduce[0] == "sum" and message[2] == reduce[1]
            self.dstdata[reduce[2]] = _SpMM.apply(self.m, self.srcdata[message[1]])

    # graphs as handler/dataset.py:211-222 builds them: bond edges in both directions + a self loop per atom
    # (smiles_to_bigraph(add_self_loop=True)), 512 - n virtual nodes appended, then add_self_loop() over ALL nodes — DGL's
    # add_self_loop does not de-duplicate, so real atoms end up with TWO self loops, virtual nodes with one
    rng = np.random.RandomState(5)
    B, N = 3, 512
    n_atoms = [23, 77, 150]
    src, dst = [], []
    h = np.zeros((B, N, 75), np.float32)
    adj = np.zeros((B, max(n_atoms), max(n_atoms)), np.float32)
    for b, n in enumerate(n_atoms):
        bonds = [(i, i + 1) for i in range(n - 1)] + [tuple(rng.randint(0, n, 2)) for _ in range(n // 5)]
        bonds = sorted({(min(u, v), max(u, v)) for u, v in bonds if u != v})
        e = [(u, v) for u, v in bonds] + [(v, u) for u, v in bonds] + [(i, i) for i in range(n)] + [(i, i) for i in range(N)]
        for u, v in e:
            src.append(b * N + u); dst.append(b * N + v)
            if u < adj.shape[1] and v < adj.shape[1]:
                adj[b, u, v] += 1.0                   # adj[source][destination] over the first max(n_atoms) nodes, multiplicities
                                                      # kept: diagonal 2 for real atoms, 1 for the virtual nodes inside the block
        h[b, :n, :74] = (rng.rand(n, 74) < 0.1)
        h[b, n:, 74] = 1.0
    gcn = fill_module(MolecularGCN(75, 128, True, [128] * 3), salt=41)
    with torch.no_grad():                              # padding=True zeroes the last output row at construction (:141-143); keep it
        gcn.init_transform.weight[-1].fill_(0)
    gcn.train()
    ht = torch.from_numpy(h.reshape(B * N, 75))
    g = BatchedGraph(np.asarray(src), np.asarray(dst), B * N, B, ht)
    out = gcn(g)
    w = T("gcn.cot", (B, N, 128), 1.0)
    (out * w).sum().backward()
    grads = {k: v.grad.detach().clone() for k, v in gcn.named_parameters()}
    save("gcn", sd=sd_spec(gcn), h=h, adj=adj, n_atoms=np.asarray(n_atoms), out=out.detach(),
         g_init=grads["init_transform.weight"], g_conv1=grads["gnn.gnn_layers.1.graph_conv.weight"],
         g_res2=grads["gnn.gnn_layers.2.res_connection.weight"], g_bn0=grads["gnn.gnn_layers.0.bn_layer.weight"],
         gnorm={k.replace(".", "_"): v.norm() for k, v in grads.items()},
         rm2=gcn.gnn.gnn_layers[2].bn_layer.running_mean.detach())


if __name__ == "__main__":
    which = sys.argv[1:] or ["pmma", "pgca", "mhla", "losses", "models", "sslcm", "train", "pmma_long", "model_long", "gcn", "pmma_drop"]
    table = dict(pmma=gen_pmma, pgca=gen_pgca, mhla=gen_mhla, losses=gen_losses, models=gen_models, sslcm=gen_ssl_cm,
                 train=gen_train_steps, pmma_long=gen_pmma_long, model_long=gen_model_long, gcn=gen_gcn, pmma_drop=gen_pmma_dropout)
    for w in which:
        table[w]()
