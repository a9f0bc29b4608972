"""Evaluation path and config-level checks on the GPU (BASELINE configs 1 and 2):
  * Trainer.evaluate on 64 synthetic pairs: AUROC / AUPRC equal sklearn on the ORACLE's scores for the same weights and
    inputs (reference metrics: trainer.py:109-119,256-292); batch size of the evaluation does not matter (the reference
    validates at batch 1, main.py:146-153);
  * Trainer.fit keeps the best-`val_ausum` parameters and stops after patience = epochs / 4 epochs without improvement
    (reference ModelCheckpoint / EarlyStopping, trainer.py:150-163,134);
  * config 2 at the bench's size: bf16 eval scores of DrugLAMP at batch 256 against the oracle on a 16-sample slice;
  * config 1: DrugLAMPwoLLM on REAL rows of datasets/human/random (tests/golden/human_random_rows.npz), batch 32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model(kind, dtype, seed=0, dropout=None):
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    torch.manual_seed(seed)
    cfg = load_yaml_into(get_cfg_defaults(), kind)
    m = MInterface(kind, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(DEV)
    if dropout is not None:
        m.pmma.p_drop = dropout
        m.pmma.embeddings.p_drop = dropout
    m.set_compute_dtype(dtype)
    return m, cfg


def _oracle_scores(m, kind, vd, vp, xd, xp):
    from oracle import druglamp_oracle as O
    sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        return O.model_forward(sd, kind, vd.float().cpu(), vp.cpu(), None if xd is None else xd.float().cpu(), xp.float().cpu())["score"]


def test_evaluate_matches_sklearn_on_oracle_scores():
    from sklearn.metrics import average_precision_score, roc_auc_score
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer
    m, cfg = _model("DrugLAMP", torch.float32)
    tr = Trainer(m, cfg, device=DEV, compute_dtype=torch.float32)
    (vd, vp, y, xd, xp), _ = make_batch(64, DEV, seed=11, with_graph=False)
    # scores of an untrained model are nearly constant: spread them with a few training steps so that ranks are meaningful
    tr.set_lrs(1e-3)
    for _ in range(3):
        tr.training_step((vd, vp, y, xd, xp), cur_epoch=1)
    chunks = lambda bs: [tuple(t[i:i + bs] for t in (vd, vp, y, xd, xp)) for i in range(0, 64, bs)]   # noqa: E731
    ev = tr.evaluate(chunks(16))
    ev1 = tr.evaluate(chunks(1)[:64])                     # the reference's batch-1 validation
    ref = torch.sigmoid(_oracle_scores(m, "DrugLAMP", vd, vp, xd, xp)).squeeze(1).numpy()
    yy = y.cpu().numpy()
    assert abs(ev["auroc"] - roc_auc_score(yy, ref)) <= 1e-6 and abs(ev["auprc"] - average_precision_score(yy, ref)) <= 1e-6
    assert abs(ev["ausum"] - (ev["auroc"] + ev["auprc"])) <= 1e-12
    assert abs(ev1["auroc"] - ev["auroc"]) <= 1e-6 and abs(ev1["auprc"] - ev["auprc"]) <= 1e-6
    bce = float(torch.nn.functional.binary_cross_entropy(torch.from_numpy(ref), y.cpu()))
    assert abs(ev["loss"] - bce) <= 1e-4 * max(bce, 1.0) and abs(ev1["loss"] - bce) <= 1e-4 * max(bce, 1.0)


def test_fit_keeps_best_val_ausum_and_stops_on_patience():
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer
    m, cfg = _model("DrugLAMP", torch.bfloat16)
    cfg["SOLVER"]["MAX_EPOCH"] = 8                         # patience = 8 / 4 = 2
    tr = Trainer(m, cfg, device=DEV, compute_dtype=torch.bfloat16)
    train, meta = make_batch(16, DEV, seed=1, with_graph=True, llm_dtype=torch.bfloat16)
    val, _ = make_batch(32, DEV, seed=2, with_graph=True, llm_dtype=torch.bfloat16)
    script = iter([1.2, 1.5, 1.4, 1.3, 1.9, 1.0, 1.0, 1.0])     # scripted validation scores: best at epoch 2, stop at 4
    snaps = {}
    real_eval = tr.evaluate

    def fake_eval(batches):
        out = real_eval(batches)
        out["ausum"] = next(script)
        return out
    tr.evaluate = fake_eval
    res = tr.fit(lambda: [(train, meta)], lambda: [val], on_epoch=lambda ep, v: snaps.__setitem__(ep, tr.flat.arena.clone()))
    assert res["best_epoch"] == 2 and res["epochs_run"] == 4 and res["best_val_ausum"] == 1.5
    assert torch.equal(tr.flat.arena, snaps[2]) and not torch.equal(snaps[4], snaps[2])     # best parameters reloaded
    # epoch means of the step losses ride along (the reference logs them with on_epoch=True, sync_dist=True)
    h = res["history"][0]
    assert 0.0 < h["train_loss"] < 10.0 and h["all_loss"] == h["train_loss"] + h["ssl_loss"] + h["cm_loss"]


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 3e-2), (torch.float32, 1e-4)])
def test_config2_eval_scores_at_batch_256_vs_oracle_slice(dtype, tol):
    """The bench's configuration (DrugLAMP, batch 256): scores of 16 samples taken from the batch-256 forward against the
    fp32 oracle run on those 16 samples alone (eval mode: per-sample independent) — bf16 at the bf16 tolerance, fp32 at the
    north-star tolerance 1e-4 (the large-tile / many-tile kernel forms only occur at this size)."""
    from druglamp_amd.synthetic import make_batch
    m, cfg = _model("DrugLAMP", dtype)
    (vd, vp, y, xd, xp), _ = make_batch(256, DEV, seed=21, with_graph=False, llm_dtype=dtype)
    m.eval()
    with torch.no_grad():
        score = m(vd, vp, xd, xp)[4].float().cpu()
    sl = slice(100, 116)
    ref = _oracle_scores(m, "DrugLAMP", vd[sl], vp[sl], xd[sl], xp[sl])
    assert float((score[sl] - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    if dtype == torch.float32:
        return
    with torch.no_grad():
        alone = m(vd[sl], vp[sl], xd[sl], xp[sl])[4].float().cpu()
    assert float((alone - score[sl]).abs().max()) <= 2e-2          # batch-size independence of the eval path (bf16 tiles differ)


def test_config1_wollm_on_real_human_rows_batch_32():
    """BASELINE config 1: DrugLAMPwoLLM, datasets/human random split (real rows, ids, labels, residue codes; synthetic graph
    and embedding features), batch 32.  fp32 eval scores of the first batch equal the oracle's; a few training steps run
    and the evaluation over the validation rows is finite and identical for two evaluation batch sizes."""
    from druglamp_amd.data import RowTable
    from druglamp_amd.trainer import Trainer
    tab = RowTable(os.path.join(GOLD, "human_random_rows.npz"), DEV, llm_dtype=torch.float32)
    assert tab.n_rows("train") == 1024 and tab.n_rows("val") == 256
    m, cfg = _model("DrugLAMPwoLLM", torch.float32, dropout=0.0)
    (feat_d, vp, y, xd, xp), meta = next(tab.batches("train", 32))
    assert vp.shape == (32, 2304) and vp.dtype == torch.float64 and xp.shape == (32, 2304, 640) and meta[0]["Prot_ID"] == 0
    m.eval()
    with torch.no_grad():
        out = m(feat_d, vp, xd, xp)
        score = out[4].cpu()
    vd_nodes = out[0]                                   # what the dense GCN produced: the oracle takes post-GCN features
    ref = _oracle_scores(m, "DrugLAMPwoLLM", vd_nodes, vp, None, xp)
    assert float((score - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    tr = Trainer(m, cfg, device=DEV, compute_dtype=torch.float32)
    tr.set_lrs(1e-4)
    losses = [float(tr.training_step(b, meta=mt, cur_epoch=1)["cls"]) for b, mt in list(tab.batches("train", 32, shuffle_seed=0, drop_last=True))[:4]]
    assert all(np.isfinite(losses))
    e32 = tr.evaluate(tab.batches_only("val", 32))
    e7 = tr.evaluate(tab.batches_only("val", 7))
    assert np.isfinite(e32["auroc"]) and abs(e32["auroc"] - e7["auroc"]) <= 1e-6 and abs(e32["loss"] - e7["loss"]) <= 1e-5


def test_config2_training_step_at_batch_256_vs_oracle():
    """Training-mode parity at the bench's FULL size (the golden fixtures stop at batch 8; BatchNorm couples the samples of a
    training batch, so no slice of it can stand in): one cls step of DrugLAMP at batch 256 in the fp32 pipeline against the CPU
    oracle's step on the same weights and inputs (oracle/druglamp_oracle.py::OracleTrainer, pinned by the reference's
    train_steps golden; run in fp64) — loss at the north-star tolerance 1e-4, every parameter's gradient against the oracle's by relative
    L2 error.  The large-tile GEMMs, the many-chunk reductions and the compact row forms only occur at this size."""
    import time
    from oracle import druglamp_oracle as O
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer
    (vd, vp, y, xd, xp), meta = make_batch(256, DEV, seed=33, with_graph=False)
    # the bench's own dtype on the same weights and inputs (same seed -> same initial weights), checked against the same oracle step below
    mb, cfgb = _model("DrugLAMP", torch.bfloat16, dropout=0.0)
    trb = Trainer(mb, cfgb, device=DEV, compute_dtype=torch.bfloat16)
    nb = {id(p): n for n, p in mb.named_parameters()}
    outb = trb.training_step((vd, vp, y, xd.bfloat16(), xp.bfloat16()), meta=meta, cur_epoch=1)
    trb.check_device_flags()
    gotb = {nb[id(p)]: g.detach().double().cpu().clone() for p, g in zip(trb.flat.params, trb.flat.grad_views) if id(p) in nb}
    del mb, trb
    m, cfg = _model("DrugLAMP", torch.float32, dropout=0.0)
    tr = Trainer(m, cfg, device=DEV, compute_dtype=torch.float32)
    # (the oracle runs in fp64 here: at 590 000 conv rows per BatchNorm channel an fp32 CPU reduction is no better a yardstick than
    #  the kernels under test)
    sd = {k: (v.detach().double() if v.is_floating_point() else v.detach()).cpu().clone() for k, v in m.state_dict().items()}
    names = {id(p): n for n, p in m.named_parameters()}
    out = tr.training_step((vd, vp, y, xd, xp), meta=meta, cur_epoch=1)
    tr.check_device_flags()
    got = {names[id(p)]: g.detach().double().cpu().clone() for p, g in zip(tr.flat.params, tr.flat.grad_views) if id(p) in names}
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    t0 = time.time()
    ot = O.OracleTrainer(sd, "DrugLAMP", use_ssl=False, use_cm=False)
    rec = ot.step(vd.double().cpu(), vp.cpu(), xd.double().cpu(), xp.double().cpu(), y.double().cpu(), cur_epoch=1)
    print("oracle step at batch 256: %.1f s" % (time.time() - t0))
    assert abs(float(out["cls"]) - rec["cls"]) <= 1e-4 * max(1.0, abs(rec["cls"])), (float(out["cls"]), rec["cls"])
    rows = []
    gmax = max(float(v.grad.norm()) for v in sd.values() if getattr(v, "grad", None) is not None)
    for k, v in sd.items():
        if getattr(v, "grad", None) is None or k not in got:
            continue
        ref = v.grad
        # (parameters whose true gradient is zero carry rounding noise only: the embedding Linear the reference never uses, and
        #  every bias in front of a training-mode BatchNorm — the batch mean absorbs it)
        if float(ref.norm()) < 1e-4 * gmax:
            continue
        rows.append((float((got[k] - ref).norm() / ref.norm()), k, float(ref.norm())))
    rows.sort(reverse=True)
    # The ProteinCNN parameters in front of its last BatchNorm are ill-conditioned in fp32 at this size: that BatchNorm's backward
    # subtracts the batch mean and the projection on y-hat from an upstream gradient that is mostly exactly those two components
    # (torch's own fp32 CPU path lands at 2e-3 - 3.5e-3 against fp64 on the same layer, these kernels at 4e-4 - 2e-3:
    # DESIGN section 5); everything else is held to 5e-4 (measured: 1.4e-4).
    soft = lambda k: k.startswith("protein_extractor.") and ".bn3." not in k     # noqa: E731
    worst_soft = max((r for r in rows if soft(r[1])), default=(0.0, None, 0.0))
    worst_rest = max((r for r in rows if not soft(r[1])), default=(0.0, None, 0.0))
    print("gradients checked: %d of %d; worst relative L2 error: ProteinCNN before its last BatchNorm %.1e (%s), all others %.1e (%s)" % (
        len(rows), len(got), worst_soft[0], worst_soft[1], worst_rest[0], worst_rest[1]))
    assert len(rows) >= 100 and worst_soft[0] <= 5e-3 and worst_rest[0] <= 5e-4, (worst_soft, worst_rest)
    # bf16 pipeline (the bench's): loss at the bf16 tolerance; direction of the whole gradient and of every large parameter's
    assert abs(float(outb["cls"]) - rec["cls"]) <= 3e-2 * max(1.0, abs(rec["cls"])), (float(outb["cls"]), rec["cls"])
    keys = [k for _, k, _ in rows]
    a = torch.cat([gotb[k].reshape(-1) for k in keys])
    b = torch.cat([sd[k].grad.reshape(-1) for k in keys])
    cos_all = float(torch.dot(a, b) / (a.norm() * b.norm()))
    big = [k for _, k, n in rows if n >= 0.05 * gmax]
    cos_min = min((float(torch.dot(gotb[k].reshape(-1), sd[k].grad.reshape(-1)) / (gotb[k].norm() * sd[k].grad.norm())), k) for k in big)
    print("bf16 step: loss %.6f (fp64 %.6f), cosine of the whole gradient %.5f, lowest cosine among the %d largest parameters %.4f (%s)" % (
        float(outb["cls"]), rec["cls"], cos_all, len(big), cos_min[0], cos_min[1]))
    assert cos_all >= 0.995 and cos_min[0] >= 0.98, (cos_all, cos_min)
