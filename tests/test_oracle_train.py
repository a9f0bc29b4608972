"""Pins the oracle's training-step ordering (three AdamW on one parameter list, interleaved zero_grad)
against the sequence recorded from the reference model driven by the reference's own step logic."""
import numpy as np
import torch

from oracle import druglamp_oracle as O
from tests.helpers import det_state_dict, load, model_inputs


def _unpack(bits, shape):
    return torch.from_numpy(np.unpackbits(bits)[:int(np.prod(shape))].reshape(shape).astype(bool))


def check_update_sample(update, g, step, cos_min, frac_min, tol=2e-2):
    """The per-step parameter UPDATE on the golden's fixed 4096-element sub-sample: direction (cosine) and the share
    of elements within `tol` of the largest reference update (AdamW updates are ~lr in magnitude; the few elements
    whose gradient is ~eps flip sign with rounding and are what the share allows for)."""
    assert update.numel() == int(g["n_total"]), (update.numel(), int(g["n_total"]))
    got = update.double().cpu()[torch.from_numpy(g["didx"])]
    ref = torch.from_numpy(g["dsample"][step]).double()
    cos = float((got * ref).sum() / (got.norm() * ref.norm() + 1e-300))
    frac = float(((got - ref).abs() <= tol * ref.abs().max()).double().mean())
    assert cos >= cos_min and frac >= frac_min, (step, cos, frac)


def test_training_step_sequence():
    g = load("train_steps")
    sd = det_state_dict(g)
    # shared ProteinCNN: make the ssl_model.extractor.* entries the SAME tensors as protein_extractor.*
    for k in list(sd):
        if k.startswith("ssl_model.extractor."):
            sd[k] = sd["protein_extractor." + k[len("ssl_model.extractor."):]]
    B = 8
    vd, vp, xd, xp, y = model_inputs("train", B)
    meta = [{"Prot_ID": [0, 1, 0, 2, 3, 1, 4, 0][t], "Drug_ID": [5, 5, 6, 7, 5, 8, 9, 7][t], "Y": float(y[t])} for t in range(B)]
    tr = O.OracleTrainer(sd, "DrugLAMP2C2P")
    before = torch.cat([p.detach().flatten() for p in tr.params]).double()
    mi = 0
    for step, ep in enumerate([1, 5, 5, 6]):
        mask = replace = None
        if ep % 5 == 0:
            mask, replace = _unpack(g["masks"][mi], (B, 2304)), _unpack(g["replaces"][mi], (B, 2304))
            mi += 1
        rec = tr.step(vd, vp, xd, xp, y, meta=meta, cur_epoch=ep, mask=mask, replace=replace)
        after = torch.cat([p.detach().flatten() for p in tr.params]).double()
        assert abs(rec["cls"] - g["cls"][step]) <= 2e-4 * abs(g["cls"][step]), (step, rec["cls"], g["cls"][step])
        assert abs(rec["ssl"] - g["ssl"][step]) <= 2e-4 * max(abs(g["ssl"][step]), 1e-6), (step, rec["ssl"], g["ssl"][step])
        assert abs(rec["cm"] - g["cm"][step]) <= 2e-3 * max(abs(g["cm"][step]), 1e-6), (step, rec["cm"], g["cm"][step])
        assert rec["cm_weight"] == g["cm_weight"][step]
        delta = float((after - before).norm())
        assert abs(delta - g["delta"][step]) <= 2e-2 * g["delta"][step], (step, delta, g["delta"][step])
        check_update_sample(after - before, g, step, cos_min=0.9999, frac_min=0.995)
        before = after
