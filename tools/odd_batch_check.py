"""Odd batch sizes through the product path against the CPU oracle (tile-edge / ragged-shape sanity): eval-mode score in
fp32 and bf16, plus one finite bf16 training step each.  Test-infrastructure style use of oracle/ (like smoke())."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
from oracle import druglamp_oracle as O
dev = torch.device("cuda:0")
for name in ("DrugLAMP", "DrugLAMPwoLLM"):
    for B in (1, 3, 5, 13):
        torch.manual_seed(B)
        cfg = load_yaml_into(get_cfg_defaults(), name)
        model = MInterface(name, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
        (feat_d, vp, y, xd, xp), meta = make_batch(B, dev, seed=B, with_graph=True)
        model.eval()
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        errs = []
        for cdt in (torch.float32, torch.bfloat16):
            model.set_compute_dtype(cdt)
            with torch.no_grad():
                _, _, _, _, score = model(feat_d, vp, xd.to(cdt), xp.to(cdt))
            with torch.no_grad():
                vd = model.drug_extractor(feat_d).float().cpu() if not torch.is_tensor(feat_d) else feat_d.cpu()
                ref = O.model_forward(sd, name, vd, vp.cpu(), xd.cpu(), xp.cpu())["score"]
            errs.append(float((score.float().cpu() - ref).abs().max()))
        tr = Trainer(model, cfg, compute_dtype=torch.bfloat16)
        tr.set_lrs(1e-4, 1e-4, 1e-4)
        if B > 1:                                   # BatchNorm needs more than one row in training mode
            out = tr.training_step((feat_d, vp, y, xd.to(torch.bfloat16), xp.to(torch.bfloat16)), meta=meta, cur_epoch=1)
            loss = float(out["cls"])
        else:
            loss = float("nan")
        print("%-14s B=%2d  |score - oracle| fp32 %.2e  bf16 %.2e   train loss %.4f" % (name, B, errs[0], errs[1], loss), flush=True)
        assert errs[0] <= 1e-4 and errs[1] <= 5e-2 and (B == 1 or (loss == loss and loss < 10))
print("odd batch sizes ok")
