"""Inference (eval-mode forward) throughput at the benchmark batch; not the headline metric."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name in ("DrugLAMP", "DrugLAMP2C2P", "DrugLAMPwoLLM"):
    cfg = load_yaml_into(get_cfg_defaults(), name)
    model = MInterface(name, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    batch, meta = make_batch(B, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
    feat_d, feat_p, labels, llm_d, llm_p = batch
    with torch.no_grad():
        for _ in range(3):
            out = model(feat_d, feat_p, llm_d, llm_p, mode="eval")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            out = model(feat_d, feat_p, llm_d, llm_p, mode="eval")
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%-14s eval forward %.2f ms / batch of %d = %.0f pairs/s  (score %s)" % (name, dt * 1e3, B, B / dt, tuple(out[2].shape)), flush=True)
