"""How long the HOST takes to enqueue one training step (no synchronisation) vs the GPU time of the step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
tr.set_lrs(1e-4, 1e-4, 1e-4)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch, meta = make_batch(B, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(3): tr.training_step(batch, meta=meta, cur_epoch=1)
torch.cuda.synchronize()
# GPU-bound timing
t0 = time.perf_counter()
for _ in range(10): tr.training_step(batch, meta=meta, cur_epoch=1)
torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / 10
# host enqueue time: tiny batch makes the GPU side negligible
small, meta_s = make_batch(2, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(3): tr.training_step(small, meta=meta_s, cur_epoch=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.training_step(small, meta=meta_s, cur_epoch=1)
host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print("step at B=%d: %.2f ms   host enqueue time per step (B=2, no sync): %.2f ms" % (B, gpu * 1e3, host * 1e3))
