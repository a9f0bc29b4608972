import os, sys, torch
sys.path.insert(0, "/root/repo")
from collections import defaultdict
from torch.profiler import profile, ProfilerActivity
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
tr.set_lrs(1e-4, 1e-4, 1e-4)
batch, meta = make_batch(256, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(2): tr.training_step(batch, meta=meta, cur_epoch=5)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.training_step(batch, meta=meta, cur_epoch=5)
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_time_total > 0 and ev.device_type.name != "CPU":
        agg[ev.name[:90]][0] += ev.device_time_total; agg[ev.name[:90]][1] += 1
tot = sum(v[0] for v in agg.values())
print("total %.1f ms" % (tot / 1e3))
for k, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print("%8.2f ms %4d  %s" % (us / 1e3, n, k))
