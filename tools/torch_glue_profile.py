"""Attribute the non-HIP-library (torch glue) kernels of one training step to Python call sites."""
import os, sys, torch
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
trainer = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
EPOCH = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 5: an SSL-epoch step (cls forward + SSL heads)
batch, meta = make_batch(B, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(3):
    trainer.training_step(batch, meta=meta, cur_epoch=EPOCH)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    trainer.training_step(batch, meta=meta, cur_epoch=EPOCH)
    torch.cuda.synchronize()
agg = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue  # count leaves only
    site = "?"
    for fr in ev.stack or []:
        if "druglamp_amd" in fr and "ops.py" not in fr:
            site = fr.split("druglamp_amd/")[-1]
            break
    if site == "?" and ev.stack:
        site = "|".join(f.split("/")[-1] for f in ev.stack[:2])
    if site == "?":
        site = "thread %s seq %s" % (ev.thread, getattr(ev, "sequence_nr", "-"))
    key = (ev.name, str(ev.input_shapes)[:70], site[:70])
    agg[key][0] += ev.device_time_total
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print("torch-op device time %.2f ms" % (tot / 1e3))
for k, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:90]:
    print("%8.1f us %3d  %-18s %-70s %s" % (us, n, k[0], k[1], k[2]))
