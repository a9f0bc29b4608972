"""cProfile of the host side of training steps (tiny batch, no syncs inside): where the ~10 ms of enqueue time go."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
batch, meta = make_batch(2, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(5):
    tr.training_step(batch, meta=meta, cur_epoch=1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.training_step(batch, meta=meta, cur_epoch=1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
