"""hipGraph-replayed training steps issued back to back WITHOUT any host synchronisation (what bench.py's timed loop does; a
training loop that reads the loss every step synchronises and never sees this regime): hundreds to thousands of replays of one or
two live graphs per step kind.  Round 5 found a configuration (the cls+ssl+cm step of DrugLAMP2C2P with an experimental gathered
masked-LM head) that raised a GPU hardware exception after ~300 unsynchronised replays while every variant with a synchronisation
each 8 steps, every eager run, and every other step kind ran 2000-3000 replays clean; that head was removed
and this soak is the regression check for the regime.
    python tools/graph_nosync_soak.py MODEL EPOCH BATCH N_DISTINCT_BATCHES     (env: NSTEPS=2000 SYNC_EVERY=0 GRAPH=1)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
name, epoch, B, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = load_yaml_into(get_cfg_defaults(), name)
model = MInterface(name, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16, graph_steps=os.environ.get('GRAPH', '1') == '1')
tr.set_lrs(1e-4, 3e-5, 1e-5)
bs = [make_batch(B, dev, seed=100 + 1009 * i, with_graph=True, llm_dtype=torch.bfloat16) for i in range(nb)]
for i in range(int(os.environ.get('NSTEPS', '40'))):
    b, m = bs[i % nb]
    n0 = tr.graph_captures
    out = tr.training_step(b, meta=m, cur_epoch=epoch)
    every = int(os.environ.get("SYNC_EVERY", "1"))
    if every and i % every == every - 1:
        torch.cuda.synchronize()
    gk = [(g.block, g.rows_cap, g.replays) for g in tr._graphs.values()]
    print("step", i, "batch", i % nb, "captures", tr.graph_captures, "graphs", gk, flush=True)
torch.cuda.synchronize()
print("done")
