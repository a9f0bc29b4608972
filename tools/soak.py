"""Soak: many steps incl. SSL/CM epochs; prints allocator high-water marks and finiteness (leak / drift check)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "DrugLAMP2C2P"
graph = "--graph" in sys.argv                      # cls-only steps replayed as hipGraphs (two batch tensors -> input copies)
B = 32 if graph else 256
cfg = load_yaml_into(get_cfg_defaults(), name)
model = MInterface(name, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16, graph_steps=graph)
tr.set_lrs(cfg["SOLVER"]["LR"], cfg["SOLVER"]["SSL_LR"], cfg["SOLVER"]["CM_LR"])
batches = [make_batch(B, dev, seed=100 + i, with_graph=True, llm_dtype=torch.bfloat16) for i in range(2)]
t0 = time.perf_counter()
for epoch in range(1, 13):
    for it in range(100 if graph else 10):
        batch, meta = batches[it % 2]
        out = tr.training_step(batch, meta=meta, cur_epoch=epoch)
    tr.on_train_epoch_end(epoch)
    torch.cuda.synchronize()
    vals = {k: float(v) for k, v in out.items()}
    assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
    print("epoch %2d  %s  alloc %.2f GB  reserved %.2f GB  max %.2f GB" % (
        epoch, {k: round(v, 4) for k, v in vals.items()}, torch.cuda.memory_allocated() / 2**30,
        torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30), flush=True)
for g_ in tr._graphs.values():      # DL_GRAPH_PTR_AUDIT=1 python tools/soak.py --graph : pointer audit of every captured step
    if getattr(g_, "audit", None):
        print("pointer audit of a captured step: %d device pointers, %d in the graph's pool, %d in pinned buffers, 0 elsewhere" % g_.audit)
print("%d steps in %.1f s%s" % (12 * (100 if graph else 10), time.perf_counter() - t0,
                              "  (graph replays: %d)" % sum(g.replays for g in tr._graphs.values()) if graph else ""))
