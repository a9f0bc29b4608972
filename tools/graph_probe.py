"""Feasibility probe: capture one whole cls training step (forward + backward + pack + AdamW) in a HIP graph
(torch.cuda.CUDAGraph drives hipStreamBeginCapture; every libdruglamp_hip launch goes to the capturing stream) and
compare replay time with the eager step at several per-GPU batches.  Dropout seeds / AdamW step counts are frozen at
their capture-time values here — this measures launch cost only, not a valid training run."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer

dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
batches = [int(a) for a in sys.argv[1:]] or [32, 64, 256]
for B in batches:
    torch.manual_seed(0)
    model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
    tr.set_lrs(1e-4, 1e-4, 1e-4)
    batch, meta = make_batch(B, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
    for _ in range(3):
        tr.training_step(batch, meta=meta, cur_epoch=1)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        tr.training_step(batch, meta=meta, cur_epoch=1)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e3
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                tr.training_step(batch, meta=meta, cur_epoch=1)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = tr.training_step(batch, meta=meta, cur_epoch=1)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / n * 1e3
        # host cost of one replay call alone
        t0 = time.perf_counter()
        g.replay()
        host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        print("batch %4d: eager %.2f ms/step   graph replay %.2f ms/step   (host side of one replay %.2f ms)   loss %.4f"
              % (B, eager, graph, host, float(out["cls"])), flush=True)
        del g
    except Exception as e:                                  # noqa: BLE001
        print("batch %4d: eager %.2f ms/step   graph capture FAILED: %s: %s" % (B, eager, type(e).__name__, str(e)[:400]), flush=True)
    del tr, model
    torch.cuda.empty_cache()
