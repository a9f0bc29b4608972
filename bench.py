"""bench.py — drug-protein pairs/sec of the full DrugLAMP training step on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = trainer.Trainer.training_step on one synthetic batch that is already resident in HBM:
DrugLAMP forward (dense MolecularGCN, ProteinCNN, adaptors, PGCA x2, MHLA x2, PMMA, classifier) +
BCE backward + gradient all-reduce (RCCL, N > 1) + fused AdamW.  Per-GPU batch 256 (the batch
BASELINE.json's metric is quoted on); N ranks process N x 256 pairs per step (weak scaling, like the
reference's DDP which keeps SOLVER.BATCH_SIZE per rank).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HOT_FLOPS_PER_PAIR_STEP = 18.52e9      # PMMA + PGCA, fwd + bwd
MODEL_FLOPS_PER_PAIR_STEP = 24.7e9     # whole model
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
BF16_PEAK_TFLOPS = 2500.0              # dense MFMA peak, MI355X_MICROARCH.md
F32_PEAK_TFLOPS = 157.3


def cpu_baseline(batch_size: int, steps: int, budget_s: float = 25.0):
    """The oracle's training step (oracle/druglamp_oracle.py, pinned to the reference) timed on the
    host cores: cls-loss step + AdamW, fp32, GCN bypassed (post-GCN features) exactly like the survey's
    in-container probe of the real reference."""
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.synthetic import make_batch
    from oracle import druglamp_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    m = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k in list(sd):
        if k.startswith("ssl_model.extractor."):
            sd[k] = sd["protein_extractor." + k[len("ssl_model.extractor."):]]
    (vd, vp, y, xd, xp), _ = make_batch(batch_size, "cpu", seed=3, with_graph=False)
    tr = O.OracleTrainer(sd, "DrugLAMP", use_cm=False)
    t0 = time.perf_counter()
    tr.step(vd, vp, xd, xp, y, cur_epoch=1)          # warm-up (also sizes the bounded sample)
    warm = time.perf_counter() - t0
    steps = max(1, min(steps, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(vd, vp, xd, xp, y, cur_epoch=1)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(batch_size / dt, 3), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d cls-only training steps (fwd+bwd+AdamW) of the CPU oracle at batch %d after 1 warm-up, fp32, "
                      "torch %d threads, post-GCN drug features" % (steps, batch_size, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="DrugLAMP")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=7,
                    help="HIP-event pairs around every N-th launch of a kernel family in the timed region (an event pair "
                         "costs ~6 us of stream time; N=1 times every launch and slows the step by ~6 %%)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=5)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with --nproc-per-node == --gpus (WORLD_SIZE=%d, --gpus %d)" % (world, args.gpus)
    # DL_DIST_BACKEND=gloo lets the N>1 path be smoke-tested on a 1-GPU box (ranks then share device 0)
    backend = os.environ.get("DL_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from druglamp_amd import _lib
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer

    from druglamp_amd import ops
    L = _lib.lib()
    torch.manual_seed(1234)                       # identical initial weights on every rank
    ops.manual_seed(1000 + rank)                  # ... but rank-specific dropout streams
    cfg = load_yaml_into(get_cfg_defaults(), args.model)
    model = MInterface(args.model, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    trainer = Trainer(model, cfg, device=dev, compute_dtype=cdt)
    trainer.set_lrs(cfg["SOLVER"]["LR"], cfg["SOLVER"]["SSL_LR"], cfg["SOLVER"]["CM_LR"])
    batch, meta = make_batch(args.batch, dev, seed=100 + rank, with_graph=True, llm_dtype=cdt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.training_step(batch, meta=meta, cur_epoch=1)
    timing = not args.no_kernel_timing
    sync()
    if timing:
        for fam in (0, 1, 2):
            L.dl_prof_enable(fam, max(args.time_every, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.training_step(batch, meta=meta, cur_epoch=1)
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)

    fam_stats = {}
    if timing:
        for fam, name in ((0, "gemm"), (1, "attn_fwd"), (2, "attn_bwd")):
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            L.dl_prof_collect(fam, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
            na, fla, bya = C.c_int64(), C.c_double(), C.c_double()
            L.dl_prof_totals(fam, C.byref(na), C.byref(fla), C.byref(bya))
            # (timed launches, their ms / flops / bytes, all launches of the timed region, their flops / bytes)
            fam_stats[name] = (n.value, ms.value, fl.value, by.value, na.value, fla.value, bya.value)
            L.dl_prof_enable(fam, 0)

    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / dt
        peak = BF16_PEAK_TFLOPS if args.dtype == "bf16" else F32_PEAK_TFLOPS
        out = {
            "metric": "drug-protein pairs/sec training step",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s training step (fwd+bwd+grad all-reduce+AdamW), BindingDB-shaped synthetic pairs "
                                   "(512 drug nodes/tokens, 2304 protein tokens, pre-extracted 384-d/640-d LLM embeddings), "
                                   "per-GPU batch %d" % (args.model, args.batch),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "hot_path_tflops_per_gpu": round(value * HOT_FLOPS_PER_PAIR_STEP / world / 1e12, 2),
            "hot_path_frac_of_peak": round(value * HOT_FLOPS_PER_PAIR_STEP / world / 1e12 / peak, 4),
        }
        if timing and fam_stats["gemm"][0] > 0:
            n, ms, fl, by, n_all, fl_all, by_all = fam_stats["gemm"]
            ms_all = ms * n_all / n            # family time over the whole timed region, from the timed sample
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r1_pmc_summary.json")
            if os.path.exists(pmc) and args.batch == 256 and args.dtype == "bf16" and args.model == "DrugLAMP":
                # HBM bytes per dl_gemm launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
                # workload (tools/pmc_summary.py; x2 gfx950 read correction), committed under profiles/
                traffic = round(json.load(open(pmc))["families"]["gemm"]["traffic_bytes_per_launch"])
            # Which roofline binds the family: the algorithmic bytes of all launches at the HBM peak vs their flops at
            # the dense MFMA peak.  For this workload (most products have K <= 512) the HBM floor is the larger one.
            t_hbm = by_all / (HBM_PEAK_GBPS * 1e9)
            t_mfma = fl_all / (peak * 1e12)
            mfma_obj = {"achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
            gbps = by / (ms * 1e-3) / 1e9
            hbm_obj = {"achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4)}
            bound = "hbm" if t_hbm >= t_mfma else "mfma"
            out["roofline"] = {"kernel": "dl_gemm (all layouts: fwd / dgrad / wgrad)", "bound": bound}
            out["roofline"].update(hbm_obj if bound == "hbm" else mfma_obj)
            out["roofline"].update({
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
                "algorithmic_bytes_per_launch": round(by_all / n_all), "algorithmic_flops_per_launch": round(fl_all / n_all),
                "launches": n_all, "timed_launches": n, "avg_launch_us": round(ms * 1e3 / n, 2),
                "time_share_of_step": round(ms_all / (dt * 1e3), 3),
                "floor_ms_per_step": {"hbm": round(t_hbm * 1e3 / args.steps, 3), "mfma": round(t_mfma * 1e3 / args.steps, 3),
                                      "measured": round(ms_all / args.steps, 3)},
                "mfma": mfma_obj, "hbm": hbm_obj})
            for name in ("attn_fwd", "attn_bwd"):
                n2, ms2, fl2, _, n2_all, _, _ = fam_stats[name]
                if n2:
                    out["roofline"][name] = {"achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 2), "launches": n2_all,
                                             "timed_launches": n2, "avg_launch_us": round(ms2 * 1e3 / n2, 2),
                                             "time_share_of_step": round(ms2 * n2_all / n2 / (dt * 1e3), 3)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
