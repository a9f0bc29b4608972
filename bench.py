"""bench.py — drug-protein pairs/sec of the full DrugLAMP training step on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = trainer.Trainer.training_step on one synthetic batch that is already resident in HBM:
DrugLAMP forward (dense MolecularGCN, ProteinCNN, adaptors, PGCA x2, MHLA x2, PMMA, classifier) +
BCE backward + gradient all-reduce (RCCL, N > 1) + fused AdamW.

BASELINE.json's metric is quoted on a GLOBAL batch of 256 pairs on 1/2/4/8 GPUs (B = 256 / n_gpu per rank):
that is what `--gpus N` measures ("scaling": "strong", per-GPU batch 256 / N); the line also carries a short weak-scaling
measurement (256 pairs per GPU, the reference DDP's "fixed SOLVER.BATCH_SIZE per rank" regime) under "weak" when N > 1.
`--batch B` fixes the per-GPU batch instead ("scaling": "weak").  `--epoch E` picks the step kind the reference's epoch
gating gives (trainer.py:192-221: SSL heads every RS.EPOCH_STEP-th epoch, CM head from RS.INIT_EPOCH) and
`--global-batch-heads` turns on RS.GLOBAL_BATCH (cross-modal triplets and, with `--drug-ssl simclr`, the NT-Xent denominator
over the all-gathered global batch: config C3).
`--graph auto` (Trainer(graph_steps="auto")): per-GPU batches <= 128 replay the step as a hipGraph (the eager step is host-enqueue
bound there), and so do the steps with the cross-modality head at any batch (600+ small launches); the rest runs eager.
Prints ONE JSON line on rank 0.
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


# the REAL reference's cls-only training step on the build container's 8-core Xeon (survey probe 8.4 /
# 8.8 / 8.2 pairs/s at batch 16 / 32 / 64; the round-5 review re-timed it at 9.3 pairs/s at batch 16, and the oracle below at
# 17.4 pairs/s on the same 8 cores: the oracle runs ~1.9x the reference's speed on equal cores — it calls the same torch CPU
# kernels without the reference's per-op Python / materialised (B, H, L, L) copies — so this baseline errs on the fast side).
REFERENCE_PROBE_PAIRS_PER_S = (8.4, 9.3)
ORACLE_OVER_REFERENCE_EQUAL_CORES = 1.9


def cpu_baseline(batch_sizes, steps: int, budget_s: float = 30.0, threads=(8, 16, 32, 64)):
    """The oracle's training step (oracle/druglamp_oracle.py, pinned to the reference) timed on the
    host cores: cls-loss step + AdamW, fp32, GCN bypassed (post-GCN features) exactly like the survey's
    in-container probe of the real reference.  A bounded sample: first a THREAD sweep at the reference's
    default batch 16 (round 5's 64-thread figure was slower than 8 threads in the build container: oversubscribed), then the
    other batch sizes at the best thread count; `value` is the best sample, the sweeps are reported alongside."""
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.synthetic import make_batch
    from oracle import druglamp_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, cores)
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    m = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k in list(sd0):
        if k.startswith("ssl_model.extractor."):
            sd0[k] = sd0["protein_extractor." + k[len("ssl_model.extractor."):]]
    t_begin = time.perf_counter()

    def sample(bs, nthreads, nsteps):
        torch.set_num_threads(nthreads)
        sd = {k: v.clone() for k, v in sd0.items()}
        (vd, vp, y, xd, xp), _ = make_batch(bs, "cpu", seed=3, with_graph=False)
        tr = O.OracleTrainer(sd, "DrugLAMP", use_cm=False)
        t0 = time.perf_counter()
        tr.step(vd, vp, xd, xp, y, cur_epoch=1)          # warm-up (also sizes the bounded sample)
        warm = time.perf_counter() - t0
        left = budget_s - (time.perf_counter() - t_begin)
        n = max(1, min(nsteps, int(left / max(warm, 1e-3) / 3)))
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(vd, vp, xd, xp, y, cur_epoch=1)
        dt = (time.perf_counter() - t0) / n
        return {"batch": bs, "threads": nthreads, "steps": n, "value": round(bs / dt, 3)}

    tlist = sorted({min(t, cores) for t in threads})
    b0 = batch_sizes[0] if batch_sizes else 16
    thread_sweep = [sample(b0, t, min(steps, 2)) for t in tlist]
    best_t = max(thread_sweep, key=lambda r: r["value"])["threads"]
    sweep = [max((r for r in thread_sweep if r["threads"] == best_t), key=lambda r: r["value"])]
    for bs in batch_sizes[1:]:
        if time.perf_counter() - t_begin > budget_s * 0.8:
            break
        sweep.append(sample(bs, best_t, steps))
    best = max(sweep, key=lambda r: r["value"])
    lo, hi = REFERENCE_PROBE_PAIRS_PER_S
    return {"value": best["value"], "unit": "pairs/s", "cores": best["threads"], "host_cores": cores, "kind": "port",
            "pairs_per_s_per_core": round(best["value"] / best["threads"], 3),
            "thread_sweep": thread_sweep, "sweep": sweep,
            "vs_reference_probe": {"reference_pairs_per_s_8_cores": [lo, hi], "ratio_low_high": [round(best["value"] / hi, 2), round(best["value"] / lo, 2)],
                                   "oracle_over_reference_on_equal_cores": ORACLE_OVER_REFERENCE_EQUAL_CORES},
            "sample": "cls-only training steps (fwd+bwd+AdamW) of the CPU oracle, fp32, post-GCN drug features: thread sweep %s at batch %d "
                      "(1 warm-up + <= %d timed steps each; %s pairs/s), then batch %s at the best thread count %d; value = the best sample "
                      "(batch %d, %d threads of %d host cores).  The real reference measured %.1f-%.1f pairs/s on 8 cores of the build "
                      "container (BASELINE.md section 2); the oracle is ~%.1fx faster than the reference on equal cores, so this is a "
                      "fast-side stand-in for the reference's CPU path" % (
                          "/".join(str(r["threads"]) for r in thread_sweep), b0, min(steps, 2),
                          "/".join("%.1f" % r["value"] for r in thread_sweep), "/".join(str(r["batch"]) for r in sweep), best_t,
                          best["batch"], best["threads"], cores, lo, hi, ORACLE_OVER_REFERENCE_EQUAL_CORES)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--global-batch", type=int, default=256, help="pairs per step over all GPUs (BASELINE.json: 256)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch; 0 = global batch / gpus (strong scaling)")
    ap.add_argument("--epoch", type=int, default=1, help="1-based epoch the step belongs to (selects cls / +SSL / +CM steps)")
    ap.add_argument("--global-batch-cm", "--global-batch-heads", dest="global_batch_cm", action="store_true",
                    help="RS.GLOBAL_BATCH: the batch-level heads see the all-gathered global batch (CM latents; NT-Xent rows)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay steps as a hipGraph (auto: per-GPU batch <= 128, and steps with the CM head at any batch)")
    ap.add_argument("--no-weak", action="store_true", help="skip the extra weak-scaling measurement at N > 1")
    ap.add_argument("--no-projection", action="store_true",
                    help="skip projected_strong_scaling (N = 1 only: the step at per-GPU batch 256 / N for N = 2, 4, 8 on this one GPU)")
    ap.add_argument("--seq-len", type=int, default=2304, help="PROTEIN.SEQ_LEN (9216 = 1024 sites: BASELINE config 5, long proteins)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="DrugLAMP")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-every", type=int, default=7,
                    help="HIP-event pairs around every N-th launch of a kernel family in the timed region (an event pair "
                         "costs ~6 us of stream time; N=1 times every launch and slows the step by ~6 %%)")
    ap.add_argument("--cpu-batch", default="16,32,64", help="batch sizes of the CPU-oracle baseline (comma separated)")
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--drug-ssl", default="simsiam", choices=["simsiam", "simclr"],
                    help="RS.DRUG_SSL_TYPE: simclr = the reference's nt_xent_loss over the B*512 node rows (config C3 with "
                         "--global-batch-heads: its denominator runs over the all-gathered global batch)")
    ap.add_argument("--distinct-batches", type=int, default=8,
                    help="distinct synthetic (batch, meta) pairs cycled through the warm-up and the timed steps — other protein "
                         "lengths, token counts and ids every step, so the per-batch host work (layout spec, table refill, label "
                         "blocks) and the capacity logic of captured graphs are inside the timed region (ADVICE r4)")
    ap.add_argument("--min-busy-seconds", type=float, default=2.0,
                    help="after the K timed steps, keep stepping (untimed) until the GPU has been busy this long in total, so "
                         "that a short --steps run is visible to an external utilisation sampler")
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
    scaling = "weak" if args.batch > 0 else "strong"
    if args.batch <= 0:
        assert args.global_batch % world == 0, "--global-batch must be divisible by --gpus"
        args.batch = args.global_batch // world
    use_graph = {"on": True, "off": False, "auto": "auto"}[args.graph]     # auto: Trainer.wants_graph (batch <= 128, or a CM step)
    torch.manual_seed(1234)                       # identical initial weights on every rank
    ops.manual_seed(1000 + rank)                  # ... but rank-specific dropout streams
    cfg = load_yaml_into(get_cfg_defaults(), args.model)
    if args.global_batch_cm:
        cfg["RS"]["GLOBAL_BATCH"] = True
    cfg["PROTEIN"]["SEQ_LEN"] = args.seq_len
    cfg["RS"]["DRUG_SSL_TYPE"] = args.drug_ssl
    model = MInterface(args.model, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    trainer = Trainer(model, cfg, device=dev, compute_dtype=cdt, graph_steps=use_graph)
    trainer.set_lrs(cfg["SOLVER"]["LR"], cfg["SOLVER"]["SSL_LR"], cfg["SOLVER"]["CM_LR"])
    ep = args.epoch
    kinds = ["cls"] + (["ssl"] if trainer.use_ssl and ep % trainer.ssl_epoch_step == 0 else []) + \
        (["cm"] if trainer.use_cm and ep >= trainer.cm_init_epoch else [])
    # cls, SSL-epoch and CM steps replay a hipGraph; eager: the epoch the CM head starts in (its loss weight is scaled on
    # the host there) and the global-batch CM form at N > 1 (object collectives)
    def graphed_at(per_gpu_batch):
        return bool(trainer.wants_graph(per_gpu_batch, "cm" in kinds) and ("cm" not in kinds or ep > trainer.cm_init_epoch) and
                    trainer.graphed_kind_ok("ssl" in kinds, "cm" in kinds))

    graphed = graphed_at(args.batch)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    batch_sets = {}

    def batches_of(per_gpu_batch):
        """`--distinct-batches` synthetic batches of one size, resident in HBM (generated once per size)."""
        if per_gpu_batch not in batch_sets:
            batch_sets[per_gpu_batch] = [
                make_batch(per_gpu_batch, dev, seed=100 + rank + 1009 * i, with_graph=True, llm_dtype=cdt, seq_len=args.seq_len,
                           max_prot_len=1022 if args.seq_len <= 2304 else args.seq_len // 2 - 2)
                for i in range(max(args.distinct_batches, 1))]
        return batch_sets[per_gpu_batch]

    def measure(per_gpu_batch, steps, warmup, with_events, busy=False):
        """W untimed + K timed steps at one per-GPU batch, cycling through the distinct batches (a replayed graph receives each
        as a device-to-device copy into its static inputs — the stand-in for a loader that assembles the next batch there);
        returns (seconds, max over ranks; family statistics)."""
        pairs = batches_of(per_gpu_batch)
        it = [0]

        def step():
            batch, meta = pairs[it[0] % len(pairs)]
            it[0] += 1
            trainer.training_step(batch, meta=meta, cur_epoch=ep)

        # Untimed, before the W warm-up steps: every distinct batch is stepped once (its row-bucket shapes reach the caching
        # allocator for the first time there, a hipMalloc each) — in graph mode graph_warmup + 1 times, so that the captures the
        # capacity classes need have all been made before the timed region starts.  Reported as config.first_touch_steps.
        first_touch = (trainer.graph_warmup + 1) * len(pairs) if graphed_at(per_gpu_batch) else (len(pairs) if len(pairs) > 1 else 0)
        measure.first_touch_steps = first_touch
        for _ in range(first_touch + warmup):
            step()
        sync()
        if with_events:
            for fam in (0, 1, 2):
                L.dl_prof_enable(fam, max(args.time_every, 1))
        captures0 = trainer.graph_captures
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        measure.captures_in_timed_region = trainer.graph_captures - captures0
        trainer.check_device_flags()              # the padding guards of the compact forms (raises if one tripped)
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        stats = {}
        if with_events:
            for fam, name in ((0, "gemm"), (1, "attn_fwd"), (2, "attn_bwd")):
                n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
                L.dl_prof_collect(fam, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
                na, fla, bya = C.c_int64(), C.c_double(), C.c_double()
                L.dl_prof_totals(fam, C.byref(na), C.byref(fla), C.byref(bya))
                # (timed launches, their ms / flops / bytes, all launches of the region, their flops / bytes)
                stats[name] = (n.value, ms.value, fl.value, by.value, na.value, fla.value, bya.value)
                if fam == 0:
                    # sub-families of dl_gemm (dl_gemm_args.prof_tag): QKV / fc / out projections, FFN, ProteinCNN
                    # convolutions (implicit im2col), weight gradients, everything else
                    sub = {}
                    for tag, tname in _lib.TAG_NAMES.items():
                        L.dl_prof_collect_tag(0, tag, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by), C.byref(na), C.byref(fla), C.byref(bya))
                        if na.value:
                            sub[tname] = (n.value, ms.value, fl.value, by.value, na.value, fla.value, bya.value)
                    stats["gemm_sub"] = sub
                L.dl_prof_enable(fam, 0)
        if busy and float(t) < args.min_busy_seconds:
            # untimed, after the K timed steps and the event collection: keeps the GPU busy long enough for an external
            # utilisation sampler to see a short --steps run (every rank runs the same count: t is the max over ranks)
            for _ in range(int((args.min_busy_seconds - float(t)) / max(float(t) / steps, 1e-6)) + 1):
                step()
            sync()
        return float(t), stats

    timing = not args.no_kernel_timing
    # The timed region runs the product configuration: on cls steps the forward's independent branches (and their backward
    # passes) share the chip on side HIP streams, per-GPU batches <= 128 replay a hipGraph.  Kernel timings for the roofline
    # object come from a SEPARATE instrumented pass of eager ONE-STREAM steps run right after it (same process, same data,
    # same kernels): HIP events bracket a launch on its stream, and a launch that shares the chip with another stream's
    # kernels — or is a node of a replayed graph — has no duration of its own to bracket.
    dt, fam_stats = measure(args.batch, args.steps, args.warmup, False, busy=True)
    timed_captures, live_graphs = getattr(measure, "captures_in_timed_region", 0), len(trainer._graphs)
    first_touch_steps = getattr(measure, "first_touch_steps", 0)
    events_steps, events_dt, events_note = args.steps, dt, None
    if timing:
        saved = (trainer.graph_steps, model.branch_streams)
        trainer.graph_steps, model.branch_streams = False, False
        events_steps = min(args.steps, 20)
        events_dt, fam_stats = measure(args.batch, events_steps, 2, True)
        trainer.graph_steps, model.branch_streams = saved
        events_note = ("HIP events over %d eager one-stream steps run right after the timed region (%.3f ms per step there; the timed "
                       "region %s: its launches overlap or are graph nodes and cannot be bracketed one by one; same kernels, same data)"
                       % (events_steps, events_dt / events_steps * 1e3,
                          "replays a hipGraph" if graphed else "runs the forward's independent branches on side HIP streams"))
    # projected_strong_scaling: what ONE GPU says about the N-GPU strong-scaling curve of the metric — the step
    # at the per-GPU batch 256 / N of N = 2, 4, 8 (hipGraph replays at those sizes), measured here back to back with the headline.
    # ceiling(N) = ms(256) / ms(256 / N): the speed-up N GPUs reach if the gradient all-reduce were free; the driver's SCALE run
    # measures the real curve.
    projection = None
    # (also skipped by --no-kernel-timing: the lean form the A/B and profiling tools run)
    if world == 1 and scaling == "strong" and not (args.no_projection or args.no_kernel_timing) and args.batch == args.global_batch \
            and args.batch % 8 == 0:
        projection = {}
        for n_gpu in (2, 4, 8):
            b = args.batch // n_gpu
            psteps = max(20, min(args.steps, 100))
            pdt, _ = measure(b, psteps, 3, False)
            projection[str(n_gpu)] = {"per_gpu_batch": b, "ms_per_step": round(pdt / psteps * 1e3, 3), "hip_graph": graphed_at(b),
                                      "hip_graph_captures_in_timed_region": getattr(measure, "captures_in_timed_region", 0) if graphed_at(b) else None,
                                      "speedup_ceiling": round((dt / args.steps) / (pdt / psteps), 2)}
    weak = None
    if world > 1 and scaling == "strong" and not args.no_weak:
        wsteps = max(10, min(args.steps // 4, 50))
        wdt, _ = measure(args.global_batch, wsteps, 3, False)
        weak = {"per_gpu_batch": args.global_batch, "global_batch": args.global_batch * world, "steps": wsteps,
                "ms_per_step": round(wdt / wsteps * 1e3, 3), "value": round(args.global_batch * world * wsteps / wdt, 2),
                "unit": "pairs/s"}

    if rank == 0:
        pairs = args.batch * world * args.steps
        value = pairs / dt
        peak = BF16_PEAK_TFLOPS if args.dtype == "bf16" else F32_PEAK_TFLOPS
        out = {
            "metric": "drug-protein pairs/sec training step",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s training step (%s; fwd+bwd+grad all-reduce+AdamW), BindingDB-shaped synthetic pairs "
                                   "(512 drug nodes/tokens, %d protein tokens = %d sites, pre-extracted 384-d/640-d LLM embeddings), "
                                   "global batch %d = %d per GPU x %d%s%s" % (
                                       args.model, "+".join(kinds) + " step, epoch %d" % ep, args.seq_len, args.seq_len // 9,
                                       args.batch * world, args.batch, world,
                                       ", step replayed as a hipGraph" if graphed else "",
                                       (", batch-level heads over the all-gathered global batch (RS.GLOBAL_BATCH)" if args.global_batch_cm else "") +
                                       (", drug SSL = NT-Xent (simclr)" if args.drug_ssl == "simclr" else "")),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "step_kind": "+".join(kinds), "epoch": ep, "hip_graph": bool(graphed), "protein_seq_len": args.seq_len,
                       # the timed steps cycle through this many distinct batches (other lengths / token counts / ids each);
                       # graph mode: graphs live after the run and captures that fell INTO the timed region (0 = none)
                       "distinct_batches": max(args.distinct_batches, 1),
                       "first_touch_steps": first_touch_steps,      # untimed steps in front of the W warm-up steps (one per distinct batch)
                       "hip_graphs_live": live_graphs if graphed else None,
                       "hip_graph_captures_in_timed_region": timed_captures if graphed else None,
                       # identical padding rows of the drug branch (virtual GCN nodes beyond the adjacency block, zero token rows
                       # beyond the collate's Drug_Tokens) are computed once and expanded: same results as computing every row
                       # (tests/test_model_gpu.py); DL_GCN_COMPACT=0 DL_PAD_COMPACT=0 computes every row
                       "padding_rows": "computed once" if (os.environ.get("DL_GCN_COMPACT", "1") != "0" or
                                                           os.environ.get("DL_PAD_COMPACT", "1") != "0") else "every row",
                       # the tiled protein sequences (period L + 2, utils.py:392-412) carry ~L + 31 distinct ProteinCNN rows of
                       # 2304: the CNN runs on those (weighted BatchNorm, device-side periodicity guard); DL_CNN_COMPACT=0 = all
                       "protein_cnn_rows": "distinct rows" if os.environ.get("DL_CNN_COMPACT", "1") != "0" else "every position",
                       # cls steps run MolecularGCN / ProteinCNN / the LLM adaptors / the v cross-attention branch concurrently
                       "streams": "independent branches on side HIP streams" if (model.branch_streams and kinds == ["cls"]) else "one stream"},
            # (the per-pair flop count of BASELINE.md section 3 is for 256 sites; not applicable to other lengths)
            "hot_path_tflops_per_gpu": round(value * HOT_FLOPS_PER_PAIR_STEP / world / 1e12, 2) if args.seq_len == 2304 else None,
            "hot_path_frac_of_peak": round(value * HOT_FLOPS_PER_PAIR_STEP / world / 1e12 / peak, 4) if args.seq_len == 2304 else None,
        }
        if timing and fam_stats["gemm"][0] > 0:
            n, ms, fl, by, n_all, fl_all, by_all = fam_stats["gemm"]
            ms_all = ms * n_all / n            # family time over the whole instrumented region, from the timed sample
            dt_ev, steps_ev = events_dt, events_steps
            ach = fl / (ms * 1e-3) / 1e12
            traffic, traffic_source, traffic_spread = None, None, None
            if args.batch == 256 and args.dtype == "bf16" and args.model == "DrugLAMP" and kinds == ["cls"] and args.seq_len == 2304:
                # HBM bytes per dl_gemm launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same workload
                # (tools/pmc_summary.py; x2 gfx950 read correction), committed under profiles/ — a recorded counter measurement
                # of the same command, NOT re-measured by this run.  Used only when the summary was taken at THESE kernel
                # sources (content hash of csrc/ + the C-ABI header): a stale file gives traffic = null, not a wrong figure.
                from druglamp_amd.build import csrc_hash
                here = csrc_hash()
                cands = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_summary.json")), reverse=True)
                traffic_source = "no profiles/*pmc_summary.json taken at these kernel sources (csrc sha1 %s): traffic not reported" % here[:12]
                for pmc_name in cands:
                    rec = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
                    if rec.get("csrc_sha1") != here:
                        continue
                    fam = rec["families"]["gemm"]
                    traffic = round(fam["traffic_bytes_per_launch"])
                    if "traffic_bytes_per_launch_min" in fam:
                        traffic_spread = {"min": round(fam["traffic_bytes_per_launch_min"]), "median": traffic,
                                          "max": round(fam["traffic_bytes_per_launch_max"]), "passes": rec.get("passes")}
                    traffic_source = ("profiles/%s (rocprofv3 --pmc passes of this command at the same kernel sources, csrc sha1 %s; "
                                      "recorded earlier, not measured by this run)" % (pmc_name, here[:12]))
                    break
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
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_source,
                "traffic_passes": traffic_spread,
                "timing_source": events_note,
                "algorithmic_bytes_per_launch": round(by_all / n_all), "algorithmic_flops_per_launch": round(fl_all / n_all),
                "launches": n_all, "timed_launches": n, "avg_launch_us": round(ms * 1e3 / n, 2),
                "time_share_of_step": round(ms_all / (dt_ev * 1e3), 3),
                "floor_ms_per_step": {"hbm": round(t_hbm * 1e3 / steps_ev, 3), "mfma": round(t_mfma * 1e3 / steps_ev, 3),
                                      "measured": round(ms_all / steps_ev, 3)},
                "mfma": mfma_obj, "hbm": hbm_obj})
            # per sub-family: which roofline binds it (its algorithmic bytes at the HBM peak vs its flops at the MFMA peak) and
            # how far from it the launches run; `furthest` names the sub-family with the most time above its floor
            subs, worst, worst_gap = {}, None, -1.0
            for tname, (sn, sms, sfl, sby, sn_all, sfl_all, sby_all) in fam_stats.get("gemm_sub", {}).items():
                if not sn:
                    subs[tname] = {"launches": sn_all, "timed_launches": 0}
                    continue
                s_ms_all = sms * sn_all / sn
                s_hbm, s_mfma = sby_all / (HBM_PEAK_GBPS * 1e9) * 1e3, sfl_all / (peak * 1e12) * 1e3     # ms over the region
                s_bound = "hbm" if s_hbm >= s_mfma else "mfma"
                s_gbps, s_tf = sby / (sms * 1e-3) / 1e9, sfl / (sms * 1e-3) / 1e12
                subs[tname] = {"bound": s_bound, "frac": round((s_gbps / HBM_PEAK_GBPS) if s_bound == "hbm" else (s_tf / peak), 4),
                               "hbm_GBps": round(s_gbps, 1), "hbm_frac": round(s_gbps / HBM_PEAK_GBPS, 4),
                               "mfma_TFLOPs": round(s_tf, 2), "mfma_frac": round(s_tf / peak, 4),
                               "launches": sn_all, "timed_launches": sn, "avg_launch_us": round(sms * 1e3 / sn, 2),
                               "algorithmic_bytes_per_launch": round(sby_all / sn_all),
                               "ms_per_step": round(s_ms_all / steps_ev, 3),
                               "floor_ms_per_step": round(max(s_hbm, s_mfma) / steps_ev, 3)}
                gap = (s_ms_all - max(s_hbm, s_mfma)) / steps_ev
                if gap > worst_gap:
                    worst, worst_gap = tname, gap
            out["roofline"]["sub_families"] = subs
            out["roofline"]["furthest"] = {"sub_family": worst, "ms_per_step_above_floor": round(worst_gap, 3)}
            for name in ("attn_fwd", "attn_bwd"):
                n2, ms2, fl2, by2, n2_all, _, _ = fam_stats[name]
                if n2:
                    tf2, gb2 = fl2 / (ms2 * 1e-3) / 1e12, by2 / (ms2 * 1e-3) / 1e9
                    out["roofline"][name] = {"achieved": round(tf2, 2), "unit": "TFLOP/s", "mfma_frac": round(tf2 / peak, 4),
                                             "hbm_achieved_GBps": round(gb2, 1), "hbm_frac": round(gb2 / HBM_PEAK_GBPS, 4),
                                             "launches": n2_all, "timed_launches": n2, "avg_launch_us": round(ms2 * 1e3 / n2, 2),
                                             "time_share_of_step": round(ms2 * n2_all / n2 / (events_dt * 1e3), 3)}
            # attention BLOCK level (the 40 %-of-MFMA target only makes sense with the projections counted in): the
            # QKV / fc / out projections (forward and data gradients, tag qkv_out) + the attention cores of every PMMA layer
            # and of PGCA, flops over time
            sub = fam_stats.get("gemm_sub", {}).get("qkv_out")
            if sub and sub[0] and fam_stats["attn_fwd"][0] and fam_stats["attn_bwd"][0]:
                fl_b, ms_b = sub[5], sub[1] * sub[4] / sub[0]
                for name in ("attn_fwd", "attn_bwd"):
                    n2, ms2, fl2, _, n2_all, fl2_all, _ = fam_stats[name]
                    fl_b += fl2_all
                    ms_b += ms2 * n2_all / n2
                tf_b = fl_b / (ms_b * 1e-3) / 1e12
                out["roofline"]["attention_block"] = {
                    "what": "QKV / fc / out projections (fwd + dgrad) + attention cores (fwd + bwd), all PMMA layers and PGCA",
                    "achieved": round(tf_b, 2), "unit": "TFLOP/s", "mfma_frac": round(tf_b / peak, 4),
                    "ms_per_step": round(ms_b / events_steps, 3)}
        if weak is not None:
            out["weak"] = weak
        if projection is not None:
            out["projected_strong_scaling"] = dict(projection, what="one-GPU measurement of the per-GPU step at batch 256 / N; speedup_ceiling = "
                                                   "ms(256) / ms(256 / N) = the N-GPU strong-scaling speed-up with free collectives")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline([int(b) for b in str(args.cpu_batch).split(",") if b], args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
