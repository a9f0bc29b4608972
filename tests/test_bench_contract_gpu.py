"""bench.py's output contract (one JSON line on stdout with the keys the driver reads), on a short run: default workload
(eager, batch 256) incl. the roofline and cpu_baseline objects, and a graph-replayed small-batch run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=900,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys():
    d = _run("--steps", "4", "--warmup", "2", "--cpu-steps", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["config"]["global_batch"] == 256 and d["config"]["per_gpu_batch"] == 256 and "workload" in d["config"]
    assert "model" not in d["config"] and d["config"]["hip_graph"] is False
    # the timed steps cycle through distinct batches; each is stepped once, untimed, in front of the W warm-up steps
    assert d["config"]["distinct_batches"] == 8 and d["config"]["first_touch_steps"] == 8
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "timing_source"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "pairs/s"
    # a thread sweep at the reference's batch 16, then the other batch sizes (as far as the time budget reaches) at the best count
    assert c["sweep"][0]["batch"] == 16 and c["value"] == max(r["value"] for r in c["sweep"]) and len(c["thread_sweep"]) >= 1
    assert c["cores"] == c["sweep"][0]["threads"] <= c["host_cores"] and abs(c["pairs_per_s_per_core"] - c["value"] / c["cores"]) < 1e-2
    assert c["vs_reference_probe"]["reference_pairs_per_s_8_cores"] == [8.4, 9.3]
    # traffic comes from a committed counter summary taken at THESE kernel sources, or is null and says why
    assert (r["traffic"] is None) == ("not reported" in r["traffic_source"])
    # one-GPU projection of the strong-scaling curve: the step at per-GPU batch 128 / 64 / 32 as hipGraph replays
    p = d["projected_strong_scaling"]
    assert [p[k]["per_gpu_batch"] for k in ("2", "4", "8")] == [128, 64, 32] and all(p[k]["hip_graph"] for k in ("2", "4", "8"))
    assert all(p[k]["hip_graph_captures_in_timed_region"] == 0 for k in ("2", "4", "8"))
    assert 1.0 < p["2"]["speedup_ceiling"] < p["4"]["speedup_ceiling"] < p["8"]["speedup_ceiling"] < 8.0
    for name in ("attn_fwd", "attn_bwd"):
        assert 0 < r[name]["hbm_frac"] < 1 and 0 < r[name]["mfma_frac"] < 1


def test_small_batch_line_replays_a_graph_and_says_so():
    d = _run("--batch", "32", "--steps", "6", "--warmup", "4", "--no-cpu-baseline")
    assert d["config"]["hip_graph"] is True and d["scaling"] == "weak" and d["config"]["per_gpu_batch"] == 32
    assert "eager one-stream steps" in d["roofline"]["timing_source"] and "replays a hipGraph" in d["roofline"]["timing_source"]
    assert "cpu_baseline" not in d
    # varying lengths / token counts: captures happen in the untimed first-touch steps, few graphs serve all eight batches
    assert d["config"]["hip_graph_captures_in_timed_region"] == 0 and 1 <= d["config"]["hip_graphs_live"] <= 3


def test_two_rank_launch_line_over_gloo_on_one_gpu():
    """The driver's N > 1 command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)
    with two ranks sharing cuda:0 over gloo (DL_DIST_BACKEND=gloo: a one-GPU box cannot host two RCCL ranks): replica
    broadcast, agreed gradient set, all-reduce after the replayed step, barrier + max-over-ranks timing, the weak-scaling
    leg and the untimed busy tail must all run, and rank 0 prints ONE line with strong scaling of the global batch."""
    import socket
    env = dict(os.environ, DL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = None
    for attempt in range(2):
        # (the run takes ~10-60 s; a rendezvous that never completes — the freshly released port taken by someone else, a
        #  c10d store that cannot resolve the container's hostname — once cost a whole 900 s timeout: retry on another port)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        import signal
        p = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                              "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6",
                              "--warmup", "3", "--min-busy-seconds", "0.5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             env=env, cwd=ROOT, start_new_session=True)        # own process group: a timeout takes the ranks down too
        try:
            out, err = p.communicate(timeout=300)
            r = subprocess.CompletedProcess(p.args, p.returncode, out, err)
            break
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)                                   # exactly the group started above
            p.communicate()
            r = None
    assert r is not None, "two attempts of the two-rank launch timed out (300 s each)"
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 256 and d["config"]["per_gpu_batch"] == 128
    assert d["config"]["hip_graph"] is True and d["value"] > 0 and "weak" in d and d["weak"]["per_gpu_batch"] == 256
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
