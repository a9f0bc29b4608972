"""gemm_trickle_kernel (round 6) against gemm_big_kernel (the 256x256 tile of rounds 1-5) and the 128-tile kernel on the path's
forward / data-gradient shapes: bitwise agreement of all three, and kernel times.  Needs the study library for the A/B switch:

    python -m druglamp_amd.build --study
    DL_USE_STUDY_LIB=1 python tools/trickle_bench.py            # per shape and epilogue: us with DL_GEMM_TRICKLE=0 / 1

(the product library has no switch: it always takes the trickle form where it is eligible)."""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(65536, 2048, 512), (65536, 1536, 512), (65536, 512, 512), (65536, 1024, 256), (65536, 768, 256), (65536, 256, 256),
          (65536, 512, 2048), (65536, 512, 1536), (65536, 256, 1024)]
CASES = ["bias", "gelu+pre+drop", "dgelu+drop", "res+drop", "plain"]


def run_child(mode):
    import torch
    from druglamp_amd import ops
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    out_rows = {}

    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    for (M, N, K) in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(dev); w = (torch.randn(N, K, generator=g) * 0.1).to(dt).to(dev)
        b = torch.randn(N, generator=g).to(dev); res = torch.randn(M, N, generator=g).to(dt).to(dev)
        pre_in = torch.randn(M, N, generator=g).to(dt).to(dev)
        pre = torch.empty(M, N, device=dev, dtype=dt); out = torch.empty(M, N, device=dev, dtype=dt)
        kws = {"plain": dict(), "bias": dict(bias=b), "gelu+pre+drop": dict(bias=b, act=1, pre_out=pre, dropout_p=0.1, seed=5),
               "dgelu+drop": dict(dact_pre=pre_in, dropout_p=0.1, seed=5), "res+drop": dict(bias=b, residual=res, dropout_p=0.1, seed=7)}
        for name in CASES:
            kw = kws[name]
            t = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **kw))
            out.zero_(); pre.zero_()
            ops.gemm(x, w, M=M, N=N, K=K, out=out, **kw)
            torch.cuda.synchronize()
            # checksum of the result bits (and of the pre-activation copy) — compared across modes by the parent
            chk = int(out.view(torch.int16).to(torch.int64).sum()) ^ (int(pre.view(torch.int16).to(torch.int64).sum()) << 1 if "pre_out" in kw else 0)
            ref = None
            if mode == "1":     # the same call on the 128-tile kernel: bitwise
                o2 = torch.empty_like(out); p2 = torch.empty_like(pre)
                kw2 = dict(kw)
                if "pre_out" in kw2: kw2["pre_out"] = p2
                ops.gemm(x, w, M=M, N=N, K=K, out=o2, algo=1, **kw2)
                torch.cuda.synchronize()
                ref = bool(torch.equal(o2, out) and ("pre_out" not in kw or torch.equal(p2, pre)))
            out_rows["%dx%dx%d %s" % (M, N, K, name)] = (round(t, 1), chk, ref)
    print("RESULT " + json.dumps(out_rows), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run_child(sys.argv[2]); sys.exit(0)
    res = {}
    for mode in ("0", "1"):
        env = dict(os.environ, DL_GEMM_TRICKLE=mode, DL_USE_STUDY_LIB="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", mode], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("mode", mode, "failed:", r.stdout[-1500:], r.stderr[-3000:]); sys.exit(1)
        res[mode] = json.loads(line[0][7:])
    print("%-34s %10s %10s %7s %9s %9s" % ("shape / epilogue", "256x256 us", "trickle us", "ratio", "bits==256", "bits==128"))
    for k in res["0"]:
        a, b = res["0"][k], res["1"][k]
        print("%-34s %10.1f %10.1f %7.2f %9s %9s" % (k, a[0], b[0], b[0] / a[0], a[1] == b[1], b[2]))
