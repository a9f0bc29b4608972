"""Where the trickle kernel's time goes (study library): each shape / epilogue with and without its stores (DL_GEMM_DBG=1),
trickle vs the 256x256 tile.  DL_USE_STUDY_LIB=1 python tools/trickle_parts.py"""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(65536, 2048, 512), (65536, 1024, 256)]
CASES = ["plain", "bias", "gelu+pre", "gelu+pre+drop", "dgelu", "res"]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from druglamp_amd import ops
    dev = torch.device("cuda:0"); dt = torch.bfloat16
    rows = {}
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    for (M, N, K) in SHAPES:
        x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt)
        b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(dt); pre_in = torch.randn(M, N, device=dev).to(dt)
        pre = torch.empty(M, N, device=dev, dtype=dt); out = torch.empty(M, N, device=dev, dtype=dt)
        kws = {"plain": dict(), "bias": dict(bias=b), "gelu+pre": dict(bias=b, act=1, pre_out=pre),
               "gelu+pre+drop": dict(bias=b, act=1, pre_out=pre, dropout_p=0.1, seed=5),
               "dgelu": dict(dact_pre=pre_in), "res": dict(bias=b, residual=res)}
        for name in CASES:
            rows["%dx%dx%d %s" % (M, N, K, name)] = round(timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **kws[name])), 1)
    print("RESULT " + json.dumps(rows), flush=True); sys.exit(0)
res = {}
for trk in ("0", "1"):
    for dbg in ("0", "1", "2", "3", "8", "9"):
        env = dict(os.environ, DL_GEMM_TRICKLE=trk, DL_GEMM_DBG=dbg, DL_USE_STUDY_LIB="1", DL_GEMM_TRICKLE_MAXK="2048")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("failed", trk, dbg, r.stderr[-2000:]); sys.exit(1)
        res[(trk, dbg)] = json.loads(line[0][7:])
print("us per launch; dbg: 0 = complete, 1 = no stores, 2 = no operand feed after the prologue, 3 = neither, 8 = no epilogue arithmetic (trickle only), 9 = 8 + 1")
print("%-32s | %8s %8s %8s %8s | %8s %8s %8s %8s %8s %8s" % ("shape / epilogue", "256² d0", "d1", "d2", "d3", "trk d0", "d1", "d2", "d3", "d8", "d9"))
for k in res[("0", "0")]:
    print("%-32s | %8.1f %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f %8.1f %8.1f %8.1f" % ((k,) + tuple(res[("0", d)][k] for d in "0123") + tuple(res[("1", d)][k] for d in ("0", "1", "2", "3", "8", "9"))))
