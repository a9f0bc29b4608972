"""Input stage at the benchmark batch: batch assembly from the device-resident embedding store (dl_gather_pad) and the
LLM-feature ingest (dl_fill_pool).  Prints time and HBM GB/s (algorithmic bytes: rows read + batch written)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
from druglamp_amd.embedding_store import EmbeddingStore
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
g = torch.Generator().manual_seed(0)
B = 256
ps, ds = EmbeddingStore(640), EmbeddingStore(384)
plens = torch.randint(100, 1023, (2000,), generator=g).tolist()
dlens = torch.randint(10, 120, (4000,), generator=g).tolist()
for i, n in enumerate(plens): ps.add(i, torch.randn(n, 640, generator=g))
for i, n in enumerate(dlens): ds.add(i, torch.randn(n, 384, generator=g))
ps.finalize(); ds.finalize()
pk = torch.randint(0, 2000, (B,), generator=g).tolist(); dk = torch.randint(0, 4000, (B,), generator=g).tolist()
print("store: %.2f GB protein (%d entities), %.2f GB drug (%d entities)" % (ps.nbytes / 1e9, len(plens), ds.nbytes / 1e9, len(dlens)))
xp = ps.batch(pk, 2304, True); xd = ds.batch(dk, 512, False)
tp = timeit(lambda: ps.batch(pk, 2304, True)); td = timeit(lambda: ds.batch(dk, 512, False))
bp = xp.numel() * 2 + sum(plens[k] for k in pk) * 640 * 2
bd = xd.numel() * 2 + sum(min(dlens[k], 512) for k in dk) * 384 * 2
print("gather_pad protein (256, 2304, 640): %.1f us  %.0f GB/s     drug (256, 512, 384): %.1f us  %.0f GB/s" % (tp * 1e6, bp / tp / 1e9, td * 1e6, bd / td / 1e9))
po = torch.tensor([ps._index[k][0] for k in pk], dtype=torch.int64, device="cuda"); pl = torch.tensor([ps._index[k][1] for k in pk], dtype=torch.int32, device="cuda")
do_ = torch.tensor([ds._index[k][0] for k in dk], dtype=torch.int64, device="cuda"); dl_ = torch.tensor([ds._index[k][1] for k in dk], dtype=torch.int32, device="cuda")
tp = timeit(lambda: ops.gather_pad(ps._store, po, pl, 2304, True)); td = timeit(lambda: ops.gather_pad(ds._store, do_, dl_, 512, False))
print("kernel only (index tensors already on the device): protein %.1f us  %.0f GB/s     drug %.1f us  %.0f GB/s" % (tp * 1e6, bp / tp / 1e9, td * 1e6, bd / td / 1e9))
tf = timeit(lambda: ops.fill_pool(xp, 9, torch.bfloat16)); tfd = timeit(lambda: ops.fill_pool(xd, 1, torch.bfloat16))
print("fill_pool protein: %.1f us  %.0f GB/s (reads %.0f MB)     drug: %.1f us" % (tf * 1e6, xp.numel() * 2 / tf / 1e9, xp.numel() * 2 / 1e6, tfd * 1e6))
