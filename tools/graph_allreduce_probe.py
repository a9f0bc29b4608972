"""Can an RCCL all-reduce be captured into a hipGraph and replayed on this stack?  World size 1 (one GPU box): exercises the
capture path of ProcessGroupNCCL / RCCL, not the multi-rank transport."""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(14_000_000, device=dev); y = torch.zeros_like(x)
dist.all_reduce(x); torch.cuda.synchronize()              # communicator warm-up outside capture
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y.copy_(x * 2.0)
    dist.all_reduce(y, op=dist.ReduceOp.SUM)
    y.mul_(0.5)
x.fill_(3.0); g.replay(); torch.cuda.synchronize()
print("replay 1:", float(y[0]), float(y[-1]))
x.fill_(5.0); g.replay(); torch.cuda.synchronize()
print("replay 2:", float(y[0]), float(y[-1]))
t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize(); print("replay %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
dist.destroy_process_group()
print("ok")
