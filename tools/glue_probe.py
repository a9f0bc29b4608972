import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
B = 256
with torch.autocast("cuda", dtype=torch.bfloat16):
    h = torch.randn(B, 512, 75, device=dev); lin = torch.nn.Linear(75, 128, bias=False).to(dev)
    print("init_transform fwd", timeit(lambda: lin(h)))
    x = torch.randn(B, 512, 128, device=dev, dtype=torch.bfloat16); w = torch.randn(128, 128, device=dev)
    print("matmul 128x128", timeit(lambda: torch.matmul(x, w.to(x.dtype))))
    adj = torch.randn(B, 128, 128, device=dev)
    print("bmm compact", timeit(lambda: torch.bmm(adj.transpose(1, 2).to(x.dtype), x[:, :128])))
    res = torch.nn.Linear(128, 128).to(dev)
    print("res linear", timeit(lambda: res(x)))
    f = torch.randn(B, 512, device=dev); fc1 = torch.nn.Linear(512, 1024).to(dev)
    print("cls fc1", timeit(lambda: fc1(f)))
    hh = h.clone().requires_grad_(False)
    def fb():
        y = lin(hh); y.sum().backward()
    print("init_transform fwd+bwd", timeit(fb))
    xx = x.clone().requires_grad_(True)
    def fb2():
        y = res(xx); y.float().sum().backward()
    print("res fwd+bwd", timeit(fb2))
    xp = torch.randn(B, 2304, 641, device=dev, dtype=torch.bfloat16)
    print("site pool", timeit(lambda: xp.view(-1, 9, 256, 641).mean(dim=1)))
    print("fill bit", timeit(lambda: (xp[..., :640].sum(dim=-1) == 0)))
    x0 = torch.randn(B, 2304, 640, device=dev, dtype=torch.bfloat16); fb_ = torch.zeros(B, 2304, device=dev, dtype=torch.bfloat16)
    print("cat fill", timeit(lambda: torch.cat((x0, fb_.unsqueeze(-1)), dim=-1)))
