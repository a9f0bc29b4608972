import torch
dev = torch.device("cuda", 0)
y = torch.zeros(4, device=dev)
torch.rand(4, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y.copy_(torch.rand(4, device=dev))
for _ in range(3):
    g.replay(); torch.cuda.synchronize(); print(y.tolist())
g2 = torch.cuda.CUDAGraph()
z = torch.zeros(4, device=dev)
with torch.cuda.graph(g2):
    z.copy_(torch.zeros(4, device=dev).uniform_(0, 1))
for _ in range(2):
    g2.replay(); torch.cuda.synchronize(); print("g2", z.tolist())
