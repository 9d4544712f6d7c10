import sys, torch
sys.path.insert(0, '/root/repo')
from ggmlsharp_amd import device as dev
dev.init(0)
M, K, N = 32000, 1024, 512
g = torch.Generator(device="cuda"); g.manual_seed(5)
w = torch.randn((M, K), generator=g, device="cuda").half()
x = torch.randn((N, K), generator=g, device="cuda")
rows = w.view(torch.uint8).view(M, -1)
W = dev.Weight.from_device(1, rows, K)
full = dev.mul_mat(W, x)
Ws = dev.Weight.from_device(1, rows, K, row_begin=4000, row_end=8000)
print("shard equal:", torch.equal(dev.mul_mat(Ws, x), full[:, 4000:8000]))
