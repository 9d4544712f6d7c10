import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
M, K, N = 4096, 256, 4096
g = torch.Generator(device="cuda"); g.manual_seed(1)
w = torch.arange(M * K, device="cuda").reshape(M, K).remainder(251).half()   # w[m][k] distinct-ish per k
W = device.Weight.from_device(1, w.contiguous().view(torch.uint8), K)
x = torch.zeros((N, K), device="cuda")
idx = torch.arange(N, device="cuda") % K
x[torch.arange(N, device="cuda"), idx] = 1.0            # row n selects k = n % 256
out = device.mul_mat(W, x)                               # expect out[n][m] = w[m][n % 256]
wt = w.float().T.contiguous()                            # [K][M]
ok_rows = (out == wt[idx]).all(dim=1)
print("rows correct:", ok_rows.sum().item(), "of", N)
# which k does each output row actually show (compare column m=1 .. use full match)
found = []
for n in list(range(0, 72)) + [128, 129, 160, 161, 255, 256, 257, 4095]:
    match = (wt == out[n].unsqueeze(0)).all(dim=1).nonzero().flatten().tolist()
    found.append((n, match[:3]))
print(found)
