import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
from ggmlsharp_amd import device
device.init(0)
K, N = 4096, 512
g = torch.Generator(device="cuda"); g.manual_seed(1)
for M in (4096, 8192, 16000, 32000):
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((N, K), generator=g, device="cuda")
    rows = device.quantize_rows(2, w)
    W = device.Weight.from_device(2, rows, K)
    full = device.mul_mat(W, x).clone()
    full2 = device.mul_mat(W, x).clone()
    print(M, "repeat equal", torch.equal(full, full2))
    for (r0, r1) in ((0, 4000), (4000, 8000), (M - 4000, M)):
        if r1 > M: continue
        Ws = device.Weight.from_device(2, rows, K, row_begin=r0, row_end=r1)
        part = device.mul_mat(Ws, x)
        eq = torch.equal(part, full[:, r0:r1])
        d = (part - full[:, r0:r1]).abs()
        nz = (d > 0).nonzero()
        print(M, (r0, r1), "equal", eq, "ndiff", int((d > 0).sum()), "max", float(d.max()), "first", nz[:3].tolist() if len(nz) else None)
        Ws.free()
    W.free()
