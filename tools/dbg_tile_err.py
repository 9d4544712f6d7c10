#!/usr/bin/env python3
"""Where is a mat-mat result wrong?  Error map per 32x32 tile and per row-in-tile (developer tool).  usage: type M K N [force]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
from ggmlsharp_amd._lib import lib
T = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}
t, M, K, N = T[sys.argv[1]], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
device.init(0)
lib().ggml_hip_debug_force_gemm(int(sys.argv[5]) if len(sys.argv) > 5 else 0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
w = torch.randn((M, K), generator=g, device="cuda"); x = torch.randn((N, K), generator=g, device="cuda")
rows = device.quantize_rows(t, w); W = device.Weight.from_device(t, rows, K)
got = device.mul_mat(W, x)
wd = device.dequantize_rows(t, rows, K).double()
xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
ref = xq @ wd.T
err = (got.double() - ref).abs() / ref.pow(2).mean().sqrt()
print("max err/rms", err.max().item(), "bad frac", (err > 1e-4).double().mean().item())
for n0 in range(0, N, 32):
    print(" ".join(f"{err[n0:n0+32, m0:m0+32].max().item():8.1e}" for m0 in range(0, M, 32)))
e = err[:32, :32]
print("rows of tile (0,0) max:", [f"{v:.0e}" for v in e.max(dim=1).values.tolist()])
print("cols of tile (0,0) max:", [f"{v:.0e}" for v in e.max(dim=0).values.tolist()])
# per-block contribution test: is the result the sum over a subset of k-blocks?
nb = K // 32
parts = torch.stack([xq[:, b*32:(b+1)*32] @ wd[:, b*32:(b+1)*32].T for b in range(nb)])   # [nb][N][M]
sol = torch.linalg.lstsq(parts[:, :32, :32].reshape(nb, -1).T, got[:32, :32].double().reshape(-1, 1)).solution.flatten()
print("least-squares weights of the k-blocks in tile (0,0):", [f"{v:.2f}" for v in sol.tolist()])
