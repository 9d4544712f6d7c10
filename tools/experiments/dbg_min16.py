#!/usr/bin/env python3
"""r5 debugging aid: Q5_1 on 16-row batched-decode tiles against the 32-row form -- where and by how much do they differ?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device
device.init(0)
t, K, N = 7, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = 16384
g = torch.Generator(device="cuda"); g.manual_seed(1)
w = torch.randn((M, K), generator=g, device="cuda"); x = torch.randn((N, K), generator=g, device="cuda") * 2
rows = device.quantize_rows(t, w)
W = device.Weight.from_device(t, rows, K)
full = device.mul_mat(W, x)
Ws = device.Weight.from_device(t, rows, K, row_begin=0, row_end=4096)
part = device.mul_mat(Ws, x)
ref = full[:, :4096]
d = (part - ref)
print("max |diff|", d.abs().max().item(), "rms ref", ref.pow(2).mean().sqrt().item(), "differing", int((d != 0).sum()), "of", d.numel())
nz = (d != 0).nonzero()
print("first differing (n, m):", nz[:12].tolist())
print("differing per column n:", (d != 0).sum(dim=1).tolist())
print("differing per row m % 32 histogram:", torch.bincount(nz[:, 1] % 32, minlength=32).tolist())
print("sample part/ref:", part[0, :4].tolist(), ref[0, :4].tolist())
