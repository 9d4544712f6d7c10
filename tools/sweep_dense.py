#!/usr/bin/env python3
"""Randomised shape sweep of the dense F16 / F32 mat-mat kernels (dense16.hip, dense.hip) against fp64, plus the row-shard
identity (a shard's result is the bitwise slice of the unsplit product).  Developer tool for the GPU box.
usage: sweep_dense.py [seed] [shapes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
nbad = ntot = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    f16 = bool(rng.integers(0, 4))          # mostly F16
    M = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 513, 1000, 2049, 4096, 9000]))
    N = int(rng.choice([1, 4, 5, 8, 9, 31, 33, 64, 100, 128, 129, 300, 512, 513, 514, 600, 640, 641, 1000, 1025, 1537, 2048]))
    K = int(rng.choice([8, 24, 32, 40, 100, 128, 136, 256, 500, 1024, 1032, 2048, 4104]))
    g = torch.Generator(device="cuda")
    g.manual_seed(it)
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    if f16:
        w = torch.randn((M, K), generator=g, device="cuda").half()
        W = device.Weight.from_device(1, w.contiguous().view(torch.uint8).view(M, -1), K)
        ref = x.half().double() @ w.double().T
    else:
        w = torch.randn((M, K), generator=g, device="cuda")
        W = device.Weight.from_device(0, w.contiguous().view(torch.uint8).view(M, -1), K)
        ref = x.double() @ w.double().T
    got = device.mul_mat(W, x)
    err = (got.double() - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    bad = int((err > 1e-3 * ref.abs() + 1e-5 * rms).sum().item())
    shard_bad = 0
    if M >= 64:
        r0 = int(rng.integers(0, M - 32))
        r1 = int(rng.integers(r0 + 1, M + 1))
        Ws = device.Weight.from_device(1 if f16 else 0, w.contiguous().view(torch.uint8).view(M, -1), K, row_begin=r0, row_end=r1)
        shard_bad = 0 if torch.equal(device.mul_mat(Ws, x), got[:, r0:r1]) else 1
        Ws.free()
    ntot += 1
    if bad or shard_bad or not torch.isfinite(got).all():
        nbad += 1
        print(f"BAD {'f16' if f16 else 'f32'} M{M} K{K} N{N}: {bad} elements, max err/rms {(err.max() / rms).item():.2e}, shard mismatch {shard_bad}", flush=True)
    W.free()
print(f"dense sweep: {ntot} shapes, {nbad} bad")
sys.exit(1 if nbad else 0)
