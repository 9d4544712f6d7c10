#!/usr/bin/env python3
"""Randomised shape sweep of the matrix-core kernels against an fp64 evaluation of the same block arithmetic
(operands from the bit-exact device quantize / dequantize kernels).  Developer tool for the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402

device.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
edges = [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 640, 2049, 5000, 12000, 26000, 33000]
nbad = ntot = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    t = int(rng.choice(list(TYPES)))
    M = int(rng.choice(edges))
    N = int(rng.choice([1, 2, 5, 8, 9, 12, 16, 17, 24, 31, 32, 33, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300, 384, 448, 512, 513, 600, 768, 1024, 1500, 2048]))
    K = 32 * int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 31, 32, 33, 40, 47, 48, 56, 63, 64, 65, 73, 100, 129]))      # (32 k-blocks and up: the batched-decode forms K3s -- r5, it was 64; 64 and up: K3p too)
    kernel = int(rng.choice([0, 1, 2, 3]))
    lib().ggml_hip_debug_force_gemm(kernel)
    g = torch.Generator(device="cuda")
    g.manual_seed(it)
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    rows = device.quantize_rows(t, w)
    W = device.Weight.from_device(t, rows, K)
    got = device.mul_mat(W, x)
    wd = device.dequantize_rows(t, rows, K).double()
    xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
    ref = xq @ wd.T
    err = (got.double() - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    bad = int((err > 1e-3 * ref.abs() + 1e-5 * rms).sum().item())
    ntot += 1
    # a row shard must be the bitwise slice of the unsplit product (the kernel form is a function of type, N and K alone)
    shard_bad = 0
    if M >= 33:
        r0 = int(rng.integers(0, M - 16))
        r1 = int(rng.integers(r0 + 1, M + 1))
        Ws = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
        shard_bad = 0 if torch.equal(device.mul_mat(Ws, x), got[:, r0:r1]) else 1
        Ws.free()
    if bad or shard_bad or not torch.isfinite(got).all():
        nbad += 1
        print(f"BAD {TYPES[t]} M{M} K{K} N{N} kernel {kernel}: {bad} elements, max err/rms {(err.max() / rms).item():.2e}, shard mismatch {shard_bad}", flush=True)
    W.free()
lib().ggml_hip_debug_force_gemm(0)
print(f"sweep: {ntot} shapes, {nbad} bad")
sys.exit(1 if nbad else 0)
