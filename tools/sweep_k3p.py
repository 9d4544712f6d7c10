#!/usr/bin/env python3
"""Targeted sweep of K3p (gemm_qmp.hip; 257..512 src1 rows, the int8 types from 129 and up to 3072 / without bound, K / 32 >= 64): every type it serves, K from the smallest the form takes to the
largest its LDS tables allow (chunks of the min-term product: none for some waves, ragged last chunk, five per wave), ragged M and N,
persistent grids; fp64 evaluation of the same block arithmetic + a random row shard must be the bitwise slice.  Developer tool, GPU box."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
device.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}   # (r5: Q4_2 -- the two-scale form of K3p)
KB = [64, 65, 66, 71, 79, 80, 81, 95, 96, 97, 127, 128, 129, 143, 144, 200, 257, 344, 400, 512, 513, 617, 623, 624, 625, 639, 640, 641, 688, 896, 1000, 1024, 1249, 1873, 2000]   # (from 625: the scale tables in two to four slices)
nbad = ntot = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    t = int(rng.choice(list(TYPES)))
    kb = int(rng.choice(KB)); K = 32 * kb
    N = int(rng.choice([129, 130, 160, 200, 255, 256, 257, 258, 288, 300, 320, 321, 383, 384, 449, 480, 511, 512, 513, 600, 768, 1000, 1024, 1025, 1500, 2047, 2048, 2049, 3000, 4096]))   # (r4: the int8 types up to 2048 rows, Q5_1 without bound)
    M = int(rng.choice([1, 31, 100, 128, 129, 700, 1000, 4096, 5000, 9000] if kb <= 200 else [100, 700, 1000, 3000]))
    g = torch.Generator(device="cuda"); g.manual_seed(1000 + it)
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
    floor = max(1e-6, 8 * 2.0 ** -24 * (K / 32) ** 0.5) * rms          # THE tolerance (tests/oracle_lib.py)
    bad = int((err > torch.maximum(1e-3 * ref.abs(), floor)).sum().item())
    ntot += 1
    shard_bad = 0
    if M >= 33:
        r0 = int(rng.integers(0, M - 16)); r1 = int(rng.integers(r0 + 1, M + 1))
        Ws = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
        shard_bad = 0 if torch.equal(device.mul_mat(Ws, x), got[:, r0:r1]) else 1
        Ws.free()
    if bad or shard_bad or not torch.isfinite(got).all():
        nbad += 1
        print(f"BAD {TYPES[t]} M{M} K{K} N{N}: {bad} elements, max err/rms {(err.max() / rms).item():.2e}, shard mismatch {shard_bad}", flush=True)
    W.free()
# r4: the batched-decode forms (K3s, 5..64 rows) over long K -- one and two rounds of table pieces (K <= 16384 | <= 32768), beyond: the staged forms
for it in range(80):
    t = int(rng.choice(list(TYPES) + [4]))
    K = 32 * int(rng.choice([344, 345, 432, 512, 513, 640, 896, 1000, 1024, 1025]))
    N = int(rng.choice([5, 9, 17, 31, 32, 33, 64, 65, 96, 100, 128, 129, 200, 256]))   # (r4: behind K >= 11008 the one-scale types stay on the form up to 128 rows, the two-scale ones up to 256 whatever K)
    M = int(rng.choice([100, 700, 3000]))
    g = torch.Generator(device="cuda"); g.manual_seed(5000 + it)
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
    floor = max(1e-6, 8 * 2.0 ** -24 * (K / 32) ** 0.5) * rms
    bad = int((err > torch.maximum(1e-3 * ref.abs(), floor)).sum().item())
    ntot += 1
    r0 = int(rng.integers(0, M - 16)); r1 = int(rng.integers(r0 + 1, M + 1))
    Ws = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
    shard_bad = 0 if torch.equal(device.mul_mat(Ws, x), got[:, r0:r1]) else 1
    Ws.free()
    if bad or shard_bad or not torch.isfinite(got).all():
        nbad += 1
        print(f"BAD (decode form) type {t} M{M} K{K} N{N}: {bad} elements, max err/rms {(err.max() / rms).item():.2e}, shard mismatch {shard_bad}", flush=True)
    W.free()
print(f"K3p sweep: {ntot} shapes, {nbad} bad")
sys.exit(1 if nbad else 0)
