#!/usr/bin/env python3
"""r5: the k-quant extras (Q5_K, Q4_K, Q6_K) over the range where the plan picks K3s-int8 or K3p-int8 by the matrix's height (33..512 src1 rows) and
around it: a random row shard must be the bitwise slice of the unsplit product whatever family either runs, and a sample of the product must
meet the numpy restatement of the published format (tests/np_kquants.py).  Developer tool, GPU box.   usage: sweep_kq_shards.py [seed] [shapes]"""
import os
import sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "support"))
import np_kquants as KQ  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402
device.init(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
SB = {113: 176, 112: 144, 114: 210}
MM = {113: KQ.mul_mat_q5_K, 112: KQ.mul_mat_q4_K, 114: KQ.mul_mat_q6_K}


def blocks(nb, t):
    b = rng.integers(0, 256, size=(nb, SB[t]), dtype=np.uint8)
    if t == 114:
        b[:, 208:210] = (rng.random(nb).astype(np.float32) * 0.002 + 0.0001).astype(np.float16).reshape(-1, 1).view(np.uint8)
        return b
    b[:, 0:2] = (rng.random(nb).astype(np.float32) * 0.02 + 0.001).astype(np.float16).reshape(-1, 1).view(np.uint8)
    b[:, 2:4] = (rng.random(nb).astype(np.float32) * 0.05).astype(np.float16).reshape(-1, 1).view(np.uint8)
    return b


nbad = ntot = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    t = int(rng.choice([113, 112, 114]))
    K = 256 * int(rng.choice([8, 9, 16, 17, 43, 64]))
    N = int(rng.choice([9, 17, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256, 300, 512, 513, 700]))
    M = int(rng.choice([100, 700, 1024, 3000, 4096, 9000, 12288, 16384]))
    if M * K > 120e6:
        M = 4096
    rows = blocks(M * K // 256, t).reshape(M, -1)
    x = rng.standard_normal((N, K)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    W = device.Weight.from_host(t, rows, K)
    full = device.mul_mat(W, xd)
    r0 = int(rng.integers(0, max(1, M - 16))); r1 = int(rng.integers(r0 + 1, M + 1))
    Ws = device.Weight.from_host(t, rows, K, row_begin=r0, row_end=r1)
    part = device.mul_mat(Ws, xd)
    shard_bad = 0 if torch.equal(part, full[:, r0:r1]) else 1
    ms = np.sort(rng.choice(M, size=min(24, M), replace=False)); ns = np.sort(rng.choice(N, size=min(24, N), replace=False))
    ref = MM[t](rows[ms], x[ns])
    got = full.cpu().numpy()[np.ix_(ns, ms)]
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))) + 1e-30
    floor = max(1e-6, 8 * 2.0 ** -24 * (K / 32) ** 0.5) * rms
    bad = int(np.sum(np.abs(got.astype(np.float64) - ref) > np.maximum(1e-3 * np.abs(ref), floor)))
    ntot += 1
    if bad or shard_bad or not torch.isfinite(full).all():
        nbad += 1
        print(f"BAD type {t} M{M} K{K} N{N}: {bad} of the sample off, shard [{r0}, {r1}) mismatch {shard_bad}", flush=True)
    Ws.free(); W.free()
print(f"k-quant shard sweep: {ntot} shapes, {nbad} bad")
