#!/usr/bin/env python3
"""Randomised sweep of the store-phase exchange (ggml_hip_mul_mat_push_dev): G ranks' shards of random (type, M, K, N), one after the
other on the one GPU, must fill G destination buffers with the unsplit product bit for bit -- fused store phase where the plan has it,
product + column-push kernel elsewhere.  Developer tool for the GPU box.  usage: sweep_push.py [seed] [shapes]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device, dist as gdist  # noqa: E402
from ggmlsharp_amd._lib import lib, check  # noqa: E402

device.init(0)
L = lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
nbad = ntot = nfused = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    t = int(rng.choice(list(TYPES)))
    G = int(rng.choice([2, 3, 4, 7, 8]))
    M = int(rng.choice([7, 64, 100, 257, 512, 1000, 2049, 4096, 5000, 12000]))
    N = int(rng.choice([1, 3, 8, 9, 33, 64, 100, 129, 256, 257, 300, 512, 513, 700, 1024, 1500]))
    K = 32 * int(rng.choice([1, 4, 8, 16, 17, 33, 64, 65, 73, 100, 129]))
    g = torch.Generator(device="cuda")
    g.manual_seed(it)
    rows = device.quantize_rows(t, torch.randn((M, K), generator=g, device="cuda"))
    x = torch.randn((N, K), generator=g, device="cuda")
    W = device.Weight.from_device(t, rows, K)
    full = device.mul_mat(W, x).clone()
    W.free()
    peers = [torch.full((N, M), -3.0, device="cuda") for _ in range(G)]
    pp = (C.c_void_p * G)(*[p.data_ptr() for p in peers])
    work = device.alloc_work(t, K, N)
    fused_any = False
    for r in range(G):
        r0, r1 = gdist.shard_rows(M, G, r)
        if r1 <= r0:
            continue
        Wr = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
        fused_any |= bool(L.ggml_hip_mul_mat_push_fused(Wr.handle, N, G))
        check(L.ggml_hip_mul_mat_push_dev(Wr.handle, C.c_void_p(x.data_ptr()), N, K, pp, G, r, M, r0, C.c_void_p(work.data_ptr()), work.numel(), None), "push")
        torch.cuda.synchronize()
        Wr.free()
    ntot += 1
    nfused += fused_any
    bad = [r for r, p in enumerate(peers) if not torch.equal(p, full)]
    if bad:
        nbad += 1
        print(f"BAD {TYPES[t]} M{M} K{K} N{N} G{G} fused {fused_any}: buffers {bad} differ", flush=True)
print(f"push sweep: {ntot} shapes ({nfused} through the fused store phase), {nbad} bad")
sys.exit(1 if nbad else 0)
