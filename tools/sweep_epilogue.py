#!/usr/bin/env python3
"""Randomised sweep of the store-phase epilogues (ggml_hip_mul_mat_epilogue_dev: add / scale) against product-then-node, bit for bit,
on every quantized type and kernel family the automatic choice reaches, padded dst rows included.  Developer tool for the GPU box.
usage: sweep_epilogue.py [seed] [shapes]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib, check  # noqa: E402

device.init(0)
L = lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
nbad = ntot = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    t = int(rng.choice(list(TYPES)))
    M = int(rng.choice([1, 33, 64, 100, 130, 256, 300, 515, 1000, 2049, 4096, 9000]))
    N = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 16, 32, 40, 64, 65, 128, 129, 256, 257, 300, 512, 513, 700, 1100, 2048]))
    K = 32 * int(rng.choice([1, 2, 8, 16, 33, 64, 65, 73, 128, 129]))
    g = torch.Generator(device="cuda")
    g.manual_seed(it)
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    r = torch.randn((N, M), generator=g, device="cuda")
    W = device.Weight.from_device(t, device.quantize_rows(t, w), K)
    prod = device.mul_mat(W, x)
    work = device.alloc_work(t, K, N)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p1, p2 = int(rng.integers(0, 9)), int(rng.integers(0, 9))
    d1 = torch.full((N, M + p1), -3.0, device="cuda")
    d2 = torch.full((N, M + p2), -4.0, device="cuda")
    check(L.ggml_hip_mul_mat_epilogue_dev(W.handle, C.c_void_p(x.data_ptr()), N, K, C.c_void_p(d1.data_ptr()), M + p1, C.c_void_p(work.data_ptr()),
                                          work.numel(), 1, C.c_void_p(r.data_ptr()), M, C.c_void_p(d2.data_ptr()), M + p2, 1.0, st), "epilogue add")
    ok = torch.equal(d1[:, :M], prod) and bool(torch.all(d1[:, M:] == -3.0)) and torch.equal(d2[:, :M], prod + r) and bool(torch.all(d2[:, M:] == -4.0))
    d3 = torch.empty((N, M), device="cuda")
    sc = float(np.float32(rng.uniform(0.1, 2.0)))
    check(L.ggml_hip_mul_mat_epilogue_dev(W.handle, C.c_void_p(x.data_ptr()), N, K, C.c_void_p(d3.data_ptr()), M, C.c_void_p(work.data_ptr()),
                                          work.numel(), 2, None, 0, None, 0, sc, st), "epilogue scale")
    ok = ok and torch.equal(d3, prod * np.float32(sc))
    ntot += 1
    if not ok:
        nbad += 1
        print(f"BAD {TYPES[t]} M{M} K{K} N{N} (fused form {L.ggml_hip_mul_mat_epilogue_fused(W.handle, N)})", flush=True)
    W.free()
print(f"epilogue sweep: {ntot} shapes, {nbad} bad")
sys.exit(1 if nbad else 0)
