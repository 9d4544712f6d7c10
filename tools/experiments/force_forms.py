#!/usr/bin/env python3
"""Experiment (GPU box): COMPUTE time of every mat-mat kernel family (forced) for one type and shape.
usage: force_forms.py type M K N   (kernel 0 = automatic, 1 = int8 MFMA, 2 = f16 MFMA, 3 = MX)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402

TYPES = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}
device.init(0)
args = sys.argv[1:]
while args:
    tname, M, K, N = args[0], int(args[1]), int(args[2]), int(args[3])
    args = args[4:]
    t = TYPES[tname]
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    w = torch.randn((M, K), generator=g, device="cuda")
    rows = device.quantize_rows(t, w)
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    out = torch.empty((N, M), device="cuda")
    for kernel in (0, 1, 2, 3):
        lib().ggml_hip_debug_force_gemm(kernel)
        W = device.Weight.from_device(t, rows, K)
        work = device.alloc_work(t, K, N)
        device.mul_mat(W, x, out=out, work=work)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(200):
            device.mul_mat_compute(W, N, out, work)
        a.record()
        for _ in range(1000):
            device.mul_mat_compute(W, N, out, work)
        b.record()
        b.synchronize()
        print(f"{tname} M{M} K{K} N{N} kernel {kernel}: compute {a.elapsed_time(b):8.1f} us", flush=True)
        W.free()
    lib().ggml_hip_debug_force_gemm(0)
