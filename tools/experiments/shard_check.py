#!/usr/bin/env python3
"""Experiment (GPU box): is a row shard the bitwise slice of the unsplit product?  usage: shard_check.py type M K N kernel [r0 r1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402

TYPES = {"q4_0": 2, "q4_1": 3, "q4_2": 4, "q5_0": 6, "q5_1": 7, "q8_0": 8}
device.init(0)
a = sys.argv[1:]
t, M, K, N, kernel = TYPES[a[0]], int(a[1]), int(a[2]), int(a[3]), int(a[4])
r0, r1 = (int(a[5]), int(a[6])) if len(a) > 6 else (0, min(M, 4000))
lib().ggml_hip_debug_force_gemm(kernel)
g = torch.Generator(device="cuda")
g.manual_seed(1)
w = torch.randn((M, K), generator=g, device="cuda")
x = torch.randn((N, K), generator=g, device="cuda") * 2
rows = device.quantize_rows(t, w)
W = device.Weight.from_device(t, rows, K)
full = device.mul_mat(W, x)
Ws = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
part = device.mul_mat(Ws, x)
d = part != full[:, r0:r1]
print(f"{a[0]} M{M} K{K} N{N} kernel {kernel} rows [{r0}, {r1}): {'bitwise equal' if not d.any() else f'{int(d.sum())} of {d.numel()} elements differ'}")
