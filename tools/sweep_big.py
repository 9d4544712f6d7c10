#!/usr/bin/env python3
"""A few large, ragged shapes for every matrix-core kernel (the big tile configurations), fp64 block-arithmetic check."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402
device.init(0)
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
nbad = 0
for (M, K, N) in ((4096, 288, 1537), (5000, 4096, 2000), (2048, 1056, 3073), (8192, 512, 4096), (4097, 32, 1600)):
    for t in TYPES:
        for kernel in (0, 1, 2, 3):
            lib().ggml_hip_debug_force_gemm(kernel)
            g = torch.Generator(device="cuda"); g.manual_seed(M + N + t)
            w = torch.randn((M, K), generator=g, device="cuda")
            # src1 as a strided view (row stride K + 64 elements): the INIT kernels take ld1 != K
            xbig = torch.randn((N, K + 64), generator=g, device="cuda") * 2
            x = xbig[:, :K]
            rows = device.quantize_rows(t, w)
            W = device.Weight.from_device(t, rows, K)
            got = device.mul_mat(W, x)
            wd = device.dequantize_rows(t, rows, K).double()
            xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
            ref = xq @ wd.T
            err = (got.double() - ref).abs(); rms = ref.pow(2).mean().sqrt()
            bad = int((err > 1e-3 * ref.abs() + 1e-5 * rms).sum().item())
            if bad or not torch.isfinite(got).all():
                nbad += 1
                print(f"BAD {TYPES[t]} M{M} K{K} N{N} kernel {kernel}: {bad}", flush=True)
            W.free()
lib().ggml_hip_debug_force_gemm(0)
print("big sweep bad:", nbad)
sys.exit(1 if nbad else 0)
