#!/usr/bin/env python3
"""north_star wording ("wavefront-level dequant+dot reductions" for the Q path, "MFMA only for the dense case") against
measurement: the VALU sdot4 mat-vec kernel serving N src1 rows in passes of 8 columns (weights re-streamed per pass, the
only VALU form there is) next to the matrix-core kernels the library selects, Q4_0 4096 x 4096, N = 8 .. 512."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402


def med_us(fn, iters=30):
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))


if __name__ == "__main__":
    device.init(0)
    M = K = 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    W = device.Weight.from_device(2, device.quantize_rows(2, torch.randn((M, K), generator=g, device="cuda")), K)
    print("| N | VALU sdot4 kernel, passes of 8 columns (us) | matrix-core kernel the library selects (us) | ratio |")
    print("|---|---|---|---|")
    for N in (8, 16, 64, 512):
        x = torch.randn((N, K), generator=g, device="cuda")
        out = torch.empty((N, M), device="cuda")
        work = device.alloc_work(2, K, N)

        def valu():
            for a in range(0, N, 8):
                device.mul_mat(W, x[a:a + 8], out=out[a:a + 8], work=work)

        def lib():
            device.mul_mat(W, x, out=out, work=work)
        for f in (valu, lib):
            f()
        torch.cuda.synchronize()
        gv, gl = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gv):
            valu()
        with torch.cuda.graph(gl):
            lib()
        tv, tl = med_us(gv.replay), med_us(gl.replay)
        print(f"| {N} | {tv:.1f} | {tl:.1f} | {tv / tl:.1f} |", flush=True)
