#!/usr/bin/env python3
"""Developer stress test for the fused mat-vec (N <= 4): many launches on alternating inputs, other kernels in between, must be
bit-identical per input, equal to the two-step form (INIT planes + mat-vec kernel: same summation tree) and close to fp64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for (M, K) in ((4096, 4096), (32000, 4096), (19999, 4096), (8200, 2048), (4096, 11008), (12345, 8192), (70000, 1024)):
    for t in (2, 3, 4, 6, 7, 8):
        for N in (1, 2, 3, 4):
            g = torch.Generator(device="cuda"); g.manual_seed(M + 7 * N + t)
            rows = device.quantize_rows(t, torch.randn((M, K), generator=g, device="cuda"))
            W = device.Weight.from_device(t, rows, K)
            wd = device.dequantize_rows(t, rows, K).double()
            xs = [torch.randn((N, K), generator=g, device="cuda") * (1 + i) for i in range(2)]
            work = device.alloc_work(t, K, N)
            first = [None, None]
            for it in range(iters):
                i = it & 1
                torch.randn((1 + it % 5) * 150000, device="cuda").sum()
                out = device.mul_mat(W, xs[i], work=work).clone()
                if first[i] is None:
                    first[i] = out
                    ref = xs[i].double() @ wd.T
                    err = float((out.double() - ref).abs().max() / ref.abs().mean())
                    device.mul_mat_init(W, xs[i], work)
                    two = torch.empty_like(out)
                    device.mul_mat_compute(W, N, two, work)
                    same = bool(torch.equal(two, out))
                    if err > 0.05 or not same:
                        bad += 1
                        print(f"BAD M{M} K{K} N{N} t{t}: err {err:.3g} two-step equal {same}", flush=True)
                elif not torch.equal(out, first[i]):
                    bad += 1
                    print(f"NONDETERMINISTIC M{M} K{K} N{N} t{t} launch {it}", flush=True)
                    break
            W.free()
    print(f"M{M} K{K} done", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
