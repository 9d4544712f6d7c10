#!/usr/bin/env python3
"""Developer stress test: every mul_mat kernel form, launched repeatedly on alternating inputs with unrelated kernels in
between, must return bit-identical results for the same input (the kernels have fixed summation trees), and the first
result of each input is checked against an fp64 evaluation.  Catches races that only show in some launches."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402
device.init(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
SHAPES = [(5000, 2048, 2000), (4096, 4096, 4096), (4096, 1024, 512), (777, 352, 130), (4096, 4096, 1), (32000, 1024, 5), (300, 11008, 513), (4096, 4096, 16), (1000, 2080, 29), (4096, 4096, 64), (777, 11008, 128), (16384, 1024, 100),
          # the batched-decode forms (K3s / K3s-i8): one, two and four tiles per workgroup, one and two column slices, K in slots and in rounds
          (4096, 4096, 6), (4096, 4096, 32), (8492, 4160, 50), (20000, 2048, 20), (300, 11008, 64), (4096, 2112, 9),
          # K3p (gemm_qmp.hip): one round, several rounds of persistent workgroups, ragged M / N, K ranges with a short last one
          (4096, 4096, 512), (4096, 11008, 300), (33000, 2048, 512), (700, 2336, 257), (11008, 4096, 400)]
nbad = 0
for (M, K, N) in SHAPES:
    for t in TYPES:
        for kernel in ((0,) if N <= 8 or t == 4 else (0, 1, 2, 3)):
            lib().ggml_hip_debug_force_gemm(kernel)
            g = torch.Generator(device="cuda"); g.manual_seed(M + 3 * N + t)
            w = torch.randn((M, K), generator=g, device="cuda")
            rows = device.quantize_rows(t, w)
            W = device.Weight.from_device(t, rows, K)
            wd = device.dequantize_rows(t, rows, K).double()
            xs = [torch.randn((N, K), generator=g, device="cuda") * (1 + i) for i in range(2)]
            first = [None, None]
            fails = 0
            for it in range(iters):
                i = it & 1
                torch.randn((1 + it % 5) * 150000, device="cuda").sum()
                got = device.mul_mat(W, xs[i])
                if first[i] is None:
                    first[i] = got.clone()
                    xq = device.dequantize_rows(8, device.quantize_rows(8, xs[i]), K).double()
                    ref = xq @ wd.T
                    err = (got.double() - ref).abs(); rms = ref.pow(2).mean().sqrt()
                    if int((err > 1e-3 * ref.abs() + 1e-5 * rms).sum().item()) or not torch.isfinite(got).all():
                        fails += 1
                        print(f"WRONG {TYPES[t]} M{M} K{K} N{N} kernel {kernel} input {i}", flush=True)
                elif not torch.equal(got, first[i]):
                    fails += 1
                    if fails <= 2:
                        d = (got != first[i])
                        print(f"DIFF {TYPES[t]} M{M} K{K} N{N} kernel {kernel} launch {it}: {int(d.sum())} elements, rows {d.any(dim=1).nonzero().flatten()[:6].tolist()}", flush=True)
            nbad += fails
            W.free()
            del wd, xs, first
lib().ggml_hip_debug_force_gemm(0)
# dense weights: F16 (dense16.hip on full grids, dense.hip otherwise) and F32 (dense.hip)
for (M, K, N) in ((4096, 4096, 4096), (2048, 1000, 300), (512, 4096, 7), (4096, 4096, 64), (16384, 2048, 100), (4096, 2048, 300), (3000, 4096, 3),
                  (1000, 4104, 700), (9000, 1032, 1537), (300, 11008, 513)):      # the 16 x 16 x 32 forms (N > 512): 128 x 128 and 256 x 128 tiles, ragged edges
    for t, dt in ((1, torch.float16), (0, torch.float32)):
        g = torch.Generator(device="cuda"); g.manual_seed(M + N + t)
        w = torch.randn((M, K), generator=g, device="cuda").to(dt)
        W = device.Weight.from_device(t, w.view(torch.uint8).view(M, -1), K)
        xs = [torch.randn((N, K), generator=g, device="cuda") * (1 + i) for i in range(2)]
        first = [None, None]
        fails = 0
        for it in range(iters):
            i = it & 1
            torch.randn((1 + it % 5) * 150000, device="cuda").sum()
            got = device.mul_mat(W, xs[i])
            if first[i] is None:
                first[i] = got.clone()
                xr = xs[i].to(dt).double() if t == 1 else xs[i].double()     # INIT rounds src1 to half for F16 weights
                ref = xr @ w.double().T
                err = (got.double() - ref).abs(); rms = ref.pow(2).mean().sqrt()
                if int((err > 1e-3 * ref.abs() + 2e-5 * rms).sum().item()):
                    fails += 1
                    print(f"WRONG dense type {t} M{M} K{K} N{N} input {i}: max err/rms {(err.max() / rms).item():.2e}", flush=True)
            elif not torch.equal(got, first[i]):
                fails += 1
                print(f"DIFF dense type {t} M{M} K{K} N{N} launch {it}", flush=True)
        nbad += fails
        W.free()
print("determinism stress: failures", nbad)
sys.exit(1 if nbad else 0)
