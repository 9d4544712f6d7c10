#!/usr/bin/env python3
"""A/B of the matrix-core kernel forms (0 default, 1 int8, 2 f16, 3 MX) per type and shape: compute-kernel time from HIP events.
usage: ab_kernels.py type:M:K:N ..."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402
TYPES = {"q4_0": 2, "q4_1": 3, "q4_2": 4, "q5_0": 6, "q5_1": 7, "q8_0": 8}
device.init(0)
for cfg in sys.argv[1:]:
    tn, M, K, N = cfg.split(":")
    t, M, K, N = TYPES[tn], int(M), int(K), int(N)
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((N, K), generator=g, device="cuda")
    rows = device.quantize_rows(t, w)
    out = torch.empty((N, M), device="cuda")
    res = []
    for kernel in (0, 1, 2, 3, 0):
        lib().ggml_hip_debug_force_gemm(kernel)
        W = device.Weight.from_device(t, rows, K)      # the MX digit planes of Q5_0 / Q8_0 exist only for weights uploaded under force 3
        kind = lib().ggml_hip_act_image_kind(t, K, N)
        work = device.alloc_work(t, K, N)
        try:
            device.mul_mat_init(W, x, work)
            device.mul_mat_compute(W, N, out, work)
        except Exception as e:
            res.append(f"{kernel}:err")
            W.free()
            continue
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        ti = []
        for i in range(40):
            device.mul_mat_init(W, x, work)
            ev[i].record(); device.mul_mat_compute(W, N, out, work); e2 = torch.cuda.Event(enable_timing=True); e2.record(); ti.append((ev[i], e2))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ti)
        res.append(f"force {kernel} (image {kind}): {ts[len(ts)//2]*1e3:7.1f} us")
        W.free()
    lib().ggml_hip_debug_force_gemm(0)
    print(f"{tn} {M}x{K}x{N}: " + " | ".join(res), flush=True)
