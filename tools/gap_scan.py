#!/usr/bin/env python3
"""COMPUTE-only loop period against the kernel's event duration over problem sizes: does the per-launch boundary cost scale
with the kernel's duration?  Developer tool."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
for (M, K, N) in ((4096, 4096, 512), (4096, 4096, 1024), (4096, 4096, 2048), (4096, 4096, 4096), (8192, 4096, 4096), (4096, 4096, 8192), (4096, 8192, 4096), (4096, 1024, 4096)):
    W = device.Weight.from_device(2, device.quantize_rows(2, torch.randn((M, K), generator=g, device="cuda")), K)
    x = torch.randn((N, K), generator=g, device="cuda")
    out = torch.empty((N, M), device="cuda"); work = device.alloc_work(2, K, N)
    device.mul_mat_init(W, x, work)
    f = lambda: device.mul_mat_compute(W, N, out, work)
    for _ in range(5): f()
    it = 30
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); b.synchronize()
    period = a.elapsed_time(b) / it * 1e3
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for e0, e1 in ev:
        e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    dur = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[it // 2] * 1e3
    print(f"{M}x{K}x{N}: loop period {period:7.1f} us, event duration {dur:7.1f} us, difference {period - dur:5.1f} us ({(period - dur) / dur * 100:4.1f} %)", flush=True)
    W.free()
