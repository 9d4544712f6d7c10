#!/usr/bin/env python3
"""Experiment (GPU box): what does a dependent launch boundary cost a batch-1 mat-vec, and how much of it disappears when the
next mat-vec is ALREADY running?  Chains of `n` Q4_0 4096 x 4096 mat-vecs (distinct weights, 10.5 MB each), captured in a hipGraph:
  serial   -- one stream, every launch behind the previous one (what a decoder's dependency chain does today)
  two-way  -- alternate launches on two streams (independent pairs may overlap: the second kernel's ramp hides behind the first)
  four-way -- four streams
The per-call time of the k-way forms is what a hand-off between co-resident kernels (flag instead of a launch boundary) could reach."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
M = K = 4096
shapes = [(4096, 4096), (11008, 4096), (4096, 11008)]
for (M, K) in shapes:
    n = 32
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    ws = []
    for i in range(n):
        w = torch.randn((M, K), generator=g, device="cuda")
        ws.append(device.Weight.from_device(2, device.quantize_rows(2, w), K))
        del w
    x = torch.randn((1, K), generator=g, device="cuda")
    outs = [torch.empty((1, M), device="cuda") for _ in range(n)]
    works = [device.alloc_work(2, K, 1) for _ in range(n)]
    for i in range(n):
        device.mul_mat(ws[i], x, out=outs[i], work=works[i])
    torch.cuda.synchronize()
    res = {}
    for ways in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(ways)]
        cap = torch.cuda.Stream()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(gr, stream=cap):
                ev0 = torch.cuda.Event()
                ev0.record(cap)
                for s in streams:
                    s.wait_event(ev0)
                for i in range(n):
                    with torch.cuda.stream(streams[i % ways]):
                        device.mul_mat(ws[i], x, out=outs[i], work=works[i])
                for s in streams:
                    e = torch.cuda.Event()
                    e.record(s)
                    cap.wait_event(e)
        for _ in range(20):
            gr.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        a.record()
        for _ in range(reps):
            gr.replay()
        b.record()
        b.synchronize()
        res[ways] = a.elapsed_time(b) / reps / n * 1e3
    mb = M * (K // 32) * 20 / 1e6
    print(f"Q4_0 {M} x {K} batch 1 ({mb:.1f} MB per call): " + "  ".join(f"{w}-way {t:6.2f} us/call = {mb / t / 1e-6 / 1e6 / 1e6:5.2f} TB/s" for w, t in res.items()), flush=True)
    for w_ in ws:
        w_.free()
