#!/usr/bin/env python3
"""Experiment (GPU box): why does the headline COMPUTE kernel take 160-165 us inside the INIT, COMPUTE, INIT, ... sequence of a step and
144-146 us launched back to back?  Arms (Q4_0 4096^3, events around COMPUTE only, 2 s of preheat each):
  cc        COMPUTE back to back, one dst
  cc_4dst   COMPUTE back to back, four dst buffers in turn (every launch writes 64 MB that is not the 64 MB just written)
  ic        INIT, COMPUTE alternating (the step)
  xc        a 64-MB device-to-device copy, COMPUTE alternating (memory traffic of INIT's size, nothing else of INIT)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
M = K = N = 4096
g = torch.Generator(device="cuda")
g.manual_seed(7)
w = torch.randn((M, K), generator=g, device="cuda")
W = device.Weight.from_device(2, device.quantize_rows(2, w), K)
x = torch.randn((N, K), generator=g, device="cuda") * 2
outs = [torch.empty((N, M), device="cuda") for _ in range(4)]
work = device.alloc_work(2, K, N)
junk_a = torch.empty((16 * 1024 * 1024,), device="cuda")
junk_b = torch.empty_like(junk_a)
device.mul_mat(W, x, out=outs[0], work=work)
st = {"i": 0}


def arm(name, pre, ndst):
    def one(ev=None):
        pre()
        o = outs[st["i"] % ndst]
        st["i"] += 1
        if ev:
            ev[0].record()
        device.mul_mat_compute(W, N, o, work)
        if ev:
            ev[1].record()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(16):
            one()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(300)]
    for e in evs:
        one(e)
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
    print(f"{name:8s} COMPUTE median {np.median(ts):7.1f} us  p10 {np.percentile(ts, 10):7.1f}  p90 {np.percentile(ts, 90):7.1f}", flush=True)


for rep in range(2):
    arm("cc", lambda: None, 1)
    arm("cc_4dst", lambda: None, 4)
    arm("ic", lambda: device.mul_mat_init(W, x, work), 1)
    arm("xc", lambda: junk_b.copy_(junk_a), 1)
