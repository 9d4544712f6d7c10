#!/usr/bin/env python3
"""Mat-vec (N <= 8, one fused launch) period per call: a hipGraph of one call per rotating weight copy, median over replays.
usage: [GGML_HIP_LIB=variant.so] python tools/gemv_time.py [type:M:K:N ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402

TYPES = {"q4_0": 2, "q4_1": 3, "q4_2": 4, "q5_0": 6, "q5_1": 7, "q8_0": 8}
BLK = {2: 20, 3: 24, 4: 20, 6: 22, 7: 24, 8: 36}


def run(tname, M, K, N):
    t = TYPES[tname]
    copies = max(2, min(32, int(400e6 // (M * K // 32 * BLK[t]))))      # > 256 MB of distinct weights: no Infinity Cache hits
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    ws = []
    for c in range(copies):
        w = torch.randn((M, K), generator=g, device="cuda")
        ws.append(device.Weight.from_device(t, device.quantize_rows(t, w), K))
        del w
    x = torch.randn((N, K), generator=g, device="cuda")
    out = torch.empty((N, M), device="cuda")
    work = device.alloc_work(t, K, N)
    for w in ws:
        device.mul_mat(w, x, out=out, work=work)
    torch.cuda.synchronize()
    nodes = copies * max(1, -(-32 // copies))          # >= 32 calls per replay: its own fixed cost is amortised
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(nodes):
            device.mul_mat(ws[i % copies], x, out=out, work=work)
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / nodes * 1e3)
    us = float(np.median(ts))
    ab = M * (K // 32) * BLK[t] + 4 * K * N + 4 * M * N
    print(f"{tname} M{M} K{K} N{N}: {us:7.2f} us/call  {ab / us / 1e6:6.2f} TB/s algorithmic  ({copies} copies)", flush=True)
    for w in ws:
        w.free()


if __name__ == "__main__":
    device.init(0)
    cfgs = sys.argv[1:] or ["q4_0:4096:4096:1", "q4_0:4096:4096:2", "q4_0:4096:4096:4", "q4_0:4096:4096:8", "q4_0:4096:11008:1",
                            "q4_0:11008:4096:1", "q4_0:32000:4096:1", "q4_0:65536:4096:1", "q8_0:4096:4096:1", "q5_0:4096:4096:1",
                            "q8_0:32000:4096:1", "q4_0:32000:4096:8"]
    print("lib:", os.environ.get("GGML_HIP_LIB", "product"))
    for c in cfgs:
        tn, M, K, N = c.split(":")
        run(tn, int(M), int(K), int(N))
