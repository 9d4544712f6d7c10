#!/usr/bin/env python3
"""What does one dependent node of a captured / replayed named scope cost?  A chain of n element-wise adds on a [1, 4096] row
through ggml_graph_compute of the host mirror (host tensors in the registered pool), n = 4, 12, 20, 40: the slope is the cost per node,
the intercept the fixed cost of a graph compute (launch, copies home, synchronise).
usage: python tools/experiments/node_cost.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "support"))
import ggml_mirror as G  # noqa: E402  (test support: the host mirror)
from ggmlsharp_amd import device
device.init(0)
res = []
for n in (4, 12, 20, 40):
    ctx = G.ggml_init(64 * 1024 * 1024)
    x = G.ggml_new_tensor_2d(ctx, G.F32, 4096, 1)
    y = G.ggml_new_tensor_2d(ctx, G.F32, 4096, 1)
    G.tensor_f32(x)[:] = 1.0
    G.tensor_f32(y)[:] = 0.5
    cur = x
    for _ in range(n):
        cur = G.ggml_add(ctx, cur, y)
    gf = G.ggml_build_forward(cur)
    for _ in range(6):
        G.ggml_graph_compute(ctx, gf)
    ts = []
    for i in range(200):
        G.tensor_f32(x)[0, 0, 0, 0] = float(i)
        t0 = time.perf_counter()
        G.ggml_graph_compute(ctx, gf)
        ts.append((time.perf_counter() - t0) * 1e6)
    assert abs(float(G.tensor_f32(cur)[0, 0, 0, 1]) - (1.0 + 0.5 * n)) < 1e-5
    res.append((n, float(np.median(ts))))
    print(f"{n:3d} add nodes: {np.median(ts):7.1f} us per graph compute (p10 {np.percentile(ts, 10):.1f}, p90 {np.percentile(ts, 90):.1f})", flush=True)
    G.ggml_free(ctx)
(n0, t0), (n1, t1) = res[0], res[-1]
print(f"slope {(t1 - t0) / (n1 - n0):.2f} us per node, intercept {t0 - n0 * (t1 - t0) / (n1 - n0):.1f} us")
