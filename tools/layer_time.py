#!/usr/bin/env python3
"""Drop-in decode latency: one LLaMA-7B-shaped decoder layer at batch 1 (attention itself replaced by adds -- soft_max / rope are
outside this path), host tensors in a registered context pool, through ggml_graph_compute of the host mirror.  Prints the wall
time per graph compute and per node, beside the sum of the kernels' own periods (tools/gemv_time.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device, ggml as G  # noqa: E402

D, F, N = 4096, 11008, int(sys.argv[1]) if len(sys.argv) > 1 else 1
device.init(0)
rng = np.random.default_rng(1)
ctx = G.ggml_init(700 * 1024 * 1024)


def qweight(K, M):
    t = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
    b = G.tensor_bytes(t).reshape(M * (K // 32), 20)
    b[:, 4:] = rng.integers(0, 256, (M * (K // 32), 16), dtype=np.uint8)
    b[:, :4] = (rng.random(M * (K // 32), dtype=np.float32) * 0.02 + 0.001).view(np.uint8).reshape(-1, 4)
    return t


def f32(K, n, scale=1.0):
    t = G.ggml_new_tensor_2d(ctx, G.F32, K, n)
    G.tensor_f32(t)[:] = (rng.standard_normal((n, K)).astype(np.float32) * scale).reshape(1, 1, n, K)
    return t


x, g1, g2 = f32(D, N), f32(D, N), f32(D, N)
wq, wk, wv, wo = (qweight(D, D) for _ in range(4))
w1, w3, w2 = qweight(D, F), qweight(D, F), qweight(F, D)
cur = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, x), g1)
q, k, v = G.ggml_mul_mat(ctx, wq, cur), G.ggml_mul_mat(ctx, wk, cur), G.ggml_mul_mat(ctx, wv, cur)
a = G.ggml_add(ctx, G.ggml_add(ctx, q, k), v)
h = G.ggml_add(ctx, G.ggml_mul_mat(ctx, wo, a), x)
cur2 = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, h), g2)
u, gt = G.ggml_mul_mat(ctx, w1, cur2), G.ggml_mul_mat(ctx, w3, cur2)
s = G.ggml_mul(ctx, G.ggml_silu(ctx, u), gt)
out = G.ggml_add(ctx, G.ggml_mul_mat(ctx, w2, s), h)
gf = G.ggml_build_forward(out)
for _ in range(5):
    G.ggml_graph_compute(ctx, gf)
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    G.ggml_graph_compute(ctx, gf)
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.sort(ts)
print(f"N={N}: {gf.n_nodes} nodes, graph compute median {np.median(ts):.1f} us (p10 {ts[5]:.1f}, p90 {ts[45]:.1f}) = {np.median(ts) / gf.n_nodes:.1f} us per node; "
      f"result checksum {float(np.abs(G.tensor_f32(out)).sum()):.6g}")
import ctypes as C  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402
v = [C.c_uint64() for _ in range(4)]
lib().ggml_hip_debug_scope_counters(*[C.byref(x) for x in v])
print("named scopes: observed %d, captured %d, replayed %d, refused %d" % tuple(x.value for x in v))
G.ggml_free(ctx)
