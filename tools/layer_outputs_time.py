#!/usr/bin/env python3
"""The 7B-shaped decoder layer of bench.py's dropin_decode_layer driven node by node through the seams inside one graph scope,
with every node's data copied home (the reference's contract) against ggml_hip_graph_outputs(last node only) -- the opt-in that
leaves the other results on the device.  usage: python tools/layer_outputs_time.py [N ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "support"))
import ggml_mirror as G  # noqa: E402  (test support: the host mirror)
from ggmlsharp_amd import device, _lib  # noqa: E402

device.init(0)
L = _lib.lib()
D, F = 4096, 11008
rng = np.random.default_rng(1)


def run(N):
    ctx = G.ggml_init(1200 * 1024 * 1024)
    try:
        def qweight(K, M):
            t = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
            b = G.tensor_bytes(t).reshape(M * (K // 32), 20)
            b[:, 4:] = rng.integers(0, 256, (M * (K // 32), 16), dtype=np.uint8)
            b[:, :4] = (rng.random(M * (K // 32), dtype=np.float32) * 0.02 + 0.001).view(np.uint8).reshape(-1, 4)
            return t

        def f32(K, n):
            t = G.ggml_new_tensor_2d(ctx, G.F32, K, n)
            G.tensor_f32(t)[:] = rng.standard_normal((n, K)).astype(np.float32).reshape(1, 1, n, K)
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
        p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
        fs = {_lib.GGML_OP_MUL_MAT: L.ggml_hip_compute_forward_mul_mat, _lib.GGML_OP_ADD: L.ggml_hip_compute_forward_add,
              _lib.GGML_OP_MUL: L.ggml_hip_compute_forward_mul}
        want = (C.c_void_p * 1)(out.contents.data)

        def compute(outputs_only):
            _lib.check(L.ggml_hip_graph_begin(), "begin")
            if outputs_only:
                _lib.check(L.ggml_hip_graph_outputs(want, 1), "outputs")
            for i in range(gf.n_nodes):
                n = gf.nodes[i].contents
                if n.op in fs:
                    rc = fs[n.op](C.byref(p), n.src0, n.src1, gf.nodes[i])
                elif n.op == _lib.GGML_OP_RMS_NORM:
                    rc = L.ggml_hip_compute_forward_rms_norm(C.byref(p), n.src0, gf.nodes[i])
                else:
                    rc = L.ggml_hip_compute_forward_silu(C.byref(p), n.src0, gf.nodes[i])
                _lib.check(rc, f"node {i}")
            _lib.check(L.ggml_hip_graph_end(), "end")

        res = {}
        for mode in (False, True):
            for _ in range(3):
                compute(mode)
            ts = []
            for _ in range(12):
                t0 = time.perf_counter()
                compute(mode)
                ts.append((time.perf_counter() - t0) * 1e6)
            res[mode] = float(np.median(ts))
        print(f"batch {N}: every node's data home {res[False]:9.1f} us | the last node's only {res[True]:9.1f} us  (node by node through the seams, no fused pairs)", flush=True)
    finally:
        G.ggml_free(ctx)


for n in [int(a) for a in sys.argv[1:]] or [32, 128, 512]:
    run(n)
