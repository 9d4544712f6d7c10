#!/usr/bin/env python3
"""Kernel micro-bench + full-size parity check for the quantized mul_mat kernels (developer tool, GPU box).

For each (type, M, K, N): checks the HIP result against an fp64 evaluation of the same integer/scale arithmetic
(dequantised weights x dequantised Q8 activations, both produced by the bit-exact device kernels) and times the
INIT (K1) and COMPUTE kernels separately with HIP events."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402

GRAPH = False
RESULT = {}          # the last run()'s timings (tools/grid_quant.py reads graph_compute_us)
TYPES = {"q4_0": 2, "q4_1": 3, "q4_2": 4, "q5_0": 6, "q5_1": 7, "q8_0": 8, "q5_k": 113, "q4_k": 112, "q6_k": 114}   # (the k-quant extras: timing only, --no-check -- their activations follow the Q8_K rule)


def ev_ms(fn, iters):
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def run_f16(M, K, N, iters, f32=False):
    """dense f16 x f32 (ggml_compute_forward_mul_mat_f16_f32): the f16 MFMA kernel incl. its INIT (src1 -> Half); f32: the F32 x F32 product"""
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    w = torch.randn((M, K), generator=g, device="cuda")
    if not f32:
        w = w.half()
    x = torch.randn((N, K), generator=g, device="cuda")
    W = device.Weight.from_device(0 if f32 else 1, w.contiguous().view(torch.uint8), K)
    out = torch.empty((N, M), device="cuda")
    work = device.alloc_work(0 if f32 else 1, K, N)
    device.mul_mat(W, x, out=out, work=work)
    ref = (x.double() if f32 else x.half().double()) @ w.double().T
    err = (out.double() - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    t = ev_ms(lambda: device.mul_mat(W, x, out=out, work=work), iters)
    print(f"{'f32' if f32 else 'f16'} M{M} K{K} N{N}: init+compute {t * 1e3:8.1f} us  {2.0 * M * K * N / t / 1e9:9.1f} TFLOP/s  "
          f"max_err/rms {err.max().item() / rms.item():.2e}", flush=True)
    W.free()


def run(tname, M, K, N, iters, check=True, copies=1):
    if tname in ("f16", "f32"):
        return run_f16(M, K, N, iters, f32=tname == "f32")
    t = TYPES[tname]
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    ws = []
    for c in range(copies):
        w = torch.randn((M, K), generator=g, device="cuda")
        rows = device.quantize_rows(t, w)
        ws.append(device.Weight.from_device(t, rows, K))
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    out = torch.empty((N, M), device="cuda")
    work = device.alloc_work(t, K, N)
    W = ws[0]
    device.mul_mat(W, x, out=out, work=work)
    msg = ""
    if check:
        wd = device.dequantize_rows(t, rows if copies == 1 else device.quantize_rows(t, w), K).double()
        if copies > 1:
            W = ws[-1]
            device.mul_mat(W, x, out=out, work=work)
        xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
        ref = xq @ wd.T
        err = (out.double() - ref).abs()
        rms = ref.pow(2).mean().sqrt()
        bad = (err > 1e-3 * ref.abs() + 1e-5 * rms).sum().item()
        msg = f"max_err/rms {err.max().item() / rms.item():.2e} bad {bad}"
    st = {"i": 0}

    def init():
        device.mul_mat_init(ws[st["i"] % copies], x, work)

    def comp():
        st["i"] += 1
        device.mul_mat_compute(ws[st["i"] % copies], N, out, work)

    t_init = ev_ms(init, iters)
    t_comp = ev_ms(comp, iters)
    RESULT.clear()
    RESULT.update(init_us=t_init * 1e3, compute_us=t_comp * 1e3)
    if N <= 8:   # launch-bound from Python: replay a captured graph of `reps` whole mul_mat calls instead
        reps = max(copies, 32)

        def full():
            st["i"] += 1
            device.mul_mat(ws[st["i"] % copies], x, out=out, work=work)
        full()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                full()
        t_graph = ev_ms(gr.replay, 20) / reps
        ab1 = M * (K // 32) * {"q4_0": 20, "q4_1": 24, "q4_2": 20, "q5_0": 22, "q5_1": 24, "q8_0": 36, "q5_k": 22, "q4_k": 18, "q6_k": 26.25}[tname] + 4 * K * N + 4 * M * N
        print(f"   graph-replayed whole mul_mat (fused, {copies} rotating weight copies): {t_graph * 1e3:7.2f} us/call  "
              f"{ab1 / t_graph / 1e6:8.1f} GB/s algorithmic", flush=True)
    if GRAPH:   # r5: GPU-side times -- the per-call figures above are host-bound below ~8 us per call (ctypes + launch); replay captured graphs of 64 launches
        reps = max(copies, 64)

        def cap(fn):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(reps):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            return min(ev_ms(gr.replay, 10) for _ in range(3)) / reps

        def both():
            init()
            comp()
        g_init, g_comp, g_both = cap(init), cap(comp), cap(both)
        RESULT.update(graph_init_us=g_init * 1e3, graph_compute_us=g_comp * 1e3, graph_both_us=g_both * 1e3)
        print(f"   graph-replayed ({copies} rotating weight copies, {reps} launches per graph): init {g_init * 1e3:6.2f} us  compute {g_comp * 1e3:6.2f} us  "
              f"init + compute {g_both * 1e3:6.2f} us", flush=True)
    flops = 2.0 * M * K * N
    blk = {"q4_0": 20, "q4_1": 24, "q4_2": 20, "q5_0": 22, "q5_1": 24, "q8_0": 36, "q5_k": 22, "q4_k": 18, "q6_k": 26.25}[tname]   # bytes per 32 weights
    ab = M * (K // 32) * blk + 4 * K * N + 4 * M * N
    print(f"{tname} M{M} K{K} N{N}: init {t_init * 1e3:8.1f} us  compute {t_comp * 1e3:8.1f} us  "
          f"{flops / t_comp / 1e9:9.1f} TOP/s  {ab / (t_init + t_comp) / 1e6:8.1f} GB/s  {msg}", flush=True)
    for w_ in ws:
        w_.free()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", nargs="*", default=["q4_0:4096:4096:4096", "q4_0:4096:4096:512", "q4_0:4096:4096:1:32",
                                                 "q8_0:4096:11008:512", "q5_0:4096:11008:512", "q4_0:32000:4096:512"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--graph", action="store_true", help="also time INIT / COMPUTE / both as replayed hipGraphs of 64 launches (GPU-side time: small kernels are host-bound per call)")
    a = ap.parse_args()
    GRAPH = a.graph
    device.init(0)
    for c in a.cfg:
        p = c.split(":")
        run(p[0], int(p[1]), int(p[2]), int(p[3]), a.iters, check=not a.no_check, copies=int(p[4]) if len(p) > 4 else 1)
