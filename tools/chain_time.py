#!/usr/bin/env python3
"""Decode-sized pre-projection chain rms_norm -> mul -> mul_mat -> add, device-resident, per chain:
  (a) ONE launch: the fused mat-vec with its rms_norm*g prologue and add epilogue (ggml_hip_norm_mul_mat_dev, N <= 4);
  (b) two launches: the pair kernel (fused.hip) + the mat-vec with the add epilogue;
  (c) three launches: pair kernel + plain mat-vec + torch's add (a stand-in for the separate add kernel);
  (d) the plain mat-vec alone.
A hipGraph of one chain per rotating weight copy, median over replays."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import check, lib  # noqa: E402


def replay_us(fn, copies):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) / copies * 1e3)
    return float(np.median(ts))


if __name__ == "__main__":
    device.init(0)
    L = lib()
    M = K = 4096
    copies = 32
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2)
    ws = [device.Weight.from_device(2, device.quantize_rows(2, torch.randn((M, K), generator=gen, device="cuda")), K) for _ in range(copies)]
    print("| N | one launch (us) | pair + mat-vec with add epilogue (us) | pair + mat-vec + add (us) | mat-vec alone (us) |")
    print("|---|---|---|---|---|")
    for N in (1, 2, 4):
        x = torch.randn((N, K), generator=gen, device="cuda")
        g = torch.randn((N, K), generator=gen, device="cuda")
        r = torch.randn((N, M), generator=gen, device="cuda")
        nrm, y = torch.empty((N, K), device="cuda"), torch.empty((N, K), device="cuda")
        d1, d2 = torch.empty((N, M), device="cuda"), torch.empty((N, M), device="cuda")
        work = device.alloc_work(2, K, N)
        P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

        def st():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def one():
            for w in ws:
                check(L.ggml_hip_norm_mul_mat_dev(w.handle, P(x), K, P(g), K, N, P(nrm), P(y), P(d1), M, P(work), work.numel(), 1, P(r), M, P(d2), M, 1.0, st()), "a")

        def two():
            for w in ws:
                check(L.ggml_hip_rms_norm_mul_rows_dev(P(x), P(g), P(nrm), P(y), N, K, st()), "b1")
                check(L.ggml_hip_mul_mat_epilogue_dev(w.handle, P(y), N, K, P(d1), M, P(work), work.numel(), 1, P(r), M, P(d2), M, 1.0, st()), "b2")

        def three():
            for w in ws:
                check(L.ggml_hip_rms_norm_mul_rows_dev(P(x), P(g), P(nrm), P(y), N, K, st()), "c1")
                device.mul_mat(w, y, out=d1, work=work)
                torch.add(d1, r, out=d2)

        def alone():
            for w in ws:
                device.mul_mat(w, y, out=d1, work=work)
        print(f"| {N} | {replay_us(one, copies):.2f} | {replay_us(two, copies):.2f} | {replay_us(three, copies):.2f} | {replay_us(alone, copies):.2f} |", flush=True)
