#!/usr/bin/env python3
"""Developer sweep of the fused mat-vec forms (N <= 8): random ragged shapes x six types -- the single call against fp64 and
against the two-step form (same bits), the multi-matrix call (with and without the rms_norm -> mul prologue) against the single
calls (same bits)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib, check  # noqa: E402

device.init(0)
L = lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(iters):
    t = int(rng.choice([2, 3, 4, 6, 7, 8]))
    N = int(rng.choice([1, 1, 2, 3, 4, 5, 8]))
    K = 32 * int(rng.choice([1, 2, 4, 8, 16, 64, 128, 129, 136, 256, 344]))
    nw = int(rng.choice([1, 2, 3, 4]))
    Ms = [int(rng.choice([1, 15, 16, 17, 100, 255, 1000, 4096, 4097, 8200, 20000])) for _ in range(nw)]
    g = torch.Generator(device="cuda"); g.manual_seed(it)
    Ws, wds = [], []
    for M in Ms:
        rows = device.quantize_rows(t, torch.randn((M, K), generator=g, device="cuda"))
        Ws.append(device.Weight.from_device(t, rows, K))
        wds.append(device.dequantize_rows(t, rows, K).double())
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
    work = device.alloc_work(t, K, N)
    singles = []
    for W, wd, M in zip(Ws, wds, Ms):
        got = device.mul_mat(W, x, work=work).clone()
        ref = xq @ wd.T
        err = (got.double() - ref).abs()
        rms = ref.pow(2).mean().sqrt()
        nb = int((err > 1e-3 * ref.abs() + 1e-5 * rms).sum().item())
        device.mul_mat_init(W, x, work)
        two = torch.empty_like(got)
        device.mul_mat_compute(W, N, two, work)
        if nb or not torch.equal(two, got):
            bad += 1
            print(f"BAD single t{t} M{M} K{K} N{N}: {nb} outside tolerance, two-step equal {torch.equal(two, got)}", flush=True)
        singles.append(got)
    if nw >= 2 and N <= 4:
        hw = (C.c_void_p * nw)(*[w.handle for w in Ws])
        assert L.ggml_hip_mul_mat_multi_fused(hw, nw, N) == 1
        outs = [torch.full((N, M + 1), 7.0, device="cuda") for M in Ms]
        dp = (C.c_void_p * nw)(*[o.data_ptr() for o in outs])
        ld = (C.c_int64 * nw)(*[M + 1 for M in Ms])
        check(L.ggml_hip_mul_mat_multi_dev(hw, nw, C.c_void_p(x.data_ptr()), K, N, dp, ld, None, 0, None, None, st), "multi")
        for o, s, M in zip(outs, singles, Ms):
            if not (torch.equal(o[:, :M], s) and bool(torch.all(o[:, M:] == 7.0))):
                bad += 1
                print(f"BAD multi t{t} Ms{Ms} K{K} N{N}", flush=True)
        gvec = torch.randn((N, K), generator=g, device="cuda")
        n_ref, y_ref = torch.empty((N, K), device="cuda"), torch.empty((N, K), device="cuda")
        check(L.ggml_hip_rms_norm_mul_rows_dev(C.c_void_p(x.data_ptr()), C.c_void_p(gvec.data_ptr()), C.c_void_p(n_ref.data_ptr()), C.c_void_p(y_ref.data_ptr()), N, K, st), "pair")
        s2 = [device.mul_mat(W, y_ref, work=work).clone() for W in Ws]
        nrm, y = torch.empty((N, K), device="cuda"), torch.empty((N, K), device="cuda")
        for o in outs:
            o.fill_(7.0)
        check(L.ggml_hip_mul_mat_multi_dev(hw, nw, C.c_void_p(x.data_ptr()), K, N, dp, ld, C.c_void_p(gvec.data_ptr()), K, C.c_void_p(nrm.data_ptr()), C.c_void_p(y.data_ptr()), st), "multi+pro")
        okp = torch.equal(nrm, n_ref) and torch.equal(y, y_ref)
        for o, s, M in zip(outs, s2, Ms):
            okp = okp and torch.equal(o[:, :M], s) and bool(torch.all(o[:, M:] == 7.0))
        if not okp:
            bad += 1
            print(f"BAD multi+prologue t{t} Ms{Ms} K{K} N{N}", flush=True)
    for W in Ws:
        W.free()
    if it % 50 == 49:
        print(f"{it + 1} cases, bad {bad}", flush=True)
print("bad:", bad)
sys.exit(1 if bad else 0)
