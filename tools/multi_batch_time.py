#!/usr/bin/env python3
"""Projection groups of a batched decoder's step: ggml_hip_mul_mat_multi_work_dev (one INIT, one launch where gemm_qmx.hip has the
form) against one ggml_hip_mul_mat_dev per matrix, weights cold (a ring of weight sets larger than the caches), inside replayed
hipGraphs.  usage: python tools/multi_batch_time.py [N]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import check, lib  # noqa: E402

device.init(0)
L = lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(os.environ.get("MULTI_TYPE", "2"))      # 2 = Q4_0, 8 = Q8_0


def run(Ms, K, sets=12, reps=20):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    rows = [device.quantize_rows(T, torch.randn((M, K), generator=g, device="cuda")) for M in Ms]
    W = [[device.Weight.from_device(T, r, K) for r in rows] for _ in range(sets)]
    x = torch.randn((N, K), generator=g, device="cuda")
    outs = [torch.empty((N, M), device="cuda") for M in Ms]
    work = device.alloc_work(T, K, N)
    dp = (C.c_void_p * len(Ms))(*[o.data_ptr() for o in outs])
    ld = (C.c_int64 * len(Ms))(*Ms)
    s = torch.cuda.Stream()
    res = {}
    for mode in ("single calls", "one call"):
        with torch.cuda.stream(s):
            st = C.c_void_p(s.cuda_stream)

            def body():
                for ws in W:
                    if mode == "one call":
                        hw = (C.c_void_p * len(Ms))(*[w.handle for w in ws])
                        check(L.ggml_hip_mul_mat_multi_work_dev(hw, len(Ms), C.c_void_p(x.data_ptr()), K, N, dp, ld, C.c_void_p(work.data_ptr()), work.numel(), st), "multi")
                    else:
                        for w, o, M in zip(ws, outs, Ms):
                            check(L.ggml_hip_mul_mat_dev(w.handle, C.c_void_p(x.data_ptr()), N, K, C.c_void_p(o.data_ptr()), M, C.c_void_p(work.data_ptr()), work.numel(), st), "single")
            body()
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                body()
            gr.replay(); s.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s); gr.replay(); e1.record(s); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / sets)
        res[mode] = float(np.median(ts))
    print(f"{len(Ms)} x ({Ms[0]} x {K}) x {N}: {res['single calls']:7.2f} us as single calls | {res['one call']:7.2f} us as one call", flush=True)
    for ws in W:
        for w in ws:
            w.free()


run((4096, 4096, 4096), 4096)
run((11008, 11008), 4096, sets=8)
run((4096, 4096), 11008, sets=8)
