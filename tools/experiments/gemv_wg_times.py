#!/usr/bin/env python3
"""Developer experiment (needs a variant build of gemv.hip that stamps wall_clock64 per workgroup into g_gv_dbg):
when do the workgroups of one mat-vec launch start, finish staging and end?"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402
from ggmlsharp_amd._lib import lib  # noqa: E402

device.init(0)
L = C.CDLL(os.environ["GGML_HIP_LIB"])
for M in (4096, 32000, 65536):
    K, N, t = 4096, 1, 2
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    copies = max(2, min(32, int(400e6 // (M * K // 32 * 20))))
    ws = [device.Weight.from_device(t, device.quantize_rows(t, torch.randn((M, K), generator=g, device="cuda")), K) for _ in range(copies)]
    x = torch.randn((N, K), generator=g, device="cuda"); out = torch.empty((N, M), device="cuda"); work = device.alloc_work(t, K, N)
    for w in ws: device.mul_mat(w, x, out=out, work=work)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(32): device.mul_mat(ws[i % copies], x, out=out, work=work)
    for _ in range(3): graph.replay()
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 4096)()
    L.ggml_hip_debug_gemv_times(buf, 4096)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    nwg = min(int(os.environ.get("GV_WGS", "512")), (M + 15) // 16)
    a = a[:nwg]
    t0 = a[:, 0].min()
    st, stg, en = (a[:, 0] - t0) / 100.0, (a[:, 2] - t0) / 100.0, (a[:, 1] - t0) / 100.0
    q = lambda v: " ".join(f"{np.percentile(v, p):6.2f}" for p in (0, 10, 50, 90, 100))
    print(f"M={M}: {nwg} workgroups; us after the first start, percentiles 0/10/50/90/100\n  start  {q(st)}\n  staged {q(stg)}\n  end    {q(en)}\n  life   {q(en - st)}", flush=True)
    life = en - st
    print("  mean life by XCD (blockIdx % 8):", " ".join(f"{life[i::8].mean():6.2f}" for i in range(8)))
    print("  mean life by blockIdx octile   :", " ".join(f"{life[i * nwg // 8:(i + 1) * nwg // 8].mean():6.2f}" for i in range(8)))
    ntl = (M + 15) // 16
    cnt = np.array([(ntl - 1 - b) // nwg + 1 for b in range(nwg)])
    print("  tiles per workgroup:", np.unique(cnt, return_counts=True), " mean life per tile count:", [round(float(life[cnt == c].mean()), 2) for c in np.unique(cnt)])
    print("  corr(start, life) = %.2f" % np.corrcoef(st, life)[0, 1])
    for w in ws: w.free()
