#!/usr/bin/env python3
"""Developer stress test for K3p's LDS-DMA scale tables (gemm_qmp.hip load_scale_table): a DMA's LDS write is complete for a reader only
after vmcnt(0) AND lgkmcnt(0) (+ the barrier); with less, a table row now and then still holds what the previous kernel left in LDS.
Every K3p form, thousands of launches on alternating inputs, other LDS-heavy kernels in between: the same input must give the same bits."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device  # noqa: E402
device.init(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
TYPES = {2: "q4_0", 3: "q4_1", 4: "q4_2", 6: "q5_0", 7: "q5_1", 8: "q8_0"}
g = torch.Generator(device="cuda"); g.manual_seed(5)
# a different kernel that fills LDS with other contents between the launches under test (staged MX form, 64 KB of stages per workgroup)
wo = device.Weight.from_device(2, device.quantize_rows(2, torch.randn((2048, 1024), generator=g, device="cuda")), 1024)
xo = torch.randn((1024, 1024), generator=g, device="cuda") * 37.0
nbad = 0
for (M, K, N) in ((4096, 4096, 512), (4096, 11008, 512), (700, 2336, 257), (9000, 2048, 300), (4096, 4096, 1024), (9000, 4096, 2048), (11008, 4096, 129), (5000, 2048, 200),
                  (1000, 28672, 300), (700, 20512, 257)):    # (r4: the sliced forms -- the tables are refilled by DMA inside the K loop)
    for t in TYPES:
        w = torch.randn((M, K), generator=g, device="cuda")
        W = device.Weight.from_device(t, device.quantize_rows(t, w), K)
        xs = [torch.randn((N, K), generator=g, device="cuda") * (1 + 3 * i) for i in range(2)]
        work = device.alloc_work(t, K, N)
        first = [None, None]
        out = torch.empty((N, M), device="cuda")
        fails = 0
        for it in range(iters):
            i = it & 1
            if it % 3 == 0:
                device.mul_mat(wo, xo)
            device.mul_mat(W, xs[i], out=out, work=work)
            if first[i] is None:
                first[i] = out.clone()
            elif it % 8 < 2 or it > iters - 16:          # (comparing every launch would serialise the stream: most launches run back to back)
                if not torch.equal(out, first[i]):
                    fails += 1
                    if fails <= 2:
                        d = out != first[i]
                        print(f"DIFF {TYPES[t]} M{M} K{K} N{N} launch {it}: {int(d.sum())} elements, src1 rows {d.any(dim=1).nonzero().flatten()[:6].tolist()}", flush=True)
        nbad += fails
        print(f"{TYPES[t]} M{M} K{K} N{N}: {iters} launches, {fails} differing", flush=True)
        W.free()
print(f"K3p stress: failures {nbad}")
sys.exit(1 if nbad else 0)
