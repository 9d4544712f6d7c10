#!/usr/bin/env python3
"""sha1 of a product's result bytes + COMPUTE time, for A/B of variant builds that must keep the bits (GGML_HIP_LIB=...).
usage: out_hash.py type:M:K:N ..."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402
TYPES = {"q4_0": 2, "q4_1": 3, "q4_2": 4, "q5_0": 6, "q5_1": 7, "q8_0": 8}
device.init(0)
for cfg in sys.argv[1:]:
    tn, M, K, N = cfg.split(":")
    t, M, K, N = TYPES[tn], int(M), int(K), int(N)
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((N, K), generator=g, device="cuda") * 2
    W = device.Weight.from_device(t, device.quantize_rows(t, w), K)
    work = device.alloc_work(t, K, N)
    out = torch.empty((N, M), device="cuda")
    device.mul_mat(W, x, out=out, work=work)
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
    for _ in range(300):
        device.mul_mat_compute(W, N, out, work)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(1000):
        device.mul_mat_compute(W, N, out, work)
    e.record(); e.synchronize()
    print(f"{cfg}: sha1 {h}  compute {s.elapsed_time(e):8.2f} us back to back", flush=True)
