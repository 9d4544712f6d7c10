#!/usr/bin/env python3
"""How exact is the min-term part of K3p-int8 (gemm_qmp.hip)?  Q5_1 weights with every block scale d set to 0: the product is then
sum_b m[r][b] * (d1 * sum(a))[n][b] alone.  Compared with an fp64 evaluation from the planes K1 wrote.  (developer tool, GPU box;
GGML_HIP_LIB selects a variant build)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device

Q5_1 = 7
for (M, K, N) in ((4096, 4096, 512), (4096, 11008, 512)):
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((N, K), generator=g, device="cuda")
    rows = device.quantize_rows(Q5_1, w).view(M, K // 32, 24).clone()
    rows[:, :, 0:2] = 0                                   # d = +0.0 (f16)
    mn = rows[:, :, 2:4].contiguous().view(torch.float16).reshape(M, K // 32).double()
    W = device.Weight.from_device(Q5_1, rows.reshape(M, -1).contiguous(), K)
    work = device.alloc_work(Q5_1, K, N)
    out = torch.empty((N, M), device="cuda")
    device.mul_mat_init(W, x, work)
    device.mul_mat_compute(W, N, out, work)
    torch.cuda.synchronize()
    nbk, Npad = K // 32, (N + 255) // 256 * 256
    nba = (nbk + 3) // 4 * 4
    img = nba * 4 * Npad * 16
    ad = work[img: img + nba * Npad * 4].view(torch.float32).reshape(nba, Npad)[:nbk, :N]
    asum = work[img + nba * Npad * 4: img + 2 * nba * Npad * 4].view(torch.int32).reshape(nba, Npad)[:nbk, :N]
    s = (ad * asum.float()).double()                      # the f32 product the kernel's operand is
    ref = s.T @ mn.T                                      # [N][M]
    err = (out.double() - ref).abs()
    rms = ref.pow(2).mean().sqrt().item()
    print(f"M{M} K{K} N{N}: min-term only: max_err/rms {err.max().item() / rms:.3e}  rms_err/rms {err.pow(2).mean().sqrt().item() / rms:.3e}  "
          f"mean signed err/rms {(out.double() - ref).mean().item() / rms:+.3e}", flush=True)
    W.free()
