"""developer experiment: the Q8_0 batched-decode form with one against two tiles per workgroup (GGML_HIP_Q8S_TILES): same bits?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device
device.init(0)
g = torch.Generator(device="cuda"); g.manual_seed(5)
for (M, K, N) in ((8492, 4160, 64), (8492, 4160, 20), (8492, 2048, 64), (2048, 4160, 64), (2048, 11008, 20)):
    w = torch.randn((M, K), generator=g, device="cuda"); x = torch.randn((N, K), generator=g, device="cuda") * 2
    rows = device.quantize_rows(8, w)
    W = device.Weight.from_device(8, rows, K)
    out = device.mul_mat(W, x)
    wd = device.dequantize_rows(8, rows, K).double()
    xq = device.dequantize_rows(8, device.quantize_rows(8, x.contiguous()), K).double()
    ref = xq @ wd.T
    err = (out.double() - ref).abs().max().item() / ref.pow(2).mean().sqrt().item()
    torch.save(out.cpu(), f"/tmp/q8_{os.environ.get('GGML_HIP_Q8S_TILES','0')}_{M}_{K}_{N}.pt")
    print(os.environ.get('GGML_HIP_Q8S_TILES'), M, K, N, "max err / rms", err, flush=True)
