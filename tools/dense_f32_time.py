"""Developer probe: time of the dense F32 / F16 mul_mat paths at a few shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
for t, dt, name in ((0, torch.float32, "f32"), (1, torch.float16, "f16")):
    for (M, K, N) in ((4096, 4096, 512), (4096, 4096, 4096), (4096, 4096, 1), (64, 128, 256)):
        w = torch.randn((M, K), device="cuda").to(dt)
        W = device.Weight.from_device(t, w.view(torch.uint8).view(M, -1), K)
        x = torch.randn((N, K), device="cuda")
        out = torch.empty((N, M), device="cuda")
        for _ in range(3):
            device.mul_mat(W, x, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            device.mul_mat(W, x, out=out)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / it
        print(f"{name} M{M} K{K} N{N}: {ms * 1e3:9.1f} us  {2.0 * M * K * N / ms / 1e9:8.1f} TFLOP/s", flush=True)
        W.free()
