"""Developer probe: the dense F32 path at a few shapes -- time, a checksum of the result bits (run once with and once without
GGML_HIP_DENSE_OLD=1: the 128 x 128 kernel must give the 64 x 64 kernel's bits) and the error against fp64."""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
g = torch.Generator(device="cuda"); g.manual_seed(5)
for (M, K, N) in ((4096, 4096, 4096), (5000, 1024, 4097), (2048, 4096, 2048), (11008, 4096, 2048), (4096, 4096, 512)):
    w = torch.randn((M, K), generator=g, device="cuda")
    W = device.Weight.from_device(0, w.view(torch.uint8).view(M, -1), K)
    x = torch.randn((N, K), generator=g, device="cuda")
    out = torch.empty((N, M), device="cuda")
    for _ in range(2):
        device.mul_mat(W, x, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 5
    e0.record()
    for _ in range(it):
        device.mul_mat(W, x, out=out)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / it
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    ref = (x[:64].double() @ w.double().T)
    err = ((out[:64].double() - ref).abs().max() / ref.pow(2).mean().sqrt()).item()
    print(f"f32 M{M} K{K} N{N}: {ms * 1e3:9.1f} us  {2.0 * M * K * N / ms / 1e9:8.1f} TFLOP/s  bits {h}  max err/rms {err:.1e}", flush=True)
    W.free()
