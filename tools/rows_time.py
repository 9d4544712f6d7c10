"""Developer probe: bandwidth of the row kernels (K9 quantize, K8 dequantize, add_q_f32) on a 4096 x 4096 matrix."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
M, K = 4096, 4096
x = torch.randn((M, K), device="cuda")
BLK = {2: 20, 3: 24, 4: 20, 6: 22, 7: 24, 8: 36}


def t(fn, it=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for ty in (2, 3, 4, 6, 7, 8):
    q = device.quantize_rows(ty, x)
    bq = M * K // 32 * BLK[ty]
    tq = t(lambda: device.quantize_rows(ty, x))
    td = t(lambda: device.dequantize_rows(ty, q, K))
    ta = t(lambda: device.add_q_f32_rows(ty, q, x))
    print(f"type {ty}: quantize {tq:7.1f} us ({(M * K * 4 + bq) / tq / 1e3:6.0f} GB/s)  dequantize {td:7.1f} us ({(M * K * 4 + bq) / td / 1e3:6.0f} GB/s)  "
          f"add_q_f32 {ta:7.1f} us ({(M * K * 4 + 2 * bq) / ta / 1e3:6.0f} GB/s)", flush=True)
