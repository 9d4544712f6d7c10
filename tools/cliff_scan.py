"""Developer probe: whole-call time (INIT + COMPUTE) over a grid of shapes and types -- looks for cliffs (time not monotone in N)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
TYPES = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q8_0": 8, "q4_2": 4, "q5_1": 7, "f16": 1, "f32": 0}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(TYPES)
for name in names:
    t = TYPES[name]
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(4096, 4096), (11008, 4096)]
    for (M, K) in shapes:
        w = torch.randn((M, K), device="cuda")
        if t == 1:
            W = device.Weight.from_device(t, w.half().view(torch.uint8).view(M, -1), K)
        elif t == 0:
            W = device.Weight.from_device(t, w.view(torch.uint8).view(M, -1), K)
        else:
            W = device.Weight.from_device(t, device.quantize_rows(t, w), K)
        line = []
        for N in (1, 4, 8, 9, 16, 32, 64, 128, 129, 256, 512, 513, 1024, 2048, 4096):
            x = torch.randn((N, K), device="cuda")
            out = torch.empty((N, M), device="cuda")
            work = device.alloc_work(t, K, N)
            for _ in range(2):
                device.mul_mat(W, x, out=out, work=work)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 8
            e0.record()
            for _ in range(it):
                device.mul_mat(W, x, out=out, work=work)
            e1.record(); e1.synchronize()
            line.append(f"{N}:{e0.elapsed_time(e1) / it * 1e3:.0f}")
        print(f"{name} M{M} K{K} us per call  " + "  ".join(line), flush=True)
        W.free()
