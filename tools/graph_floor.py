"""Developer probe: time per kernel node of a replayed hipGraph for a trivial kernel -- the floor under the batch-1 mat-vec number."""
import torch
x = torch.zeros(64, device="cuda")
y = torch.zeros(1 << 20, device="cuda")
s = torch.cuda.Stream()
for name, fn in (("1 element add", lambda: x.add_(1.0)), ("4 MB add", lambda: y.add_(1.0))):
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(32):
                fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            g.replay()
        e1.record(s)
        e1.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) * 1e3 / (20 * 32):.2f} us per node")
