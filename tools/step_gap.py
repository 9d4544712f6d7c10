#!/usr/bin/env python3
"""Headline step (INIT + COMPUTE, Q4_0 4096^3): loop period against the two kernels' own durations -- what the kernel
boundaries cost.  Developer tool (select an A/B build with GGML_HIP_LIB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggmlsharp_amd import device
device.init(0)
M = K = N = 4096
g = torch.Generator(device="cuda"); g.manual_seed(1)
ZERO = os.environ.get("STEP_GAP_DATA", "")        # "zero": all-zero operands, "const": one value everywhere (low toggle rate = low power)
wsrc = torch.randn((M, K), generator=g, device="cuda")
x = torch.randn((N, K), generator=g, device="cuda")
if ZERO == "zero":
    wsrc.zero_(); x.zero_()
elif ZERO == "const":
    wsrc.fill_(0.5); x.fill_(1.0)
W = device.Weight.from_device(2, device.quantize_rows(2, wsrc), K)
out = torch.empty((N, M), device="cuda"); work = device.alloc_work(2, K, N)
def step():
    device.mul_mat_init(W, x, work); device.mul_mat_compute(W, N, out, work)
for _ in range(10): step()
it = 50
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(it): step()
b.record(); b.synchronize()
period = a.elapsed_time(b) / it
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(it)]
for e0, e1, e2 in ev:
    e0.record(); device.mul_mat_init(W, x, work); e1.record(); device.mul_mat_compute(W, N, out, work); e2.record()
torch.cuda.synchronize()
ti = sum(e[0].elapsed_time(e[1]) for e in ev) / it; tc = sum(e[1].elapsed_time(e[2]) for e in ev) / it
print(f"{(os.environ.get('GGML_HIP_LIB', 'product') + ' ' + ZERO)[-28:]:>28}: period {period*1e3:7.1f} us   init {ti*1e3:6.1f}  compute {tc*1e3:6.1f}  period - kernels {(period-ti-tc)*1e3:5.1f} us", flush=True)
# --- which boundary costs what: loops of one kernel, and of COMPUTE + a trivial kernel
def period_of(fn, it=50):
    for _ in range(5): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it * 1e3
tiny = torch.zeros(64, device="cuda")
print(f"   COMPUTE only loop {period_of(lambda: device.mul_mat_compute(W, N, out, work)):7.1f} us | INIT only loop {period_of(lambda: device.mul_mat_init(W, x, work)):6.1f} us | "
      f"tiny-kernel loop {period_of(lambda: tiny.add_(1.0)):5.1f} us | COMPUTE + tiny {period_of(lambda: (device.mul_mat_compute(W, N, out, work), tiny.add_(1.0))):7.1f} us | "
      f"INIT + tiny {period_of(lambda: (device.mul_mat_init(W, x, work), tiny.add_(1.0))):6.1f} us", flush=True)
