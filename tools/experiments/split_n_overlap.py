#!/usr/bin/env python3
"""r4 experiment: does running the headline step as two src1-row halves on two streams (INIT of the second half and the store
tail of the first under the other half's COMPUTE) beat the single INIT + COMPUTE pair?  Needs a dev build whose K3m form can be
forced (GGML_HIP_MX_TILE=30) so that the halves run the kernel form of the whole.
usage: GGML_HIP_LIB=ggmlsharp_amd/lib/dbg/libggml_hip_n30.so GGML_HIP_MX_TILE=30 python tools/experiments/split_n_overlap.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import device  # noqa: E402

M = K = N = 4096
device.init(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
w = torch.randn((M, K), generator=g, device="cuda")
x = torch.randn((N, K), generator=g, device="cuda") * 2
W = device.Weight.from_device(2, device.quantize_rows(2, w), K)
ref = torch.empty((N, M), device="cuda")
work = device.alloc_work(2, K, N)
device.mul_mat(W, x, out=ref, work=work)
torch.cuda.synchronize()


def timed(fn, iters=400, pre=2000):
    for _ in range(pre): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def whole():
    device.mul_mat_init(W, x, work); device.mul_mat_compute(W, N, ref, work)


print(f"whole, one stream: {timed(whole):7.1f} us/step", flush=True)


def graphed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): fn()
    return timed(gr.replay, iters=40, pre=200) / reps


print(f"whole, one stream, graph of 20 steps: {graphed(whole):7.1f} us/step", flush=True)
for nch, nst, join in ((2, 1, True), (2, 2, True), (2, 2, False), (4, 2, True), (4, 4, True)):
    out = torch.zeros((N, M), device="cuda")
    c = N // nch
    works = [device.alloc_work(2, K, c) for _ in range(nch)]
    streams = [torch.cuda.Stream() for _ in range(nst)]
    evs = [torch.cuda.Event() for _ in range(nst)]
    main = torch.cuda.current_stream()
    ev0 = torch.cuda.Event()

    def step():
        if join:
            ev0.record(main)
        for i in range(nch):
            s = streams[i % nst]
            if join and i < nst: s.wait_event(ev0)
            with torch.cuda.stream(s):
                device.mul_mat_init(W, x[i * c:(i + 1) * c], works[i])
                device.mul_mat_compute(W, c, out[i * c:(i + 1) * c], works[i])
        if join:
            for k, s in enumerate(streams):
                evs[k].record(s); main.wait_event(evs[k])
    t = timed(step)
    torch.cuda.synchronize()
    if join:
        try:
            print(f"   the same as a replayed graph of 20 steps: {graphed(step):7.1f} us/step", flush=True)
        except Exception as e:
            print("   graph capture failed:", str(e).splitlines()[0], flush=True)
    same = torch.equal(out, ref)
    print(f"{nch} chunks on {nst} streams, join per step {join}: {t:7.1f} us/step   bitwise == whole: {same}", flush=True)
