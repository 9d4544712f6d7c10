#!/usr/bin/env python3
"""bench.py -- the hot path (quantized mul_mat) on MI355X, one JSON line on rank 0.

Workload (BASELINE.json `metric`): Q4_0 mul_mat M=4096, K=4096, N=4096, f32 activations and weights already
resident in HBM.  One step = one pass of ggml_compute_forward_mul_mat_q_f32 over that input:
INIT phase (quantize the 4096 src1 rows to Q8_0) + COMPUTE phase (block-scaled exact-integer MFMA mat-mat on the MX
matrix path, gemm_qmx.hip) [+ the all-gather of dst shards and the re-layout when --gpus > 1].
  value  = effective GFLOP/s = 2*M*K*N*(ranks) / step time  (whole job)
  roofline = the dominant kernel (COMPUTE phase) alone, timed with HIP events on its stream
  cpu_baseline = the oracle's scalar CPU path (reference algorithm) on the host cores, bounded sample
Multi-GPU (--gpus G, launched by torch.distributed.run): the weight matrix is row-split, rank r owns rows
[r*4096, (r+1)*4096) of a (4096*G) x 4096 matrix (weak scaling: per-GPU work fixed), every rank holds all of
src1, dst shards are exchanged with one RCCL all-gather per step and re-laid-out to the reference's [N][M].
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

Q4_0 = 2
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
I8_MFMA_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x BF16 per clock, ~5 PF dense.  The MX kernel issues bf6 MFMAs
                            # (4x BF16 per clock, ~10 PF) over twice the algorithmic K (two digits per Q8 activation): the same
                            # 5 PF ceiling in algorithmic FLOPs.


def algorithmic_bytes(M, K, N, blk=20):
    # reference formats; Q4_0 block = 20 B / 32 weights; the Q8 scratch is not counted
    return M * (K // 32) * blk + 4 * K * N + 4 * M * N


def event_time_ms(fn, iters, stream):
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(iters):
        fn()
    end.record(stream)
    end.synchronize()
    return start.elapsed_time(end) / iters


Q5_0, Q8_0 = 6, 8
BLOCK_BYTES = {Q4_0: 20, Q5_0: 22, Q8_0: 36}          # Ggml.cs:76-82
TYPE_NAME = {Q4_0: "Q4_0", Q5_0: "Q5_0", Q8_0: "Q8_0"}


def make_weights_q4_0(M, K, seed, qtype=Q4_0):
    """f32 N(0,1) weights quantized on the device with the bit-exact K9 kernel -> reference block rows (device)."""
    from ggmlsharp_amd import device
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    w = torch.randn((M, K), generator=g, device="cuda", dtype=torch.float32)
    return device.quantize_rows(qtype, w)


def cpu_baseline(M, K, n_cols, threads):
    """The reference's CPU algorithm (oracle = C port, scalar block dots, INIT on one thread, rows split over
    `threads`), timed on the host cores on a bounded sample of the same workload."""
    import oracle_lib as O
    rng = np.random.default_rng(0)
    w = rng.standard_normal((M, K)).astype(np.float32)
    x = rng.standard_normal((n_cols, K)).astype(np.float32)
    wq = O.quantize_row(O.Q4_0, w)
    O.mul_mat(O.Q4_0, wq[:64], x[:8], 64, K, 8, nth=1)  # warm the library
    t0 = time.perf_counter()
    O.mul_mat(O.Q4_0, wq, x, M, K, n_cols, nth=threads)
    dt = time.perf_counter() - t0
    return {"value": round(2.0 * M * K * n_cols / dt / 1e9, 2), "unit": "GFLOP/s", "cores": threads, "kind": "port",
            "sample": f"Q4_0 mul_mat M={M} K={K} N={n_cols} (first {n_cols} of 4096 src1 rows), {dt:.2f} s wall, "
                      f"oracle/ggml_oracle.c scalar path, host has {os.cpu_count()} logical CPUs"}


def side_config(device, M, K, N, copies, iters, qtype=Q4_0):
    """Extra measured configs (the other single-GPU configs of BASELINE.json); weights rotate over `copies` distinct
    matrices so that the weight stream is not served from the 256 MB Infinity Cache."""
    ws = []
    for c in range(copies):
        rows = make_weights_q4_0(M, K, 100 + c, qtype)
        ws.append(device.Weight.from_device(qtype, rows, K))
        del rows
    x = torch.randn((N, K), device="cuda", dtype=torch.float32)
    out = torch.empty((N, M), device="cuda", dtype=torch.float32)
    work = device.alloc_work(qtype, K, N)
    if N > 8:
        device.mul_mat_init(ws[0], x, work)   # so that the compute-only timing has valid scratch
    stream = torch.cuda.current_stream()
    state = {"i": 0}

    def step():
        w = ws[state["i"] % copies]
        state["i"] += 1
        device.mul_mat(w, x, out=out, work=work)

    def compute_only():
        w = ws[state["i"] % copies]
        state["i"] += 1
        device.mul_mat_compute(w, N, out, work)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # Launch-bound from Python below ~10 us per call: replay a captured hipGraph of `copies` whole mul_mat calls
    # (one per distinct weight matrix) so the GPU-side rate is what is timed.
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(copies):
            step()
    t_step = event_time_ms(graph.replay, max(3, iters // copies), stream) / copies
    t_comp = event_time_ms(compute_only, iters, stream) if N > 8 else None
    ab = algorithmic_bytes(M, K, N, BLOCK_BYTES[qtype])
    flops = 2.0 * M * K * N
    res = {"workload": f"{TYPE_NAME[qtype]} mul_mat M={M} K={K} N={N}", "ms_per_step": round(t_step, 5),
           "gflops": round(flops / t_step / 1e6, 1), "algorithmic_GBs": round(ab / t_step / 1e6, 1),
           "hbm_frac": round(ab / t_step / 1e6 / HBM_PEAK_GBS, 4), "weight_copies_rotated": copies,
           "timing": "hipGraph replay of one call per weight copy"}
    if t_comp is not None:
        res["compute_kernel_ms"] = round(t_comp, 5)
    for w in ws:
        w.free()
    return res


def dense_config(device, wtype, M, K, N, iters):
    """The dense mul_mat case (F16 / F32 weights): whole call (INIT + COMPUTE), matrix-pipe roofline."""
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    w = torch.randn((M, K), generator=g, device="cuda", dtype=torch.float32)
    raw = (w.half() if wtype == 1 else w).contiguous().view(torch.uint8).view(M, -1)
    W = device.Weight.from_device(wtype, raw, K)
    x = torch.randn((N, K), generator=g, device="cuda", dtype=torch.float32)
    out = torch.empty((N, M), device="cuda", dtype=torch.float32)
    work = device.alloc_work(wtype, K, N)
    stream = torch.cuda.current_stream()
    for _ in range(3):
        device.mul_mat(W, x, out=out, work=work)
    t = event_time_ms(lambda: device.mul_mat(W, x, out=out, work=work), iters, stream)
    tf = 2.0 * M * K * N / t / 1e9
    peak = 2500.0 if wtype == 1 else 157.0      # MI355X_MICROARCH.md: dense f16 MFMA ~2.5 PF; f32 matrix 157 TF
    W.free()
    return {"workload": f"{'F16' if wtype == 1 else 'F32'} mul_mat M={M} K={K} N={N} (INIT + COMPUTE)", "ms_per_step": round(t, 5),
            "gflops": round(tf * 1e3, 1),
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N > 1 flow on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if "GGML_BENCH_FORCE_DEVICE" in os.environ:      # rehearsal of the N > 1 flow with every rank on one GPU
        local_rank = int(os.environ["GGML_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from ggmlsharp_amd import device, dist as gdist
    device.init(local_rank)

    M, K, N = 4096, 4096, 4096
    rows = make_weights_q4_0(M, K, seed=1000 + rank)       # this rank's row shard of the (M*world) x K matrix
    W = device.Weight.from_device(Q4_0, rows, K)
    del rows
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randn((N, K), generator=g, device="cuda", dtype=torch.float32)   # replicated src1
    runner = gdist.RowSplitMulMat(W, N, world, rank, chunks=4 if world > 1 else 1)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.step(x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step(x)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    flops_step = 2.0 * M * K * N * world
    value = flops_step / (ms_per_step * 1e-3) / 1e9

    out = {
        "metric": "effective GFLOP/s, Q4_0 mul_mat 4096x4096x4096 (2*M*K*N / step time)",
        "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i8", "data": "synthetic",
        "config": {"workload": f"Q4_0 mul_mat M={M * world} K={K} N={N} f32 src1, weights resident",
                   "per_gpu": f"M={M} row shard, all of src1", "block_bytes": 20,
                   "parallelism": f"row-split x{world} + all-gather" if world > 1 else "single GPU"},
    }

    if rank == 0:
        # dominant kernel alone (COMPUTE phase), HIP events on the launch stream
        stream = torch.cuda.current_stream()
        # timed inside the step sequence (INIT, COMPUTE, INIT, ...), as the kernels run in the measured loop: the same
        # kernel launched back to back on its own holds a higher clock and reads ~10 % faster
        iters = 20
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
        for e0, e1, e2 in ev:
            e0.record(stream)
            device.mul_mat_init(W, x, runner.work)
            e1.record(stream)
            device.mul_mat_compute(W, N, runner.shard, runner.work)
            e2.record(stream)
        torch.cuda.synchronize()
        t_init = sum(e0.elapsed_time(e1) for e0, e1, _ in ev) / iters
        t_comp = sum(e1.elapsed_time(e2) for _, e1, e2 in ev) / iters
        achieved = 2.0 * M * K * N / (t_comp * 1e-3) / 1e12
        ab = algorithmic_bytes(M, K, N)
        traffic = None   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_traffic.json)
        kernels = {0: ("gemm_q_kernel<Q4_0,2,2>", "v_mfma_i32_32x32x32_i8 + f32 block-scale epilogue on the VALU"),
                   1: ("gemm_q16_kernel<Q4_0,2,4,4,1>", "2 x v_mfma_f32_32x32x16_f16 per tile and block + f32 block-scale epilogue on the VALU"),
                   3: ("gemm_qmx_kernel<Q4_0,2,4,4,1>", "1 x v_mfma_scale_f32_32x32x64_f8f6f4 (bf6 digits, exact) per tile and block + f32 block-scale epilogue on the VALU")}
        from ggmlsharp_amd._lib import lib
        kname, kdesc = kernels.get(lib().ggml_hip_act_image_kind(Q4_0, M, K, N), kernels[0])
        try:
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
                traffic = json.load(f)[f"{kname} M=4096 K=4096 N=4096"]["traffic_bytes"]
        except Exception:
            pass
        out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / I8_MFMA_PEAK_TOPS, 4), "traffic": traffic,
                           "kernel": f"{kname} ({kdesc})",
                           "kernel_ms": round(t_comp, 5), "init_kernel_ms": round(t_init, 5),
                           "algorithmic_bytes": ab,
                           "note": "the binding unit is the VALU, not the matrix pipe: the reference applies two f32 scales per 32-element "
                                   "block (Ggml.cs:1158) = 32 VALU instructions per 32x32 tile and block, floor ~55 us for this shape",
                           "hbm_view": {"achieved_GBs": round(ab / ((t_init + t_comp) * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                                        "frac": round(ab / ((t_init + t_comp) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic bytes / (INIT + COMPUTE kernel time); this shape is MFMA/VALU-bound, not HBM-bound"}}
    if world == 1 and rank == 0:
        if not args.no_side_configs:
            out["other_configs"] = {
                "batch1": side_config(device, 4096, 4096, 1, copies=32, iters=200),
                "prompt512": side_config(device, 4096, 4096, 512, copies=32, iters=50),
                "batch1_M32000": side_config(device, 32000, 4096, 1, copies=8, iters=100),   # the same mat-vec kernel on an 82 MB matrix
                # BASELINE.json configs[3] (the reference has no k-quants: its 5-bit type Q5_0 stands in) and
                # configs[4] on ONE GPU (the 8-GPU row split is `--gpus 8` of the headline shape)
                "q8_0_ffn512": side_config(device, 4096, 11008, 512, copies=6, iters=30, qtype=Q8_0),
                "q5_0_ffn512": side_config(device, 4096, 11008, 512, copies=8, iters=30, qtype=Q5_0),
                "vocab512": side_config(device, 32000, 4096, 512, copies=3, iters=20),
                # the dense case of the path (north_star: MFMA utilisation for the dense f16 / f32 mul_mat)
                "dense_f16": dense_config(device, 1, 4096, 4096, 4096, iters=20),
                "dense_f32": dense_config(device, 0, 4096, 4096, 4096, iters=5),
            }
            for k in ("batch1_M32000",):
                out["other_configs"][k]["roofline"] = {"bound": "hbm", "achieved": out["other_configs"][k]["algorithmic_GBs"],
                                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out["other_configs"][k]["hbm_frac"]}
            out["other_configs"]["batch1"]["roofline"] = {"bound": "hbm", "achieved": out["other_configs"]["batch1"]["algorithmic_GBs"],
                                                          "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                          "frac": out["other_configs"]["batch1"]["hbm_frac"]}
        if not args.no_cpu_baseline:
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(M, K, 2048 if threads >= 8 else 256, threads)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
