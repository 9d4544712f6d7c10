#!/usr/bin/env python3
"""bench.py -- the hot path (quantized mul_mat) on MI355X, one JSON line on rank 0.

Workload (BASELINE.json `metric`): Q4_0 mul_mat M=4096, K=4096, N=4096, f32 activations and weights already
resident in HBM.  One step = one pass of ggml_compute_forward_mul_mat_q_f32 over that input:
INIT phase (quantize the 4096 src1 rows to Q8_0) + COMPUTE phase (block-scaled exact-integer MFMA mat-mat on the MX
matrix path, gemm_qmx.hip) [+ the exchange of dst shards when --gpus > 1].
  value  = effective GFLOP/s = 2*M*K*N / step time  (whole job)
  roofline = the dominant kernel (COMPUTE phase) alone, timed with HIP events on its stream
  cpu_baseline = the oracle's scalar CPU path (reference algorithm) on the host cores, bounded sample, T = 1 and T = cores
Multi-GPU (--gpus G): `python bench.py --gpus G` launches its own rank processes (python -m torch.distributed.run, one per
GPU, before anything in this process touches a GPU) unless it already runs under one (WORLD_SIZE set).  The SAME 4096^3
problem is row-split (strong scaling, BASELINE's 1 -> 8 metric): rank r owns rows [dr*r, dr*(r+1)) of W (Ggml.cs:6665-6672),
every rank holds all of src1, and every rank ends with the whole reference-layout dst [N][M] (ggmlsharp_amd/dist.py: RCCL
all-gather + re-layout, or direct peer stores, whichever the run verified and measured faster).  `other_configs` then
carries the weak-scaling run (M = 4096 per rank) and BASELINE config 5 (Q4_0 32000 x 4096 x 512, row-split), each with
compute-only / exchange-only / whole-step times.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "support"))     # ggml_mirror: the stand-in for the C# host (the drop-in side configs only)

Q4_0 = 2
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
I8_MFMA_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x BF16 per clock, ~5 PF dense.  The MX kernel issues bf6 MFMAs
                            # (4x BF16 per clock, ~10 PF) over twice the algorithmic K (two digits per Q8 activation): the same
                            # 5 PF ceiling in algorithmic FLOPs.
PROFILE_TRAFFIC = ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")   # HBM bytes per launch from the committed rocprofv3 PMC passes


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


def per_call_ms(fn, iters, stream):
    """One HIP event pair per call on the launch stream -> list of per-call durations."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def stats(ts):
    """median and p10 / p90 over >= 100 launches."""
    a = np.sort(np.asarray(ts, dtype=np.float64))
    return {"median_ms": round(float(np.median(a)), 5), "p10_ms": round(float(np.percentile(a, 10)), 5),
            "p90_ms": round(float(np.percentile(a, 90)), 5), "launches": int(a.size)}


Q5_0, Q5_1, Q8_0 = 6, 7, 8
Q5_K, Q4_K, Q6_K = 113, 112, 114                      # extension types (upstream k-quant format; absent from the reference)
BLOCK_BYTES = {Q4_0: 20, Q5_0: 22, Q5_1: 24, Q8_0: 36, Q5_K: 22, Q4_K: 18, Q6_K: 26.25}   # Ggml.cs:76-82; per 32 weights -- Q5_K: 176 B per 256, Q4_K: 144, Q6_K: 210
TYPE_NAME = {Q4_0: "Q4_0", Q5_0: "Q5_0", Q5_1: "Q5_1", Q8_0: "Q8_0", Q5_K: "Q5_K", Q4_K: "Q4_K", Q6_K: "Q6_K"}


def make_weights_q4_0(M, K, seed, qtype=Q4_0):
    """f32 N(0,1) weights quantized on the device with the bit-exact K9 kernel -> reference block rows (device)."""
    from ggmlsharp_amd import device
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    # (r4: the extension types too -- kquants.hip quantize_kq_kernel, the published reference quantizers restated; rounds 2-3 fed random bytes)
    w = torch.randn((M, K), generator=g, device="cuda", dtype=torch.float32)
    return device.quantize_rows(qtype, w)


def physical_cores():
    """Physical host cores this process may use: lscpu's sockets x cores-per-socket, capped by the CPU affinity mask."""
    n = None
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {ln.split(":")[0].strip(): ln.split(":")[1].strip() for ln in out.splitlines() if ":" in ln}
        n = int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])
        model = kv.get("Model name", "?")
    except Exception:
        model = "?"
    logical = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except Exception:
        allowed = logical
    if not n:
        n = max(1, logical // 2)
    return max(1, min(n, allowed)), logical, allowed, model


def cpu_baseline_leg(M, K, n_cols, threads, reps=3):
    """`reps` repetitions of the oracle's mul_mat on the first n_cols src1 rows; returns (median GFLOP/s, all wall times)."""
    import oracle_lib as O
    rng = np.random.default_rng(0)
    w = rng.standard_normal((M, K)).astype(np.float32)
    x = rng.standard_normal((n_cols, K)).astype(np.float32)
    wq = O.quantize_row(O.Q4_0, w)
    O.mul_mat(O.Q4_0, wq[:64], x[:8], 64, K, 8, nth=1)  # warm the library
    dts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        O.mul_mat(O.Q4_0, wq, x, M, K, n_cols, nth=threads)
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    return round(2.0 * M * K * n_cols / dt / 1e9, 2), dts


def cpu_baseline(M, K, N):
    """The reference's CPU algorithm (oracle = C port, scalar block dots, INIT on one thread, rows split over T threads,
    Ggml.cs:6641-6698), timed on the host cores as BASELINE.md section 2 states it: three repetitions, the median -- all N src1
    rows at T = physical cores (a few seconds), and a stated sample of the rows at T = 1."""
    cores, logical, allowed, model = physical_cores()
    nT = N if cores >= 8 else 384                     # (a host with a handful of cores: a bounded sample there too)
    vT, dtsT = cpu_baseline_leg(M, K, nT, cores)
    n1 = 256
    v1, dts1 = cpu_baseline_leg(M, K, n1, 1)
    fmt = lambda ds: " / ".join(f"{d:.2f}" for d in ds)   # noqa: E731
    return {"value": vT, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": f"Q4_0 mul_mat M={M} K={K}, {'all' if nT == N else 'first'} {nT} of {N} src1 rows on T={cores} threads, median of 3 repetitions "
                      f"({fmt(dtsT)} s wall); oracle/ggml_oracle.c scalar path; host: {model}, {logical} logical CPUs, {allowed} usable",
            "single_thread": {"value": v1, "unit": "GFLOP/s", "cores": 1,
                              "sample": f"first {n1} of {N} src1 rows, median of 3 repetitions ({fmt(dts1)} s wall)"}}


FAMILY_NAME = {1: "gemv_fused_kernel: fused mat-vec, INIT + COMPUTE in one launch (gemv.hip)", 2: "gemv_q_kernel: two-step mat-vec (gemv.hip)",
               3: "gemm_qmx_small_kernel: K3s, stage-free batched decode, MX bf6 MFMA (gemm_qmx.hip)",
               4: "gemm_q8_small_kernel: K3s on the int8 cores (gemm_q8s.hip)",
               5: "gemm_qmx_mid_kernel: K3p, 128x64 tiles per wave, K split over 8 waves, MX bf6 MFMA (gemm_qmp.hip)",
               6: "gemm_q8_mid_kernel: K3p, 128x64 tiles per wave, K split over 8 waves, v_mfma_i32_32x32x32_i8 off resident int8 planes (gemm_qmp.hip)",
               7: "gemm_qmx_kernel: K3m staged MX form (gemm_qmx.hip)", 8: "gemm_q16_kernel: staged f16 form (gemm_q16.hip)",
               9: "gemm_q_kernel: staged int8 form (gemm_q.hip)", 10: "dense_kernel (dense.hip)", 11: "dense_gemv_kernel (dense.hip)",
               12: "dense16_kernel: F16 on the f16 cores (dense16.hip)", 13: "dense32s_kernel: F32 as split bf16 (dense16.hip)"}


def compute_kernel_name(qtype, M, K, N):
    """which COMPUTE kernel serves the product: asked of the library's own plan (csrc/plan.cpp through ggml_hip_mm_plan), not restated here"""
    import ctypes as C
    from ggmlsharp_amd import _lib
    pl = _lib.ggml_hip_mm_plan_t()
    if _lib.lib().ggml_hip_mm_plan(qtype, M, K, N, C.byref(pl)) != 0:
        return "?"
    name = FAMILY_NAME.get(pl.family, f"family {pl.family}")
    if pl.family == 6 and qtype in (Q5_K, Q4_K, 7, 3):
        name += " + the min terms of 16 k-blocks as one bf16-piece matrix product (5 / 6 v_mfma_f32_32x32x16_bf16 per tile) ahead of the K loop (r4)"
    return f"{name}; plan: form {pl.form}, tile {pl.tile_m}x{pl.tile_n}, {pl.waves} waves, K in {pl.ksplit} partial sum(s), {pl.workgroups} workgroups"


# ---------------------------------------------------------------------------------------------- side configs (bench_configs.py)
def _configs():
    """The side configurations live in bench_configs.py (they are not the contract: `value` never depends on them); it uses this module's helpers."""
    import bench_configs
    bench_configs.B = sys.modules[__name__]
    return bench_configs


def side_config(*a, **k):
    return _configs().side_config(*a, **k)


def dense_config(*a, **k):
    return _configs().dense_config(*a, **k)


def seam1_host_config(*a, **k):
    return _configs().seam1_host_config(*a, **k)


def dropin_decode_layer(*a, **k):
    return _configs().dropin_decode_layer(*a, **k)


def pcie_probe(*a, **k):
    return _configs().pcie_probe(*a, **k)


def projection_group_config(*a, **k):
    return _configs().projection_group_config(*a, **k)


# ---------------------------------------------------------------------------------------------- multi-GPU
def self_launch(args):
    """`python bench.py --gpus N` from a plain shell: start N rank processes (one per GPU) and relay rank 0's JSON line.
    Nothing in THIS process has touched a GPU (torch.cuda.device_count() does not initialise it on this image)."""
    ndev = torch.cuda.device_count()
    env = os.environ.copy()
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = args.backend
    if ndev < args.gpus:
        # fewer devices than ranks (a one-GPU box): rehearse the whole N > 1 flow with ranks sharing devices; RCCL cannot put
        # two ranks on one device, so gloo carries the collectives
        env["GGML_BENCH_SHARE_DEVICES"] = str(max(ndev, 1))
        backend = "gloo"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup",
           str(args.warmup), "--backend", backend, "--exchange", args.exchange, "--preheat-s", str(args.preheat_s)]
    if args.no_side_configs:
        cmd.append("--no-side-configs")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    return p.returncode if p.returncode else (0 if line else 1)


PREHEAT_S = 0.6            # seconds of untimed steps ahead of a timed region (--preheat-s; recorded in the full record)


def preheat(fn, seconds=None, world=1):
    """Untimed: run `fn` back to back for about `seconds` so that the timed region that follows starts on a chip that is already busy.
    The clock of an MI355X that was idle (or busy for milliseconds only) is not the clock it holds in service: the same
    4096^3 kernel on the same data measured 177 us per launch over 20 launches after 3 warm-up launches, 152 us over 200,
    144 us over 5000 (docs/experiments/README.md).  A handful of warm-up steps is over in a millisecond; this is the part
    of the warm-up that is measured in time rather than in steps.  With several ranks `fn` holds collectives, so every rank must
    make the SAME number of calls: rank 0 times one chunk, the chunk count is broadcast.  Returns the number of calls made."""
    seconds = PREHEAT_S if seconds is None else seconds
    if seconds <= 0:
        return 0
    chunk = 16
    t0 = time.perf_counter()
    for _ in range(chunk):
        fn()
    torch.cuda.synchronize()
    dt = max(time.perf_counter() - t0, 1e-6)
    n_chunks = max(0, min(int(seconds / dt), 100000))
    if world > 1:
        t = torch.tensor([n_chunks], dtype=torch.int64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.broadcast(t, src=0)
        n_chunks = int(t.item())
    for _ in range(n_chunks):
        for _ in range(chunk):
            fn()
        torch.cuda.synchronize()
    return chunk * (1 + n_chunks)


def timed_steps(fn, warmup, steps, world):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt * 1e3 / steps


def make_runner(device, gdist, M_total, K, N, world, rank, exchange, chunks, seed):
    r0, r1 = gdist.shard_rows(M_total, world, rank)
    # every rank draws ITS rows from a stream seeded by the row range, so any split of the same matrix holds the same weights
    rows = make_weights_q4_0(max(r1 - r0, 1), K, seed=seed + r0)
    W = device.Weight.from_device(Q4_0, rows, K, row_begin=0, row_end=r1 - r0)
    del rows
    return gdist.RowSplitMulMat(W, N, world, rank, M_total=M_total, chunks=chunks if world > 1 else 1, exchange=exchange), W


def pick_exchange(device, gdist, world, rank, requested, K):
    """Which exchange form runs: "rccl" always works; "push" (direct peer stores through IPC-shared dst buffers) is used
    when every rank could set it up AND its result is bit for bit the RCCL form's on a small problem."""
    import torch.distributed as dist
    if world == 1:
        return "none", "single GPU"
    if requested == "rccl":
        return "rccl", "requested"
    ok = 1.0
    why = "verified against the all-gather form"
    want = "push" if requested == "push" else "push_fused"
    N, M = 96, 64 * world + 8
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.randn((N, K), generator=g, device="cuda")
    ra, wa = make_runner(device, gdist, M, K, N, world, rank, "rccl", 2, 4242)
    a = ra.step(x).clone()
    try:
        # (the set-up keeps every rank in step with its peers whatever fails locally and raises on ALL ranks or on none)
        rb, wb = make_runner(device, gdist, M, K, N, world, rank, want, 2, 4242)
    except Exception as e:  # noqa: BLE001 -- the IPC set-up is unavailable: the RCCL form runs
        return "rccl", f"push set-up failed: {type(e).__name__}: {e}"[:200]
    b1 = rb.step(x).clone()
    b2 = rb.step(x).clone()
    torch.cuda.synchronize()
    if not (torch.equal(a, b1) and torch.equal(a, b2)):
        ok, why = 0.0, "push result differs from the all-gather result"
    dist.barrier()
    rb.close()
    t = torch.tensor([ok], dtype=torch.float32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if float(t.item()) >= 1.0:
        return want, why
    return "rccl", why if ok == 0.0 else "another rank could not use push"


def multi_config(device, gdist, name, M_total, K, N, world, rank, exchange, chunks, steps, warmup):
    """One row-split problem: whole step, compute only, exchange only (max over ranks each)."""
    runner, W = make_runner(device, gdist, M_total, K, N, world, rank, exchange, chunks, seed=9000)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randn((N, K), generator=g, device="cuda", dtype=torch.float32)
    t_step = timed_steps(lambda: runner.step(x), warmup, steps, world)
    t_comp = timed_steps(lambda: runner.compute_only(x), max(2, warmup // 2), steps, world)
    t_xchg = timed_steps(runner.exchange_only, max(2, warmup // 2), steps, world)
    flops = 2.0 * M_total * K * N
    S = 4 * N * gdist.shard_width(M_total, world)
    res = {"workload": f"Q4_0 mul_mat M={M_total} K={K} N={N} row-split x{world} ({name})", "ms_per_step": round(t_step, 5),
           "gflops": round(flops / t_step / 1e6, 1), "compute_only_ms": round(t_comp, 5), "exchange_only_ms": round(t_xchg, 5),
           "overlap_gain_ms": round(t_comp + t_xchg - t_step, 5), "exchange": exchange, "chunks": chunks, "ranks": world,
           "shard_bytes": S, "bytes_in_per_gpu": S * (world - 1),
           "exchange_GBs_in_per_gpu": round(S * (world - 1) / t_xchg / 1e6, 1) if t_xchg > 0 else None}
    runner.close() if hasattr(runner, "close") else None
    W.free()
    return res


LINE_LIMIT = 4096     # the driver reads the LAST stdout line; round 4's 20.8 KB line did not parse


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def compact_line(out):
    """The ONE stdout line: headline + roofline + cpu_baseline + one {frac, ms_per_step, kernel_ms, bound} per BASELINE config.
    Everything else bench.py measures (other_configs, step_stats, protocol prose) goes to the side file / stderr (emit())."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "cold_ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: _short(cfg[k], 160) for k in ("workload", "parallelism") if k in cfg}
    r = out.get("roofline")
    if r:
        rk = ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "kernel_ms", "init_kernel_ms")
        line["roofline"] = {k: r[k] for k in rk if k in r}
        line["roofline"]["kernel"] = _short(r.get("kernel", "?").split(" (")[0], 80)
        if r.get("traffic_from_profile"):
            line["roofline"]["traffic_from"] = _short(r["traffic_from_profile"], 60)
        if r.get("note"):
            line["roofline"]["note"] = _short(r["note"], 120)
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": _short(c["sample"], 200)}
        if "single_thread" in c:
            line["cpu_baseline"]["single_thread_value"] = c["single_thread"]["value"]
    b = out.get("baseline_config_rooflines")
    if b:
        line["baseline_config_rooflines"] = {
            name: {k: v[k] for k in ("frac", "ms_per_step", "kernel_ms", "bound") if k in v} for name, v in b.items()}
    m = out.get("multi_gpu")
    if m:
        mk = ("compute_only_ms", "exchange_only_ms", "overlap_gain_ms", "exchange", "chunks", "ranks_observed", "backend",
              "exchange_GBs_in_per_gpu", "devices_shared_by_ranks")
        line["multi_gpu"] = {k: m[k] for k in mk if k in m}
        oc = out.get("other_configs") or {}
        line["multi_gpu"]["other"] = {name: {k: v[k] for k in ("ms_per_step", "compute_only_ms", "exchange_only_ms", "exchange") if k in v}
                                      for name, v in oc.items() if isinstance(v, dict) and "ms_per_step" in v}
    if out.get("detail_file"):
        line["detail_file"] = out["detail_file"]
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:          # never again: drop the optional parts, longest first, until it fits
        for k in ("multi_gpu", "baseline_config_rooflines"):
            if k in line and len(s) >= LINE_LIMIT:
                line[k] = {"dropped": "line too long; see detail_file"}
                s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def emit(out):
    """Full record -> side file next to the script (gpurun_out/, merged back by gpurun) and stderr; compact line -> stdout, LAST."""
    name = f"bench_full_n{out.get('n_gpus', 1)}.json"
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                json.dump(out, f, indent=1)
            out["detail_file"] = os.path.relpath(os.path.join(d, name), ROOT) if d != "/tmp" else os.path.join(d, name)
            break
        except OSError:
            continue
    print("bench_full: " + json.dumps(out), file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)


def main():
    global PREHEAT_S
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N > 1 flow on one GPU)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "push", "push_fused"],
                    help="dst shard exchange for --gpus > 1: RCCL all-gather + re-layout, direct peer stores by a kernel behind the product (push), or "
                         "peer stores from the GEMM's own store phase (push_fused); auto: push_fused when verified bit for bit against the all-gather, else rccl")
    ap.add_argument("--preheat-s", type=float, default=PREHEAT_S, help="seconds of untimed steps of the same workload ahead of each timed region")
    args = ap.parse_args()
    PREHEAT_S = args.preheat_s

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "GGML_BENCH_SHARE_DEVICES" in os.environ:      # rehearsal of the N > 1 flow with ranks sharing the box's device(s)
        local_rank = local_rank % int(os.environ["GGML_BENCH_SHARE_DEVICES"])
    if "GGML_BENCH_FORCE_DEVICE" in os.environ:
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
    exchange, exchange_why = pick_exchange(device, gdist, world, rank, args.exchange, K)
    # src1-row chunks (chunk i's exchange overlaps chunk i + 1's kernels): only where a chunk's kernels still fill the chip.
    # One launch of the MX mat-mat runs the whole K loop per workgroup (>= 40 us at K = 4096 whatever the grid): measured on
    # one GPU, 512 x 4096 x 4096 45 us in one launch against 4 x 38 us in chunks of 1024 rows; 4096 x 4096 x 4096 160 against
    # 4 x 54 us.  So a row shard of the strong-scaling problem (Ms = 4096 / G) is not chunked; the weak-scaling problem is.
    chunks = 1 if (world == 1 or gdist.shard_width(M, world) < 4096) else 4
    runner, W = make_runner(device, gdist, M, K, N, world, rank, exchange, chunks, seed=1000)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randn((N, K), generator=g, device="cuda", dtype=torch.float32)   # replicated src1

    # the round-2 protocol first (W warm-up steps on a chip that was busy for milliseconds), reported as "cold"; then the same
    # W + K steps behind PREHEAT_S seconds of the same step: the steady-state figure, which is `value`
    cold_ms_per_step = timed_steps(lambda: runner.step(x), args.warmup, args.steps, world)
    n_pre = preheat(lambda: runner.step(x), world=world)
    ms_per_step = timed_steps(lambda: runner.step(x), args.warmup, args.steps, world)
    flops_step = 2.0 * M * K * N
    value = flops_step / (ms_per_step * 1e-3) / 1e9

    out = {
        "metric": "effective GFLOP/s, Q4_0 mul_mat 4096x4096x4096 (2*M*K*N / step time)",
        "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        # the SAME W + K steps WITHOUT the untimed preheat (the protocol of rounds 1-2): compare rounds on this pair, not on `value` alone
        "cold_ms_per_step": round(cold_ms_per_step, 5), "cold_value": round(flops_step / (cold_ms_per_step * 1e-3) / 1e9, 1),
        "timing_protocol": f"value / ms_per_step: {PREHEAT_S} s of untimed steps of the same workload, then W warm-up steps, then exactly K timed steps between "
                           "barrier + synchronize; cold_*: W warm-up steps then K timed steps on a chip that was idle (no preheat)",
        "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
        "dtype": "i8", "data": "synthetic",
        "config": {"workload": f"Q4_0 mul_mat M={M} K={K} N={N} f32 src1, weights resident",
                   "per_gpu": f"rows [{gdist.shard_rows(M, world, 0)[0]}, {gdist.shard_rows(M, world, 0)[1]}) of W on rank 0, all of src1, whole dst [N][M] on every rank",
                   "block_bytes": 20,
                   "parallelism": f"row-split x{world} (Ggml.cs:6665-6672) + exchange '{exchange}' ({exchange_why})" if world > 1 else "single GPU"},
    }
    out["clock_preheat"] = {"seconds": PREHEAT_S, "untimed_steps": n_pre, "cold_ms_per_step": round(cold_ms_per_step, 5),
                            "note": "untimed steps of the same workload ahead of the W warm-up steps, so that the K timed steps run at the clock the chip "
                                    "holds in service; cold_ms_per_step = the same W + K steps without them (the protocol of rounds 1-2)"}
    stream = torch.cuda.current_stream()
    # per-step HIP events -> median / p10 / p90 (after the wall-clock region, same loop body)
    n_ev = max(args.steps, 100) if world == 1 else args.steps
    out["step_stats"] = stats(per_call_ms(lambda: runner.step(x), n_ev, stream))

    if world > 1:
        t_comp = timed_steps(lambda: runner.compute_only(x), 3, args.steps, world)
        t_xchg = timed_steps(runner.exchange_only, 3, args.steps, world)
        S = 4 * N * gdist.shard_width(M, world)
        out["multi_gpu"] = {"compute_only_ms": round(t_comp, 5), "exchange_only_ms": round(t_xchg, 5),
                            "overlap_gain_ms": round(t_comp + t_xchg - ms_per_step, 5), "exchange": exchange, "exchange_choice": exchange_why,
                            "chunks": chunks, "ranks_observed": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
                            "shard_bytes": S, "bytes_in_per_gpu": S * (world - 1),
                            "exchange_GBs_in_per_gpu": round(S * (world - 1) / t_xchg / 1e6, 1) if t_xchg > 0 else None,
                            "devices_shared_by_ranks": "GGML_BENCH_SHARE_DEVICES" in os.environ}
        # the dominant kernel per GPU: one rank's row shard (INIT + COMPUTE, no exchange; max over ranks) against ONE GPU's matrix peak
        Ms = gdist.shard_width(M, world)
        tf_shard = 2.0 * Ms * K * N / (t_comp * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "achieved": round(tf_shard, 2), "peak": I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s",
                           "frac": round(tf_shard / I8_MFMA_PEAK_TOPS, 4), "traffic": None, "algorithmic_bytes": algorithmic_bytes(Ms, K, N),
                           "kernel": compute_kernel_name(Q4_0, Ms, K, N), "kernel_ms": round(t_comp, 5),
                           "note": f"per GPU: INIT + COMPUTE of one rank's {Ms}-row shard (no exchange), max over ranks, vs one GPU's peak"}
        if hasattr(runner, "close"):
            torch.distributed.barrier()
            runner.close()
        W.free()
        if not args.no_side_configs:
            small = max(5, args.steps // 4)
            oc = {}
            # the weak-scaling run of round 1 (per-GPU work fixed: rank r owns 4096 rows of a (4096 * G) x 4096 matrix)
            oc["weak_4096_rows_per_gpu"] = multi_config(device, gdist, "weak scaling", 4096 * world, K, N, world, rank, exchange, 4, small, 3)
            # BASELINE.json configs[4]: Q4_0 32000 x 4096 x 512 row-split (Ms = 4000 at 8 GPUs)
            oc["config5_vocab512"] = multi_config(device, gdist, "BASELINE config 5", 32000, K, 512, world, rank, exchange, 2, small, 3)
            if exchange in ("push", "push_fused"):   # the other forms beside it, same problem: what fusing the exchange into the store phase buys
                oc["headline_rccl_allgather"] = multi_config(device, gdist, "strong scaling, RCCL all-gather + re-layout", M, K, N, world, rank, "rccl", chunks, small, 3)
                other = "push" if exchange == "push_fused" else "push_fused"
                oc[f"headline_{other}"] = multi_config(device, gdist, f"strong scaling, exchange '{other}'", M, K, N, world, rank, other, chunks, small, 3)
            out["other_configs"] = oc

    if world == 1 and rank == 0:
        # dominant kernel alone (COMPUTE phase), HIP events on the launch stream
        # timed inside the step sequence (INIT, COMPUTE, INIT, ...), as the kernels run in the measured loop: the same
        # kernel launched back to back on its own holds a higher clock and reads ~10 % faster
        iters = 100
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(iters)]
        for e0, e1, e2 in ev:
            e0.record(stream)
            device.mul_mat_init(W, x, runner.work)
            e1.record(stream)
            device.mul_mat_compute(W, N, runner.shard, runner.work)
            e2.record(stream)
        torch.cuda.synchronize()
        init_ts = [e0.elapsed_time(e1) for e0, e1, _ in ev]
        comp_ts = [e1.elapsed_time(e2) for _, e1, e2 in ev]
        t_init_ev, t_comp_ev = float(np.mean(init_ts)), float(np.mean(comp_ts))
        # an event pair around EVERY launch exposes ~10 us between a kernel's dispatch and its first tiles (r3: kernel_ms
        # came out larger than ms_per_step).  The dominant kernel's duration AS THE STEP RUNS IT: ONE event pair around `blk` back-to-back
        # INIT + COMPUTE pairs, minus INIT's own duration measured the same way (one pair around `blk` back-to-back INIT launches) --
        # so init_kernel_ms + kernel_ms is the step the headline times, and the rocprofv3 --kernel-trace average of the same command
        # (profiles/r04_bench_kernel_stats.csv) is the cross-check.  The per-launch event figures stay as *_per_launch_events.
        blk, nblk = 25, 8

        def pairs():
            device.mul_mat_init(W, x, runner.work)
            device.mul_mat_compute(W, N, runner.shard, runner.work)

        def inits():
            device.mul_mat_init(W, x, runner.work)
        t_pairs = [event_time_ms(pairs, blk, stream) for _ in range(nblk)]
        t_inits = [event_time_ms(inits, blk, stream) for _ in range(nblk)]
        t_init = float(np.median(t_inits))
        t_comp = float(np.median(t_pairs)) - t_init
        achieved = 2.0 * M * K * N / (t_comp * 1e-3) / 1e12
        ab = algorithmic_bytes(M, K, N)
        traffic, traffic_src = None, None
        kernels = {0: ("gemm_q_kernel<Q4_0,2,2>", "v_mfma_i32_32x32x32_i8 + f32 block-scale epilogue on the VALU"),
                   1: ("gemm_q16_kernel<Q4_0,2,4,4,1>", "2 x v_mfma_f32_32x32x16_f16 per tile and block + f32 block-scale epilogue on the VALU"),
                   3: ("gemm_qmx_kernel<Q4_0,2,4,4,1>", "1 x v_mfma_scale_f32_32x32x64_f8f6f4 (bf6 digits, exact) per tile and block + f32 block-scale epilogue on the VALU")}
        from ggmlsharp_amd._lib import lib
        kname, kdesc = kernels.get(lib().ggml_hip_act_image_kind(Q4_0, K, N), kernels[0])
        for fn in PROFILE_TRAFFIC:
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    traffic = json.load(f)[f"{kname} M=4096 K=4096 N=4096"]["traffic_bytes"]
                traffic_src = f"profiles/{fn}"
                break
            except Exception:
                pass
        out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s",
                           "frac": round(achieved / I8_MFMA_PEAK_TOPS, 4), "traffic": traffic,
                           # NOT measured by this run: HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
                           # (tools/profile_round.sh -> tools/summarize_round.py, the guide's FETCH/WRITE_SIZE recipe and gfx950 corrections)
                           "traffic_from_profile": traffic_src,
                           "kernel": f"{kname} ({kdesc})", "kernel_plan": compute_kernel_name(Q4_0, M, K, N),
                           "kernel_ms": round(t_comp, 5), "init_kernel_ms": round(t_init, 5),
                           "kernel_ms_method": f"median over {nblk} blocks of (one HIP event pair around {blk} back-to-back INIT + COMPUTE pairs) - (the same around {blk} INIT launches)",
                           "step_from_kernels_ms": round(t_comp + t_init, 5),
                           "kernel_ms_per_launch_events": round(t_comp_ev, 5), "kernel_ms_per_launch_events_stats": stats(comp_ts),
                           "init_kernel_ms_per_launch_events": round(t_init_ev, 5),
                           "algorithmic_bytes": ab,
                           "note": "the binding unit is the VALU, not the matrix pipe: the reference applies two f32 scales per 32-element "
                                   "block (Ggml.cs:1158) = 32 VALU instructions per 32x32 tile and block, floor ~55 us for this shape",
                           "hbm_view": {"achieved_GBs": round(ab / ((t_init + t_comp) * 1e-3) / 1e9, 1), "peak_GBs": HBM_PEAK_GBS,
                                        "frac": round(ab / ((t_init + t_comp) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "note": "algorithmic bytes / (INIT + COMPUTE kernel time); this shape is MFMA/VALU-bound, not HBM-bound"}}
        if not args.no_side_configs:
            # the other BASELINE configs, the batched-decode / dense / k-quant shapes and the drop-in path with host tensors: bench_configs.py
            out["other_configs"], out["baseline_config_rooflines"] = _configs().single_gpu_side_configs(device)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, K, N)
    if rank == 0:
        emit(out)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
