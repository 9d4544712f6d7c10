"""bench_configs.py -- the SIDE configurations of bench.py (r5: moved out of bench.py, which keeps the contract: headline, roofline, cpu_baseline,
the one stdout line, the multi-GPU flow).  Nothing here feeds `value`.  The other single-GPU configs of BASELINE.json (2, 3, 4 in both
orientations and with Q5_K / Q5_0, 5 as a total and as the shard a rank runs), batched decode, dense F16 / F32, the k-quant extras, and the
drop-in path with HOST tensors (Seam 1 through the host mirror: PCIe-inclusive).  Uses bench.py's helpers through `B` (bound by bench._configs())."""
import time

import numpy as np
import torch

B = None          # the bench module (its helpers and constants): set by bench._configs()


def side_config(device, M, K, N, copies, iters, qtype=2):
    """Extra measured configs (the other single-GPU configs of BASELINE.json); weights rotate over `copies` distinct
    matrices so that the weight stream is not served from the 256 MB Infinity Cache."""
    ws = []
    for c in range(copies):
        rows = B.make_weights_q4_0(M, K, 100 + c, qtype)
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
    # (at least 32 calls per graph: a replay has a fixed cost of several microseconds of its own, which two or four calls of
    # a large shape do not amortise -- M = 32000 read 1.5-2 us per call high with one call per copy)
    nodes = copies * max(1, -(-32 // copies))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(nodes):
            step()
    reps = max(20, iters // nodes)
    B.preheat(graph.replay, B.PREHEAT_S / 2)
    per = [t / nodes for t in B.per_call_ms(graph.replay, reps, stream)]
    t_step = float(np.median(per))
    t_comp = float(np.median(B.per_call_ms(compute_only, iters, stream))) if N > 8 else None
    ab = B.algorithmic_bytes(M, K, N, B.BLOCK_BYTES[qtype])
    flops = 2.0 * M * K * N
    st = B.stats(per)
    res = {"workload": f"{B.TYPE_NAME[qtype]} mul_mat M={M} K={K} N={N}", "ms_per_step": round(t_step, 5),
           "p10_ms": st["p10_ms"], "p90_ms": st["p90_ms"],
           "gflops": round(flops / t_step / 1e6, 1), "algorithmic_GBs": round(ab / t_step / 1e6, 1),
           "hbm_frac": round(ab / t_step / 1e6 / B.HBM_PEAK_GBS, 4), "weight_copies_rotated": copies,
           "timing": f"median over {reps} hipGraph replays of {nodes} calls rotating over the weight copies (dependent launches: includes the inter-kernel boundary)"}
    if t_comp is not None:
        res["compute_kernel_ms"] = round(t_comp, 5)
        # the binding roof of the mat-mat shapes is the matrix unit: the COMPUTE kernel alone against the dense int8-class peak
        tops = flops / (t_comp * 1e-3) / 1e12
        res["roofline"] = {"bound": "mfma", "achieved": round(tops, 1), "peak": B.I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s", "frac": round(tops / B.I8_MFMA_PEAK_TOPS, 4),
                           "kernel_ms": round(t_comp, 5), "kernel": B.compute_kernel_name(qtype, M, K, N)}
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
    # (r4: the clock B.preheat every other config has -- the weights above are made on an idle chip, and 20 launches straight after an idle
    # stretch read 0.138 ms where the same launches a second into a busy stretch read 0.120: tools/experiments/dense_order.py)
    B.preheat(lambda: device.mul_mat(W, x, out=out, work=work), B.PREHEAT_S / 2)
    t = float(np.median(B.per_call_ms(lambda: device.mul_mat(W, x, out=out, work=work), iters, stream)))
    tf = 2.0 * M * K * N / t / 1e9
    # MI355X_MICROARCH.md: dense f16 / bf16 MFMA ~2.5 PF.  F32 (above 256 src1 rows): each operand split exactly into three bf16 pieces,
    # six bf16 MFMAs per f32 product (dense16.hip K10d) -- the roof of that form is 2.5 PF / 6; the f32 matrix instruction itself peaks at 157 TF
    peak = 2500.0 if wtype == 1 else 2500.0 / 6
    W.free()
    res = {"workload": f"{'F16' if wtype == 1 else 'F32'} mul_mat M={M} K={K} N={N} (INIT + COMPUTE)", "ms_per_step": round(t, 5),
           "gflops": round(tf * 1e3, 1), "timing": f"median of {iters} launches after {B.PREHEAT_S / 2:.1f} s of untimed launches of the same call",
           "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4)}}
    if wtype != 1:
        res["roofline"]["note"] = "f32-equivalent TFLOP/s against 2.5 PF bf16 / 6 MFMAs per product; the f32 matrix instruction's own peak is 157 TF"
        res["roofline"]["vs_f32_mfma_peak"] = round(tf / 157.0, 4)
    return res


def seam1_host_config(M, K, N, iters):
    """The DROP-IN path: ggml_graph_compute of one mul_mat node with HOST tensors (Seam 1, host pointers in and out), as
    the C# host would run it -- the context pool registered for DMA (ggml_hip_register_host_pool, done by the mirror's
    ggml_init), src1 / dst moved in chunks of src1 rows that overlap the kernels.  PCIe-inclusive: never the headline."""
    import ggml_mirror as G
    rows = B.make_weights_q4_0(M, K, 77).cpu().numpy()
    pool = (M * (K // 32) * 20) + 4 * K * N + 4 * M * N + (1 << 20)
    ctx = G.ggml_init(pool)
    try:
        W = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        G.tensor_bytes(W)[:] = rows.reshape(-1)
        G.tensor_f32(X)[:] = np.random.default_rng(3).standard_normal((1, 1, N, K)).astype(np.float32)
        Y = G.ggml_mul_mat(ctx, W, X)
        gf = G.ggml_build_forward(Y)
        G.ggml_graph_compute(ctx, gf)          # uploads and caches the weights
        G.ggml_graph_compute(ctx, gf)
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            G.ggml_graph_compute(ctx, gf)
            ts.append((time.perf_counter() - t0) * 1e3)
        t = float(np.median(ts))
        moved = 4.0 * K * N + 4.0 * M * N
        return {"workload": f"B.Q4_0 mul_mat M={M} K={K} N={N}, host tensors through ggml_graph_compute (Seam 1)",
                "ms_per_step": round(t, 4), "p10_ms": round(float(np.percentile(ts, 10)), 4), "p90_ms": round(float(np.percentile(ts, 90)), 4),
                "pcie_GBs_both_directions": round(moved / t / 1e6, 1), "gflops": round(2.0 * M * K * N / t / 1e6, 1),
                "host_bytes_in": 4 * K * N, "host_bytes_out": 4 * M * N}
    finally:
        G.ggml_free(ctx)


def dropin_decode_layer(N, iters, layers=1):
    """The DROP-IN path at decode size: one LLaMA-7B-shaped decoder layer (D = 4096, F = 11008, B.Q4_0; the attention itself is
    replaced by adds -- soft_max / rope are outside this path) at batch N through ggml_graph_compute of the host mirror, host
    tensors in the registered context pool.  What is timed is the wall clock of a whole graph compute (17 nodes): the seams,
    the launches, the device -> host copies that leave every node's data in host memory."""
    import ctypes as C
    import ggml_mirror as G
    from ggmlsharp_amd._lib import lib
    D, F = 4096, 11008
    rng = np.random.default_rng(1)
    ctx = G.ggml_init((200 + 500 * layers) * 1024 * 1024)
    try:
        def qweight(K, M):
            t = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
            b = G.tensor_bytes(t).reshape(M * (K // 32), 20)
            b[:, 4:] = rng.integers(0, 256, (M * (K // 32), 16), dtype=np.uint8)
            b[:, :4] = (rng.random(M * (K // 32), dtype=np.float32) * 0.02 + 0.001).view(np.uint8).reshape(-1, 4)
            return t

        def f32(K, n):
            t = G.ggml_new_tensor_2d(ctx, G.F32, K, n)
            G.tensor_f32(t)[:] = rng.standard_normal((n, K)).astype(np.float32).reshape(1, 1, n, K)
            return t

        x = f32(D, N)
        out = x
        for _ in range(layers):             # (several layers in ONE graph: the fixed cost of a graph compute is paid once, as in a real decoder)
            xin, g1, g2 = out, f32(D, N), f32(D, N)
            wq, wk, wv, wo = (qweight(D, D) for _ in range(4))
            w1, w3, w2 = qweight(D, F), qweight(D, F), qweight(F, D)
            cur = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, xin), g1)
            q, k, v = G.ggml_mul_mat(ctx, wq, cur), G.ggml_mul_mat(ctx, wk, cur), G.ggml_mul_mat(ctx, wv, cur)
            a = G.ggml_add(ctx, G.ggml_add(ctx, q, k), v)
            h = G.ggml_add(ctx, G.ggml_mul_mat(ctx, wo, a), xin)
            cur2 = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, h), g2)
            u, gt = G.ggml_mul_mat(ctx, w1, cur2), G.ggml_mul_mat(ctx, w3, cur2)
            s = G.ggml_mul(ctx, G.ggml_silu(ctx, u), gt)
            out = G.ggml_add(ctx, G.ggml_mul_mat(ctx, w2, s), h)
        gf = G.ggml_build_forward(out)
        for _ in range(5):
            G.ggml_graph_compute(ctx, gf)
        ts = []
        for i in range(iters):
            G.tensor_f32(x)[0, 0, 0, 0] = float(i)          # a new token every time: the captured scope re-reads its leaves
            t0 = time.perf_counter()
            G.ggml_graph_compute(ctx, gf)
            ts.append((time.perf_counter() - t0) * 1e6)
        cnt = [C.c_uint64() for _ in range(4)]
        lib().ggml_hip_debug_scope_counters(*[C.byref(c) for c in cnt])
        wbytes = (4 * D * D + 3 * D * F) // 32 * 20 * layers
        t = float(np.median(ts))
        return {"workload": f"{layers} x 7B-shaped decoder layer (B.Q4_0, 7 mul_mat + 10 element-wise nodes each) in one graph, batch {N}, host tensors through ggml_graph_compute",
                "us_per_graph": round(t, 1), "us_per_layer": round(t / layers, 1), "p10_us": round(float(np.percentile(ts, 10)), 1), "p90_us": round(float(np.percentile(ts, 90)), 1),
                "nodes": int(gf.n_nodes), "weight_bytes": wbytes, "weight_stream_GBs": round(wbytes / t / 1e3, 1),
                "hbm_frac": round(wbytes / t / 1e3 / B.HBM_PEAK_GBS, 4),
                "named_scopes": {"observed": cnt[0].value, "captured": cnt[1].value, "replayed": cnt[2].value, "refused": cnt[3].value}}
    finally:
        G.ggml_free(ctx)


def pcie_probe():
    """Pinned host <-> device copy rates of this box (what bounds the drop-in path)."""
    n = 64 << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t_in = float(np.median(B.per_call_ms(lambda: d.copy_(h, non_blocking=True), 10, stream)))
    t_out = float(np.median(B.per_call_ms(lambda: h.copy_(d, non_blocking=True), 10, stream)))
    # both directions at once (what a pipelined Seam-1 call asks of the link): two streams, one copy each way
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        s1.wait_event(a)
        s2.wait_event(a)
        with torch.cuda.stream(s1):
            d.copy_(h, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(d2, non_blocking=True)
        stream.wait_stream(s1)
        stream.wait_stream(s2)
        b.record(stream)
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return {"h2d_GBs": round(n / t_in / 1e6, 1), "d2h_GBs": round(n / t_out / 1e6, 1),
            "both_ways_total_GBs": round(2 * n / float(np.median(ts)) / 1e6, 1)}



def projection_group_config(device, Ms, K, N, sets=12, reps=20):
    """q / k / v (or gate / up) of a batched decoder's step: the matrices of one group behind ONE quantization of src1 and one
    launch (ggml_hip_mul_mat_multi_work_dev) next to one ggml_hip_mul_mat_dev per matrix; a ring of weight sets larger than the
    caches, each form inside a replayed hipGraph."""
    import ctypes as C
    from ggmlsharp_amd._lib import check, lib
    L = lib()
    g = torch.Generator(device="cuda")
    g.manual_seed(21)
    rows = [device.quantize_rows(B.Q4_0, torch.randn((M, K), generator=g, device="cuda")) for M in Ms]
    W = [[device.Weight.from_device(B.Q4_0, r, K) for r in rows] for _ in range(sets)]
    x = torch.randn((N, K), generator=g, device="cuda")
    outs = [torch.empty((N, M), device="cuda") for M in Ms]
    work = device.alloc_work(B.Q4_0, K, N)
    dp = (C.c_void_p * len(Ms))(*[o.data_ptr() for o in outs])
    ld = (C.c_int64 * len(Ms))(*Ms)
    s = torch.cuda.Stream()
    res = {}
    try:
        for mode in ("single_calls", "one_call"):
            with torch.cuda.stream(s):
                st = C.c_void_p(s.cuda_stream)

                def body():
                    for ws in W:
                        if mode == "one_call":
                            hw = (C.c_void_p * len(Ms))(*[w.handle for w in ws])
                            check(L.ggml_hip_mul_mat_multi_work_dev(hw, len(Ms), C.c_void_p(x.data_ptr()), K, N, dp, ld, C.c_void_p(work.data_ptr()),
                                                                    work.numel(), st), "multi")
                        else:
                            for w, o, M in zip(ws, outs, Ms):
                                check(L.ggml_hip_mul_mat_dev(w.handle, C.c_void_p(x.data_ptr()), N, K, C.c_void_p(o.data_ptr()), M,
                                                             C.c_void_p(work.data_ptr()), work.numel(), st), "single")
                body()
                s.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    body()
                gr.replay()
                s.synchronize()
                ts = []
                for _ in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    gr.replay()
                    e1.record(s)
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) / sets)
            res[mode] = B.stats(ts)
    finally:
        for ws in W:
            for w in ws:
                w.free()
    wbytes = sum(M * (K // 32) * 20 for M in Ms)
    t = res["one_call"]["median_ms"]
    return {"workload": f"B.Q4_0 {len(Ms)} x mul_mat M={Ms[0]} K={K} N={N} behind one src1 (a projection group of a batched decoder's step)",
            "ms_per_group": round(t, 5), "p10_ms": res["one_call"]["p10_ms"], "p90_ms": res["one_call"]["p90_ms"],
            "ms_as_single_calls": round(res["single_calls"]["median_ms"], 5), "weight_sets_rotated": sets,
            "algorithmic_GBs": round(wbytes / (t * 1e-3) / 1e9, 1), "hbm_frac": round(wbytes / (t * 1e-3) / 1e9 / B.HBM_PEAK_GBS, 4)}


def single_gpu_side_configs(device):
    """Every side measurement of the N = 1 run: (other_configs, baseline_config_rooflines)."""
    oc = {
        "batch1": side_config(device, 4096, 4096, 1, copies=32, iters=200),
        "batch32": side_config(device, 4096, 4096, 32, copies=32, iters=100),      # a batched decoder's step: INIT + the stage-free MX form (gemm_qmx.hip K3s)
        "prompt512": side_config(device, 4096, 4096, 512, copies=32, iters=100),
        "batch1_M32000": side_config(device, 32000, 4096, 1, copies=8, iters=100),   # the same mat-vec kernel on an 82 MB matrix
        # BASELINE.json configs[3] (the reference has no k-quants: its 5-bit type Q5_0 stands in) and
        # configs[4] on ONE GPU (the row split is `--gpus 8`: other_configs.config5_vocab512 of that line)
        "q8_0_ffn512": side_config(device, 4096, 11008, 512, copies=6, iters=60, qtype=B.Q8_0),
        "q5_0_ffn512": side_config(device, 4096, 11008, 512, copies=8, iters=60, qtype=B.Q5_0),
        # config 4 in its other orientation ("bench both orientations"): the up projection, M = 11008, K = 4096
        "q8_0_ffn512_up": side_config(device, 11008, 4096, 512, copies=6, iters=60, qtype=B.Q8_0),
        "q5_0_ffn512_up": side_config(device, 11008, 4096, 512, copies=8, iters=60, qtype=B.Q5_0),
        # ... and Q5_K itself as an UNPINNED EXTRA (upstream format, no oracle in the reference; ggml_hip.h GGML_HIP_TYPE_Q5_K)
        "q5_k_ffn512_unpinned_extra": side_config(device, 4096, 11008, 512, copies=8, iters=60, qtype=B.Q5_K),
        # r4, beyond BASELINE's sizes: what the widened plan ranges serve -- K3p-int8 above 512 rows, K3s-int8 for Q5_0
        "q8_0_ffn1024": side_config(device, 4096, 11008, 1024, copies=6, iters=40, qtype=B.Q8_0),
        "q5_k_ffn2048_unpinned_extra": side_config(device, 4096, 11008, 2048, copies=6, iters=20, qtype=B.Q5_K),
        "q4_k_ffn512_unpinned_extra": side_config(device, 4096, 11008, 512, copies=8, iters=60, qtype=B.Q4_K),   # (r4: Q4_K, the same resident form and kernels)
        "q6_k_ffn512_unpinned_extra": side_config(device, 4096, 11008, 512, copies=6, iters=40, qtype=B.Q6_K),   # (r4: Q6_K in the planar Q4_2 form on int8 planes: the staged int8 kernel)
        "q6_k_batch1_unpinned_extra": side_config(device, 4096, 4096, 1, copies=32, iters=100, qtype=B.Q6_K),    #     ... and its decode step (its own mat-vec on the int8 planes, the Q8_K rule fused)
        "q5_k_batch1_unpinned_extra": side_config(device, 4096, 4096, 1, copies=32, iters=100, qtype=B.Q5_K),    # (r4: the k-quants' fused mat-vec -- the Q8_K rule inside the kernel)
        "q5_0_batch32": side_config(device, 4096, 4096, 32, copies=32, iters=100, qtype=B.Q5_0),
        # r5: prompt chunks whose grid of 128-row K3p tiles left CUs idle -- 64-row wave tiles on K3p's tree
        "q8_0_prompt192": side_config(device, 4096, 4096, 192, copies=16, iters=60, qtype=B.Q8_0),
        "q8_0_ffn_down256": side_config(device, 4096, 11008, 256, copies=6, iters=60, qtype=B.Q8_0),
        "vocab512": side_config(device, 32000, 4096, 512, copies=3, iters=30),
        "vocab512_shard_of_8": side_config(device, 4000, 4096, 512, copies=24, iters=100),      # what each rank of config 5's 8-GPU split computes
        # the dense case of the path (north_star: MFMA utilisation for the dense f16 / f32 mul_mat)
        "dense_f16": dense_config(device, 1, 4096, 4096, 4096, iters=20),
        "dense_f32": dense_config(device, 0, 4096, 4096, 4096, iters=5),
    }
    for k in ("batch1", "batch1_M32000"):
        oc[k]["roofline"] = {"bound": "hbm", "achieved": oc[k]["algorithmic_GBs"], "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": oc[k]["hbm_frac"]}
    oc["vocab512"]["note"] = ("config 5's TOTAL on one GPU runs the kernel form of its 4000-row shards (K split eight ways, 7.8 rounds of persistent "
                              "workgroups) so that the 8-GPU result is bit for bit the 1-GPU result; each of the 8 ranks runs the 4000 x 4096 x 512 shard")
    # the BASELINE configs' rooflines where the driver's record shows them: top level, one line each
    br = {
        "config2_batch1_Q4_0_4096x4096": dict(oc["batch1"]["roofline"], ms_per_step=oc["batch1"]["ms_per_step"]),
        "config3_prompt512_Q4_0_4096x4096x512": dict(oc["prompt512"]["roofline"], ms_per_step=oc["prompt512"]["ms_per_step"]),
        "config4_Q8_0_4096x11008x512": dict(oc["q8_0_ffn512"]["roofline"], ms_per_step=oc["q8_0_ffn512"]["ms_per_step"]),
        "config4_Q5_K_4096x11008x512_unpinned_extra": dict(oc["q5_k_ffn512_unpinned_extra"]["roofline"], ms_per_step=oc["q5_k_ffn512_unpinned_extra"]["ms_per_step"]),
        "config4_Q5_0_standin_4096x11008x512": dict(oc["q5_0_ffn512"]["roofline"], ms_per_step=oc["q5_0_ffn512"]["ms_per_step"]),
        "config5_total_on_one_gpu_Q4_0_32000x4096x512": dict(oc["vocab512"]["roofline"], ms_per_step=oc["vocab512"]["ms_per_step"]),
        "config5_shard_of_8_Q4_0_4000x4096x512": dict(oc["vocab512_shard_of_8"]["roofline"], ms_per_step=oc["vocab512_shard_of_8"]["ms_per_step"]),
    }
    # r5: one summation tree for the stage-free forms -- the family follows M: a grouped-query k / v projection at prompt size (short: K3s),
    # the output projection of a batched decode step (tall: K3p), a batch-128 decode projection (K3s), a min-term type on 16-row tiles.  Guarded one by
    # one: nothing here feeds the line, and a side measurement must not take it down.
    for name, (M_, K_, N_, copies_, iters_, qt_) in {"q8_0_kv_proj_prompt512": (1024, 4096, 512, 32, 80, B.Q8_0), "q8_0_lm_head_batch64": (32000, 4096, 64, 3, 40, B.Q8_0),
                                                     "q4_0_batch128": (4096, 4096, 128, 24, 80, B.Q4_0), "q5_1_batch16": (4096, 4096, 16, 24, 100, B.Q5_1)}.items():
        try:
            oc[name] = side_config(device, M_, K_, N_, copies=copies_, iters=iters_, qtype=qt_)
        except Exception as e:  # noqa: BLE001
            oc[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # the drop-in path with HOST tensors (PCIe-inclusive; never `value`)
    try:
        pc = pcie_probe()
        s1 = {"pcie_probe": pc,
              "seam1_host_4096x4096x4096": seam1_host_config(4096, 4096, 4096, 10),
              "seam1_host_4096x4096x512": seam1_host_config(4096, 4096, 512, 30)}
        for k in ("seam1_host_4096x4096x4096", "seam1_host_4096x4096x512"):
            c = s1[k]
            # ms: each direction at its one-way rate, and both together at the rate the link holds with both busy
            bound = max(c["host_bytes_in"] / pc["h2d_GBs"], c["host_bytes_out"] / pc["d2h_GBs"],
                        (c["host_bytes_in"] + c["host_bytes_out"]) / pc["both_ways_total_GBs"]) / 1e6
            c["pcie_bound_ms"] = round(bound, 4)
            c["step_over_pcie_bound"] = round(c["ms_per_step"] / bound, 3)
        oc.update(s1)
        oc["dropin_decode_layer_batch1"] = dropin_decode_layer(1, 200)
        # four such layers as ONE graph (what a decoder computes per token is all its layers in one graph: the fixed cost of a
        # graph compute -- launch, copies home, synchronise -- is paid once)
        oc["dropin_decode_4layers_batch1"] = dropin_decode_layer(1, 100, layers=4)
        oc["batch32_qkv_group"] = projection_group_config(device, (4096, 4096, 4096), 4096, 32)
    except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down
        oc["seam1_host_error"] = f"{type(e).__name__}: {e}"[:300]

    return oc, br
