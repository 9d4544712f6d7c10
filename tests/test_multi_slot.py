"""GPU tests (-m gpu) of the per-device contexts behind the C-ABI (SURVEY 8(b) "n_devices", 8(e)): the row split of
Ggml.cs:6665-6672 over several device SLOTS of one process, rehearsed with two / three slots on the one physical GPU
of the test box.  The bar: the split result is BIT FOR BIT the single-slot result (the exchange only moves data, the
kernel form never depends on M), through Seam 1 (host pointers, also inside a graph scope) and through the device-level
split entry; the pinned-pool pipeline, the RCCL exchange form's one-rank self test, cache invalidation."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

RNG = np.random.default_rng(77)


def _rand(shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


def assert_close(got, ref, what, K):
    """THE mul_mat tolerance (tests/oracle_lib.py: SURVEY 8(c) with its one stated floor); K = the product's inner dimension"""
    O.assert_mul_mat_close(got, ref, K, what)


@pytest.fixture()
def slots():
    """(re)initialise the library on a given slot list; always back to one slot on device 0 afterwards"""
    from ggmlsharp_amd._lib import lib, check
    L = lib()

    def set_slots(ids):
        L.ggml_hip_shutdown()
        arr = (C.c_int * len(ids))(*ids)
        check(L.ggml_hip_init_devices(len(ids), arr), "ggml_hip_init_devices")
        assert L.ggml_hip_n_slots() == len(ids)
    yield set_slots
    L.ggml_hip_shutdown()
    check(L.ggml_hip_init(0), "ggml_hip_init")


def _program(G, t, wraw, x, K, M, N, graph_chain=False, w2raw=None, M2=0):
    """Y = mul_mat(W, X) [; Y2 = mul_mat(W2, Y)] as a reference-style program; returns numpy copies of the results."""
    ctx = G.ggml_init(256 * 1024 * 1024)
    assert ctx
    try:
        W = G.ggml_new_tensor_2d(ctx, t, K, M)
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        G.tensor_bytes(W)[:] = wraw.reshape(-1).view(np.uint8)
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        Y = G.ggml_mul_mat(ctx, W, X)
        out = Y
        if graph_chain:
            W2 = G.ggml_new_tensor_2d(ctx, t, M, M2)
            G.tensor_bytes(W2)[:] = w2raw.reshape(-1).view(np.uint8)
            out = G.ggml_mul_mat(ctx, W2, G.ggml_silu(ctx, Y))
        gf = G.ggml_build_forward(out)
        G.ggml_graph_compute(ctx, gf)
        res = [G.tensor_f32(Y)[0, 0].copy()]
        if graph_chain:
            res.append(G.tensor_f32(out)[0, 0].copy())
        return res
    finally:
        G.ggml_free(ctx)


@pytest.mark.parametrize("t", [O.Q4_0, O.Q8_0, O.Q5_1, O.F16, O.F32])
def test_seam1_row_split_over_slots_is_bitwise_the_single_slot_result(slots, t):
    import ggml_mirror as G
    for (M, K, N) in ((515, 256, 40), (96, 512, 1), (992, 1024, 300)):      # (300 src1 rows: F32 runs the split-bf16 matrix-core kernel)
        w = _rand((M, K))
        x = _rand((N, K))
        wraw = w.astype(np.float16).view(np.uint16) if t == O.F16 else w if t == O.F32 else O.quantize_row(t, w)
        w2 = _rand((64, M))
        w2raw = w2.astype(np.float16).view(np.uint16) if t == O.F16 else None
        chain = t not in (O.F16, O.F32) and M % 32 == 0
        if chain:
            w2raw = O.quantize_row(t, w2)
        slots([0])
        one = _program(G, t, wraw, x, K, M, N, chain, w2raw, 64)
        ref = O.mul_mat(t, wraw, x, M, K, N, nth=3)[0, 0]
        assert_close(one[0], ref, f"single slot type {t}", K=K)
        for ids in ([0, 0], [0, 0, 0]):
            slots(ids)
            many = _program(G, t, wraw, x, K, M, N, chain, w2raw, 64)
            for a, b in zip(one, many):
                assert np.array_equal(a, b), f"type {t} {M}x{K}x{N} over {len(ids)} slots differs from one slot"


def test_split_dev_entry_matches_single_device_bitwise(slots):
    from ggmlsharp_amd import device
    from ggmlsharp_amd._lib import lib, check
    L = lib()
    M, K, N = 1030, 512, 130
    for t in (O.Q4_0, O.Q8_0):
        wq = O.quantize_row(t, _rand((M, K)))
        x = torch.from_numpy(_rand((N, K))).cuda()
        slots([0])
        W = device.Weight.from_host(t, wq, K)
        single = device.mul_mat(W, x).cpu().numpy()
        torch.cuda.synchronize()
        for G in (1, 2, 3):
            slots([0] * G)
            h = C.c_void_p()
            check(L.ggml_hip_split_weight_upload(t, wq.ctypes.data_as(C.c_void_p), K, M, wq.shape[1], C.byref(h)), "split upload")
            a, b = C.c_int64(), C.c_int64()
            check(L.ggml_hip_split_weight_rows(h, G - 1, C.byref(a), C.byref(b)), "rows")
            dr = (M + G - 1) // G
            assert (a.value, b.value) == (min(dr * (G - 1), M), M)          # Ggml.cs:6665-6672
            outs = [torch.full((N, M + 16), -5.0, device="cuda") for _ in range(G)]
            torch.cuda.synchronize()
            xs = (C.c_void_p * G)(*[x.data_ptr()] * G)
            ds = (C.c_void_p * G)(*[o.data_ptr() for o in outs])
            for mode in (0, 2):                                          # peer copies behind the products; the GEMMs' own store phase (r4)
                for o in outs:
                    o.fill_(-5.0)
                torch.cuda.synchronize()
                check(L.ggml_hip_set_exchange(mode), "set_exchange")
                check(L.ggml_hip_mul_mat_split_dev(h, xs, N, K, ds, M + 16), "split mul_mat")
                check(L.ggml_hip_sync_slots(), "sync")
                for o in outs:
                    assert np.array_equal(o[:, :M].cpu().numpy(), single), f"type {t} G {G}"
                    assert torch.all(o[:, M:] == -5.0)
            L.ggml_hip_set_exchange(0)
            L.ggml_hip_split_weight_free(h)
        # the RCCL form refuses slots that share a device instead of hanging
        slots([0, 0])
        assert L.ggml_hip_set_exchange(1) == 0
        h = C.c_void_p()
        check(L.ggml_hip_split_weight_upload(t, wq.ctypes.data_as(C.c_void_p), K, M, wq.shape[1], C.byref(h)), "split upload")
        outs = [torch.zeros((N, M), device="cuda") for _ in range(2)]
        xs = (C.c_void_p * 2)(*[x.data_ptr()] * 2)
        ds = (C.c_void_p * 2)(*[o.data_ptr() for o in outs])
        assert L.ggml_hip_mul_mat_split_dev(h, xs, N, K, ds, M) == -4
        assert b"distinct devices" in L.ggml_hip_last_error()
        L.ggml_hip_sync_slots()
        L.ggml_hip_split_weight_free(h)
        L.ggml_hip_set_exchange(0)


def test_rccl_exchange_form_one_rank_selftest(slots):
    """librccl loaded at run time, ncclCommInitAll over slot 0's device, ncclAllGather of one rank, re-layout: bytes checked
    inside the library.  (More ranks need more GPUs: the driver's multi-GPU node.)"""
    from ggmlsharp_amd._lib import lib, check
    slots([0])
    check(lib().ggml_hip_debug_rccl_selftest(), "rccl selftest")


def test_bound_threads_run_on_their_own_slot_concurrently(slots):
    """SURVEY 8(b) threading: two host threads bound to two slots run graphs at the same time; an unbound thread splits."""
    import threading
    import ggml_mirror as G
    from ggmlsharp_amd._lib import lib, check
    slots([0, 0])
    M, K, N = 256, 256, 33
    wq = O.quantize_row(O.Q4_0, _rand((M, K)))
    xs = [_rand((N, K)) for _ in range(2)]
    refs = [O.mul_mat(O.Q4_0, wq, x, M, K, N)[0, 0] for x in xs]
    got, errs = [None, None], []

    def run(i):
        try:
            check(lib().ggml_hip_bind_thread(i), "bind")
            for _ in range(5):
                got[i] = _program(G, O.Q4_0, wq, xs[i], K, M, N)[0]
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert_close(got[i], refs[i], f"thread {i}", K=K)
    assert lib().ggml_hip_bind_thread(5) == -4


def test_pinned_pool_pipeline_matches_the_oracle_and_counts_bytes(slots):
    """ggml_hip_register_host_pool (the mirror's ggml_init does it): src1 / dst move by asynchronous DMA in chunks of src1
    rows that overlap the kernels.  Same values (within the mul_mat tolerance; chunks are whole calls of their own N) and
    exactly the same PCIe byte counts as the unchunked path."""
    import ggml_mirror as G
    from ggmlsharp_amd._lib import lib
    slots([0])
    M, K, N = 512, 4096, 600            # 9.8 MB of activations: two chunks
    wq = O.quantize_row(O.Q4_0, _rand((M, K)))
    x = _rand((N, K))
    c0 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
    c1 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
    lib().ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c0])
    got = _program(G, O.Q4_0, wq, x, K, M, N)[0]
    lib().ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c1])
    ref = O.mul_mat(O.Q4_0, wq, x, M, K, N, nth=8)[0, 0]
    assert_close(got, ref, "pinned pipeline", K=K)
    assert c1[0].value - c0[0].value == N * K * 4 and c1[1].value - c0[1].value == N * M * 4
    # the same pipeline again and again on the same tensors -- same bits
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        W = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        G.tensor_bytes(W)[:] = wq.reshape(-1)
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        Y = G.ggml_mul_mat(ctx, W, X)
        gf = G.ggml_build_forward(Y)
        outs = []
        for i in range(5):
            G.tensor_f32(Y)[:] = -1.0
            G.ggml_graph_compute(ctx, gf)
            outs.append(G.tensor_f32(Y)[0, 0].copy())
        for o in outs:
            assert np.array_equal(o, got)
        # new activations in the SAME tensor
        x2 = _rand((N, K))
        G.tensor_f32(X)[:] = x2.reshape(1, 1, N, K)
        G.ggml_graph_compute(ctx, gf)
        assert_close(G.tensor_f32(Y)[0, 0], O.mul_mat(O.Q4_0, wq, x2, M, K, N, nth=8)[0, 0], "replay with new activations", K=K)
        # rewritten weights (host mirror invalidates on its own writers; a direct store needs the explicit call)
        wq2 = O.quantize_row(O.Q4_0, _rand((M, K)))
        G.tensor_bytes(W)[:] = wq2.reshape(-1)
        lib().ggml_hip_invalidate(W.contents.data)
        G.ggml_graph_compute(ctx, gf)
        assert_close(G.tensor_f32(Y)[0, 0], O.mul_mat(O.Q4_0, wq2, x2, M, K, N, nth=8)[0, 0], "after invalidate", K=K)
    finally:
        G.ggml_free(ctx)
    # pageable memory (a caller-provided, unregistered buffer): same values, no chunking
    buf = np.zeros(64 * 1024 * 1024, dtype=np.uint8)
    ctx = G.ggml_init(buf.nbytes, mem_buffer=buf.ctypes.data_as(C.c_void_p))
    try:
        lib().ggml_hip_unregister_host_pool(buf.ctypes.data_as(C.c_void_p))
        W = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        G.tensor_bytes(W)[:] = wq.reshape(-1)
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        Y = G.ggml_mul_mat(ctx, W, X)
        G.ggml_graph_compute(ctx, G.ggml_build_forward(Y))
        assert_close(G.tensor_f32(Y)[0, 0], ref, "pageable", K=K)
    finally:
        G.ggml_free(ctx)


def test_weight_cache_is_invalidated_by_writers_and_not_used_for_computed_src0(slots):
    """(1) a leaf rewritten between two computes (the reference's Test1 / Test2 pattern: set_f32, compute
    again) must not be served from the cache; (2) a src0 that a node of the graph computes must be rebuilt each time."""
    import ggml_mirror as G
    slots([0])
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        K, M, N = 64, 48, 7
        W = G.ggml_new_tensor_2d(ctx, G.F32, K, M)
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        x = _rand((N, K))
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        G.ggml_set_f32(W, 0.5)
        Y = G.ggml_mul_mat(ctx, W, X)
        gf = G.ggml_build_forward(Y)
        G.ggml_graph_compute(ctx, gf)
        assert_close(G.tensor_f32(Y)[0, 0], np.repeat(0.5 * x.astype(np.float64).sum(1, keepdims=True), M, 1), "first", K=K)
        G.ggml_set_f32(W, -2.0)                                  # rewritten leaf, same pointer and shape
        G.ggml_graph_compute(ctx, gf)
        assert_close(G.tensor_f32(Y)[0, 0], np.repeat(-2.0 * x.astype(np.float64).sum(1, keepdims=True), M, 1), "after set_f32", K=K)
        # (2) src0 = a computed tensor: A = add(P, Q) [K x M], Y2 = mul_mat(A, X); change P and recompute
        P = G.ggml_new_tensor_2d(ctx, G.F32, K, M)
        Q = G.ggml_new_tensor_2d(ctx, G.F32, K, M)
        p, q = _rand((M, K)), _rand((M, K))
        G.tensor_f32(P)[:] = p.reshape(1, 1, M, K)
        G.tensor_f32(Q)[:] = q.reshape(1, 1, M, K)
        A = G.ggml_add(ctx, P, Q)
        Y2 = G.ggml_mul_mat(ctx, A, X)
        g2 = G.ggml_build_forward(Y2)
        for trial in range(2):
            G.ggml_graph_compute(ctx, g2)
            a = (p + q).astype(np.float32)
            ref = O.mul_mat(O.F32, a, x, M, K, N)[0, 0]
            assert_close(G.tensor_f32(Y2)[0, 0], ref, f"computed src0, trial {trial}", K=K)
            p = _rand((M, K))
            G.tensor_f32(P)[:] = p.reshape(1, 1, M, K)           # direct store through the data pointer: P is not cached (src1-side operand)
    finally:
        G.ggml_free(ctx)


def test_reinit_on_another_device_set_is_refused(slots):
    from ggmlsharp_amd._lib import lib
    L = lib()
    slots([0])
    two = (C.c_int * 2)(0, 0)
    assert L.ggml_hip_init_devices(2, two) == -4 and b"shutdown" in L.ggml_hip_last_error()
    assert L.ggml_hip_init(0) == 0                               # the same device again is fine
    if L.ggml_hip_device_count() > 1:
        assert L.ggml_hip_init(1) == -4


# ---------------------------------------------------------------- G = 8 without a node
# BASELINE config 5's exact partition (Q4_0 32000 x 4096 x 512 -> eight shards of 4000 rows) and the strong-scaling partition of
# the headline (4096^3 -> eight shards of 512 rows), both on the ONE GPU of the test box: eight device slots of one process
# (the form the C# host uses), and the two exchange forms' data movement with eight ranks' worth of shards in one process.
G8_CASES = [("config 5", 32000, 4096, 512), ("headline 4096^3", 4096, 4096, 4096)]


def _q4_rows_on_device(M, K, seed):
    from ggmlsharp_amd import device
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    w = torch.randn((M, K), generator=g, device="cuda")
    x = torch.randn((512, K), generator=g, device="cuda")
    return device.quantize_rows(O.Q4_0, w), x


@pytest.mark.parametrize("name,M,K,N", G8_CASES)
def test_eight_slots_on_one_device_exact_partition_is_bitwise_the_unsplit_result(slots, name, M, K, N):
    from ggmlsharp_amd import device
    from ggmlsharp_amd._lib import lib, check
    L = lib()
    slots([0])
    rows, x512 = _q4_rows_on_device(M, K, seed=M + N)
    x = x512 if N == 512 else torch.randn((N, K), device="cuda")
    W = device.Weight.from_device(O.Q4_0, rows, K)
    single = device.mul_mat(W, x).clone()
    torch.cuda.synchronize()
    W.free()
    host_rows = rows.cpu().numpy()
    G = 8
    slots([0] * G)
    h = C.c_void_p()
    check(L.ggml_hip_split_weight_upload(O.Q4_0, host_rows.ctypes.data_as(C.c_void_p), K, M, host_rows.shape[1], C.byref(h)), "split upload")
    dr = (M + G - 1) // G
    for g in range(G):                                           # the reference's thread partition, Ggml.cs:6665-6672
        a, b = C.c_int64(), C.c_int64()
        check(L.ggml_hip_split_weight_rows(h, g, C.byref(a), C.byref(b)), "rows")
        assert (a.value, b.value) == (dr * g, min(dr * (g + 1), M)), (name, g)
    assert dr == (4000 if M == 32000 else 512)
    outs = [torch.full((N, M), -5.0, device="cuda") for _ in range(G)]
    torch.cuda.synchronize()
    xs = (C.c_void_p * G)(*[x.data_ptr()] * G)
    ds = (C.c_void_p * G)(*[o.data_ptr() for o in outs])
    # exchange form 0: peer copies behind the products (the in-process RCCL form refuses slots that share a device);
    # form 2 (r4): no exchange pass -- every slot's GEMM stores its rows into every slot's dst from its own store phase
    try:
        for mode in (0, 2):
            for o in outs:
                o.fill_(-5.0)
            torch.cuda.synchronize()
            check(L.ggml_hip_set_exchange(mode), "set_exchange")
            check(L.ggml_hip_mul_mat_split_dev(h, xs, N, K, ds, M), "split mul_mat")
            check(L.ggml_hip_sync_slots(), "sync")
            for g, o in enumerate(outs):
                assert torch.equal(o, single), f"{name}, exchange form {mode}: slot {g}'s copy of dst differs from the unsplit result"
    finally:
        L.ggml_hip_set_exchange(0)
        L.ggml_hip_split_weight_free(h)


@pytest.mark.parametrize("name,M,K,N", G8_CASES)
def test_both_exchange_forms_move_eight_ranks_shards_into_the_unsplit_result(name, M, K, N):
    """One process per GPU (ggmlsharp_amd/dist.py): rank r computes rows [dr r, dr (r + 1)) into a [N][Ms] shard; "rccl" = all-gather
    to [G][N][Ms] + ggml_hip_relayout_gathered_dev, "push" = ggml_hip_push_columns_dev into every rank's [N][M].  Here the eight
    ranks' shards are computed one after the other on the one GPU (row_begin / row_end weights: exactly what a rank holds) and both
    kernels must assemble the unsplit result bit for bit, in each of eight destination buffers."""
    from ggmlsharp_amd import device
    from ggmlsharp_amd import dist as gdist
    from ggmlsharp_amd._lib import lib, check
    device.init(0)
    L = lib()
    G = 8
    rows, x512 = _q4_rows_on_device(M, K, seed=M + N + 1)
    x = x512 if N == 512 else torch.randn((N, K), device="cuda")
    W = device.Weight.from_device(O.Q4_0, rows, K)
    full = device.mul_mat(W, x).clone()
    W.free()
    Ms = gdist.shard_width(M, G)
    gathered = torch.zeros((G, N, Ms), device="cuda")
    shards = []
    for r in range(G):
        r0, r1 = gdist.shard_rows(M, G, r)
        Wr = device.Weight.from_device(O.Q4_0, rows, K, row_begin=r0, row_end=r1)
        device.mul_mat(Wr, x, out=gathered[r, :, : r1 - r0])
        shards.append((r0, r1))
        Wr.free()
    # "rccl": what ncclAllGather leaves on every rank, then the re-layout kernel
    out = device.relayout_gathered(gathered, G, N, Ms, M)
    assert torch.equal(out, full), f"{name}: re-layout of eight gathered shards differs from the unsplit result"
    # "push": every rank stores its columns into every rank's dst
    peers = [torch.full((N, M), -3.0, device="cuda") for _ in range(G)]
    pp = (C.c_void_p * G)(*[p.data_ptr() for p in peers])
    for r, (r0, r1) in enumerate(shards):
        check(L.ggml_hip_push_columns_dev(C.c_void_p(gathered[r].data_ptr()), Ms, N, r1 - r0, pp, G, M, r0, None), "push")
    torch.cuda.synchronize()
    for r, p in enumerate(peers):
        assert torch.equal(p, full), f"{name}: rank {r}'s pushed dst differs from the unsplit result"


FUSED_PUSH_CASES = [("config 5 (K3p-MX shards)", O.Q4_0, 32000, 4096, 512, True), ("headline 4096^3 (64 x 64 MX shards)", O.Q4_0, 4096, 4096, 4096, True),
                    ("weak scaling (256 x 128 MX shards)", O.Q4_0, 32768, 2048, 1024, True), ("config 4's type (1024-row shards: K3s-int8, r5 -- on K3p's tree, with the store-phase exchange too)", O.Q8_0, 8192, 4096, 512, True),
                    ("Q5_1, ragged last shard (K3s-int8 + min term)", 7, 4000, 2048, 300, True), ("config 4's type, taller (K3p-int8 shards)", O.Q8_0, 16384, 4096, 512, True),
                    ("batched decode (K3s-MX shards, 16-row tiles)", O.Q4_0, 4096, 4096, 32, True), ("batched decode, two-scale type (K3s-int8)", 4, 8192, 4096, 24, True), ("Q4_1 above 512 rows (staged MX with the min-term MFMA)", 3, 4096, 1024, 640, True),
                    ("Q8_0, 640 rows: a family without the store-phase exchange", O.Q8_0, 4096, 1024, 640, False),
                    ("decode batch: the mat-vec", O.Q4_0, 4096, 4096, 2, False)]


@pytest.mark.parametrize("name,t,M,K,N,fused", FUSED_PUSH_CASES)
def test_exchange_fused_into_the_store_phase_fills_eight_ranks_buffers_with_the_unsplit_result(name, t, M, K, N, fused):
    """r4 (SURVEY 8(e) "epilogue peer-writes"): ggml_hip_mul_mat_push_dev computes a rank's rows and stores every
    element into EVERY rank's reference-layout dst [N][M] from the GEMM's store phase (mm_epilogue mode 3: up to eight destination
    bases).  Eight ranks' shards, one after the other on the one GPU, the eight destination buffers standing in for the peers' IPC
    mappings: each buffer must end as the unsplit product, bit for bit (the checker: the column-push kernel's path, tested above)."""
    from ggmlsharp_amd import device
    from ggmlsharp_amd import dist as gdist
    from ggmlsharp_amd._lib import lib, check
    device.init(0)
    L = lib()
    G = 8
    g = torch.Generator(device="cuda")
    g.manual_seed(M + N + t)
    rows = device.quantize_rows(t, torch.randn((M, K), generator=g, device="cuda"))
    x = torch.randn((N, K), generator=g, device="cuda")
    W = device.Weight.from_device(t, rows, K)
    full = device.mul_mat(W, x).clone()
    W.free()
    peers = [torch.full((N, M), -3.0, device="cuda") for _ in range(G)]
    pp = (C.c_void_p * G)(*[p.data_ptr() for p in peers])
    work = device.alloc_work(t, K, N)
    for r in range(G):
        r0, r1 = gdist.shard_rows(M, G, r)
        if r1 <= r0:
            continue
        Wr = device.Weight.from_device(t, rows, K, row_begin=r0, row_end=r1)
        assert bool(L.ggml_hip_mul_mat_push_fused(Wr.handle, N, G)) == fused, f"{name}: rank {r}"
        check(L.ggml_hip_mul_mat_push_dev(Wr.handle, C.c_void_p(x.data_ptr()), N, K, pp, G, r, M, r0, C.c_void_p(work.data_ptr()), work.numel(), None),
              "ggml_hip_mul_mat_push_dev")
        torch.cuda.synchronize()
        Wr.free()
    for r, p in enumerate(peers):
        assert torch.equal(p, full), f"{name}: rank {r}'s dst differs from the unsplit result"
    # argument checks: this rank's own buffer must be among the destinations; the row must hold the columns
    Wr = device.Weight.from_device(t, rows, K, row_begin=0, row_end=min(64, M))
    none = (C.c_void_p * G)(*[None] * G)
    assert L.ggml_hip_mul_mat_push_dev(Wr.handle, C.c_void_p(x.data_ptr()), N, K, none, G, 0, M, 0, C.c_void_p(work.data_ptr()), work.numel(), None) == -4
    assert L.ggml_hip_mul_mat_push_dev(Wr.handle, C.c_void_p(x.data_ptr()), N, K, pp, G, 0, M, M - 8, C.c_void_p(work.data_ptr()), work.numel(), None) == -3
    Wr.free()
