"""GPU tests (-m gpu) of the fused neighbours of mul_mat (include/ggml_hip.h "fused neighbours"): every
fused form must give, bit for bit, what the separate seams / kernels give, and both nodes' data must reach host memory."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ggml_mirror as G
from ggmlsharp_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
RNG = np.random.default_rng(99)


def _rand(shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


@pytest.fixture(scope="module")
def dev():
    from ggmlsharp_amd import device
    device.init(0)
    return device


def _params(phase=_lib.GGML_TASK_COMPUTE):
    return _lib.ggml_compute_params(phase, 0, 1, 0, None)


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q8_0])
def test_epilogue_device_form_is_bitwise_product_then_node(dev, t):
    from ggmlsharp_amd._lib import lib, check
    L = lib()
    for (M, K, N) in ((96, 256, 1), (130, 512, 3), (64, 256, 4), (200, 256, 6), (515, 512, 40), (260, 1024, 300), (128, 256, 1100),
                      (130, 2048, 12), (260, 4096 + 64, 32), (96, 2048, 6), (200, 2048, 40), (130, 2048, 64),      # (the last three: the batched-decode form, gemm_qmx.hip K3s)
                      (260, 2048, 300), (130, 2336, 512)):                                                          # (K3p, gemm_qmp.hip: the epilogue in its reduction's store phase)
        wq = O.quantize_row(t, _rand((M, K)))
        W = dev.Weight.from_host(t, wq, K)
        x = torch.from_numpy(_rand((N, K))).cuda()
        r = torch.from_numpy(_rand((N, M))).cuda()
        prod = dev.mul_mat(W, x)
        work = dev.alloc_work(t, K, N)
        fused = L.ggml_hip_mul_mat_epilogue_fused(W.handle, N)
        assert fused == (1 if (N <= 4 or ((N > 8 or K >= 2048) and t in (O.Q4_0, O.Q4_1)) or (t == O.Q8_0 and 5 <= N <= 64 and 2048 <= K <= 16384) or (t == O.Q8_0 and 256 < N <= 512 and K >= 2048)) else 0), (t, N)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        d1 = torch.full((N, M + 8), -3.0, device="cuda")
        d2 = torch.full((N, M + 4), -4.0, device="cuda")
        check(L.ggml_hip_mul_mat_epilogue_dev(W.handle, C.c_void_p(x.data_ptr()), N, K, C.c_void_p(d1.data_ptr()), M + 8, C.c_void_p(work.data_ptr()),
                                              work.numel(), 1, C.c_void_p(r.data_ptr()), M, C.c_void_p(d2.data_ptr()), M + 4, 1.0, st), "epilogue add")
        assert torch.equal(d1[:, :M], prod) and torch.all(d1[:, M:] == -3.0), (t, M, K, N)
        assert torch.equal(d2[:, :M], prod + r) and torch.all(d2[:, M:] == -4.0), (t, M, K, N)
        d3 = torch.empty((N, M), device="cuda")
        check(L.ggml_hip_mul_mat_epilogue_dev(W.handle, C.c_void_p(x.data_ptr()), N, K, C.c_void_p(d3.data_ptr()), M, C.c_void_p(work.data_ptr()),
                                              work.numel(), 2, None, 0, None, 0, 0.3125, st), "epilogue scale")
        assert torch.equal(d3, prod * np.float32(0.3125)), (t, M, K, N)


def _block_graph(ctx, t, K, M1, N, w1q, x, g, r, s):
    """out = add(mul_mat(W1, mul(rms_norm(X), Gn)), R);  sc = scale(mul_mat(W1, X), S);  gate = mul(silu(out), R)"""
    X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
    Gn = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
    R = G.ggml_new_tensor_2d(ctx, G.F32, M1, N)
    S = G.ggml_new_tensor_1d(ctx, G.F32, 1)
    W1 = G.ggml_new_tensor_2d(ctx, t, K, M1)
    G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
    G.tensor_f32(Gn)[:] = g.reshape(1, 1, N, K)
    G.tensor_f32(R)[:] = r.reshape(1, 1, N, M1)
    G.tensor_f32(S)[:] = s
    G.tensor_bytes(W1)[:] = w1q.reshape(-1)
    t_norm = G.ggml_rms_norm(ctx, X)
    t_mul = G.ggml_mul(ctx, t_norm, Gn)
    t_mm = G.ggml_mul_mat(ctx, W1, t_mul)
    t_add = G.ggml_add(ctx, t_mm, R)
    t_mm2 = G.ggml_mul_mat(ctx, W1, X)
    t_sc = G.ggml_scale(ctx, t_mm2, S)
    t_silu = G.ggml_silu(ctx, t_add)
    t_gate = G.ggml_mul(ctx, t_silu, R)
    t_fin = G.ggml_add(ctx, t_gate, t_sc)
    return [t_norm, t_mul, t_mm, t_add, t_mm2, t_sc, t_silu, t_gate, t_fin], t_fin


@pytest.mark.parametrize("N", [1, 3, 6, 40, 300])
def test_fused_graph_equals_the_node_by_node_seams_bitwise(dev, N):
    """ggml_graph_compute takes the fused seams for (rms_norm, mul), (mul_mat, add), (mul_mat, scale), (silu, mul); the same
    graph run node by node through the unfused seams gives identical bytes for EVERY node, and the values match the oracle
    chain (element-wise nodes bit-exact given their inputs)."""
    L = _lib.lib()
    K, M1 = 256, 128
    t = O.Q4_0
    w1q = O.quantize_row(t, _rand((M1, K)))
    x, g, r = _rand((N, K), 3.0), _rand((N, K)), _rand((N, M1))
    res = {}
    for mode in ("fused", "unfused"):
        ctx = G.ggml_init(64 * 1024 * 1024)
        try:
            nodes, fin = _block_graph(ctx, t, K, M1, N, w1q, x, g, r, 0.125)
            gf = G.ggml_build_forward(fin)
            assert gf.n_nodes == 9
            order = [gf.nodes[i] for i in range(gf.n_nodes)]
            if mode == "fused":
                G.ggml_graph_compute(ctx, gf)
            else:
                p = _params()
                _lib.check(L.ggml_hip_graph_begin(), "begin")
                for nd in order:
                    c = nd.contents
                    op = c.op
                    if op == _lib.GGML_OP_MUL_MAT:
                        rc = L.ggml_hip_compute_forward_mul_mat(C.byref(p), c.src0, c.src1, nd)
                    elif op == _lib.GGML_OP_ADD:
                        rc = L.ggml_hip_compute_forward_add(C.byref(p), c.src0, c.src1, nd)
                    elif op == 4:
                        rc = L.ggml_hip_compute_forward_mul(C.byref(p), c.src0, c.src1, nd)
                    elif op == 21:
                        rc = L.ggml_hip_compute_forward_scale(C.byref(p), c.src0, c.src1, nd)
                    elif op == 19:
                        rc = L.ggml_hip_compute_forward_rms_norm(C.byref(p), c.src0, nd)
                    elif op == _lib.GGML_OP_SILU:
                        rc = L.ggml_hip_compute_forward_silu(C.byref(p), c.src0, nd)
                    else:
                        raise AssertionError(op)
                    _lib.check(rc, f"op {op}")
                _lib.check(L.ggml_hip_graph_end(), "end")
            res[mode] = [G.tensor_f32(nd)[0, 0].copy() for nd in nodes]
        finally:
            G.ggml_free(ctx)
    for a, b, name in zip(res["fused"], res["unfused"], ["norm", "mul", "mm", "add", "mm2", "scale", "silu", "gate", "fin"]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"node {name} differs between the fused and the unfused path (N = {N})"
    norm, mul, mm, add, mm2, sc, silu, gate, fin = res["fused"]
    assert np.allclose(norm, O.eltwise("rms_norm", x), rtol=2e-7, atol=0)
    assert np.array_equal(mul, O.eltwise("mul", norm, g))
    ref_mm = O.mul_mat(t, w1q, np.ascontiguousarray(mul), M1, K, N)[0, 0]
    O.assert_mul_mat_close(mm, ref_mm, K, "mul_mat behind the fused prologue")
    assert np.array_equal(add, O.eltwise("add", mm, r))
    assert np.array_equal(mm2, sc)                                              # the scale node is a view: both hold product * s
    ref_mm2 = O.mul_mat(t, w1q, x, M1, K, N)[0, 0]
    O.assert_mul_mat_close(sc, 0.125 * ref_mm2.astype(np.float64), K, "mul_mat with the scale epilogue")
    assert np.array_equal(silu, O.eltwise("silu", add))
    assert np.array_equal(gate, O.eltwise("mul", silu, r))
    assert np.array_equal(fin, O.eltwise("add", gate, sc))


def test_fused_seams_ignore_other_phases_and_threads(dev):
    L = _lib.lib()
    ctx = G.ggml_init(4 * 1024 * 1024)
    try:
        a = G.ggml_new_tensor_2d(ctx, G.F32, 64, 3)
        b = G.ggml_new_tensor_2d(ctx, G.F32, 64, 3)
        n = G.ggml_rms_norm(ctx, a)
        y = G.ggml_mul(ctx, n, b)
        G.ggml_set_f32(a, 1.0)
        G.ggml_set_f32(b, 2.0)
        G.tensor_f32(n)[:] = -9.0
        G.tensor_f32(y)[:] = -9.0
        for phase, ith in ((_lib.GGML_TASK_INIT, 0), (_lib.GGML_TASK_FINALIZE, 0), (_lib.GGML_TASK_COMPUTE, 1)):
            p = _lib.ggml_compute_params(phase, ith, 2, 0, None)
            assert L.ggml_hip_compute_forward_rms_norm_mul(C.byref(p), a, b, n, y) == 0
            assert np.all(G.tensor_f32(y) == -9.0)
        p = _params()
        assert L.ggml_hip_compute_forward_rms_norm_mul(C.byref(p), a, b, n, y) == 0
        assert np.allclose(G.tensor_f32(n), 1.0, rtol=1e-6) and np.allclose(G.tensor_f32(y), 2.0, rtol=1e-6)
        bad = G.ggml_new_tensor_2d(ctx, G.F32, 32, 3)
        assert L.ggml_hip_compute_forward_silu_mul(C.byref(p), a, bad, n, y) == _lib.ERR_SHAPE
    finally:
        G.ggml_free(ctx)


def test_chain_device_form_one_launch_for_decode_batches(dev):
    """ggml_hip_norm_mul_mat_dev: rms_norm -> mul -> mul_mat -> add on device rows; N <= 4 is one launch (prologue + epilogue of
    the fused mat-vec), larger N the pair kernel + the mat-mul: the same bits either way, equal to the separate steps."""
    from ggmlsharp_amd._lib import lib, check
    L = lib()
    M, K = 200, 512
    for t in (O.Q4_0, O.Q5_1, O.Q8_0):
        wq = O.quantize_row(t, _rand((M, K)))
        W = dev.Weight.from_host(t, wq, K)
        for N in (1, 2, 3, 4, 6, 40):
            assert L.ggml_hip_norm_mul_mat_fused(W.handle, N) == (1 if N <= 4 else 0)
            x = torch.from_numpy(_rand((N, K), 2.0)).cuda()
            g = torch.from_numpy(_rand((N, K))).cuda()
            r = torch.from_numpy(_rand((N, M))).cuda()
            nrm = torch.empty((N, K), device="cuda")
            y = torch.empty((N, K), device="cuda")
            d1 = torch.empty((N, M), device="cuda")
            d2 = torch.empty((N, M), device="cuda")
            work = dev.alloc_work(t, K, N)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            check(L.ggml_hip_norm_mul_mat_dev(W.handle, C.c_void_p(x.data_ptr()), K, C.c_void_p(g.data_ptr()), K, N, C.c_void_p(nrm.data_ptr()),
                                              C.c_void_p(y.data_ptr()), C.c_void_p(d1.data_ptr()), M, C.c_void_p(work.data_ptr()), work.numel(), 1,
                                              C.c_void_p(r.data_ptr()), M, C.c_void_p(d2.data_ptr()), M, 1.0, st), "norm_mul_mat")
            want_n = O.eltwise("rms_norm", x.cpu().numpy())
            assert np.allclose(nrm.cpu().numpy(), want_n, rtol=2e-7, atol=0)
            assert np.array_equal(y.cpu().numpy(), O.eltwise("mul", nrm.cpu().numpy(), g.cpu().numpy()))
            prod = dev.mul_mat(W, y)                       # the separate mul_mat on the same y
            assert torch.equal(d1, prod), (t, N)
            assert torch.equal(d2, prod + r), (t, N)


@pytest.mark.parametrize("t", [O.Q4_0, O.Q5_1, O.Q8_0])
def test_multi_weight_device_form_is_bitwise_the_single_calls(dev, t):
    """2..4 weight matrices behind one activation matrix in one launch (ggml_hip_mul_mat_multi_dev), with and without the
    rms_norm -> mul prologue: every dst equals the single-matrix call's, the norm / mul nodes equal the pair kernel's."""
    from ggmlsharp_amd._lib import lib, check
    L = lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (Ms, K, N) in (((96, 130), 256, 1), ((64, 64, 200), 512, 3), ((33, 500, 16, 129), 256, 4), ((4096, 4096, 4096), 4096, 1), ((300, 40), 4352, 2)):
        Ws = [dev.Weight.from_host(t, O.quantize_row(t, _rand((M, K))), K) for M in Ms]
        x = torch.from_numpy(_rand((N, K), 2.0)).cuda()
        g = torch.from_numpy(_rand((N, K))).cuda()
        hw = (C.c_void_p * len(Ws))(*[w.handle for w in Ws])
        assert L.ggml_hip_mul_mat_multi_fused(hw, len(Ws), N) == 1
        assert L.ggml_hip_mul_mat_multi_fused(hw, len(Ws), 5) == 0
        singles = [dev.mul_mat(w, x) for w in Ws]
        outs = [torch.full((N, M + 4), -2.0, device="cuda") for M in Ms]
        dp = (C.c_void_p * len(Ws))(*[o.data_ptr() for o in outs])
        ld = (C.c_int64 * len(Ws))(*[M + 4 for M in Ms])
        check(L.ggml_hip_mul_mat_multi_dev(hw, len(Ws), C.c_void_p(x.data_ptr()), K, N, dp, ld, None, 0, None, None, st), "multi")
        for o, s, M in zip(outs, singles, Ms):
            assert torch.equal(o[:, :M], s) and torch.all(o[:, M:] == -2.0), (t, Ms, K, N)
        # with the prologue
        nrm, y = torch.empty((N, K), device="cuda"), torch.empty((N, K), device="cuda")
        n_ref, y_ref = torch.empty((N, K), device="cuda"), torch.empty((N, K), device="cuda")
        check(L.ggml_hip_rms_norm_mul_rows_dev(C.c_void_p(x.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(n_ref.data_ptr()), C.c_void_p(y_ref.data_ptr()),
                                               N, K, st), "pair")
        singles = [dev.mul_mat(w, y_ref) for w in Ws]
        for o in outs:
            o.fill_(-2.0)
        check(L.ggml_hip_mul_mat_multi_dev(hw, len(Ws), C.c_void_p(x.data_ptr()), K, N, dp, ld, C.c_void_p(g.data_ptr()), K, C.c_void_p(nrm.data_ptr()),
                                           C.c_void_p(y.data_ptr()), st), "multi + prologue")
        assert torch.equal(nrm, n_ref) and torch.equal(y, y_ref), (t, Ms, K, N)
        for o, s, M in zip(outs, singles, Ms):
            assert torch.equal(o[:, :M], s) and torch.all(o[:, M:] == -2.0), (t, Ms, K, N, "prologue")
        for w in Ws:
            w.free()


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q8_0])
def test_multi_weight_batch_form_is_bitwise_the_single_calls(dev, t):
    """ggml_hip_mul_mat_multi_work_dev: 1..4 matrices behind ONE quantization of src1 for a batch of any size -- one launch where
    gemm_qmx.hip / gemm_q8s.hip have the form (5..64 rows, Q4_0 / Q4_1 / Q8_0, K >= 2048: one / two / three / four tiles per workgroup by the tiles of all the
    matrices together), else one COMPUTE after the other.  Every dst equals the single-matrix call's, bit for bit."""
    from ggmlsharp_amd._lib import lib, check
    L = lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # (r5: three tiles per workgroup where that keeps the group inside one round -- gate / up of a 7B model, ragged rows that overhang the last 96-row tile -- and four beyond)
    for (Ms, K, N) in (((11008, 11008), 4096, 16), ((8192 + 40, 9000 + 7), 2048, 24), ((12000, 12000, 9000), 2048, 32), ((96, 130), 2048, 12), ((4096, 4096, 4096), 2048, 32), ((300, 8200, 40), 2048 + 64, 9), ((8192, 8192 + 300), 2048, 20),
                       ((64, 200), 512, 20), ((130, 70, 33), 2048, 3), ((100, 60), 2048, 7), ((128, 96), 4096 + 128, 70), ((77,), 2048, 16), ((4096, 300, 4096), 2048, 50), ((8192 + 40, 200), 2048 + 64, 64)):
        Ws = [dev.Weight.from_host(t, O.quantize_row(t, _rand((M, K))), K) for M in Ms]
        x = torch.from_numpy(_rand((N, K), 2.0)).cuda()
        hw = (C.c_void_p * len(Ws))(*[w.handle for w in Ws])
        singles = [dev.mul_mat(w, x) for w in Ws]
        outs = [torch.full((N, M + 4), -2.0, device="cuda") for M in Ms]
        dp = (C.c_void_p * len(Ws))(*[o.data_ptr() for o in outs])
        ld = (C.c_int64 * len(Ws))(*[M + 4 for M in Ms])
        work = dev.alloc_work(t, K, N)
        check(L.ggml_hip_mul_mat_multi_work_dev(hw, len(Ws), C.c_void_p(x.data_ptr()), K, N, dp, ld, C.c_void_p(work.data_ptr()), work.numel(), st),
              "multi with work")
        for o, s, M in zip(outs, singles, Ms):
            assert torch.equal(o[:, :M], s) and torch.all(o[:, M:] == -2.0), (t, Ms, K, N)
        for w in Ws:
            w.free()


@pytest.mark.parametrize("N", [6, 12, 32, 50])
def test_projection_groups_of_a_batch_go_down_together(dev, N):
    """q / k / v and gate / up of a batched decoder's step through ggml_graph_compute (K = 2048: the one-launch form) against the
    same nodes one by one through the single seams: identical bytes for every node, over an observed, a captured and two
    replayed computes."""
    rng = np.random.default_rng(40 + N)
    K, M, F = 2048, 96, 160
    ctx = G.ggml_init(128 * 1024 * 1024)
    try:
        def qw(t, k, m):
            w = G.ggml_new_tensor_2d(ctx, t, k, m)
            G.tensor_bytes(w)[:] = O.quantize_row(t, (rng.standard_normal((m, k)) * 0.3).astype(np.float32)).reshape(-1)
            return w
        x = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        g1 = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        G.tensor_f32(g1)[:] = rng.standard_normal((N, K)).astype(np.float32).reshape(1, 1, N, K)
        cur = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, x), g1)
        q, k, v = (G.ggml_mul_mat(ctx, qw(G.Q4_0, K, M), cur) for _ in range(3))
        a = G.ggml_add(ctx, G.ggml_add(ctx, q, k), v)
        u, gt = G.ggml_mul_mat(ctx, qw(G.Q4_1, K, F), x), G.ggml_mul_mat(ctx, qw(G.Q4_1, K, F), x)
        s = G.ggml_mul(ctx, G.ggml_silu(ctx, u), gt)
        out = G.ggml_add(ctx, G.ggml_mul_mat(ctx, qw(G.Q4_0, F, M), s), a)
        gf = G.ggml_build_forward(out)
        L = _lib.lib()
        p = _params()

        def one_by_one():
            for i in range(gf.n_nodes):
                n = gf.nodes[i].contents
                if n.op == _lib.GGML_OP_MUL_MAT:
                    rc = L.ggml_hip_compute_forward_mul_mat(C.byref(p), n.src0, n.src1, gf.nodes[i])
                elif n.op == _lib.GGML_OP_ADD:
                    rc = L.ggml_hip_compute_forward_add(C.byref(p), n.src0, n.src1, gf.nodes[i])
                elif n.op == _lib.GGML_OP_MUL:
                    rc = L.ggml_hip_compute_forward_mul(C.byref(p), n.src0, n.src1, gf.nodes[i])
                elif n.op == _lib.GGML_OP_RMS_NORM:
                    rc = L.ggml_hip_compute_forward_rms_norm(C.byref(p), n.src0, gf.nodes[i])
                else:
                    rc = L.ggml_hip_compute_forward_silu(C.byref(p), n.src0, gf.nodes[i])
                _lib.check(rc, f"node {i}")
        for it in range(4):
            G.tensor_f32(x)[:] = (rng.standard_normal((N, K)) * (1 + it)).astype(np.float32).reshape(1, 1, N, K)
            L.ggml_hip_invalidate_range(x.contents.data, N * K * 4)
            G.ggml_graph_compute(ctx, gf)
            got = [np.array(G.tensor_f32(gf.nodes[i]), copy=True) for i in range(gf.n_nodes)]
            one_by_one()
            for i in range(gf.n_nodes):
                assert np.array_equal(got[i].view(np.uint32), G.tensor_f32(gf.nodes[i]).view(np.uint32)), (N, it, i, gf.nodes[i].contents.op)
    finally:
        G.ggml_free(ctx)
