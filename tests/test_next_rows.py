"""SURVEY.md 8(f) "next" rows: ggml_cpy f32/f16 -> quantized (the only public way to make a Q tensor in the reference)
and ggml_add with a quantized src0 (add_q_f32).  CPU: the oracle composes its own row functions; host-mirror node
construction.  GPU: the HIP path through the C-ABI, bit-exact against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ggmlsharp_amd import _lib
import ggml_mirror as G

RNG = np.random.default_rng(77)
QT = [O.Q4_0, O.Q4_1, O.Q5_0, O.Q8_0, O.Q4_2, O.Q5_1]


def _rand(shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


# ------------------------------------------------------------------ CPU: oracle
@pytest.mark.parametrize("t", QT)
def test_oracle_cpy_is_rowwise_quantize(t):
    x = _rand((2, 3, 5, 64))
    got = O.cpy_to_q(t, x)
    assert np.array_equal(got, O.quantize_row(t, x))
    xh = x.astype(np.float16)
    got_h = O.cpy_to_q(t, xh)
    assert np.array_equal(got_h, O.quantize_row(t, xh.astype(np.float32)))   # f16 widened exactly, then quantized


@pytest.mark.parametrize("t", QT)
def test_oracle_add_q_f32_is_dequant_add_requant(t):
    w = _rand((4, 96), 2.0)
    b = O.quantize_row(t, w)
    x = _rand((4, 96), 0.5)
    got = O.add_q_f32(t, b, x)
    want = O.quantize_row(t, (O.dequantize_row(t, b, 96) + x).astype(np.float32))
    assert np.array_equal(got, want)
    # adding zero to a Q8_0 row is the identity on its values (quantize . dequantize fixed point)
    if t == O.Q8_0:
        same = O.add_q_f32(t, b, np.zeros_like(x))
        assert np.array_equal(O.dequantize_row(t, same, 96), O.dequantize_row(t, b, 96))


def test_host_mirror_cpy_and_add_nodes():
    ctx = G.ggml_init(4 * 1024 * 1024)
    try:
        a = G.ggml_new_tensor_2d(ctx, G.F32, 64, 6)
        b = G.ggml_new_tensor_2d(ctx, G.Q4_0, 64, 6)
        c = G.ggml_cpy(ctx, a, b)
        cc = c.contents
        assert cc.op == _lib.GGML_OP_CPY and cc.type == G.Q4_0 and cc.data == b.contents.data      # a view of b (Ggml.cs:8292)
        assert (cc.nb[0], cc.nb[1]) == (20, 40)
        assert C.addressof(cc.src0.contents) == C.addressof(a.contents) and C.addressof(cc.src1.contents) == C.addressof(b.contents)
        assert not G.ggml_cpy(ctx, a, G.ggml_new_tensor_2d(ctx, G.Q4_0, 64, 5))                       # element counts differ
        x = G.ggml_new_tensor_2d(ctx, G.F32, 64, 6)
        s = G.ggml_add(ctx, b, x)
        sc = s.contents
        assert sc.op == _lib.GGML_OP_ADD and sc.type == G.Q4_0 and sc.data != b.contents.data        # a dup of a (Ggml.cs:7883)
        assert not G.ggml_add(ctx, b, G.ggml_new_tensor_2d(ctx, G.F32, 64, 5))
        gf = G.ggml_build_forward(G.ggml_mul_mat(ctx, s, x))
        assert (gf.n_nodes, gf.n_leafs) == (2, 2)                                                     # add, mul_mat ; b, x
    finally:
        G.ggml_free(ctx)


# ------------------------------------------------------------------ GPU: HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("t", QT)
def test_device_quantize_from_f16_and_strided_rows(t):
    torch = pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)
    x = _rand((9, 256))
    big = torch.zeros((9, 320), dtype=torch.float32, device="cuda")
    big[:, :256] = torch.from_numpy(x).cuda()
    got = device.quantize_rows_from(t, big[:, :256]).cpu().numpy()            # row stride 320 elements
    assert np.array_equal(got, O.quantize_row(t, x))
    xh = x.astype(np.float16)
    got_h = device.quantize_rows_from(t, torch.from_numpy(xh).cuda()).cpu().numpy()
    assert np.array_equal(got_h, O.cpy_to_q(t, xh))


@pytest.mark.gpu
@pytest.mark.parametrize("t", QT)
def test_device_add_q_f32_bit_exact(t):
    torch = pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)
    w = _rand((33, 512), 3.0)
    x = _rand((33, 512))
    x[0, :32] = 0.0
    b = O.quantize_row(t, w)
    got = device.add_q_f32_rows(t, torch.from_numpy(b).cuda(), torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, O.add_q_f32(t, b, x))


@pytest.mark.gpu
def test_ggml_program_cpy_then_mul_mat_then_add():
    """Reference-style program: make Q weights with ggml_cpy (f32 -> Q4_0 and f16 -> Q8_0), multiply, then add_q_f32."""
    pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)
    K, M, N = 128, 40, 6
    ctx = G.ggml_init(32 * 1024 * 1024)
    try:
        for src_t, dst_t, npdt in ((G.F32, G.Q4_0, np.float32), (G.F16, G.Q8_0, np.float16)):
            w = _rand((M, K)).astype(npdt)
            Wf = G.ggml_new_tensor_2d(ctx, src_t, K, M)
            G.tensor_bytes(Wf)[:] = w.view(np.uint8).reshape(-1)
            Wq = G.ggml_new_tensor_2d(ctx, dst_t, K, M)
            X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
            x = _rand((N, K))
            G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
            cp = G.ggml_cpy(ctx, Wf, Wq)
            gf = G.ggml_build_forward(cp)
            G.ggml_graph_compute(ctx, gf)
            want_q = O.cpy_to_q(dst_t, w)
            assert np.array_equal(G.tensor_bytes(Wq), want_q.reshape(-1))            # bit-exact quantized weights
            Y = G.ggml_mul_mat(ctx, Wq, X)
            G.ggml_graph_compute(ctx, G.ggml_build_forward(Y))
            ref = O.mul_mat(dst_t, want_q, x, M, K, N)[0, 0]
            got = G.tensor_f32(Y)[0, 0]
            O.assert_mul_mat_close(got, ref, K, "mul_mat on cpy-quantized weights")
            # add_q_f32: Wq + delta -> new quantized tensor
            D = G.ggml_new_tensor_2d(ctx, G.F32, K, M)
            delta = _rand((M, K), 0.3)
            G.tensor_f32(D)[:] = delta.reshape(1, 1, M, K)
            S = G.ggml_add(ctx, Wq, D)
            G.ggml_graph_compute(ctx, G.ggml_build_forward(S))
            assert np.array_equal(G.tensor_bytes(S), O.add_q_f32(dst_t, want_q, delta).reshape(-1))
            # rewriting the weights through cpy invalidates the cached device copy: the next mul_mat sees new data
            w2 = _rand((M, K)).astype(npdt)
            G.tensor_bytes(Wf)[:] = w2.view(np.uint8).reshape(-1)
            G.ggml_graph_compute(ctx, gf)
            G.ggml_graph_compute(ctx, G.ggml_build_forward(Y))
            ref2 = O.mul_mat(dst_t, O.cpy_to_q(dst_t, w2), x, M, K, N)[0, 0]
            got2 = G.tensor_f32(Y)[0, 0]
            O.assert_mul_mat_close(got2, ref2, K, "mul_mat after the weights were rewritten")
    finally:
        G.ggml_free(ctx)


@pytest.mark.gpu
def test_silu_bit_exact_for_every_half_and_swiglu_graph():
    """silu (Ggml.cs:5705-5748, half-table form): all 63488 finite halves through the device kernel, bit for bit; then the
    gate of a LLaMA-style FFN, out = mul_mat(W2, mul(silu(mul_mat(W1, x)), mul_mat(W3, x))), as one graph."""
    pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)
    h = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    f = h[np.isfinite(h)].astype(np.float32)
    n = f.size
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        A = G.ggml_new_tensor_1d(ctx, G.F32, n)
        G.tensor_f32(A)[:] = f.reshape(1, 1, 1, n)
        Sn = G.ggml_silu(ctx, A)
        assert Sn and Sn.contents.op == _lib.GGML_OP_SILU and Sn.contents.data != A.contents.data
        G.ggml_graph_compute(ctx, G.ggml_build_forward(Sn))
        got = G.tensor_f32(Sn).reshape(-1)
        want = O.eltwise("silu", f.reshape(1, -1)).reshape(-1)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # in-place form: the result is a view of its argument
        B = G.ggml_new_tensor_2d(ctx, G.F32, 96, 7)
        xb = _rand((7, 96), 3.0)
        G.tensor_f32(B)[:] = xb.reshape(1, 1, 7, 96)
        Si = G.ggml_silu_inplace(ctx, B)
        assert Si.contents.data == B.contents.data
        G.ggml_graph_compute(ctx, G.ggml_build_forward(Si))
        assert np.array_equal(G.tensor_f32(B)[0, 0], O.eltwise("silu", xb))
        # SwiGLU gate
        K, F, N = 128, 256, 9
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        W1 = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, F)
        W3 = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, F)
        W2 = G.ggml_new_tensor_2d(ctx, G.Q4_0, F, K)
        x = _rand((N, K))
        w1, w3, w2 = (O.quantize_row(O.Q4_0, _rand(s)) for s in ((F, K), (F, K), (K, F)))
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        for T, w in ((W1, w1), (W3, w3), (W2, w2)):
            G.tensor_bytes(T)[:] = w.reshape(-1)
        gate = G.ggml_silu(ctx, G.ggml_mul_mat(ctx, W1, X))
        up = G.ggml_mul_mat(ctx, W3, X)
        out = G.ggml_mul_mat(ctx, W2, G.ggml_mul(ctx, gate, up))
        gf = G.ggml_build_forward(out)
        assert gf.n_nodes == 5
        G.ggml_graph_compute(ctx, gf)
        got_gate = G.tensor_f32(gate)[0, 0]
        h1 = np.ascontiguousarray(G.tensor_f32(gate.contents.src0)[0, 0])
        assert np.array_equal(got_gate, O.eltwise("silu", h1))                       # bit-exact given its input
        ref_h1 = O.mul_mat(O.Q4_0, w1, x, F, K, N)[0, 0]
        O.assert_mul_mat_close(h1, ref_h1, K, "gate projection")
        prod = O.eltwise("mul", got_gate, np.ascontiguousarray(G.tensor_f32(up)[0, 0]))
        ref_out = O.mul_mat(O.Q4_0, w2, prod, K, F, N)[0, 0]
        O.assert_mul_mat_close(G.tensor_f32(out)[0, 0], ref_out, F, "down projection")
    finally:
        G.ggml_free(ctx)


# ---------------------------------------------------------------- row 4: f32 neighbours of mul_mat, chained on the device
@pytest.mark.gpu
def test_transformer_style_chain_stays_on_device():
    """cur = rms_norm(x); cur = mul(cur, g); y = mul_mat(W1, cur); y = scale(y, s) [in place]; out = add(mul_mat(W2, y), r)
    in ONE graph: every intermediate is consumed from HBM, every node's data is in host memory afterwards, add / mul / scale
    bit-exact against the oracle chain, rms_norm and mul_mat within their documented tolerances."""
    pytest.importorskip("torch")
    from ggmlsharp_amd import device
    from ggmlsharp_amd._lib import lib
    device.init(0)

    def assert_close(got, ref, what, K):     # THE mul_mat tolerance (tests/oracle_lib.py)
        O.assert_mul_mat_close(got, ref, K, what)

    K, M1, M2, N = 256, 128, 64, 20
    rng = np.random.default_rng(5)
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        Gn = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        W1 = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M1)
        W2 = G.ggml_new_tensor_2d(ctx, G.Q8_0, M1, M2)
        S = G.ggml_new_tensor_1d(ctx, G.F32, 1)
        R = G.ggml_new_tensor_2d(ctx, G.F32, M2, N)
        x = rng.standard_normal((N, K)).astype(np.float32) * 3
        g = rng.standard_normal((N, K)).astype(np.float32)
        r = rng.standard_normal((N, M2)).astype(np.float32)
        w1q = O.quantize_row(O.Q4_0, rng.standard_normal((M1, K)).astype(np.float32))
        w2q = O.quantize_row(O.Q8_0, rng.standard_normal((M2, M1)).astype(np.float32))
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        G.tensor_f32(Gn)[:] = g.reshape(1, 1, N, K)
        G.tensor_f32(R)[:] = r.reshape(1, 1, N, M2)
        G.tensor_f32(S)[:] = 0.125
        G.tensor_bytes(W1)[:] = w1q.reshape(-1)
        G.tensor_bytes(W2)[:] = w2q.reshape(-1)
        t_norm = G.ggml_rms_norm(ctx, X)
        t_mul = G.ggml_mul(ctx, t_norm, Gn)
        t_y1 = G.ggml_mul_mat(ctx, W1, t_mul)
        t_sc = G.ggml_scale(ctx, t_y1, S)                   # a view of t_y1: in place
        t_y2 = G.ggml_mul_mat(ctx, W2, t_sc)
        t_out = G.ggml_add(ctx, t_y2, R)
        assert t_norm and t_mul and t_y1 and t_sc and t_y2 and t_out
        assert t_sc.contents.data == t_y1.contents.data
        gf = G.ggml_build_forward(t_out)
        assert gf.n_nodes == 6
        c0 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
        c1 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
        lib().ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c0])
        G.ggml_graph_compute(ctx, gf)
        lib().ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c1])
        # oracle chain
        o_norm = O.eltwise("rms_norm", x)
        got_norm = G.tensor_f32(t_norm)[0, 0]
        assert np.allclose(got_norm, o_norm, rtol=2e-7, atol=0)                 # f64 summation order only
        o_mul = O.eltwise("mul", got_norm, g)
        assert np.array_equal(G.tensor_f32(t_mul)[0, 0], o_mul)                 # bit-exact given its input
        o_y1 = O.mul_mat(O.Q4_0, w1q, o_mul, M1, K, N, nth=2)[0, 0]
        o_sc = O.eltwise("scale", o_y1, v=0.125)
        got_sc = G.tensor_f32(t_sc)[0, 0]
        assert_close(got_sc, o_sc, "scale(mul_mat)", K)
        o_y2 = O.mul_mat(O.Q8_0, w2q, np.ascontiguousarray(got_sc), M2, M1, N, nth=2)[0, 0]
        got_y2 = G.tensor_f32(t_y2)[0, 0]
        assert_close(got_y2, o_y2, "second mul_mat", M1)
        assert np.array_equal(G.tensor_f32(t_out)[0, 0], O.eltwise("add", got_y2, r))   # bit-exact given its input
        # PCIe: only the four leaf activations go down (X, Gn, R; S is read on the host), every intermediate is a hit
        assert c1[0].value - c0[0].value == (2 * N * K + N * M2) * 4
        # (the pairs (rms_norm, mul), (mul_mat, scale), (mul_mat, add) are fused calls: what is left to look up are the two
        # src1 operands of the mul_mats, both found resident)
        assert c1[2].value - c0[2].value >= 2
    finally:
        G.ggml_free(ctx)


def test_host_mirror_silu_nodes():
    """ggml_silu / ggml_silu_inplace node construction (Ggml.cs:8154-8174): dup vs view, op id, graph order."""
    ctx = G.ggml_init(4 * 1024 * 1024)
    try:
        a = G.ggml_new_tensor_2d(ctx, G.F32, 64, 3)
        s = G.ggml_silu(ctx, a)
        si = G.ggml_silu_inplace(ctx, a)
        assert s.contents.op == _lib.GGML_OP_SILU and si.contents.op == _lib.GGML_OP_SILU
        assert s.contents.data != a.contents.data and si.contents.data == a.contents.data          # dup vs view (:8166)
        assert C.addressof(s.contents.src0.contents) == C.addressof(a.contents) and not s.contents.src1
        assert tuple(s.contents.ne) == tuple(a.contents.ne) and tuple(s.contents.nb) == tuple(a.contents.nb)
        gf = G.ggml_build_forward(G.ggml_mul(ctx, s, G.ggml_silu(ctx, s)))
        assert (gf.n_nodes, gf.n_leafs) == (3, 1)                                                   # silu, silu, mul ; a
    finally:
        G.ggml_free(ctx)
