"""Q5_K as an UNPINNED EXTRA (include/ggml_hip.h GGML_HIP_TYPE_Q5_K; ggmlsharp_amd/csrc/kquants.hip).  The reference has no
k-quants and there is no oracle for them: the checker is tests/np_kquants.py, a numpy restatement of the published upstream
format.  CPU tests: the restatement is self-consistent.  GPU tests: the device path against it -- dequantize, upload /
download and the Q8_K activation image bit-exact, mul_mat within the path's tolerance."""
import ctypes as C

import numpy as np
import pytest

import np_kquants as KQ
import oracle_lib as O

RNG = np.random.default_rng(4242)
Q5_K, Q4_K, Q6_K = 113, 112, 114
# per type: super-block bytes, the test encoder, the restated dequantize and mul_mat
KQT = {Q5_K: (176, KQ.quantize_q5_K, KQ.dequantize_q5_K, KQ.mul_mat_q5_K), Q4_K: (144, KQ.quantize_q4_K, KQ.dequantize_q4_K, KQ.mul_mat_q4_K),
       Q6_K: (210, KQ.quantize_q6_K, KQ.dequantize_q6_K, KQ.mul_mat_q6_K)}
# the restated quantizer the device quantizer follows
REFQ = {Q5_K: lambda x: KQ.quantize_kq_reference(x, 5), Q4_K: lambda x: KQ.quantize_kq_reference(x, 4), Q6_K: KQ.quantize_q6_K}
ALL_KQ = [Q5_K, Q4_K, Q6_K]


def _rand(shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


def _random_blocks(nb, t=Q5_K):
    """raw super-blocks: every bit pattern of scales / qh / qs, finite half scales"""
    b = RNG.integers(0, 256, size=(nb, KQT[t][0]), dtype=np.uint8)
    if t == Q6_K:                                          # every bit pattern of ql / qh / the signed scales; d a small finite half
        b[:, 208:210] = (RNG.random(nb).astype(np.float32) * 0.002 + 0.0001).astype(np.float16).reshape(-1, 1).view(np.uint8)
        return b
    b[:, 0:2] = (RNG.random(nb).astype(np.float32) * 0.02 + 0.001).astype(np.float16).reshape(-1, 1).view(np.uint8)
    b[:, 2:4] = (RNG.random(nb).astype(np.float32) * 0.05).astype(np.float16).reshape(-1, 1).view(np.uint8)
    return b


# ---------------------------------------------------------------- CPU: the restatement itself
def test_scale_packing_roundtrip_and_value_unpacking():
    sc = RNG.integers(0, 64, size=(50, 8))
    m = RNG.integers(0, 64, size=(50, 8))
    s2, m2 = KQ.unpack_scales(KQ.pack_scales(sc, m))
    assert np.array_equal(s2, sc) and np.array_equal(m2, m)
    # one block by hand: element 64 g + l = low nibble of qs[32 g + l] + bit 2g of qh[l]; 64 g + 32 + l = high nibble + bit 2g+1
    b = np.zeros((1, 176), dtype=np.uint8)
    b[0, 48 + 32 * 1 + 5] = 0xA7          # g = 1, l = 5: element 69 -> 7, element 101 -> 10
    b[0, 16 + 5] = 0b00001100             # bits 2, 3 of qh[5]: fifth bit of elements 64 + 5 and 96 + 5
    q = KQ.q5_values(b)[0]
    assert q[69] == 7 + 16 and q[101] == 10 + 16 and q.sum() == 7 + 10 + 32


def test_simple_quantizer_is_a_valid_encoding_close_to_its_input():
    x = _rand((40, 256), 2.0)
    x[3] = 0.0
    x[4, :32] = 5.0                        # a constant sub-block above zero: offset 0, scale 5/31
    blocks = KQ.quantize_q5_K(x)
    y = KQ.dequantize_q5_K(blocks)
    err = np.abs(y - x).reshape(40, 8, 32).max(axis=2)
    rng = (np.maximum(x.reshape(40, 8, 32).max(axis=2), 0) - np.minimum(x.reshape(40, 8, 32).min(axis=2), 0))
    assert np.all(err <= 0.08 * np.maximum(rng.max(axis=1, keepdims=True), 1e-6) + 1e-6)   # 5-bit code with 6-bit super-scales


def test_q4_K_restatement_by_hand_and_its_encoder():
    """Q4_K (r4): the super-block of Q5_K without the fifth-bit bytes -- one block by hand, then the encoder round trip"""
    b = np.zeros((1, 144), dtype=np.uint8)
    b[0, 0:2] = np.array([0.5], np.float16).view(np.uint8)            # d
    b[0, 2:4] = np.array([0.25], np.float16).view(np.uint8)           # dmin
    sc = np.array([[1, 2, 3, 40, 5, 6, 7, 8]]); m = np.array([[0, 1, 2, 3, 50, 5, 6, 7]])
    b[0, 4:16] = KQ.pack_scales(sc, m)
    b[0, 16 + 32 * 1 + 5] = 0xA7          # g = 1, l = 5: element 69 -> 7 under (sc, m)[2], element 101 -> 10 under (sc, m)[3]
    y = KQ.dequantize_q4_K(b)[0]
    assert y[69] == np.float32(0.5 * 3 * 7 - 0.25 * 2) and y[101] == np.float32(0.5 * 40 * 10 - 0.25 * 3)
    assert y[0] == 0.0 and y[33] == np.float32(-0.25) and y[128] == np.float32(-0.25 * 50)     # q = 0: minus the sub-block's min
    x = _rand((20, 256), 2.0)
    err = np.abs(KQ.dequantize_q4_K(KQ.quantize_q4_K(x)) - x).max()
    assert err <= 0.15 * (x.max() - x.min())                          # 4-bit code with 6-bit super-scales
    w = KQ.quantize_q4_K(_rand((4, 256))).reshape(1, -1)
    xx = _rand((1, 1024))
    d8, q8, _ = KQ.quantize_q8_K(xx.reshape(-1, 256))
    exact = float(KQ.dequantize_q4_K(w).astype(np.float64).reshape(-1) @ (q8.astype(np.float64) * d8.astype(np.float64)[:, None]).reshape(-1))
    assert abs(float(KQ.mul_mat_q4_K(w, xx)[0, 0]) - exact) <= 1e-5 * max(1.0, abs(exact))


def test_published_quantizers_restated_are_valid_and_tighter_than_the_simple_encoder():
    """quantize_row_q5_K_reference / _q4_K_reference as restated (r4; what the device quantizer follows): valid super-blocks whose
    decode is at least as close to the input as the simple test encoder's; upstream's own corner: a constant sub-block decodes to 0
    (make_qkx1_quants returns scale 0, min 0 when max == min), an all-zero super-block to zeros with d = dmin = 0."""
    x = _rand((60, 256), 2.0)
    x[3] = 0.0
    x[5] = np.abs(x[5])                                               # no negative value: min stays 0
    for bits, simple, deq, nbytes in ((5, KQ.quantize_q5_K, KQ.dequantize_q5_K, 176), (4, KQ.quantize_q4_K, KQ.dequantize_q4_K, 144)):
        b = KQ.quantize_kq_reference(x, bits)
        assert b.shape == (60, nbytes) and not b[3].any()
        y = deq(b)
        e_ref, e_simple = np.sqrt(((y - x) ** 2).mean()), np.sqrt(((deq(simple(x)) - x) ** 2).mean())
        assert e_ref <= 1.02 * e_simple and np.abs(y - x).max() <= (0.12 if bits == 5 else 0.25) * 2.0 * 3.5
        sc, m = KQ.unpack_scales(b[:, 4:16])
        assert sc.max() <= 63 and m.max() <= 63 and (sc.max(axis=1)[np.arange(60) != 3] == 63).all()   # the largest scale of a super-block maps to 63
    c = np.full((1, 256), 5.0, dtype=np.float32)
    assert not KQ.dequantize_q5_K(KQ.quantize_kq_reference(c, 5)).any()


def test_q6_K_restatement_by_hand_and_its_encoder():
    """Q6_K (r4): 16 sub-blocks of 16 six-bit weights (q - 32) with signed 8-bit scales -- one element by hand, pack / unpack, the encoder"""
    b = np.zeros((1, 210), dtype=np.uint8)
    b[0, 64 + 32 + 5] = 0xB0               # n = 1, c = 3 (high nibble of ql[64 n + 32 (c & 1) + l]), l = 5: element 128 + 96 + 5 = 229
    b[0, 128 + 32 + 5] = 0b10000000        # bits 6, 7 of qh[32 n + l]: the two high bits = 2
    sc = np.zeros(16, dtype=np.int8)
    sc[229 // 16] = -7
    b[0, 192:208] = sc.view(np.uint8)
    b[0, 208:210] = np.array([0.5], np.float16).view(np.uint8)
    v = KQ.q6_values(b)[0]
    assert v[229] == (0xB | (2 << 4)) - 32 and (v != -32).sum() == 1
    y = KQ.dequantize_q6_K(b)[0]
    assert y[229] == np.float32(0.5 * -7 * 11) and y[224] == np.float32(0.5 * -7 * -32) and y[0] == 0.0
    L = RNG.integers(0, 64, size=(5, 256))
    ql, qh = KQ.pack_q6(L)
    bb = np.zeros((5, 210), dtype=np.uint8)
    bb[:, :128], bb[:, 128:192] = ql, qh
    assert np.array_equal(KQ.q6_values(bb) + 32, L)
    x = _rand((30, 256), 2.0)
    x[3] = 0.0
    blocks = KQ.quantize_q6_K(x)
    assert not blocks[3].any()
    assert np.abs(KQ.dequantize_q6_K(blocks) - x).max() <= 0.04 * np.abs(x).max()          # 6-bit code with 8-bit super-scales
    w = KQ.quantize_q6_K(_rand((4, 256))).reshape(1, -1)
    xx = _rand((1, 1024))
    d8, q8, _ = KQ.quantize_q8_K(xx.reshape(-1, 256))
    exact = float(KQ.dequantize_q6_K(w).astype(np.float64).reshape(-1) @ (q8.astype(np.float64) * d8.astype(np.float64)[:, None]).reshape(-1))
    assert abs(float(KQ.mul_mat_q6_K(w, xx)[0, 0]) - exact) <= 1e-5 * max(1.0, abs(exact))


def test_q8_K_rule():
    x = np.zeros((3, 256), dtype=np.float32)
    x[0, 7] = -4.0
    x[0, 9] = 4.0                          # equal magnitudes: the FIRST one (negative) sets the sign: iscale = 32, d = 1/32
    x[0, 10] = 1.0 / 64                    # 0.5 -> tie -> 0
    x[0, 11] = 3.0 / 64                    # 1.5 -> 2
    x[1] = 0.0                             # all-zero super-block: d = 0
    x[2, 0] = 2.0                          # iscale = -64: the max element maps to -128, its negation clamps to 127
    x[2, 1] = -2.0
    d, q, bs = KQ.quantize_q8_K(x)
    assert d[0] == np.float32(1 / 32) and q[0, 7] == -128 and q[0, 9] == 127 and q[0, 10] == 0 and q[0, 11] == 2
    assert d[1] == 0 and not q[1].any()
    assert d[2] == np.float32(-1 / 64) and q[2, 0] == -128 and q[2, 1] == 127
    assert bs[0, 0] == -128 + 127 + 2 and bs[2, 0] == -1


def test_vec_dot_restatement_against_dequantized_product():
    K = 1024
    w = KQ.quantize_q5_K(_rand((K // 256, 256)))
    x = _rand((1, K))
    d8, q8, bs = KQ.quantize_q8_K(x.reshape(-1, 256))
    got = KQ.vec_dot_q5_K_q8_K(w, d8, q8, bs)
    exact = float(KQ.dequantize_q5_K(w).astype(np.float64).reshape(-1) @ (q8.astype(np.float64) * d8.astype(np.float64)[:, None]).reshape(-1))
    assert abs(float(got) - exact) <= 1e-5 * max(1.0, abs(exact))
    mm = KQ.mul_mat_q5_K(w.reshape(1, -1), x)
    assert abs(float(mm[0, 0]) - exact) <= 1e-5 * max(1.0, abs(exact))


# ---------------------------------------------------------------- GPU: the device path against the restatement
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    pytest.importorskip("torch")
    from ggmlsharp_amd import device
    device.init(0)
    return device


@gpu
@pytest.mark.parametrize("t", ALL_KQ)
def test_dequantize_q5_K_bit_exact(dev, t):
    import torch
    for nb in (1, 7, 64):
        b = np.concatenate([_random_blocks(nb, t), KQT[t][1](_rand((nb, 256), 3.0))])
        want = KQT[t][2](b)
        got = dev.dequantize_rows(t, torch.from_numpy(b.reshape(1, -1)).cuda(), b.shape[0] * 256).cpu().numpy().reshape(-1, 256)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@gpu
@pytest.mark.parametrize("t", ALL_KQ)
def test_device_quantizer_writes_the_restated_reference_quantizers_bytes(dev, t):
    """ggml_hip_quantize_rows_dev for the extension types (r4): byte for byte the super-blocks of tests/np_kquants.py quantize_kq_reference
    -- normal data at several scales, a zero row, a non-negative row, constant sub-blocks, a single outlier, ragged tail of the grid"""
    import torch
    for (nrows, K, scale) in ((1, 256, 1.0), (7, 768, 3.0), (33, 2048, 0.01), (5, 11008, 40.0)):
        x = _rand((nrows, K), scale)
        x[0, :256] = 0.0
        if nrows > 1:
            x[1] = np.abs(x[1])
            x[2, 32:64] = -1.5
            x[2, 300] = 1000.0
        want = REFQ[t](x.reshape(-1, 256)).reshape(nrows, -1)
        got = dev.quantize_rows(t, torch.from_numpy(x).cuda()).cpu().numpy()
        assert got.shape == want.shape
        bad = np.nonzero((got != want).reshape(-1, KQT[t][0]).any(axis=1))[0]
        assert bad.size == 0, f"type {t} {nrows}x{K}: super-blocks {bad[:8]} differ"
        # and the weight made from them multiplies like the restatement says
    x = _rand((64, 1024))
    rows = dev.quantize_rows(t, torch.from_numpy(x).cuda())
    a = _rand((20, 1024))
    got = dev.mul_mat(dev.Weight.from_device(t, rows, 1024), torch.from_numpy(a).cuda()).cpu().numpy()
    _close(got, KQT[t][3](rows.cpu().numpy(), a), f"k-quant {t} from the device quantizer", 1024)


@gpu
@pytest.mark.parametrize("t", ALL_KQ)
def test_upload_download_roundtrip_is_byte_exact_and_type_reported(dev, t):
    from ggmlsharp_amd._lib import lib
    M, K = 70, 768
    rows = _random_blocks(M * K // 256, t).reshape(M, -1)
    W = dev.Weight.from_host(t, rows, K)
    assert lib().ggml_hip_weight_type(W.handle) == t and lib().ggml_hip_weight_rows(W.handle) == M
    assert lib().ggml_hip_type_size(t) == KQT[t][0] and lib().ggml_hip_blck_size(t) == 256
    assert np.array_equal(W.download().reshape(M, -1), rows)
    shard = dev.Weight.from_host(t, rows, K, row_begin=11, row_end=40)
    assert np.array_equal(shard.download().reshape(29, -1), rows[11:40])
    h = C.c_void_p()
    assert lib().ggml_hip_weight_upload(t, rows.ctypes.data_as(C.c_void_p), 700, M, 528, 0, M, None, C.byref(h)) == -3   # K % 256


@gpu
@pytest.mark.parametrize("kind", [0, 1])
def test_q8_K_activation_image_matches_the_rule_bitwise(dev, kind):
    """INIT with the Q8_K rule (image kind + 16): decode the operand image back to quants and scales."""
    import torch
    from ggmlsharp_amd._lib import lib, check
    N, K = 37, 1024
    x = _rand((N, K), 2.0)
    x[3, 256:512] = 0.0
    x[5, 0] = 9.0
    x[5, 200] = -9.0                                   # equal magnitudes: the first decides the sign
    d8, q8, bs = KQ.quantize_q8_K(x.reshape(-1, 256))
    d8, q8 = d8.reshape(N, K // 256), q8.reshape(N, K)
    work = dev.alloc_work(Q5_K, K, N)
    xd = torch.from_numpy(x).cuda()
    check(lib().ggml_hip_quantize_act_dev(C.c_void_p(xd.data_ptr()), N, K, K, C.c_void_p(work.data_ptr()), work.numel(), 16 + kind, None), "quantize_act")
    torch.cuda.synchronize()
    nbk, Npad = K // 32, 256
    raw = work.cpu().numpy()
    img_bytes = nbk * 4 * Npad * 16
    ad = raw[img_bytes: img_bytes + nbk * Npad * 4].view(np.float32).reshape(nbk, Npad)[:, :N].T          # [N, nbk]
    asum = raw[img_bytes + nbk * Npad * 4: img_bytes + 2 * nbk * Npad * 4]
    assert np.array_equal(ad, np.repeat(d8, 8, axis=1))                                                  # one scale per 256 elements
    if kind == 0:
        a8 = raw[: nbk * 2 * Npad * 16].view(np.int8).reshape(nbk, 2, Npad, 16)
        q = np.empty((N, nbk, 32), dtype=np.int8)
        q[:, :, 0::2] = a8[:, 0, :N, :].transpose(1, 0, 2)
        q[:, :, 1::2] = a8[:, 1, :N, :].transpose(1, 0, 2)
        assert np.array_equal(q.reshape(N, K), q8)
        sums = asum.view(np.int32).reshape(nbk, Npad)[:, :N].T
        assert np.array_equal(sums, q8.reshape(N, nbk, 32).astype(np.int32).sum(axis=2))
    else:
        sums = asum.view(np.float32).reshape(nbk, Npad)[:, :N].T
        want = (np.repeat(d8, 8, axis=1) * q8.reshape(N, nbk, 32).astype(np.int32).sum(axis=2).astype(np.float32)).astype(np.float32)
        assert np.array_equal(sums, want)


def _close(got, ref, what, K):
    ref = np.asarray(ref, np.float64)
    O.assert_mul_mat_close(got, ref, K, what, normwise=1e-5 if ref.size >= 256 else 1e-3)   # THE mul_mat tolerance (tests/oracle_lib.py)


@gpu
@pytest.mark.parametrize("t", ALL_KQ)
def test_mul_mat_q5_K_matches_the_restatement(dev, t):
    """(r4: Q4_K through the same resident form and kernels)  every kernel form behind the Q5_1 image of a Q5_K weight: mat-vec (N <= 8, two-step), f16 MFMA with 4- / 2-way K split
    and unsplit, int8 MFMA; ragged M and N; raw random super-blocks as well as quantized normal data"""
    import torch
    for (M, K, N) in ((96, 256, 1), (300, 1024, 3), (128, 512, 8), (515, 768, 40), (256, 2048, 130), (640, 1024, 300),
                      (130, 512, 600), (257, 768, 1100),
                      (200, 2048, 300), (130, 4352, 512),       # r4: K3p on the int8 planes of the planar Q5_1 form (from 257 rows, K >= 2048)
                      (200, 2048, 1100), (130, 2304, 2500),      #     ... which has no upper bound for this type
                      (300, 2048, 33), (130, 4352, 64), (515, 2304, 9),    # r4: the batched-decode form K3s-int8 (9..64 rows, K >= 2048)
                      (130, 11008, 100), (96, 2048, 250)):                 #     ... up to 128 rows behind K >= 11008 (Q6_K: up to 256 whatever K)
        for raw in (False, True):
            rows = _random_blocks(M * K // 256, t).reshape(M, -1) if raw else KQT[t][1](_rand((M * K // 256, 256))).reshape(M, -1)
            x = _rand((N, K))
            ref = KQT[t][3](rows, x)
            W = dev.Weight.from_host(t, rows, K)
            got = dev.mul_mat(W, torch.from_numpy(x).cuda()).cpu().numpy()
            _close(got, ref, f"k-quant {t} {M}x{K}x{N} raw={raw}", K)
    # a row shard is bitwise a column slice of the whole (the multi-GPU promise holds for the extension too)
    M, K, N = 300, 1024, 70
    rows = KQT[t][1](_rand((M * K // 256, 256))).reshape(M, -1)
    xd = torch.from_numpy(_rand((N, K))).cuda()
    whole = dev.mul_mat(dev.Weight.from_host(t, rows, K), xd)
    part = dev.mul_mat(dev.Weight.from_host(t, rows, K, row_begin=100, row_end=260), xd)
    assert torch.equal(part, whole[:, 100:260])
    M, K, N = 300, 2048, 40                                # ... in the batched-decode form as well
    rows = KQT[t][1](_rand((M * K // 256, 256))).reshape(M, -1)
    xd = torch.from_numpy(_rand((N, K))).cuda()
    whole = dev.mul_mat(dev.Weight.from_host(t, rows, K), xd)
    part = dev.mul_mat(dev.Weight.from_host(t, rows, K, row_begin=100, row_end=260), xd)
    assert torch.equal(part, whole[:, 100:260])


@gpu
@pytest.mark.parametrize("t", ALL_KQ)
def test_fused_mat_vec_with_the_Q8_K_rule_equals_the_two_step_form_bitwise(dev, t):
    """up to 4 src1 rows the one-call entry runs the fused mat-vec with the Q8_K rule inside the kernel (gemv.hip K8, r4; K <= 32768): the
    same quants, the same summation tree as INIT + the mat-vec on K1's image -- so the same bits; and both against the restatement.
    One chunk of K (<= 4096), several, a ragged last chunk, the largest K the form takes, a K beyond it (two-step by plan)."""
    import torch
    from ggmlsharp_amd._lib import lib
    import ctypes as C
    for (M, K, N) in ((100, 256, 1), (515, 4096, 1), (515, 4096, 2), (300, 4096, 3), (300, 4096, 4), (130, 11008, 1), (130, 11008, 4),
                      (70, 32768, 2), (70, 33024, 1), (4096, 4096, 1)):
        rows = KQT[t][1](_rand((M * K // 256, 256))).reshape(M, -1)
        x = _rand((N, K), 2.0)
        x[0, 256:512] = 0.0                                   # an all-zero super-block: scale 0
        if K >= 1024:
            x[0, 700] = -x[0, 900]                            # equal magnitudes inside one super-block: the first decides the sign
        W = dev.Weight.from_host(t, rows, K)
        xd = torch.from_numpy(x).cuda()
        pl = _lib_plan(t, M, K, N)
        assert (pl.family == 1) == (K <= 32768), (K, pl.family)            # 1 = the fused mat-vec
        one = dev.mul_mat(W, xd)
        work = dev.alloc_work(t, K, N)
        dev.mul_mat_init(W, xd, work)
        two = torch.empty_like(one)
        dev.mul_mat_compute(W, N, two, work)
        assert torch.equal(one, two), (t, M, K, N)
        _close(one.cpu().numpy(), KQT[t][3](rows, x), f"k-quant {t} fused mat-vec {M}x{K}x{N}", K)
        W.free()


def _lib_plan(t, M, K, N):
    import ctypes as C
    from ggmlsharp_amd import _lib
    out = _lib.ggml_hip_mm_plan_t()
    assert _lib.lib().ggml_hip_mm_plan(t, M, K, N, C.byref(out)) == 0
    return out


@gpu
@pytest.mark.parametrize("t", ALL_KQ)
def test_mul_mat_q5_K_at_config_4s_size_on_a_sample(dev, t):
    """BASELINE config 4 names Q5_K at 4096 x 11008 x 512: the literal type at its literal size.  64 weight rows x 64 src1 rows of
    the device result against the numpy restatement (there is no oracle for k-quants: unpinned extra), and a 512-row shard is the
    bitwise column slice of the whole.  Weights: valid super-blocks quantized from normal data for the sampled rows, random
    super-block bytes (finite scales) elsewhere."""
    import torch
    M, K, N = 4096, 11008, 512
    rs = np.random.default_rng(20264)
    rows = _random_blocks(M * K // 256, t).reshape(M, -1)
    ms = np.sort(rs.choice(M, size=64, replace=False))
    ns = np.sort(rs.choice(N, size=64, replace=False))
    rows[ms[::2]] = KQT[t][1](_rand((32 * K // 256, 256))).reshape(32, -1)      # half of the sample: real quantized data
    x = _rand((N, K))
    W = dev.Weight.from_host(t, rows, K)
    xd = torch.from_numpy(x).cuda()
    got = dev.mul_mat(W, xd)
    ref = KQT[t][3](rows[ms], x[ns])
    _close(got.cpu().numpy()[np.ix_(ns, ms)], ref, f"k-quant {t} {M}x{K}x{N} (64 x 64 sample)", K)
    Ws = dev.Weight.from_host(t, rows, K, row_begin=1024, row_end=1536)
    assert torch.equal(dev.mul_mat(Ws, xd), got[:, 1024:1536])
    Ws.free()
    W.free()
