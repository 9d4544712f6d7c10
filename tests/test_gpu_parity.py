"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Bit-exact for quantize / dequantize / re-layout (byte and integer work); for mul_mat
SURVEY 8(c)'s metric with its one stated floor (tests/oracle_lib.py assert_mul_mat_close) per element and <= 1e-5 normwise
(only the order of the f32 block additions differs from the scalar reference, north_star tolerance is 1e-3 relative)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

QTYPES = [O.Q4_0, O.Q4_1, O.Q5_0, O.Q8_0]
# Q4_2 / Q5_1: unusable in the C# as written (numeric casts of the half scales); built to the intent, IEEE-half bit
# patterns as in the upstream scalar code.  Q4_2: mat-vec and int8 kernels only; Q5_1 also has the f16 form (no MX form).
QTYPES_D7 = [O.Q4_2, O.Q5_1]
ALLQ = QTYPES + QTYPES_D7
RNG = np.random.default_rng(20240613)


@pytest.fixture(scope="module")
def dev():
    from ggmlsharp_amd import device
    device.init(0)
    return device


def _rand(shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


def _special_rows(k):
    """rows that exercise the edge cases of the block quantizers."""
    x = _rand((8, k))
    x[0, :32] = 0.0                              # all-zero block: d = -0.0 / 0, id = 0
    x[1, :32] = 2.5                              # constant block
    x[2, :64] = np.round(x[2, :64] * 4) / 2      # many exact .5 ties after scaling
    x[3, :32] = -np.abs(x[3, :32])               # all negative: max is negative, d positive
    x[4, :32] *= 1e-30                           # tiny magnitudes
    x[5, :32] *= 1e20                            # huge magnitudes
    x[6, 0:32] = np.arange(32) - 16.0            # exact integers
    x[7, 5] = -x[7, 5]
    return x


def assert_close(got, ref, K, what=""):
    """THE mul_mat tolerance (tests/oracle_lib.py: SURVEY 8(c) with its one stated floor); over whole matrices also 1e-5 norm-wise
    (a handful of elements can sit on a cancellation -- Q4_1 / Q5_1 add a scale term and a min term of opposite sign -- the matrix cannot)."""
    ref = np.asarray(ref)
    O.assert_mul_mat_close(got, ref, K, what, normwise=1e-5 if ref.size >= 256 else 1e-4)


# ---------------------------------------------------------------- K9 / K8 bit-exact
@pytest.mark.parametrize("t", ALLQ + [O.Q8_1])
def test_quantize_rows_bit_exact(dev, t):
    for k in (32, 64, 4096):
        for x in (_special_rows(max(k, 64))[:, :k], _rand((37, k)), _rand((1, k), 1e-3), _rand((5, k), 300.0)):
            x = np.ascontiguousarray(x)
            want = O.quantize_row(t, x)
            got = dev.quantize_rows(t, torch.from_numpy(x).cuda()).cpu().numpy()
            assert np.array_equal(got, want), f"type {t} k {k}"


@pytest.mark.parametrize("t", ALLQ)
def test_dequantize_rows_bit_exact(dev, t):
    for k in (32, 256, 4096):
        x = np.ascontiguousarray(np.concatenate([_special_rows(max(k, 64))[:, :k], _rand((19, k), 3.0)]))
        q = O.quantize_row(t, x)
        want = O.dequantize_row(t, q, k)
        got = dev.dequantize_rows(t, torch.from_numpy(q).cuda(), k).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # arbitrary block bytes (not produced by the quantizer): every nibble / high-bit / sign pattern
    k = 1024
    raw = RNG.integers(0, 256, size=(16, O.row_bytes(t, k)), dtype=np.uint8)
    blocks = raw.reshape(-1, O.type_size(t))
    if t in (O.Q4_0, O.Q4_1, O.Q8_0):
        blocks[:, 0:4] = _rand(blocks.shape[0]).view(np.uint8).reshape(-1, 4)
    if t == O.Q4_1:
        blocks[:, 4:8] = _rand(blocks.shape[0]).view(np.uint8).reshape(-1, 4)
    if t in (O.Q5_0, O.Q4_2, O.Q5_1):
        blocks[:, 0:2] = _rand(blocks.shape[0]).astype(np.float16).view(np.uint8).reshape(-1, 2)
    if t == O.Q5_1:
        blocks[:, 2:4] = _rand(blocks.shape[0]).astype(np.float16).view(np.uint8).reshape(-1, 2)
    want = O.dequantize_row(t, raw, k)
    got = dev.dequantize_rows(t, torch.from_numpy(raw).cuda(), k).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_empty_inputs(dev):
    x = torch.empty((0, 64), dtype=torch.float32, device="cuda")
    assert dev.quantize_rows(O.Q4_0, x).shape == (0, 40)
    w = dev.Weight.from_host(O.Q4_0, O.quantize_row(O.Q4_0, _rand((4, 64))), 64)
    out = dev.mul_mat(w, torch.empty((0, 64), dtype=torch.float32, device="cuda"))
    assert out.shape == (0, 4)


# ---------------------------------------------------------------- re-layout round trip (byte-exact)
@pytest.mark.parametrize("t", ALLQ + [O.F32, O.F16])
def test_weight_roundtrip_bytes(dev, t):
    for (M, K) in ((1, 32), (37, 256), (130, 4096)):
        if t == O.F32:
            rows = _rand((M, K)).view(np.uint8)
        elif t == O.F16:
            rows = _rand((M, K)).astype(np.float16).view(np.uint8)
        else:
            rows = RNG.integers(0, 256, size=(M, O.row_bytes(t, K)), dtype=np.uint8)
        w = dev.Weight.from_host(t, rows, K)
        assert np.array_equal(w.download(), rows.reshape(-1))
        # a row shard, and the device-source path
        if M > 4:
            ws = dev.Weight.from_device(t, torch.from_numpy(rows).cuda(), K, row_begin=3, row_end=M - 1)
            assert np.array_equal(ws.download(), rows[3:M - 1].reshape(-1))


# ---------------------------------------------------------------- K1: activation planes == oracle Q8_0 blocks
_BF6 = {0: 0, 12: 1, 16: 2, 18: 3, 20: 4, 21: 5, 22: 6, 23: 7, 24: 8}   # e3m2 code -> integer (the only codes K1 may emit)


def _decode_bf6_fragments(frag):
    """frag: uint8 [..., 24] = 32 bf6 codes, element e at bits [6e, 6e+5] -> int32 [..., 32]"""
    bits = np.unpackbits(frag, axis=-1, bitorder="little").reshape(frag.shape[:-1] + (32, 6))
    code = (bits * (1 << np.arange(6))).sum(axis=-1)
    mag = np.vectorize(lambda c: _BF6.get(int(c) & 31, 99))(code)
    assert (mag != 99).all(), "K1 emitted a bf6 code that is not a small integer"
    return np.where(code & 32, -mag, mag).astype(np.int32)


def _decode_act_image(kind, raw, nbk, Npad, N):
    """the device image K1 wrote (layouts: csrc/common.h, gemm_q16.hip, gemm_qmx.hip) -> Q8_0 quants int32 [N][nbk][32]"""
    q = np.zeros((N, nbk, 32), np.int32)
    if kind == 0:      # int8 planes: plane 0 = even elements, plane 1 = odd elements
        a8 = raw[: nbk * 2 * Npad * 16].view(np.int8).reshape(nbk, 2, Npad, 16)
        q[:, :, 0::2] = a8[:, 0, :N, :].transpose(1, 0, 2)
        q[:, :, 1::2] = a8[:, 1, :N, :].transpose(1, 0, 2)
    elif kind == 1:    # f16, panel 2kk+h, k-slots [e0, e4, e1/16, e5/16, e2, e6, e3/16, e7/16], e_i = 16h + 8kk + i
        a16 = raw[: nbk * 4 * Npad * 16].view(np.float16).reshape(nbk, 4, Npad, 8).astype(np.float64)
        for kk in range(2):
            for h in range(2):
                pan = a16[:, 2 * kk + h, :N, :].transpose(1, 0, 2)
                for slot, (i, sc) in enumerate([(0, 1), (4, 1), (1, 16), (5, 16), (2, 1), (6, 1), (3, 16), (7, 16)]):
                    v = pan[:, :, slot] * sc
                    assert (v == np.round(v)).all()
                    q[:, :, 16 * h + 8 * kk + i] = v
    elif kind == 2:    # f16, panel 2kk+h, plane h byte jj = element 2jj + h, k-slots jj = 8kk + [0, 2, 1, 3, 4, 6, 5, 7]
        a16 = raw[: nbk * 4 * Npad * 16].view(np.float16).reshape(nbk, 4, Npad, 8).astype(np.float64)
        for kk in range(2):
            for h in range(2):
                pan = a16[:, 2 * kk + h, :N, :].transpose(1, 0, 2)
                for slot, jj in enumerate([0, 2, 1, 3, 4, 6, 5, 7]):
                    q[:, :, 2 * (8 * kk + jj) + h] = pan[:, :, slot]
    else:              # bf6 digits: per k-block [2][Npad][16 B] then [2][Npad][8 B]; a = 16 * ah + al
        img = raw[: nbk * 48 * Npad].reshape(nbk, 48 * Npad)
        p16 = img[:, : 32 * Npad].reshape(nbk, 2, Npad, 16)
        p8 = img[:, 32 * Npad:].reshape(nbk, 2, Npad, 8)
        frag = np.concatenate([p16, p8], axis=-1)[:, :, :N, :]            # [nbk][2][N][24]
        dig = _decode_bf6_fragments(np.ascontiguousarray(frag))          # [nbk][2][N][32]
        assert np.abs(dig[:, 0]).max() <= 8 and dig[:, 1].min() >= -8 and dig[:, 1].max() <= 7
        q[:] = (16 * dig[:, 0] + dig[:, 1]).transpose(1, 0, 2)
    return q


def _split3_bf16(v):
    """an f32 array as three bf16 bit patterns that sum to it exactly (common.h split3: truncation, 8 + 8 + 8 significand bits)"""
    v = np.ascontiguousarray(v, dtype=np.float32)
    b0 = v.view(np.uint32) & np.uint32(0xFFFF0000)
    r1 = v - b0.view(np.float32)
    b1 = r1.view(np.uint32) & np.uint32(0xFFFF0000)
    r2 = r1 - b1.view(np.float32)
    return (b0 >> 16).astype(np.uint16), (b1 >> 16).astype(np.uint16), (r2.view(np.uint32) >> 16).astype(np.uint16)


@pytest.mark.parametrize("t", [O.Q5_1, O.Q4_1])
def test_min_term_piece_planes_hold_d_times_sum_exactly(dev, t):
    """r4, K3p-int8 behind a min-term type: beside image 0, INIT writes d * (float)sum(q) -- the Q8_1 s0 + s1 of Ggml.cs:820-821 -- as three
    bf16 pieces per k-block, eight k-blocks of a row to a 16-byte entry; the pieces must sum to the f32 value bit for bit, k-blocks past
    the end of K and the k-group that completes the last pair must be zeros."""
    from ggmlsharp_amd._lib import lib, check
    for K, N in ((2048 + 96, 300), (4096, 512)):             # 67 k-blocks: 9 k-groups (the last holds 3 blocks), a tenth of zeros; 128: 16 k-groups
        assert lib().ggml_hip_act_image_kind(t, K, N) == 64
        x = np.ascontiguousarray(np.concatenate([_special_rows(K)[:8], _rand((N - 8, K))]))
        w = dev.Weight.from_host(t, O.quantize_row(t, _rand((4, K))), K)
        work = dev.alloc_work(t, K, N)
        work.fill_(0x7F)
        dev.mul_mat_init(w, torch.from_numpy(x).cuda(), work)
        torch.cuda.synchronize()
        raw = work.cpu().numpy()
        nbk, Npad = K // 32, (N + 255) // 256 * 256
        nba, nkg = (nbk + 3) // 4 * 4, (nbk + 7) // 8
        ref = O.quantize_row(O.Q8_0, x).reshape(N, nbk, 36)
        d = ref[:, :, :4].copy().view(np.float32).reshape(N, nbk)
        sums = ref[:, :, 4:].copy().view(np.int8).astype(np.int32).sum(axis=2)
        s = np.zeros((N, (nkg + (nkg & 1)) * 8), dtype=np.float32)
        s[:, :nbk] = d * sums.astype(np.float32)
        want = _split3_bf16(s)
        base = nba * 2 * Npad * 16                            # the half of the image region the int8 image leaves free
        planes = raw[base: base + (nkg + (nkg & 1)) * 3 * Npad * 16].view(np.uint16).reshape(-1, 3, Npad, 8)
        for pc in range(3):
            got = planes[:, pc, :N, :].transpose(1, 0, 2).reshape(N, -1)
            assert np.array_equal(got, want[pc]), (K, N, pc)
        # and the pieces do sum to the value (non-finite rows aside: those travel in the third piece alone)
        fin = np.isfinite(s)
        tot = sum((wp.astype(np.uint32) << 16).view(np.float32).astype(np.float64) for wp in want)
        assert np.array_equal(tot[fin].astype(np.float32).view(np.uint32), s[fin].view(np.uint32))
        # the explicit-layout entry with the kind ggml_hip_act_image_kind names writes the same bytes; a kind the flag does not belong to is refused
        work2 = dev.alloc_work(t, K, N)
        work2.fill_(0x7F)
        xd = torch.from_numpy(x).cuda()
        check(lib().ggml_hip_quantize_act_dev(C.c_void_p(xd.data_ptr()), N, K, K, C.c_void_p(work2.data_ptr()), work2.numel(), 64, None), "quantize_act")
        torch.cuda.synchronize()
        raw2 = work2.cpu().numpy()
        assert np.array_equal(raw2[base: base + (nkg + (nkg & 1)) * 3 * Npad * 16].view(np.uint16).reshape(-1, 3, Npad, 8)[:, :, :N, :], planes[:, :, :N, :])
        assert np.array_equal(raw2[:nbk * 2 * Npad * 16].reshape(nbk, 2, Npad, 16)[:, :, :N, :], raw[:nbk * 2 * Npad * 16].reshape(nbk, 2, Npad, 16)[:, :, :N, :])
        assert lib().ggml_hip_quantize_act_dev(C.c_void_p(xd.data_ptr()), N, K, K, C.c_void_p(work2.data_ptr()), work2.numel(), 64 + 3, None) != 0
        w.free()


@pytest.mark.parametrize("t,force", [(O.Q4_0, ""), (O.Q8_0, ""), (O.Q5_0, "f16"), (O.Q8_0, "f16")])
def test_quantize_act_planes_match_oracle(dev, t, force):
    """INIT phase (Ggml.cs:6641-6654): whatever image the selected kernel wants, it must hold exactly the oracle's Q8_0 row."""
    from ggmlsharp_amd._lib import lib, check
    K = 256 + 32    # 9 k-blocks: the pad blocks up to a whole stage (12) must be written as zeros
    for N in (1, 7, 33, 130):
        x = np.ascontiguousarray(np.concatenate([_special_rows(K)[:min(N, 8)], _rand((max(N - 8, 0), K))])[:N])
        w = dev.Weight.from_host(t, O.quantize_row(t, _rand((4, K))), K)
        work = dev.alloc_work(t, K, N)
        work.fill_(0x7F)     # poison: NaN patterns in every float view
        dev.mul_mat_init(w, torch.from_numpy(x).cuda(), work)
        torch.cuda.synchronize()
        raw = work.cpu().numpy()
        nbk, Npad = K // 32, (N + 255) // 256 * 256
        nba = (nbk + 3) // 4 * 4
        img = nba * 4 * Npad * 16   # the image region is sized for the largest (f16) image
        ref = O.quantize_row(O.Q8_0, x).reshape(N, nbk, 36)
        ref_d = ref[:, :, :4].copy().view(np.float32).reshape(N, nbk)
        ref_q = ref[:, :, 4:].copy().view(np.int8).astype(np.int32)
        kind = lib().ggml_hip_act_image_kind(t, K, N)
        if force == "f16":     # the f16 images serve big grids only: write them through the explicit-layout entry
            kind = 0 if N <= 8 else (2 if t == O.Q8_0 else 1)
            if kind:
                work.fill_(0x7F)
                check(lib().ggml_hip_quantize_act_dev(C.c_void_p(torch.from_numpy(x).cuda().data_ptr()), N, K, K, C.c_void_p(work.data_ptr()), work.numel(), kind, None), "quantize_act")
                torch.cuda.synchronize()
                raw = work.cpu().numpy()
        ad = raw[img: img + nba * Npad * 4].view(np.float32).reshape(nba, Npad)
        asum = raw[img + nba * Npad * 4: img + 2 * nba * Npad * 4].reshape(nba, Npad, 4)
        assert np.array_equal(ad[:nbk, :N].T.view(np.uint32), ref_d.view(np.uint32))
        sums = ref_q.sum(axis=2)
        if kind == 0:
            assert np.array_equal(asum[:nbk, :N].copy().view(np.int32)[..., 0].T, sums)
        else:   # the MFMA images carry the float d * sum(q) (the Q8_1 s0 + s1 of Ggml.cs:820-821)
            assert np.array_equal(asum[:nbk, :N].copy().view(np.float32)[..., 0].T.view(np.uint32), (ref_d * sums.astype(np.float32)).view(np.uint32))
        q = _decode_act_image(kind, raw, nba, Npad, N)
        assert np.array_equal(q[:, :nbk], ref_q)
        if kind != 0:    # pad k-blocks: zero quants, zero scales (finite * 0 = 0 in the kernels)
            assert not q[:, nbk:].any() and not ad[nbk:, :N].any()
        assert lib().ggml_hip_mul_mat_work_size(t, K, N) == raw.size


def test_bf6_image_is_reproducible_over_many_launches(dev):
    """Regression: the lane-per-block INIT kernel once produced wrong quants in a few rows of a ragged last row group in
    ~1 of 5 launches (a register-overlap hazard of v_cvt_scalef32_2xpk16_bf6_f32, quantize.hip).  40 launches with other
    kernels in between, every image decoded and compared with the bit-exact row quantizer."""
    from ggmlsharp_amd._lib import lib, check
    N, K = 2000, 1024
    nbk, Npad = K // 32, 2048
    x = torch.randn((N, K), device="cuda") * 2
    ref = dev.quantize_rows(O.Q8_0, x).view(N, nbk, 36)[:, :, 4:].contiguous().view(torch.int8).to(torch.int32)
    mag = torch.tensor([0] * 12 + [1, 99, 99, 99, 2, 99, 3, 99, 4, 5, 6, 7, 8] + [99] * 7, device="cuda")
    sh6 = torch.arange(6, device="cuda")
    for it in range(40):
        work = dev.alloc_work(O.Q4_0, K, N)
        work.zero_()
        torch.randn((1 + it % 5) * 200000, device="cuda").sum()
        check(lib().ggml_hip_quantize_act_dev(C.c_void_p(x.data_ptr()), N, K, K, C.c_void_p(work.data_ptr()), work.numel(), 3, None), "quantize_act")
        img = work[: nbk * 48 * Npad].view(nbk, 48 * Npad)
        frag = torch.cat([img[:, : 32 * Npad].view(nbk, 2, Npad, 16), img[:, 32 * Npad:].view(nbk, 2, Npad, 8)], dim=-1)[:, :, :N, :]
        bits = ((frag.to(torch.int64).unsqueeze(-1) >> torch.arange(8, device="cuda")) & 1).view(nbk, 2, N, 32, 6)
        code = (bits * (1 << sh6)).sum(-1)
        dig = torch.where((code & 32) != 0, -mag[code & 31], mag[code & 31])
        q = (16 * dig[:, 0] + dig[:, 1]).permute(1, 0, 2)
        assert torch.equal(q, ref), f"launch {it}: {int((q != ref).sum())} wrong quants"


# ---------------------------------------------------------------- mul_mat vs oracle
SHAPES = [  # (M, K, N): ragged M / N, all kernels (N <= 8 fused mat-vec, above that MFMA -- 32-row K-split tiles up to 128 rows; Q4_2 / Q5_1: two-step mat-vec up to 16), tail stage (K/32 % 4 != 0)
    (1, 32, 1), (16, 64, 1), (17, 96, 2), (200, 256, 3), (128, 4096, 1), (130, 352, 8),
    (64, 128, 9), (33, 160, 17), (200, 256, 64), (128, 512, 128), (257, 1024, 130), (64, 11008, 40),
    (600, 288, 300),    # several workgroup tiles in both directions, ragged edges, K/32 = 9 (pad k-blocks)
    (1, 32, 9), (31, 64, 33), (257, 96, 257), (129, 32, 65),   # one k-block (three pad blocks per stage), one-row weights
    (128, 4096, 16), (130, 4096 + 64, 13), (200, 288, 10), (17, 2080, 12),   # 9..16 rows: wide mat-vec form (Q4_2 / Q5_1), several LDS chunks of K
    (100, 512, 70), (96, 1056, 128), (40, 4096, 100),                       # 32-row tiles with the four-way K split (nbk >= 16, N <= 128)
    # K3p (gemm_qmp.hip): K >= 2048, Q4_0 257..512 rows / Q8_0 65..512 rows -- eight K ranges per workgroup, ragged M and N,
    # K / 32 = 65 (ranges of 10 with the last one short), 73 (pad blocks inside the last range)
    (200, 2048, 300), (129, 4160, 257), (96, 2336, 512), (300, 2080, 65), (130, 2048, 130),
]


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])   # 0 = automatic choice, 1 int8 MFMA, 2 f16 MFMA, 3 MX (Q4_0 / Q4_1)
@pytest.mark.parametrize("t", ALLQ)
def test_mul_mat_q_matches_oracle(dev, t, kernel):
    from ggmlsharp_amd._lib import lib
    if t == O.Q4_2 and kernel:
        pytest.skip("Q4_2 has one mat-mat kernel (int8); nothing to force")
    lib().ggml_hip_debug_force_gemm(kernel)
    try:
        for (M, K, N) in SHAPES:
            if kernel and N <= 8:
                continue          # the mat-vec kernel serves N <= 8 whatever is forced
            w = _rand((M, K))
            x = _rand((N, K), 2.0)
            wq = O.quantize_row(t, w)
            ref = O.mul_mat(t, wq, x, M, K, N, nth=4)[0, 0]
            W = dev.Weight.from_host(t, wq, K)
            got = dev.mul_mat(W, torch.from_numpy(x).cuda()).cpu().numpy()
            assert_close(got, ref, K, f"type {t} M{M} K{K} N{N} kernel {kernel}")
    finally:
        lib().ggml_hip_debug_force_gemm(0)


def test_small_n_fused_path_equals_two_step_path(dev):
    # N <= 8: ggml_hip_mul_mat_dev runs the fused kernel (quantize in-kernel); init_dev + compute_dev is the two-step
    # form.  Same integer and float arithmetic, same summation tree -> bitwise equal.
    for t in ALLQ:
        for (M, K, N) in ((130, 352, 1), (64, 4096 + 64, 3), (37, 256, 8)):
            wq = O.quantize_row(t, _rand((M, K)))
            x = torch.from_numpy(_rand((N, K), 2.0)).cuda()
            W = dev.Weight.from_host(t, wq, K)
            fused = dev.mul_mat(W, x)
            work = dev.alloc_work(t, K, N)
            two = torch.empty_like(fused)
            dev.mul_mat_init(W, x, work)
            dev.mul_mat_compute(W, N, two, work)
            assert torch.equal(fused, two), (t, M, K, N)


def test_mul_mat_q_strided_src1_and_dst(dev):
    M, K, N = 96, 256, 20
    wq = O.quantize_row(O.Q4_0, _rand((M, K)))
    xbig = torch.from_numpy(_rand((N, K + 64))).cuda()
    x = xbig[:, :K]
    out_big = torch.full((N, M + 32), -7.0, dtype=torch.float32, device="cuda")
    W = dev.Weight.from_host(O.Q4_0, wq, K)
    dev.mul_mat(W, x, out=out_big[:, :M])
    ref = O.mul_mat(O.Q4_0, wq, x.cpu().numpy(), M, K, N)[0, 0]
    assert_close(out_big[:, :M].cpu().numpy(), ref, K)
    assert torch.all(out_big[:, M:] == -7.0)


def test_mul_mat_extreme_block_values(dev):
    # every weight at the extreme quant and activations at +-amax: the largest possible integer block sums
    M, K, N = 64, 128, 16
    for t, lo in ((O.Q4_0, -8), (O.Q5_0, -16), (O.Q8_0, -127)):
        w = np.full((M, K), float(lo), dtype=np.float32)
        x = np.where(RNG.random((N, K)) < 0.5, -3.0, 3.0).astype(np.float32)
        wq = O.quantize_row(t, w)
        ref = O.mul_mat(t, wq, x, M, K, N)[0, 0]
        got = dev.mul_mat(dev.Weight.from_host(t, wq, K), torch.from_numpy(x).cuda()).cpu().numpy()
        assert_close(got, ref, K, f"extreme type {t}")
    # raw Q8_0 weight bytes including -128
    raw = RNG.integers(0, 256, size=(M, K // 32 * 36), dtype=np.uint8)
    raw.reshape(-1, 36)[:, :4] = _rand(M * K // 32).view(np.uint8).reshape(-1, 4)
    x = _rand((N, K))
    ref = O.mul_mat(O.Q8_0, raw, x, M, K, N)[0, 0]
    got = dev.mul_mat(dev.Weight.from_host(O.Q8_0, raw, K), torch.from_numpy(x).cuda()).cpu().numpy()
    assert_close(got, ref, K, "raw q8_0")


@pytest.mark.parametrize("t", [O.F32, O.F16])
def test_mul_mat_dense_matches_oracle(dev, t):
    for (M, K, N) in ((64, 128, 256), (1, 32, 1), (70, 100, 33), (256, 4096, 1),  # first = BASELINE config 1
                      (300, 4096, 3), (129, 1024, 8), (100, 520, 5), (77, 516, 2), (90, 1024, 19), (64, 512, 32),     # mat-vec form (K % 8 / K % 4, K >= 512), else the tile kernel
                      (1024, 512, 200), (1000, 520, 130), (2050, 96, 64),             # F16: the matrix-core kernel on partly filled grids, ragged M / N / K
                      (1030, 1024, 64), (1500, 2080, 100), (11000, 1024, 33)):        # F16, N <= 128: four-way K split on 32-row / 128-row tiles
        w = _rand((M, K))
        x = _rand((N, K))
        wraw = w if t == O.F32 else w.astype(np.float16).view(np.uint16)
        ref = O.mul_mat(t, wraw, x, M, K, N, nth=4)[0, 0]
        W = dev.Weight.from_host(t, wraw.view(np.uint8), K)
        got = dev.mul_mat(W, torch.from_numpy(x).cuda()).cpu().numpy()
        assert_close(got, ref, K, f"dense type {t} M{M} K{K} N{N}")


def test_dense_matrix_core_forms_require_scratch_and_aligned_src1(dev):
    """the matrix-core forms of dense weights (F16 above 4 src1 rows; F32 above 256, where the reference needs no wdata,
    Ggml.cs:3371-3373) REQUIRE the scratch of ggml_hip_mul_mat_work_size and a 16-byte aligned src1: a missing / short buffer is
    GGML_HIP_ERR_ARG, a misaligned src1 GGML_HIP_ERR_SHAPE -- never a silent switch to another kernel (include/ggml_hip.h)."""
    from ggmlsharp_amd._lib import lib, ERR_ARG, ERR_SHAPE
    L = lib()
    M, K = 128, 256
    for t, N in ((O.F32, 300), (O.F16, 16)):
        w = _rand((M, K))
        wraw = w if t == O.F32 else w.astype(np.float16).view(np.uint16)
        W = dev.Weight.from_host(t, wraw.view(np.uint8), K)
        need = L.ggml_hip_mul_mat_work_size(t, K, N)
        assert need > 0
        x = torch.from_numpy(_rand((N, K + 4))).cuda()
        out = torch.empty((N, M), device="cuda")
        work = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda xp, ld, wp, wb: L.ggml_hip_mul_mat_dev(W.handle, C.c_void_p(xp), N, ld, C.c_void_p(out.data_ptr()), M, wp, wb, st)  # noqa: E731
        assert call(x.data_ptr(), K + 4, None, 0) == ERR_ARG                                     # no scratch
        assert call(x.data_ptr(), K + 4, C.c_void_p(work.data_ptr()), need - 1) == ERR_ARG       # short scratch
        assert call(x.data_ptr() + 4, K + 4, C.c_void_p(work.data_ptr()), need) == ERR_SHAPE     # base not 16-byte aligned
        assert call(x.data_ptr(), K + 3, C.c_void_p(work.data_ptr()), need) == ERR_SHAPE         # row stride not a multiple of 4
        assert call(x.data_ptr(), K + 4, C.c_void_p(work.data_ptr()), need) == 0                 # ... and the well-formed call runs
        ref = O.mul_mat(t, wraw, x[:, :K].cpu().numpy().copy(), M, K, N, nth=4)[0, 0]
        assert_close(out.cpu().numpy(), ref, K, f"dense type {t} with scratch")
        # below the matrix-core range the reference's contract holds as it is: no scratch needed
        n2 = 3
        assert L.ggml_hip_mul_mat_work_size(t, K, n2) == 0 or t == O.F16
        W.free()


# ---------------------------------------------------------------- Seam 2 host forms
def test_row_functions_host_forms(dev):
    from ggmlsharp_amd._lib import lib
    L = lib()
    k = 256
    x = _rand(k)
    for t in ALLQ:
        want = O.quantize_row(t, x)
        got = np.zeros_like(want)
        assert L.ggml_hip_quantize_row(t, x.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), k) == 0
        assert np.array_equal(got, want)
        y = np.zeros(k, dtype=np.float32)
        assert L.ggml_hip_dequantize_row(t, want.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), k) == 0
        assert np.array_equal(y.view(np.uint32), O.dequantize_row(t, want, k).view(np.uint32))
        vt = O.lib().oracle_vec_dot_type(t)
        aq = O.quantize_row(vt, _rand(k))
        s = np.zeros(1, dtype=np.float32)
        assert L.ggml_hip_vec_dot(t, k, s.ctypes.data_as(C.c_void_p), want.ctypes.data_as(C.c_void_p),
                                  aq.ctypes.data_as(C.c_void_p)) == 0
        ref = O.vec_dot(t, k, want, aq)
        assert abs(s[0] - ref) <= 1e-3 * abs(ref) + 1e-5
    # rejected types (null slots)
    s = np.zeros(1, dtype=np.float32)
    assert L.ggml_hip_vec_dot(O.Q8_1, 32, s.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                              x.ctypes.data_as(C.c_void_p)) == -2
    assert L.ggml_hip_dequantize_row(O.Q8_1, x.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), 32) == -2


# ---------------------------------------------------------------- Seam 1 through the ggml_* mirror (Test3-style program)
def test_ggml_api_program_quantized_and_batched(dev):
    import ggml_mirror as G
    K, M, N = 256, 48, 12
    ctx = G.ggml_init(64 * 1024 * 1024)
    assert ctx
    try:
        for t in (G.Q4_0, G.Q8_0, G.F32):
            Wt = G.ggml_new_tensor_3d(ctx, t, K, M, 2)
            Xt = G.ggml_new_tensor_3d(ctx, G.F32, K, N, 2)
            w = _rand((2, M, K))
            x = _rand((2, N, K))
            wraw = w.view(np.uint8).reshape(2, M, -1) if t == G.F32 else O.quantize_row(t, w)
            G.tensor_bytes(Wt)[:] = wraw.reshape(-1)
            G.tensor_f32(Xt)[:] = x.reshape(1, 2, N, K)
            Y = G.ggml_mul_mat(ctx, Wt, Xt)
            assert Y and Y.contents.ne[0] == M and Y.contents.ne[1] == N and Y.contents.ne[2] == 2
            gf = G.ggml_build_forward(Y)
            assert gf.n_nodes == 1 and gf.n_leafs == 2
            G.ggml_graph_compute(ctx, gf)
            got = G.tensor_f32(Y)[0]
            ref = O.mul_mat(t, wraw, x, M, K, N, nth=2, ne2=2)[0]
            assert_close(got, ref, K, f"ggml api type {t}")
            # second compute hits the weight cache; after rewriting the weights the caller must invalidate
            G.ggml_graph_compute(ctx, gf)
            assert_close(G.tensor_f32(Y)[0], ref, K)
        # shape errors surface as NULL / status instead of the reference's vanished Debug.Assert
        bad = G.ggml_mul_mat(ctx, G.ggml_new_tensor_2d(ctx, G.F32, 64, 4), G.ggml_new_tensor_2d(ctx, G.F32, 32, 4))
        assert not bad
    finally:
        G.ggml_free(ctx)


def _test3_data():
    """Test3's design matrix and labels (Test3/Program.cs:22-42): F f32 [ne0 = NF = 256, ne1 = NP = 4096] from the LCG of :98-107
    (seeded 0; the stream itself is pinned in test_oracle.py against the five values SURVEY 8(c) lists), generated by the build's own
    oracle_xrand, in f32 arithmetic as the C# writes it."""
    NP, NF = 1 << 12, 1 << 8
    L = O.lib()
    L.oracle_xsrand(0)
    r = np.array([L.oracle_xrand() for _ in range(NP * NF)], dtype=np.float32).reshape(NP, NF)
    ll = np.where(np.arange(NP) < NP // 2, np.float32(1.0), np.float32(-1.0)).astype(np.float32)
    i = np.arange(NF)[None, :]
    ind = np.where((ll[:, None] > 0) & (i < NF // 2), np.float32(1.0), np.where((ll[:, None] < 0) & (i >= NF // 2), np.float32(1.0), np.float32(0.0))).astype(np.float32)
    noise = ((r / np.float32(32767.0) - np.float32(0.5)) * np.float32(0.1)).astype(np.float32)
    F = ((ind + noise) / np.float32(0.5 * NF)).astype(np.float32)
    return F, ll, NP, NF


def test_the_references_own_caller_test3_product_through_seam_1(dev):
    """The ONE workload the reference itself defines on this path: Test3/Program.cs:57 `ggml_mul_mat(ctx0, F, x)` --
    F f32 [256, 4096] from the LCG, x f32 [256]: M = 4096, K = 256, N = 1 -- built with the ggml_* API of the host mirror, run by
    ggml_graph_compute through Seam 1 (ggml_hip_compute_forward_mul_mat) and compared with the oracle's
    ggml_compute_forward_mul_mat_f32 (Ggml.cs:5969-6178, f64 running sum) under THE tolerance.  x is taken at three points of the
    optimizer's path: its start (all zeros, :46), the solution the test asserts (+1 / -1, :82-88) and a point in between.  Then the
    product of its backward graph, mul_mat(cont(transpose(F)), grad) -- M = 256, K = 4096, N = 1 -- as a plain product on F^T."""
    import ggml_mirror as G
    F, ll, NP, NF = _test3_data()
    ctx = G.ggml_init(64 * 1024 * 1024)
    assert ctx
    try:
        Ft = G.ggml_new_tensor_2d(ctx, G.F32, NF, NP)
        xt = G.ggml_new_tensor_1d(ctx, G.F32, NF)
        G.tensor_f32(Ft)[:] = F.reshape(1, 1, NP, NF)
        y = G.ggml_mul_mat(ctx, Ft, xt)
        assert y and y.contents.ne[0] == NP and y.contents.ne[1] == 1          # Ggml.cs:7137-7151: {a.ne1, b.ne1}
        gf = G.ggml_build_forward(y)
        sol = np.where(np.arange(NF) < NF // 2, 1.0, -1.0).astype(np.float32)
        for name, xv in (("start", np.zeros(NF, np.float32)), ("solution", sol), ("midway", (0.37 * sol + _rand(NF, 0.05)).astype(np.float32))):
            G.tensor_f32(xt)[:] = xv.reshape(1, 1, 1, NF)
            G.ggml_graph_compute(ctx, gf)
            got = G.tensor_f32(y)[0, 0].copy()
            ref = O.mul_mat(O.F32, F, xv.reshape(1, NF), NP, NF, 1, nth=4)[0, 0]
            assert_close(got, ref, NF, f"Test3 F * x at the {name}")
            if name == "solution":                        # what the optimizer converges to: F x = l up to the noise (sanity of the recipe itself)
                assert np.max(np.abs(got - ll)) < 0.1
        # the backward graph's product: F^T (contiguous) times the residual -- a 256 x 4096 x 1 mat-vec
        FT = np.ascontiguousarray(F.T)
        Tt = G.ggml_new_tensor_2d(ctx, G.F32, NP, NF)
        gt = G.ggml_new_tensor_1d(ctx, G.F32, NP)
        G.tensor_f32(Tt)[:] = FT.reshape(1, 1, NF, NP)
        grad = (F @ (0.37 * sol) - ll).astype(np.float32)
        G.tensor_f32(gt)[:] = grad.reshape(1, 1, 1, NP)
        z = G.ggml_mul_mat(ctx, Tt, gt)
        gz = G.ggml_build_forward(z)
        G.ggml_graph_compute(ctx, gz)
        ref = O.mul_mat(O.F32, FT, grad.reshape(1, NP), NF, NP, 1, nth=4)[0, 0]
        assert_close(G.tensor_f32(z)[0, 0], ref, NP, "Test3 backward product F^T * grad")
    finally:
        G.ggml_free(ctx)


def test_seam1_weight_cache_evicts_least_recently_used_entries_under_a_budget(dev):
    """the Seam-1 weight cache is bounded -- with a budget that holds two of three weights, the third upload evicts the
    least recently used one, the entry a call runs on always stays, an evicted weight is uploaded again on its next use, and every
    result stays the oracle's.  (Without a budget the bound is the device's memory: a failed hipMalloc evicts and retries once.)"""
    import ggml_mirror as G
    from ggmlsharp_amd._lib import lib
    L = lib()
    K, M, N = 512, 256, 8
    ctx = G.ggml_init(32 * 1024 * 1024)

    def stats():
        v = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
        L.ggml_hip_debug_weight_cache_stats(*[C.byref(x) for x in v])
        return [x.value for x in v]
    try:
        L.ggml_hip_invalidate_all()
        Ws, graphs, refs = [], [], []
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        x = _rand((N, K))
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        for i in range(3):
            W = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
            wq = O.quantize_row(O.Q4_0, _rand((M, K)))
            G.tensor_bytes(W)[:] = wq.reshape(-1)
            Y = G.ggml_mul_mat(ctx, W, X)
            Ws.append(W); graphs.append((G.ggml_build_forward(Y), Y)); refs.append(O.mul_mat(O.Q4_0, wq, x, M, K, N)[0, 0])

        def run(i):
            gf, Y = graphs[i]
            G.ggml_graph_compute(ctx, gf)
            assert_close(G.tensor_f32(Y)[0, 0], refs[i], K, f"weight {i} through the bounded cache")
        run(0)
        n0, b0, e0 = stats()
        assert n0 == 1 and b0 > M * K // 2                         # one entry; its resident images are larger than the 20 B / 32 of the format
        L.ggml_hip_debug_weight_cache_budget(2 * b0 + b0 // 2)      # room for two
        run(1)
        assert stats()[0] == 2 and stats()[2] == e0
        run(0)                                                      # touch 0: now 1 is the least recently used
        run(2)                                                      # third upload: 1 goes
        n, b, e = stats()
        assert n == 2 and e == e0 + 1 and b == 2 * b0
        h0 = C.c_uint64()
        L.ggml_hip_debug_transfer_counters(C.byref(h0), None, None)
        run(0)                                                      # still cached: only src1 travels
        h1 = C.c_uint64()
        L.ggml_hip_debug_transfer_counters(C.byref(h1), None, None)
        assert h1.value - h0.value == N * K * 4
        run(1)                                                      # evicted: uploaded again (2 goes), same result
        assert stats()[2] == e0 + 2 and stats()[0] == 2
        L.ggml_hip_debug_weight_cache_budget(1)                     # a budget below any entry: the entry a call runs on still stays
        run(2)
        assert stats()[0] == 1
        run(2)
    finally:
        L.ggml_hip_debug_weight_cache_budget(0)
        G.ggml_free(ctx)


def test_graph_residency_chained_mul_mats(dev):
    """Y2 = W2 * (W1 * X) in one graph.  The intermediate is consumed from HBM (no second host ->
    device copy), both node results are in host memory when ggml_graph_compute returns, values match the oracle chain."""
    import ggml_mirror as G
    from ggmlsharp_amd._lib import lib
    K, M1, M2, N = 256, 96, 40, 24
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        for t in (G.Q4_0, G.Q8_0):
            for batch in (1, 2):
                W1 = G.ggml_new_tensor_3d(ctx, t, K, M1, batch)
                W2 = G.ggml_new_tensor_3d(ctx, t, M1, M2, batch)
                X = G.ggml_new_tensor_3d(ctx, G.F32, K, N, batch)
                w1, w2, x = _rand((batch, M1, K)), _rand((batch, M2, M1)), _rand((batch, N, K))
                w1q, w2q = O.quantize_row(t, w1), O.quantize_row(t, w2)
                G.tensor_bytes(W1)[:] = w1q.reshape(-1)
                G.tensor_bytes(W2)[:] = w2q.reshape(-1)
                G.tensor_f32(X)[:] = x.reshape(1, batch, N, K)
                Y1 = G.ggml_mul_mat(ctx, W1, X)
                Y2 = G.ggml_mul_mat(ctx, W2, Y1)
                gf = G.ggml_build_forward(Y2)
                assert gf.n_nodes == 2
                c0 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
                c1 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
                lib().ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c0])
                G.ggml_graph_compute(ctx, gf)
                lib().ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c1])
                ref1 = O.mul_mat(t, w1q, x, M1, K, N, nth=2, ne2=batch)[0]
                ref2 = O.mul_mat(t, w2q, np.ascontiguousarray(ref1), M2, M1, N, nth=2, ne2=batch)[0]
                assert_close(G.tensor_f32(Y1)[0], ref1, K, f"graph node 1 type {t} batch {batch}")
                assert_close(G.tensor_f32(Y2)[0], ref2, M1, f"graph node 2 type {t} batch {batch}")
                assert c1[2].value - c0[2].value == 1                                  # Y1 was read from HBM
                assert c1[0].value - c0[0].value == batch * N * K * 4                  # only X went host -> device
                assert c1[1].value - c0[1].value == batch * N * (M1 + M2) * 4          # both results came back
                # again: the buffers are recycled, the results are the same
                G.ggml_graph_compute(ctx, gf)
                assert_close(G.tensor_f32(Y2)[0], ref2, M1)
    finally:
        G.ggml_free(ctx)


def test_seam1_ignores_non_compute_phases_and_other_threads(dev):
    import ggml_mirror as G
    from ggmlsharp_amd._lib import lib, ggml_compute_params
    ctx = G.ggml_init(4 * 1024 * 1024)
    try:
        Wt = G.ggml_new_tensor_2d(ctx, G.F32, 32, 4)
        Xt = G.ggml_new_tensor_2d(ctx, G.F32, 32, 2)
        G.ggml_set_f32(Wt, 1.0)
        G.ggml_set_f32(Xt, 2.0)
        Y = G.ggml_mul_mat(ctx, Wt, Xt)
        G.ggml_set_f32(Y, -1.0)
        for (phase, ith) in ((0, 0), (2, 0), (1, 1)):
            p = ggml_compute_params(phase, ith, 2, 0, None)
            assert lib().ggml_hip_compute_forward_mul_mat(C.byref(p), Wt, Xt, Y) == 0
            assert G.ggml_get_f32_1d(Y, 0) == -1.0
        p = ggml_compute_params(1, 0, 2, 0, None)
        assert lib().ggml_hip_compute_forward_mul_mat(C.byref(p), Wt, Xt, Y) == 0
        assert G.ggml_get_f32_1d(Y, 0) == 64.0
    finally:
        G.ggml_free(ctx)


# ---------------------------------------------------------------- multi-GPU re-layout helper
def test_relayout_gathered(dev):
    G_, N, Ms, M = 3, 5, 7, 19
    g = _rand((G_, N, Ms))
    out = dev.relayout_gathered(torch.from_numpy(g).cuda(), G_, N, Ms, M).cpu().numpy()
    want = np.concatenate([g[r] for r in range(G_)], axis=1)[:, :M]
    assert np.array_equal(out, want)


def test_reference_style_c_program_on_gpu(dev, tmp_path):
    """tests/c/reference_style_program.c (Test1-shaped f32 mul_mat 64x128x256 through ggml_graph_compute) from plain C."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir, supdir = os.path.join(root, "ggmlsharp_amd", "lib"), os.path.join(root, "tests", "support")
    exe = str(tmp_path / "refprog")
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), "-I" + supdir, os.path.join(root, "tests", "c", "reference_style_program.c"),
                           "-L" + supdir, "-L" + libdir, "-lggml_hostmirror", "-lggml_hip", "-Wl,-rpath," + supdir, "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
