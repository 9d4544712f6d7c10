"""CPU tests that pin the oracle: hand-derived KATs, the reference's own pinnable
facts (Test0 layout, Test3 LCG), and bit-exact agreement with the independent numpy restatement."""
import ctypes as C

import numpy as np
import pytest

import np_restatement as R
import oracle_lib as O

RNG = np.random.default_rng(1234)


def _rand(shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


# ---------------- type tables (Ggml.cs:55-87, TypeDefinitions.cs:236-290) ----------------
def test_type_tables():
    sizes = {O.F32: 4, O.F16: 2, O.Q4_0: 20, O.Q4_1: 24, O.Q4_2: 10, O.Q4_3: 12, O.Q5_0: 22, O.Q5_1: 24,
             O.Q8_0: 36, O.Q8_1: 44, O.I8: 1, O.I16: 2, O.I32: 4}
    blck = {O.F32: 1, O.F16: 1, O.Q4_0: 32, O.Q4_1: 32, O.Q4_2: 16, O.Q4_3: 16, O.Q5_0: 32, O.Q5_1: 32,
            O.Q8_0: 32, O.Q8_1: 32, O.I8: 1, O.I16: 1, O.I32: 1}
    for t, s in sizes.items():
        assert O.type_size(t) == s
        assert O.blck_size(t) == blck[t]
    L = O.lib()
    assert L.oracle_vec_dot_type(O.Q4_0) == O.Q8_0
    assert L.oracle_vec_dot_type(O.Q5_0) == O.Q8_0
    assert L.oracle_vec_dot_type(O.Q8_0) == O.Q8_0
    assert L.oracle_vec_dot_type(O.Q4_1) == O.Q8_1
    assert L.oracle_vec_dot_type(O.Q4_3) == -1


# ---------------- Test3's LCG: first five values 38, 7719, 21238, 2437, 8855 ----------------
def test_test3_lcg_stream():
    L = O.lib()
    L.oracle_xsrand(0)
    assert [L.oracle_xrand() for _ in range(5)] == [38, 7719, 21238, 2437, 8855]


# ---------------- KAT1: exact Q4_0 round trip ----------------
def test_kat1_q4_0_roundtrip():
    x = np.array([((l % 16) - 8) * 0.5 for l in range(32)], dtype=np.float32)
    q = O.quantize_row(O.Q4_0, x)
    assert q[:4].tobytes() == bytes.fromhex("0000003F")  # d = 0.5
    assert q[4:].tobytes() == bytes.fromhex("1032547698BADCFE" * 2)
    y = O.dequantize_row(O.Q4_0, q, 32)
    assert np.array_equal(y.view(np.uint32), x.view(np.uint32))


# ---------------- KAT2: rounding mode, half-to-even (D1) ----------------
def test_kat2_q4_0_half_even():
    x = np.zeros(32, dtype=np.float32)
    x[:6] = [-8, 0.5, 1.5, 2.5, -0.5, -1.5]
    q = O.quantize_row(O.Q4_0, x)
    assert q[:4].copy().view(np.float32)[0] == 1.0
    # half-even: -8->0, 0.5->8, 1.5->10, 2.5->10, -0.5->8, -1.5->6 ; upstream half-away would be 90 BA 67
    assert q[4:7].tobytes() == bytes.fromhex("80AA68")
    assert q[7:].tobytes() == bytes([0x88] * 13)


def test_kat_q8_0_ties_and_sign():
    # amax = 127 -> d = 1, id = 1: quants are the rounded inputs; ties go to even; negatives stay signed (D4)
    x = np.zeros(32, dtype=np.float32)
    x[:8] = [127, -127, 0.5, 1.5, 2.5, -0.5, -1.5, -2.5]
    q = O.quantize_row(O.Q8_0, x)
    assert q[:4].copy().view(np.float32)[0] == 1.0
    assert list(q[4:12].view(np.int8)) == [127, -127, 0, 2, 2, 0, -2, -2]
    # D2: odd positions are written too
    x2 = np.arange(32, dtype=np.float32) - 16
    q2 = O.quantize_row(O.Q8_0, x2)
    d = q2[:4].copy().view(np.float32)[0]
    assert d == np.float32(16.0) / np.float32(127.0)
    deq = O.dequantize_row(O.Q8_0, q2, 32)
    assert np.max(np.abs(deq - x2)) <= d / 2 + 1e-6
    assert q2[4:].view(np.int8)[1] != 0  # an odd index


def test_kat_q5_0_bits():
    # max-abs element is -16 -> d = 1; q = min(31, (int)(x + 16.5))
    x = np.zeros(32, dtype=np.float32)
    x[0] = -16.0
    x[1] = 15.0   # -> 31, high bit set
    x[2] = 0.4    # (int)16.9 -> 16, high bit set, nibble 0
    x[3] = -0.6   # (int)15.9 -> 15, nibble 15
    x[31] = 15.7  # (int)32.2 -> clamp 31
    q = O.quantize_row(O.Q5_0, x)
    assert q[0:2].copy().view(np.float16)[0] == 1.0
    qh = int(q[2:6].copy().view(np.uint32)[0])
    assert q[6] == (0 | (15 << 4))
    assert q[7] == (0 | (15 << 4))
    assert (qh >> 0) & 1 == 0 and (qh >> 1) & 1 == 1 and (qh >> 2) & 1 == 1 and (qh >> 3) & 1 == 0
    assert (qh >> 31) & 1 == 1
    y = O.dequantize_row(O.Q5_0, q, 32)
    assert list(y[:4]) == [-16.0, 15.0, 0.0, -1.0]


def test_kat_vec_dot_q4_0_q8_0_by_hand():
    # weights: d=0.5, nibbles as KAT1 -> w = (l%16 - 8)*0.5 ; activations 1..32 scaled so d1 = 32/127
    xw = np.array([((l % 16) - 8) * 0.5 for l in range(32)] * 2, dtype=np.float32)
    xa = np.array(list(range(1, 33)) * 2, dtype=np.float32)
    wq = O.quantize_row(O.Q4_0, xw)
    aq = O.quantize_row(O.Q8_0, xa)
    got = O.vec_dot(O.Q4_0, 64, wq, aq)
    d1 = np.float32(32.0) / np.float32(127.0)
    idv = np.float32(1.0) / d1
    a_int = np.rint((xa[:32] * idv).astype(np.float32).astype(np.float64)).astype(np.int64)
    w_int = np.array([(l % 16) - 8 for l in range(32)], dtype=np.int64)
    sumi = int((a_int * w_int).sum())
    t = np.float32(np.float32(np.float32(0.5) * d1) * np.float32(sumi))
    want = np.float32(np.float32(0.0) + t)
    want = np.float32(want + t)
    assert got == want


# ---------------- half conversion vs numpy ----------------
def test_half_conversion_exhaustive_f16_and_sampled_f32():
    L = O.lib()
    hs = np.arange(65536, dtype=np.uint16)
    ref = hs.view(np.float16).astype(np.float32)
    got = np.array([L.oracle_f16_to_f32(int(h)) for h in hs], dtype=np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(got[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    assert np.all(np.isnan(got[nan]))
    xs = np.concatenate([_rand(20000), _rand(5000, 1e-6), _rand(5000, 1e4),
                         np.array([0.0, -0.0, 65504.0, 65520.0, 65519.9, 1e-8, 6e-8, 3e-8, 2.98e-8,
                                   np.inf, -np.inf, 0.00006103515625, 0.000061], dtype=np.float32)])
    # ties: halfway points between adjacent halves
    h = RNG.integers(0, 0x7BFF, size=4000).astype(np.uint16)
    lo = h.view(np.float16).astype(np.float64)
    hi = (h + 1).astype(np.uint16).view(np.float16).astype(np.float64)
    xs = np.concatenate([xs, ((lo + hi) / 2).astype(np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([L.oracle_f32_to_f16(float(v)) for v in xs], dtype=np.uint16)
    assert np.array_equal(got, want)


# ---------------- C oracle == numpy restatement, bit for bit ----------------
@pytest.mark.parametrize("name,t", [("q4_0", O.Q4_0), ("q4_1", O.Q4_1), ("q5_0", O.Q5_0), ("q8_0", O.Q8_0),
                                    ("q8_1", O.Q8_1), ("q4_2", O.Q4_2), ("q5_1", O.Q5_1)])
def test_quantize_matches_numpy(name, t):
    for scale in (1.0, 1e-3, 37.5):
        x = _rand((64, 256), scale)
        x[3, :32] = 0.0                      # an all-zero block (d = 0 -> id = 0)
        x[4, 32:64] = 2.5                    # constant block
        x[5, :64] = np.round(x[5, :64] * 2) / 2  # many exact .5 ties after scaling
        got = O.quantize_row(t, x).reshape(-1, O.type_size(t))
        want = R.QUANT[name](x)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("name,t", [("q4_0", O.Q4_0), ("q4_1", O.Q4_1), ("q5_0", O.Q5_0), ("q8_0", O.Q8_0),
                                    ("q4_2", O.Q4_2), ("q5_1", O.Q5_1)])
def test_dequantize_matches_numpy_and_roundtrip_idempotent(name, t):
    x = _rand((32, 128), 3.0)
    q = O.quantize_row(t, x)
    y = O.dequantize_row(t, q, 128)
    want = R.DEQUANT[name](q.reshape(-1, O.type_size(t))).reshape(32, 128)
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
    # quantisation error bound: half a step (the half-precision scale of Q4_2 adds 2^-11 relative)
    if name in ("q4_0", "q8_0", "q5_0", "q4_2"):
        bs = O.blck_size(t)
        amax = np.abs(x.reshape(-1, bs)).max(axis=1)
        step = amax / {"q4_0": 8, "q5_0": 16, "q8_0": 127, "q4_2": 8}[name] * (1.001 if name == "q4_2" else 1.0)
        err = np.abs(y - x).reshape(-1, bs).max(axis=1)
        assert np.all(err <= step * (1.0 if name != "q8_0" else 0.5) + 1e-6)
    # dequantize . quantize is idempotent on representable rows for the symmetric 8-bit format
    if name == "q8_0":
        q2 = O.quantize_row(t, y)
        y2 = O.dequantize_row(t, q2, 128)
        assert np.allclose(y2, y, rtol=0, atol=np.abs(y).max() * 2 ** -20)


@pytest.mark.parametrize("t,fn", [(O.Q4_0, R.vec_dot_q4_0_q8_0), (O.Q5_0, R.vec_dot_q5_0_q8_0),
                                  (O.Q8_0, R.vec_dot_q8_0_q8_0), (O.Q4_2, R.vec_dot_q4_2_q8_0),
                                  (O.Q5_1, R.vec_dot_q5_1_q8_1)])
def test_vec_dot_matches_numpy(t, fn):
    for n in (64, 256, 4096):
        w = _rand(n)
        a = _rand(n, 2.0)
        wq = O.quantize_row(t, w)
        aq = O.quantize_row(O.lib().oracle_vec_dot_type(t), a)
        if t == O.Q5_1:      # Q8_1 activations: checked bit for bit against the numpy form, exactness bound below is Q8_0's
            got = O.vec_dot(t, n, wq, aq)
            assert np.float32(got).view(np.uint32) == np.float32(fn(wq, aq)).view(np.uint32)
            continue
        got = O.vec_dot(t, n, wq, aq)
        want = fn(wq, aq)
        assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32)
        # and close to the mathematical product of the dequantised operands
        exact = float(O.dequantize_row(t, wq, n).astype(np.float64) @ O.dequantize_row(O.Q8_0, aq, n).astype(np.float64))
        assert abs(got - exact) <= 1e-5 * np.sqrt(n) * max(1.0, abs(exact))


def test_vec_dot_q4_1_q8_1_close_to_exact():
    n = 512
    w = _rand(n)
    a = _rand(n)
    wq = O.quantize_row(O.Q4_1, w)
    aq = O.quantize_row(O.Q8_1, a)
    got = O.vec_dot(O.Q4_1, n, wq, aq)
    wd = O.dequantize_row(O.Q4_1, wq, n).astype(np.float64)
    raw = aq.reshape(-1, 44)
    d = raw[:, :4].copy().view(np.float32).reshape(-1).astype(np.float64)
    qa = raw[:, 12:].copy().view(np.int8).astype(np.float64)
    ad = (qa * d[:, None]).reshape(-1)
    assert abs(got - float(wd @ ad)) <= 1e-4 * max(1.0, abs(float(wd @ ad)))
    # s0/s1 are d * signed half sums (D3)
    s0 = raw[:, 4:8].copy().view(np.float32).reshape(-1)
    assert np.allclose(s0, (d * qa[:, :16].sum(axis=1)).astype(np.float32), rtol=1e-6, atol=0)


def test_vec_dot_f32_f64_accumulate():
    x = _rand(4096)
    y = _rand(4096)
    s = np.zeros(1, dtype=np.float32)
    O.lib().oracle_vec_dot_f32(4096, s.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                               y.ctypes.data_as(C.c_void_p))
    assert s[0].view(np.uint32) == R.vec_dot_f32(x, y).view(np.uint32)


# ---------------- mul_mat driver ----------------
@pytest.mark.parametrize("t", [O.Q4_0, O.Q5_0, O.Q8_0, O.Q4_1, O.Q4_2, O.Q5_1])
@pytest.mark.parametrize("nth", [1, 3])
def test_mul_mat_q_composes_row_functions(t, nth):
    M, K, N = 24, 128, 5
    w = _rand((M, K))
    x = _rand((N, K))
    wq = O.quantize_row(t, w)
    dst = O.mul_mat(t, wq, x, M, K, N, nth=nth)[0, 0]
    vt = O.lib().oracle_vec_dot_type(t)
    xq = O.quantize_row(vt, x)
    for n in range(N):
        for m in range(M):
            assert dst[n, m].view(np.uint32) == np.float32(O.vec_dot(t, K, wq[m], xq[n])).view(np.uint32)


def test_mul_mat_f32_config1_shape_and_value():
    # BASELINE config 1: f32 64 x 128 x 256, "Test1-style" plumbing on the CPU path
    M, K, N = 64, 128, 256
    w = _rand((M, K))
    x = _rand((N, K))
    dst = O.mul_mat(O.F32, w, x, M, K, N, nth=4)[0, 0]
    want = (x.astype(np.float64) @ w.astype(np.float64).T)
    assert dst.shape == (N, M)
    assert np.allclose(dst, want, rtol=1e-5, atol=1e-5)
    assert dst[7, 3].view(np.uint32) == R.vec_dot_f32(w[3], x[7]).view(np.uint32)


def test_mul_mat_f16_and_batch_dims():
    M, K, N = 8, 64, 3
    w = _rand((2, 2, M, K)).astype(np.float16)
    x = _rand((2, 2, N, K))
    dst = O.mul_mat(O.F16, w.view(np.uint16), x, M, K, N, nth=2, ne2=2, ne3=2)
    xh = x.astype(np.float16).astype(np.float64)
    for i3 in range(2):
        for i2 in range(2):
            want = xh[i3, i2] @ w[i3, i2].astype(np.float64).T
            assert np.allclose(dst[i3, i2], want, rtol=1e-5, atol=1e-5)


def test_mul_mat_rejects_bad_types_and_shapes():
    L = O.lib()
    w = np.zeros(44 * 4, dtype=np.uint8)
    x = np.zeros((1, 128), dtype=np.float32)
    d = np.zeros((1, 1), dtype=np.float32)
    s0 = O.make_tensor(O.Q8_1, w, [128, 1])
    s1 = O.make_tensor(O.F32, x, [128, 1])
    dd = O.make_tensor(O.F32, d, [1, 1])
    work = np.zeros(4096, dtype=np.uint8)
    assert L.oracle_mul_mat(C.byref(s0), C.byref(s1), C.byref(dd), work.ctypes.data_as(C.c_void_p), 4096, 1) == -1
    s0b = O.make_tensor(O.Q4_0, w, [64, 1])
    assert L.oracle_mul_mat(C.byref(s0b), C.byref(s1), C.byref(dd), work.ctypes.data_as(C.c_void_p), 4096, 1) == -2


# ---------------- Test0's layout asserts (Test0/Program.cs:22-38) via the stride rule ----------------
def test_test0_layout_rule():
    t1 = O.make_tensor(O.F32, np.zeros(10, dtype=np.float32), [10])
    assert t1.nb[1] == 10 * 4
    t2 = O.make_tensor(O.I16, np.zeros(200, dtype=np.int16), [10, 20])
    assert t2.nb[1] == 10 * 2 and t2.nb[2] == 10 * 20 * 2
    t3 = O.make_tensor(O.I32, np.zeros(6000, dtype=np.int32), [10, 20, 30])
    assert t3.nb[1] == 40 and t3.nb[2] == 800 and t3.nb[3] == 24000


# ---------------------------------------------------------------- element-wise neighbours
def test_eltwise_restatements_match_numpy():
    rng = np.random.default_rng(11)
    x = rng.standard_normal((7, 96)).astype(np.float32)
    y = rng.standard_normal((7, 96)).astype(np.float32)
    assert np.array_equal(O.eltwise("add", x, y), x + y)                       # one IEEE f32 add per element
    assert np.array_equal(O.eltwise("mul", x, y), x * y)
    assert np.array_equal(O.eltwise("scale", x, v=0.3), x * np.float32(0.3))
    # rms_norm (Ggml.cs:5858-5920): f32 squares, f64 sequential sum, mean cast to f32, 1 / sqrtf(mean + 1e-6f)
    sq = (x * x).astype(np.float64)
    mean = np.array([np.float32(sum(row.tolist()) / 96.0) for row in sq], dtype=np.float32)     # sequential f64 sum
    scale = (np.float32(1.0) / np.sqrt(mean + np.float32(1e-6), dtype=np.float32)).astype(np.float32)
    assert np.array_equal(O.eltwise("rms_norm", x), x * scale[:, None])
    z = np.zeros((3, 32), np.float32)
    assert np.array_equal(O.eltwise("rms_norm", z), z)                          # 0 * 1000 = 0, no NaN


def _all_halves_as_f32():
    h = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    return h[np.isfinite(h)].astype(np.float32)


def test_silu_half_table_matches_numpy_for_every_half():
    """silu in the GGML_SILU_FP16 build (Ggml.cs:2737-2746 + table 1455-1471, A2 intent): 65536 possible arguments after the
    round to half -- every finite one is checked against an independent numpy evaluation (exp in f64 rounded to f32)."""
    f = _all_halves_as_f32()
    with np.errstate(over="ignore"):
        e = np.exp(-f.astype(np.float64)).astype(np.float32)
        s = (f / (np.float32(1.0) + e)).astype(np.float32)
    want = s.astype(np.float16).astype(np.float32)
    got = O.eltwise("silu", f.reshape(1, -1)).reshape(-1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # an f32 argument is rounded to half first: same result as its half neighbour
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((5, 64)) * 4).astype(np.float32)
    assert np.array_equal(O.eltwise("silu", x), O.eltwise("silu", x.astype(np.float16).astype(np.float32)))
    # known values: silu(0) = 0, silu(large) = x (as a half), silu(-large) = -0
    k = O.eltwise("silu", np.array([[0.0, 20.0, -20.0, 1.0]], np.float32))[0]
    assert k[0] == 0.0 and k[1] == 20.0 and abs(k[2]) < 1e-7 and abs(k[3] - 0.7310586) < 5e-4
