"""Known-answer tests that pin the oracle for the types round 1 left thin: Q4_1, Q5_1, Q8_1, Q4_2
quantizers and the dot products of Q4_1, Q5_0, Q5_1, Q8_0 -- one block each, with a rounding tie and a negative-max case.

Two independent routes per KAT, neither of which is the oracle:
  1. the expected bytes / value as a LITERAL, derived by hand from the cited C# lines;
  2. an evaluation of the same C# statements in exact rational arithmetic (python `fractions`), rounding to binary32
     (round-to-nearest-even on the exact value -- no double rounding through binary64) only where the C# statement
     produces a `float`, and with Math.Round = half-to-even / `(uint)(v + 0.5f)` = truncation where the C#
     has that instead.
The oracle must agree with both.  (The reference holds no vector for these functions -- SURVEY 8(c): parity stays
"unpinned by the reference"; this is the only pin available.)"""
import struct
from fractions import Fraction as Fr

import numpy as np

import oracle_lib as O


# ---------------------------------------------------------------- exact binary32 arithmetic on Fractions
def f32(x):
    """round a Fraction to the nearest binary32 (ties to even), returned as a Fraction (normal range + zero only)"""
    x = Fr(x)
    if x == 0:
        return Fr(0)
    s = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fr(2) ** e > a:
        e -= 1
    assert Fr(2) ** e <= a < Fr(2) ** (e + 1) and -126 <= e <= 127
    q = a / Fr(2) ** (e - 23)                        # in [2^23, 2^24)
    n, r = divmod(q.numerator, q.denominator)
    twice = 2 * r
    if twice > q.denominator or (twice == q.denominator and (n & 1)):
        n += 1
    return s * n * Fr(2) ** (e - 23)


def f32_bytes(x):
    return struct.pack("<f", float(x))               # exact: x is a binary32 value


def f16_bits(x):
    """binary32 value (as a Fraction that is exactly representable in binary16 in these KATs) -> half bit pattern"""
    h = np.float16(float(x))
    assert Fr(float(h)) == x, "KAT scales are chosen to be exact halves"
    return int(h.view(np.uint16))


def round_half_even(x):                              # Math.Round(double)
    n = x.numerator // x.denominator                 # floor
    r = x - n
    if r > Fr(1, 2) or (r == Fr(1, 2) and (n & 1)):
        n += 1
    return n


def trunc(x):                                        # (uint)(float) / (int)(float) for the non-negative values used here
    assert x >= 0
    return x.numerator // x.denominator


def fr_list(xs):
    return [Fr(x) for x in xs]


def as_f32_array(xs):
    a = np.array([float(x) for x in xs], dtype=np.float32)
    assert all(Fr(float(v)) == Fr(x) for v, x in zip(a, xs)), "KAT inputs are exact binary32 values"
    return a


# ---------------------------------------------------------------- second route: the C# statements in rationals
def q4_1_block(x):     # Ggml.cs:487-528
    mn, mx = min(x), max(x)
    d = f32(f32(mx - mn) / 15)
    idv = f32(Fr(1) / d) if d != 0 else Fr(0)
    q = [round_half_even(f32(f32(v - mn) * idv)) for v in x]
    assert all(0 <= v < 16 for v in q)
    return f32_bytes(d) + f32_bytes(mn) + bytes(q[l] | (q[l + 1] << 4) for l in range(0, 32, 2))


def q5_1_block(x):     # Ggml.cs:672-714 (half scales as IEEE bit patterns: SURVEY D7 intent)
    mn, mx = min(x), max(x)
    d = f32(f32(mx - mn) / 31)
    idv = f32(Fr(1) / d) if d != 0 else Fr(0)
    q = [trunc(f32(f32(f32(v - mn) * idv) + Fr(1, 2))) for v in x]          # (uint)(v + 0.5f)
    qh = 0
    for l, v in enumerate(q):
        qh |= ((v & 0x10) >> 4) << l
    return struct.pack("<HH", f16_bits(d), f16_bits(mn)) + struct.pack("<I", qh) + bytes((q[l] & 15) | ((q[l + 1] & 15) << 4) for l in range(0, 32, 2))


def q8_1_block(x):     # Ggml.cs:781-823 with the loop the upstream code has (SURVEY D3 intent): l < 16, ++l, signed sums
    amax = max(abs(v) for v in x)
    d = f32(amax / 127)
    idv = f32(Fr(1) / d) if d != 0 else Fr(0)
    q = [round_half_even(f32(v * idv)) for v in x]
    s0 = f32(d * sum(q[:16]))
    s1 = f32(d * sum(q[16:]))
    return f32_bytes(d) + f32_bytes(s0) + f32_bytes(s1) + bytes(v & 0xFF for v in q), d, q


def q8_0_block(x):     # Ggml.cs:733-762 (SURVEY D2 intent: every l)
    amax = max(abs(v) for v in x)
    d = f32(amax / 127)
    idv = f32(Fr(1) / d) if d != 0 else Fr(0)
    q = [round_half_even(f32(v * idv)) for v in x]
    return f32_bytes(d) + bytes(v & 0xFF for v in q), d, q


def q4_2_block(x):     # Ggml.cs:547-590, 16 elements, half scale (D7 intent)
    amax, mx = Fr(0), Fr(0)
    for v in x:
        if amax < abs(v):
            amax, mx = abs(v), v
    d = f32(mx / -8)
    idv = f32(Fr(1) / d) if d != 0 else Fr(0)
    q = [min(15, round_half_even(f32(v * idv)) + 8) for v in x]
    assert all(0 <= v < 16 for v in q)
    return struct.pack("<H", f16_bits(d)) + bytes(q[l] | (q[l + 1] << 4) for l in range(0, 16, 2))


def q5_0_block(x):     # Ggml.cs:609-653
    amax, mx = Fr(0), Fr(0)
    for v in x:
        if amax < abs(v):
            amax, mx = abs(v), v
    d = f32(mx / -16)
    idv = f32(Fr(1) / d) if d != 0 else Fr(0)
    q = [min(31, trunc(f32(f32(v * idv) + Fr(33, 2)))) for v in x]          # (int)(x*id + 16.5f)
    qh = 0
    for l, v in enumerate(q):
        qh |= ((v & 0x10) >> 4) << l
    return struct.pack("<H", f16_bits(d)) + struct.pack("<I", qh) + bytes((q[l] & 15) | ((q[l + 1] & 15) << 4) for l in range(0, 32, 2)), d, q


# ---------------------------------------------------------------- quantizer KATs
def test_kat_q4_1_tie_and_negative_min():
    # min = -1, max = 2.75 -> d = 0.25, id = 4: (x - min) * id = 0, 15, 0.5 (tie -> 0), 1.5 (-> 2), 2.5 (tie -> 2), 4, 3.5 (tie -> 4), 7 ...
    x = fr_list([-1, Fr(11, 4), Fr(-7, 8), Fr(-5, 8), Fr(-3, 8), 0, Fr(-1, 8), Fr(3, 4)] + [Fr(k % 16, 4) - 1 for k in range(24)])
    lit = bytes.fromhex("0000803e" "000080bf" "f0" "20" "42" "74" "10" "32" "54" "76" "98" "ba" "dc" "fe" "10" "32" "54" "76")
    assert q4_1_block(x) == lit
    assert O.quantize_row(O.Q4_1, as_f32_array(x)).tobytes() == lit
    y = O.dequantize_row(O.Q4_1, np.frombuffer(lit, dtype=np.uint8), 32)      # nib * d + m (Ggml.cs:962-987)
    assert list(y[:8]) == [-1.0, 2.75, -1.0, -0.5, -0.5, 0.0, 0.0, 0.75]


def test_kat_q5_1_truncating_round_and_high_bits():
    # min = -2, max = 5.75 -> d = 0.25 (half 0x3400), m = -2 (half 0xC000); (uint)(v + 0.5f): 0.5 -> 1 (NOT to even)
    x = fr_list([-2, Fr(23, 4), Fr(-15, 8), Fr(-13, 8), 2, Fr(17, 8), Fr(11, 2), 0] + [Fr(k, 4) - 2 for k in range(24)])
    q = [0, 31, 1, 2, 16, 17, 30, 8] + list(range(24))
    qh = sum(((v >> 4) & 1) << l for l, v in enumerate(q))
    lit = struct.pack("<HHI", 0x3400, 0xC000, qh) + bytes((q[l] & 15) | ((q[l + 1] & 15) << 4) for l in range(0, 32, 2))
    assert lit.hex() == "003400c0" "720000ff" "f0" "21" "10" "8e" "10" "32" "54" "76" "98" "ba" "dc" "fe" "10" "32" "54" "76"
    assert q5_1_block(x) == lit
    assert O.quantize_row(O.Q5_1, as_f32_array(x)).tobytes() == lit


def test_kat_q8_1_ties_signed_sums():
    # amax = 127 -> d = 1; ties to even, negatives signed (D4); s0 / s1 = d * the signed sums of the two halves (D3)
    x = fr_list([127, -127, Fr(1, 2), Fr(3, 2), Fr(5, 2), Fr(-1, 2), Fr(-3, 2), Fr(-5, 2), 3, -4, 0, 0, 0, 0, 0, 0] +
                [Fr(7, 2), Fr(-7, 2), 100, -90, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, Fr(9, 2)])
    blk, d, q = q8_1_block(x)
    assert q[:10] == [127, -127, 0, 2, 2, 0, -2, -2, 3, -4] and q[16:20] == [4, -4, 100, -90] and q[31] == 4
    lit = struct.pack("<fff", 1.0, -1.0, 18.0) + bytes(v & 0xFF for v in q)
    assert blk == lit
    assert O.quantize_row(O.Q8_1, as_f32_array(x)).tobytes() == lit


def test_kat_q8_1_inexact_scale():
    # amax = 100 -> d = 100/127 rounded to binary32, id = 1/d rounded again: quants follow THOSE roundings
    x = fr_list([100, -50, 25, Fr(25, 2), 1, -1, Fr(1, 2), 0] * 4)
    blk, d, q = q8_1_block(x)
    assert f32_bytes(d).hex() == "2693493f"
    assert q[:8] == [127, -64, 32, 16, 1, -1, 1, 0]            # -50 -> -63.5 -> -64 (even); 25 -> 31.75 -> 32; 12.5 -> 15.875 -> 16
    assert O.quantize_row(O.Q8_1, as_f32_array(x)).tobytes() == blk


def test_kat_q4_2_negative_and_positive_max():
    # block 0: the first largest |x| is -4 -> d = 0.5 (half 0x3800), id = 2: round(2x) + 8; +4 clamps to 15; ties to even
    b0 = fr_list([-4, 4, Fr(1, 4), Fr(3, 4), Fr(5, 4), Fr(-1, 4), Fr(-3, 4), 0, 1, -1, 2, -2, 3, -3, Fr(7, 2), Fr(-7, 2)])
    # block 1: max is +2 -> d = -0.25 (half 0xB400), id = -4: round(-4x) + 8
    b1 = fr_list([2, -2, 1, -1, Fr(1, 8), Fr(3, 8), Fr(-1, 8), 0] * 2)
    lit0 = bytes.fromhex("0038" "f0" "a8" "8a" "86" "6a" "4c" "2e" "1f")
    lit1 = bytes.fromhex("00b4" "f0" "c4" "68" "88" "f0" "c4" "68" "88")
    assert q4_2_block(b0) == lit0 and q4_2_block(b1) == lit1
    assert O.quantize_row(O.Q4_2, as_f32_array(b0 + b1)).tobytes() == lit0 + lit1


# ---------------------------------------------------------------- dot-product KATs (float statement order as written)
def _two_blocks(a, b):
    return np.frombuffer(a + b, dtype=np.uint8)


def test_kat_vec_dot_q4_1_q8_1_per_element_float_chain():
    """Ggml.cs:1176-1198: f0 = d0*nib + m0, f2 = d1*q, sumf += f0*f2 + f1*f3 -- every operation a binary32 rounding."""
    xw = fr_list([-1, Fr(11, 4), Fr(-7, 8), Fr(-5, 8), Fr(-3, 8), 0, Fr(-1, 8), Fr(3, 4)] + [Fr(k % 16, 4) - 1 for k in range(24)])
    xa = fr_list([100, -50, 25, Fr(25, 2), 1, -1, Fr(1, 2), 0] * 4)
    wblk = q4_1_block(xw)
    ablk, d1, qa = q8_1_block(xa)
    d0, m0 = Fr(1, 4), Fr(-1)
    nib = [b for byte in wblk[8:] for b in (byte & 15, byte >> 4)]
    sumf = Fr(0)
    for _ in range(2):                                   # two identical blocks (nb % 2 == 0, Ggml.cs:1170)
        for j in range(16):
            f0 = f32(f32(d0 * nib[2 * j]) + m0)
            f1 = f32(f32(d0 * nib[2 * j + 1]) + m0)
            f2 = f32(d1 * qa[2 * j])
            f3 = f32(d1 * qa[2 * j + 1])
            sumf = f32(sumf + f32(f32(f0 * f2) + f32(f1 * f3)))
    # third route: the same statement sequence in numpy binary32 scalars (every numpy float32 operation rounds once)
    F = np.float32
    s32 = F(0)
    for _ in range(2):
        for j in range(16):
            g0 = F(F(0.25) * F(nib[2 * j])) + F(-1)
            g1 = F(F(0.25) * F(nib[2 * j + 1])) + F(-1)
            g2 = F(float(d1)) * F(qa[2 * j])
            g3 = F(float(d1)) * F(qa[2 * j + 1])
            s32 = F(s32 + F(F(g0 * g2) + F(g1 * g3)))
    got = O.vec_dot(O.Q4_1, 64, _two_blocks(wblk, wblk), _two_blocks(ablk, ablk))
    assert Fr(float(got)) == sumf == Fr(float(s32))
    assert struct.pack("<f", got).hex() == f32_bytes(sumf).hex() == "5cae23c4"       # -654.7244...


def test_kat_vec_dot_q5_0_q8_0():
    """Ggml.cs:1270-1298: sumf += (d * sxy) * y.d with sxy the exact integer dot of (5-bit value - 16) and the signed quants."""
    xw = fr_list([-16, 15, Fr(2, 5), Fr(-3, 5), 8, -8, Fr(31, 2), 0] + [Fr(k) - 12 for k in range(24)])
    xa = fr_list([127, -127, Fr(1, 2), Fr(3, 2), Fr(5, 2), Fr(-1, 2), 64, -64] + [Fr(k - 10) for k in range(24)])
    # (2/5 and -3/5 are not binary32 values: use their roundings as the inputs)
    xw = [f32(v) for v in xw]
    wblk, dw, qw = q5_0_block(xw)
    ablk, da, qa = q8_0_block(xa)
    assert dw == 1 and da == 1 and qw[:8] == [0, 31, 16, 15, 24, 8, 31, 16]
    sxy = sum((w - 16) * a for w, a in zip(qw, qa))
    # by hand: -16*127 + 15*(-127) + (-1)*2 + 8*2 + 15*64 = -2963, plus sum_{k<24} (k-12)(k-10) = 4324 - 6072 + 2880 = 1132
    assert sxy == -2963 + 1132 == -1831
    t = f32(f32(dw * sxy) * da)
    sumf = f32(f32(Fr(0) + t) + t)
    got = O.vec_dot(O.Q5_0, 64, _two_blocks(wblk, wblk), _two_blocks(ablk, ablk))
    assert Fr(float(got)) == sumf == Fr(-3662)


def test_kat_vec_dot_q5_1_q8_1():
    """Ggml.cs:1318-1344: sumf += (d * sxy) * y.d + m * (y.s0 + y.s1), unsigned 5-bit values."""
    xw = fr_list([-2, Fr(23, 4), Fr(-15, 8), Fr(-13, 8), 2, Fr(17, 8), Fr(11, 2), 0] + [Fr(k, 4) - 2 for k in range(24)])
    xa = fr_list([100, -50, 25, Fr(25, 2), 1, -1, Fr(1, 2), 0] * 4)
    wblk = q5_1_block(xw)
    ablk, d1, qa = q8_1_block(xa)
    qw = [0, 31, 1, 2, 16, 17, 30, 8] + list(range(24))
    d, m = Fr(1, 4), Fr(-2)
    sxy = sum(w * a for w, a in zip(qw, qa))
    s0, s1 = f32(d1 * sum(qa[:16])), f32(d1 * sum(qa[16:]))
    term = f32(f32(f32(d * sxy) * d1) + f32(m * f32(s0 + s1)))
    sumf = f32(f32(Fr(0) + term) + term)
    got = O.vec_dot(O.Q5_1, 64, _two_blocks(wblk, wblk), _two_blocks(ablk, ablk))
    assert Fr(float(got)) == sumf
    assert struct.pack("<f", got).hex() == f32_bytes(sumf).hex()


def test_kat_vec_dot_q8_0_q8_0_signed_bytes():
    """Ggml.cs:1362-1378 with signed quants: sumf += (x.d * y.d) * sumi."""
    xw = fr_list([127, -127, 1, -1, 64, -64, Fr(1, 2), Fr(-3, 2)] + [Fr(k - 12) for k in range(24)])
    xa = fr_list([100, -50, 25, Fr(25, 2), 1, -1, Fr(1, 2), 0] * 4)
    wblk, dw, qw = q8_0_block(xw)
    ablk, da, qa = q8_0_block(xa)
    assert dw == 1 and qw[:8] == [127, -127, 1, -1, 64, -64, 0, -2]
    sumi = sum(w * a for w, a in zip(qw, qa))
    t = f32(f32(dw * da) * sumi)
    sumf = f32(f32(Fr(0) + t) + t)
    got = O.vec_dot(O.Q8_0, 64, _two_blocks(wblk, wblk), _two_blocks(ablk, ablk))
    assert Fr(float(got)) == sumf
    # a reading of the quants as UNSIGNED bytes (the C# as written, D4) would give a different sum
    sumi_unsigned = sum((w & 0xFF) * (a & 0xFF) for w, a in zip(qw, qa))
    assert sumi_unsigned != sumi


def test_f32_rounding_helper_against_numpy():
    rng = np.random.default_rng(11)
    for _ in range(2000):
        a, b = rng.standard_normal(2).astype(np.float32)
        for op, want in (((Fr(float(a)) * Fr(float(b))), np.float32(a) * np.float32(b)),
                         ((Fr(float(a)) + Fr(float(b))), np.float32(a) + np.float32(b)),
                         ((Fr(float(a)) / Fr(float(b))), np.float32(a) / np.float32(b))):
            if want == 0 or not np.isfinite(want) or abs(float(want)) < 1e-30:
                continue
            assert float(f32(op)) == float(want)
