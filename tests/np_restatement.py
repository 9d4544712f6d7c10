"""Independent numpy restatement of the reference's block arithmetic (vectorised, written separately
from oracle/ggml_oracle.c) used to cross-check the C oracle bit for bit.  Test infrastructure only.

Follows /root/reference/GGMLSharp/Ggml.cs: quantize 334-377, 487-528, 547-590 + 672-714 (D7 intent), 609-653, 733-762 (D2 fixed),
781-823 (D3 fixed); dequantize 886-910, 962-987, 1025-1061, 1104-1122 (D4 signed); dots 1125-1162,
1165-1201, 1258-1301, 1351-1381; dense dots 2631-2651.  Rounding: Math.Round(double) = half-to-even
(np.rint), SURVEY.md 8.1 D1.
"""
import numpy as np

f32 = np.float32


def _blocks(x, qk=32):
    x = np.asarray(x, dtype=f32)
    assert x.shape[-1] % qk == 0
    return x.reshape(-1, qk)


def _first_absmax(b):
    """value of the first element attaining max |x| in each block (strict '<' scan, Ggml.cs:349)."""
    a = np.abs(b)
    idx = np.argmax(a, axis=1)  # argmax returns the first maximal index
    mx = b[np.arange(b.shape[0]), idx]
    # an all-(+-)zero block never passes 'amax < |v|', so max keeps its initial +0.0f (then d = -0.0f)
    return np.where(a.max(axis=1) > 0, mx, f32(0.0)).astype(f32)


def _inv(d):
    with np.errstate(divide="ignore"):
        return np.where(d != 0, f32(1.0) / d, f32(0.0)).astype(f32)


def quantize_q4_0(x):
    b = _blocks(x)
    mx = _first_absmax(b)
    d = (mx / f32(-8)).astype(f32)
    idv = _inv(d)
    v = (b * idv[:, None]).astype(f32)
    q = np.minimum(15.0, np.rint(v.astype(np.float64)) + 8.0).astype(np.int64).astype(np.uint8)
    qs = (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8)
    out = np.zeros((b.shape[0], 20), dtype=np.uint8)
    out[:, :4] = d.view(np.uint8).reshape(-1, 4)
    out[:, 4:] = qs
    return out


def dequantize_q4_0(raw):
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1, 20)
    d = raw[:, :4].copy().view(f32).reshape(-1)
    qs = raw[:, 4:]
    y = np.zeros((raw.shape[0], 32), dtype=f32)
    y[:, 0::2] = ((qs & 0x0F).astype(np.int32) - 8).astype(f32) * d[:, None]
    y[:, 1::2] = ((qs >> 4).astype(np.int32) - 8).astype(f32) * d[:, None]
    return y


def quantize_q4_1(x):
    b = _blocks(x)
    ar = np.arange(b.shape[0])
    mn = b[ar, np.argmin(b, axis=1)]   # first minimal element (strict '<' scan keeps the first; -0.0 vs 0.0)
    mxv = b[ar, np.argmax(b, axis=1)]
    d = ((mxv - mn).astype(f32) / f32(15)).astype(f32)
    idv = _inv(d)
    v = ((b - mn[:, None]).astype(f32) * idv[:, None]).astype(f32)
    q = np.rint(v.astype(np.float64)).astype(np.int64).astype(np.uint8)
    out = np.zeros((b.shape[0], 24), dtype=np.uint8)
    out[:, 0:4] = d.view(np.uint8).reshape(-1, 4)
    out[:, 4:8] = mn.astype(f32).view(np.uint8).reshape(-1, 4)
    out[:, 8:] = (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8)
    return out


def dequantize_q4_1(raw):
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1, 24)
    d = raw[:, 0:4].copy().view(f32).reshape(-1)
    m = raw[:, 4:8].copy().view(f32).reshape(-1)
    qs = raw[:, 8:]
    y = np.zeros((raw.shape[0], 32), dtype=f32)
    y[:, 0::2] = ((qs & 0x0F).astype(f32) * d[:, None]).astype(f32) + m[:, None]
    y[:, 1::2] = ((qs >> 4).astype(f32) * d[:, None]).astype(f32) + m[:, None]
    return y


def quantize_q5_0(x):
    b = _blocks(x)
    mx = _first_absmax(b)
    d = (mx / f32(-16)).astype(f32)
    idv = _inv(d)
    v = (b * idv[:, None]).astype(f32)
    q = np.minimum(31, (v + f32(16.5)).astype(f32).astype(np.int32)).astype(np.uint32)
    out = np.zeros((b.shape[0], 22), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    qh = np.zeros(b.shape[0], dtype=np.uint32)
    for l in range(32):
        qh |= ((q[:, l] >> 4) & 1) << np.uint32(l)
    out[:, 2:6] = qh.view(np.uint8).reshape(-1, 4)
    out[:, 6:] = ((q[:, 0::2] & 0xF) | ((q[:, 1::2] & 0xF) << 4)).astype(np.uint8)
    return out


def _q5_0_ints(raw):
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1, 22)
    d = raw[:, 0:2].copy().view(np.float16).reshape(-1).astype(f32)
    qh = raw[:, 2:6].copy().view(np.uint32).reshape(-1)
    qs = raw[:, 6:]
    q = np.zeros((raw.shape[0], 32), dtype=np.int32)
    q[:, 0::2] = qs & 0x0F
    q[:, 1::2] = qs >> 4
    bits = ((qh[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.int32)
    return d, (q | (bits << 4)) - 16


def dequantize_q5_0(raw):
    d, q = _q5_0_ints(raw)
    return (q.astype(f32) * d[:, None]).astype(f32)


def quantize_q8_0(x):
    b = _blocks(x)
    amax = np.abs(b).max(axis=1)
    d = (amax / f32(127)).astype(f32)
    idv = _inv(d)
    v = (b * idv[:, None]).astype(f32)
    q = np.rint(v.astype(np.float64)).astype(np.int64).astype(np.int8)
    out = np.zeros((b.shape[0], 36), dtype=np.uint8)
    out[:, :4] = d.view(np.uint8).reshape(-1, 4)
    out[:, 4:] = q.view(np.uint8)
    return out


def _q8_0_parts(raw):
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1, 36)
    d = raw[:, :4].copy().view(f32).reshape(-1)
    q = raw[:, 4:].copy().view(np.int8).astype(np.int32)
    return d, q


def dequantize_q8_0(raw):
    d, q = _q8_0_parts(raw)
    return (q.astype(f32) * d[:, None]).astype(f32)


def quantize_q8_1(x):
    b = _blocks(x)
    amax = np.abs(b).max(axis=1)
    d = (amax / f32(127)).astype(f32)
    idv = _inv(d)
    v = (b * idv[:, None]).astype(f32)
    q = np.rint(v.astype(np.float64)).astype(np.int64).astype(np.int8)
    s0 = (d * q[:, :16].astype(np.int32).sum(axis=1).astype(f32)).astype(f32)
    s1 = (d * q[:, 16:].astype(np.int32).sum(axis=1).astype(f32)).astype(f32)
    out = np.zeros((b.shape[0], 44), dtype=np.uint8)
    out[:, 0:4] = d.view(np.uint8).reshape(-1, 4)
    out[:, 4:8] = s0.view(np.uint8).reshape(-1, 4)
    out[:, 8:12] = s1.view(np.uint8).reshape(-1, 4)
    out[:, 12:] = q.view(np.uint8)
    return out


def _seq_sum_f32(terms):
    """sequential f32 accumulation in index order (the scalar loops of the reference)."""
    s = f32(0.0)
    for t in terms:
        s = f32(s + f32(t))
    return s


def vec_dot_q4_0_q8_0(xraw, yraw):
    xraw = np.asarray(xraw, dtype=np.uint8).reshape(-1, 20)
    d0 = xraw[:, :4].copy().view(f32).reshape(-1)
    qs = xraw[:, 4:]
    w = np.zeros((xraw.shape[0], 32), dtype=np.int32)
    w[:, 0::2] = (qs & 0x0F).astype(np.int32) - 8
    w[:, 1::2] = (qs >> 4).astype(np.int32) - 8
    d1, a = _q8_0_parts(yraw)
    sumi = (w * a).sum(axis=1)
    terms = ((d0 * d1).astype(f32) * sumi.astype(f32)).astype(f32)
    return _seq_sum_f32(terms)


def vec_dot_q5_0_q8_0(xraw, yraw):
    d, w = _q5_0_ints(xraw)
    d1, a = _q8_0_parts(yraw)
    sxy = (w * a).sum(axis=1)
    terms = ((d * sxy.astype(f32)).astype(f32) * d1).astype(f32)
    return _seq_sum_f32(terms)


def vec_dot_q8_0_q8_0(xraw, yraw):
    d0, w = _q8_0_parts(xraw)
    d1, a = _q8_0_parts(yraw)
    sumi = (w * a).sum(axis=1)
    terms = ((d0 * d1).astype(f32) * sumi.astype(f32)).astype(f32)
    return _seq_sum_f32(terms)


def vec_dot_f32(x, y):
    p = (np.asarray(x, dtype=f32) * np.asarray(y, dtype=f32)).astype(f32).astype(np.float64)
    s = np.float64(0.0)
    for t in p:
        s += t
    return f32(s)


def mul_mat_exact(wdeq, aq_deq):
    """float64 product of dequantised weights [M,K] and dequantised Q8 activations [N,K] -> [N,M]."""
    return aq_deq.astype(np.float64) @ wdeq.astype(np.float64).T


# ---- Q4_2 / Q5_1 (intent: the half scales are IEEE bit patterns as in the upstream scalar code) ----
def quantize_q4_2(x):
    """Ggml.cs:547-590: blocks of 16, d = max/-8, nibbles min(15, round(x*id) + 8)"""
    b = _blocks(x, 16)
    mx = _first_absmax(b)
    d = (mx / f32(-8)).astype(f32)
    idv = _inv(d)
    v = (b * idv[:, None]).astype(f32)
    q = np.minimum(15.0, np.rint(v.astype(np.float64)) + 8.0).astype(np.int64).astype(np.uint8)
    out = np.zeros((b.shape[0], 10), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = (q[:, 0::2] | (q[:, 1::2] << 4)).astype(np.uint8)
    return out


def _q4_2_ints(raw):
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1, 10)
    d = raw[:, 0:2].copy().view(np.float16).reshape(-1).astype(f32)
    qs = raw[:, 2:]
    q = np.zeros((raw.shape[0], 16), dtype=np.int32)
    q[:, 0::2] = (qs & 0x0F).astype(np.int32) - 8
    q[:, 1::2] = (qs >> 4).astype(np.int32) - 8
    return d, q


def dequantize_q4_2(raw):
    d, q = _q4_2_ints(raw)
    return (q.astype(f32) * d[:, None]).astype(f32)


def quantize_q5_1(x):
    """Ggml.cs:672-714: d = (max - min)/31, q = (uint)((x - min)*id + 0.5f), low nibble + fifth bit in qh"""
    b = _blocks(x)
    ar = np.arange(b.shape[0])
    mn = b[ar, np.argmin(b, axis=1)]      # strict '<' scans keep the FIRST extreme element (matters for -0.0 vs +0.0)
    mx = b[ar, np.argmax(b, axis=1)]
    d = ((mx - mn).astype(f32) / f32(31)).astype(f32)
    idv = _inv(d)
    v = ((b - mn[:, None]).astype(f32) * idv[:, None]).astype(f32)
    q = (v + f32(0.5)).astype(f32).astype(np.int64).astype(np.uint32)
    out = np.zeros((b.shape[0], 24), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:4] = mn.astype(np.float16).view(np.uint8).reshape(-1, 2)
    qh = np.zeros(b.shape[0], dtype=np.uint32)
    for l in range(32):
        qh |= ((q[:, l] >> 4) & 1) << np.uint32(l)
    out[:, 4:8] = qh.view(np.uint8).reshape(-1, 4)
    out[:, 8:] = ((q[:, 0::2] & 0xF) | ((q[:, 1::2] & 0xF) << 4)).astype(np.uint8)
    return out


def _q5_1_parts(raw):
    raw = np.asarray(raw, dtype=np.uint8).reshape(-1, 24)
    d = raw[:, 0:2].copy().view(np.float16).reshape(-1).astype(f32)
    m = raw[:, 2:4].copy().view(np.float16).reshape(-1).astype(f32)
    qh = raw[:, 4:8].copy().view(np.uint32).reshape(-1)
    qs = raw[:, 8:]
    q = np.zeros((raw.shape[0], 32), dtype=np.int32)
    q[:, 0::2] = qs & 0x0F
    q[:, 1::2] = qs >> 4
    bits = ((qh[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.int32)
    return d, m, q | (bits << 4)


def dequantize_q5_1(raw):
    d, m, q = _q5_1_parts(raw)
    return ((q.astype(f32) * d[:, None]).astype(f32) + m[:, None]).astype(f32)


def vec_dot_q4_2_q8_0(xraw, yraw):
    """Ggml.cs:1204-1255: two 16-element blocks per Q8_0 block, each with its own scale, added in turn"""
    d, w = _q4_2_ints(xraw)
    d1, a = _q8_0_parts(yraw)
    sumi = (w.reshape(-1, 2, 16) * a.reshape(-1, 2, 16)).sum(axis=2)            # [nb][2]
    dd = d.reshape(-1, 2)
    terms = ((dd * d1[:, None]).astype(f32) * sumi.astype(f32)).astype(f32)      # (d0 * y.d) * sumi_0, (d1 * y.d) * sumi_1
    return _seq_sum_f32(terms.reshape(-1))


def vec_dot_q5_1_q8_1(xraw, yraw):
    """Ggml.cs:1304-1348: sumf += (d * sxy) * y.d + m * (y.s0 + y.s1)"""
    d, m, w = _q5_1_parts(xraw)
    yraw = np.asarray(yraw, dtype=np.uint8).reshape(-1, 44)
    yd = yraw[:, 0:4].copy().view(f32).reshape(-1)
    s0 = yraw[:, 4:8].copy().view(f32).reshape(-1)
    s1 = yraw[:, 8:12].copy().view(f32).reshape(-1)
    a = yraw[:, 12:].copy().view(np.int8).astype(np.int32)
    sxy = (w * a).sum(axis=1)
    terms = (((d * sxy.astype(f32)).astype(f32) * yd).astype(f32) + (m * (s0 + s1).astype(f32)).astype(f32)).astype(f32)
    return _seq_sum_f32(terms)


QUANT = {"q4_0": quantize_q4_0, "q4_1": quantize_q4_1, "q5_0": quantize_q5_0, "q8_0": quantize_q8_0,
         "q8_1": quantize_q8_1, "q4_2": quantize_q4_2, "q5_1": quantize_q5_1}
DEQUANT = {"q4_0": dequantize_q4_0, "q4_1": dequantize_q4_1, "q5_0": dequantize_q5_0, "q8_0": dequantize_q8_0,
           "q4_2": dequantize_q4_2, "q5_1": dequantize_q5_1}
