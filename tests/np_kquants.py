"""numpy restatement of the PUBLISHED upstream k-quant algorithms that the Q5_K extension follows (ggml k_quants.c,
2023-06: block_q5_K, block_q8_K, dequantize_row_q5_K, quantize_row_q8_K_reference, ggml_vec_dot_q5_K_q8_K scalar path).

TEST INFRASTRUCTURE, and the only checker there is: the reference (kant2002/GGMLSharp) holds no k-quants at all
(SURVEY 8(a) row K), the upstream source is not vendored anywhere in /root/reference, nothing here was run against it --
PARITY UNPINNED.  `quantize_q5_K` below is NOT upstream's search-based quantizer (make_qkx1_quants): it is a simple valid
encoder used to make test weights; the format's meaning is fixed by dequantize_q5_K alone."""
import numpy as np

QK_K = 256
Q5K_BYTES = 176
Q8K_BYTES = 292


def unpack_scales(scales):
    """get_scale_min_k4 for j = 0..7: scales [..., 12] uint8 -> (sc [..., 8], m [..., 8]) 6-bit values"""
    q = scales.astype(np.uint32)
    sc = np.empty(q.shape[:-1] + (8,), dtype=np.uint32)
    m = np.empty_like(sc)
    for j in range(4):
        sc[..., j] = q[..., j] & 63
        m[..., j] = q[..., j + 4] & 63
    for j in range(4, 8):
        sc[..., j] = (q[..., j + 4] & 0xF) | ((q[..., j - 4] >> 6) << 4)
        m[..., j] = (q[..., j + 4] >> 4) | ((q[..., j] >> 6) << 4)
    return sc, m


def pack_scales(sc, m):
    """inverse of unpack_scales (sc, m in 0..63) -> [..., 12] uint8"""
    sc = sc.astype(np.uint32)
    m = m.astype(np.uint32)
    out = np.zeros(sc.shape[:-1] + (12,), dtype=np.uint32)
    for j in range(4):
        out[..., j] = (sc[..., j] & 63) | ((sc[..., j + 4] >> 4) << 6)
        out[..., j + 4] = (m[..., j] & 63) | ((m[..., j + 4] >> 4) << 6)
        out[..., j + 8] = (sc[..., j + 4] & 0xF) | ((m[..., j + 4] & 0xF) << 4)
    return out.astype(np.uint8)


def q5_values(blocks):
    """blocks [nb, 176] uint8 -> the 5-bit values [nb, 256] in element order (dequantize_row_q5_K's unpacking)"""
    qh = blocks[:, 16:48].astype(np.uint32)           # [nb, 32]
    qs = blocks[:, 48:176].astype(np.uint32).reshape(-1, 4, 32)
    out = np.empty((blocks.shape[0], 8, 32), dtype=np.uint32)
    for g in range(4):
        out[:, 2 * g] = (qs[:, g] & 0xF) + (((qh >> (2 * g)) & 1) << 4)
        out[:, 2 * g + 1] = (qs[:, g] >> 4) + (((qh >> (2 * g + 1)) & 1) << 4)
    return out.reshape(-1, 256)


def dequantize_q5_K(blocks):
    """[nb, 176] uint8 -> [nb, 256] float32: y = (d * sc) * q - (dmin * m), every operation a binary32 rounding"""
    d = blocks[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(-1, 1)
    dmin = blocks[:, 2:4].copy().view(np.float16).astype(np.float32).reshape(-1, 1)
    sc, m = unpack_scales(blocks[:, 4:16])
    d1 = (d * sc.astype(np.float32)).astype(np.float32)          # [nb, 8]
    m1 = (dmin * m.astype(np.float32)).astype(np.float32)
    q = q5_values(blocks).reshape(-1, 8, 32).astype(np.float32)
    y = ((d1[:, :, None] * q).astype(np.float32) - m1[:, :, None]).astype(np.float32)
    return y.reshape(-1, 256)


def quantize_q8_K(x):
    """quantize_row_q8_K_reference: x [n, 256] float32 -> (d [n] f32, qs [n, 256] int8, bsums [n, 16] int16)"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    ax = np.abs(x)
    idx = np.argmax(ax, axis=1)                                   # first element of largest magnitude
    mx = x[np.arange(x.shape[0]), idx]
    amax = ax[np.arange(x.shape[0]), idx]
    with np.errstate(divide="ignore", invalid="ignore"):
        iscale = np.where(amax != 0, np.float32(-128.0) / mx, np.float32(0)).astype(np.float32)
        d = np.where(amax != 0, np.float32(1.0) / iscale, np.float32(0)).astype(np.float32)
    v = np.rint((x * iscale[:, None]).astype(np.float32))          # nearest_int: round half to even
    q = np.minimum(v, 127).astype(np.int32)
    bsums = q.reshape(-1, 16, 16).sum(axis=2).astype(np.int16)
    return d, q.astype(np.int8), bsums


def vec_dot_q5_K_q8_K(wblocks, d8, q8, bsums):
    """one weight row [nb, 176] against one activation row (d8 [nb], q8 [nb, 256], bsums [nb, 16]) -> float32.
    Per super-block, in order: sumf += (d * dy) * sum_j sc_j <q_j, a_j>  -  (dmin * dy) * sum_j m_j (bsum_2j + bsum_2j+1),
    the integer sums exact (upstream's scalar path spreads the first term over 8 partial lanes; only the order of its
    f32 additions differs)."""
    F = np.float32
    d = wblocks[:, 0:2].copy().view(np.float16).astype(F).reshape(-1)
    dmin = wblocks[:, 2:4].copy().view(np.float16).astype(F).reshape(-1)
    sc, m = unpack_scales(wblocks[:, 4:16])
    q5 = q5_values(wblocks).astype(np.int64).reshape(-1, 8, 32)
    a = q8.astype(np.int64).reshape(-1, 8, 32)
    isum = (sc.astype(np.int64) * (q5 * a).sum(axis=2)).sum(axis=1)
    bs = bsums.astype(np.int64).reshape(-1, 8, 2).sum(axis=2)
    msum = (m.astype(np.int64) * bs).sum(axis=1)
    sumf = F(0)
    for i in range(wblocks.shape[0]):
        sumf = F(sumf + F(F(d[i] * d8[i]) * F(isum[i])))
        sumf = F(sumf - F(F(dmin[i] * d8[i]) * F(msum[i])))
    return sumf


def mul_mat_q5_K(wrows, x):
    """wrows [M, K/256*176] uint8, x [N, K] f32 -> [N, M] f32 (the reference's dst layout), activations by the Q8_K rule"""
    M = wrows.shape[0]
    N, K = x.shape
    nb = K // 256
    d8, q8, bs = quantize_q8_K(x.reshape(-1, 256))
    d8, q8, bs = d8.reshape(N, nb), q8.reshape(N, nb, 256), bs.reshape(N, nb, 16)
    w = wrows.reshape(M, nb, 176)
    # vectorised over (n, m): exact integer sums in int64, then the two float terms per super-block in f64 (a checker
    # for 1e-3-relative parity, not a bit-level one)
    dw = w[:, :, 0:2].copy().view(np.float16).astype(np.float64).reshape(M, nb)
    dmin = w[:, :, 2:4].copy().view(np.float16).astype(np.float64).reshape(M, nb)
    sc, mn = unpack_scales(w[:, :, 4:16])
    q5 = q5_values(w.reshape(-1, 176)).astype(np.float64).reshape(M, nb, 8, 32)
    a = q8.astype(np.float64).reshape(N, nb, 8, 32)
    dots = np.einsum("mbjl,nbjl->nmbj", q5, a)                      # exact in f64 (|.| < 2^24)
    isum = (dots * sc.astype(np.float64)[None]).sum(axis=3)          # [N, M, nb]
    bsum = bs.astype(np.float64).reshape(N, nb, 8, 2).sum(axis=3)
    msum = np.einsum("mbj,nbj->nmb", mn.astype(np.float64), bsum)
    out = (dw[None] * d8.astype(np.float64)[:, None, :] * isum - dmin[None] * d8.astype(np.float64)[:, None, :] * msum).sum(axis=2)
    return out.astype(np.float32)


def quantize_q5_K(x):
    """A simple VALID Q5_K encoder for test data (not upstream's): per 32-element sub-block an affine 5-bit code over
    [min(0, min x), max x], its scale and offset re-quantized to 6 bits against the super-block's d / dmin."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 8, 32)
    lo = np.minimum(x.min(axis=2), 0.0)                            # the format subtracts a non-negative offset
    hi = np.maximum(x.max(axis=2), lo + 1e-30)
    scale = (hi - lo) / 31.0
    off = -lo
    d = np.maximum(scale.max(axis=1), 1e-30) / 63.0
    dmin = np.maximum(off.max(axis=1), 1e-30) / 63.0
    d16 = d.astype(np.float16)
    dmin16 = dmin.astype(np.float16)
    df, dminf = d16.astype(np.float32), dmin16.astype(np.float32)
    sc = np.clip(np.rint(scale / np.maximum(df[:, None], 1e-30)), 1, 63).astype(np.uint32)
    m = np.clip(np.rint(off / np.maximum(dminf[:, None], 1e-30)), 0, 63).astype(np.uint32)
    d1 = df[:, None] * sc
    m1 = dminf[:, None] * m
    q = np.clip(np.rint((x + m1[:, :, None]) / np.maximum(d1[:, :, None], 1e-30)), 0, 31).astype(np.uint32)
    nb = x.shape[0]
    blocks = np.zeros((nb, 176), dtype=np.uint8)
    blocks[:, 0:2] = d16.reshape(-1, 1).view(np.uint8)
    blocks[:, 2:4] = dmin16.reshape(-1, 1).view(np.uint8)
    blocks[:, 4:16] = pack_scales(sc, m)
    qh = np.zeros((nb, 32), dtype=np.uint32)
    qs = np.zeros((nb, 4, 32), dtype=np.uint32)
    for g in range(4):
        qs[:, g] = (q[:, 2 * g] & 0xF) | ((q[:, 2 * g + 1] & 0xF) << 4)
        qh |= ((q[:, 2 * g] >> 4) & 1) << (2 * g)
        qh |= ((q[:, 2 * g + 1] >> 4) & 1) << (2 * g + 1)
    blocks[:, 16:48] = qh.astype(np.uint8)
    blocks[:, 48:176] = qs.reshape(nb, 128).astype(np.uint8)
    return blocks


# ---------------------------------------------------------------- Q4_K (r4): the same super-block without the fifth bits
Q4K_BYTES = 144


def q4_K_as_q5_K(blocks):
    """[nb, 144] Q4_K super-blocks { half d; half dmin; u8 scales[12]; u8 qs[128] } -> the [nb, 176] Q5_K super-blocks with the same
    meaning (fifth-bit bytes zero).  The published dequantize_row_q4_K / ggml_vec_dot_q4_K_q8_K are Q5_K's formulas with q in 0..15:
    element 64 g + l = low nibble of qs[32 g + l] under (sc, m)[2 g], 64 g + 32 + l = its high nibble under (sc, m)[2 g + 1],
    y = (d * sc) * q - (dmin * m); the dot against Q8_K the same two terms per super-block."""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, Q4K_BYTES)
    out = np.zeros((blocks.shape[0], Q5K_BYTES), dtype=np.uint8)
    out[:, 0:16] = blocks[:, 0:16]
    out[:, 48:176] = blocks[:, 16:144]
    return out


def dequantize_q4_K(blocks):
    return dequantize_q5_K(q4_K_as_q5_K(blocks))


def mul_mat_q4_K(wrows, x):
    M = wrows.shape[0]
    return mul_mat_q5_K(q4_K_as_q5_K(wrows.reshape(-1, Q4K_BYTES)).reshape(M, -1), x)


def quantize_q4_K(x):
    """A simple VALID Q4_K encoder for test data (not upstream's search): the affine code of quantize_q5_K with 15 steps."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 8, 32)
    lo = np.minimum(x.min(axis=2), 0.0)
    hi = np.maximum(x.max(axis=2), lo + 1e-30)
    scale = (hi - lo) / 15.0
    off = -lo
    d16 = (np.maximum(scale.max(axis=1), 1e-30) / 63.0).astype(np.float16)
    dmin16 = (np.maximum(off.max(axis=1), 1e-30) / 63.0).astype(np.float16)
    df, dminf = d16.astype(np.float32), dmin16.astype(np.float32)
    sc = np.clip(np.rint(scale / np.maximum(df[:, None], 1e-30)), 1, 63).astype(np.uint32)
    m = np.clip(np.rint(off / np.maximum(dminf[:, None], 1e-30)), 0, 63).astype(np.uint32)
    d1 = df[:, None] * sc
    m1 = dminf[:, None] * m
    q = np.clip(np.rint((x + m1[:, :, None]) / np.maximum(d1[:, :, None], 1e-30)), 0, 15).astype(np.uint32)
    nb = x.shape[0]
    blocks = np.zeros((nb, Q4K_BYTES), dtype=np.uint8)
    blocks[:, 0:2] = d16.reshape(-1, 1).view(np.uint8)
    blocks[:, 2:4] = dmin16.reshape(-1, 1).view(np.uint8)
    blocks[:, 4:16] = pack_scales(sc, m)
    qs = np.zeros((nb, 4, 32), dtype=np.uint32)
    for g in range(4):
        qs[:, g] = q[:, 2 * g] | (q[:, 2 * g + 1] << 4)
    blocks[:, 16:144] = qs.reshape(nb, 128).astype(np.uint8)
    return blocks


# ---------------------------------------------------------------- r4: the published quantizers, restated (what kquants.hip quantize_kq_kernel follows)
def make_qkx1_quants(x, nmax, ntry=5):
    """x [ns, 32] f32 -> (scale [ns], the_min [ns], L [ns, 32] int32): the affine code over [min(0, min x), max x], its scale refitted by
    least squares until no code changes (at most `ntry` times).  Every float operation a binary32 operation in upstream's order."""
    F = np.float32
    x = np.ascontiguousarray(x, dtype=F)
    ns = x.shape[0]
    mn = x.min(axis=1).astype(F)
    mx = x.max(axis=1).astype(F)
    flat = mx == mn
    mn = np.where(mn > 0, F(0), mn).astype(F)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iscale = (F(nmax) / (mx - mn).astype(F)).astype(F)
        scale = (F(1) / iscale).astype(F)
        L = np.full((ns, 32), -1, dtype=np.int64)
        live = ~flat                                                   # sub-blocks still iterating
        for _ in range(ntry):
            if not live.any():
                break
            sumlx = np.zeros(ns, dtype=F)
            suml2 = np.zeros(ns, dtype=np.int64)
            changed = np.zeros(ns, dtype=bool)
            Lnew = L.copy()
            for i in range(32):
                t = (x[:, i] - mn).astype(F)
                l = np.clip(np.rint((iscale * t).astype(F)), 0, nmax).astype(np.int64)
                changed |= l != L[:, i]
                Lnew[:, i] = l
                sumlx = (sumlx + (t * l.astype(F)).astype(F)).astype(F)
                suml2 += l * l
            sc_new = (sumlx / suml2.astype(F)).astype(F)
            s = np.zeros(ns, dtype=F)
            for i in range(32):
                s = (s + (x[:, i] - (sc_new * Lnew[:, i].astype(F)).astype(F)).astype(F)).astype(F)
            mn_new = (s / F(32)).astype(F)
            mn_new = np.where(mn_new > 0, F(0), mn_new).astype(F)
            L = np.where(live[:, None], Lnew, L)
            scale = np.where(live, sc_new, scale).astype(F)
            mn = np.where(live, mn_new, mn).astype(F)
            iscale = np.where(live, (F(1) / scale).astype(F), iscale).astype(F)
            live = live & changed
    L = np.where(flat[:, None], 0, L)
    return np.where(flat, F(0), scale).astype(F), np.where(flat, F(0), -mn).astype(F), L


def quantize_kq_reference(x, bits):
    """quantize_row_q5_K_reference (bits = 5) / quantize_row_q4_K_reference (bits = 4): x [nb, 256] f32 -> super-blocks [nb, 176 | 144]"""
    F = np.float32
    nmax = (1 << bits) - 1
    x = np.ascontiguousarray(x, dtype=F).reshape(-1, 256)
    nb = x.shape[0]
    scale, mins, L = make_qkx1_quants(x.reshape(-1, 32), nmax)
    scale, mins, L = scale.reshape(nb, 8), mins.reshape(nb, 8), L.reshape(nb, 8, 32)
    max_scale = np.maximum(scale.max(axis=1), F(0)).astype(F)
    max_min = np.maximum(mins.max(axis=1), F(0)).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv_scale = np.where(max_scale > 0, (F(63) / max_scale).astype(F), F(0)).astype(F)
        inv_min = np.where(max_min > 0, (F(63) / max_min).astype(F), F(0)).astype(F)
        ls = np.rint((inv_scale[:, None] * scale).astype(F)).astype(np.int64) & 255       # (a uint8_t in upstream)
        lm = np.rint((inv_min[:, None] * mins).astype(F)).astype(np.int64) & 255
        ls, lm = np.minimum(ls, 63), np.minimum(lm, 63)
        d16 = (max_scale / F(63)).astype(F).astype(np.float16)
        dmin16 = (max_min / F(63)).astype(F).astype(np.float16)
        dd = (d16.astype(F)[:, None] * ls.astype(F)).astype(F)
        dm = (dmin16.astype(F)[:, None] * lm.astype(F)).astype(F)
        l2 = np.clip(np.rint(((x.reshape(nb, 8, 32) + dm[:, :, None]).astype(F) / dd[:, :, None]).astype(F)), 0, nmax)
    L = np.where((dd != 0)[:, :, None], np.nan_to_num(l2).astype(np.int64), L).astype(np.uint32)
    blocks = np.zeros((nb, Q5K_BYTES if bits == 5 else Q4K_BYTES), dtype=np.uint8)
    blocks[:, 0:2] = d16.reshape(-1, 1).view(np.uint8)
    blocks[:, 2:4] = dmin16.reshape(-1, 1).view(np.uint8)
    blocks[:, 4:16] = pack_scales(ls, lm)
    qs = np.zeros((nb, 4, 32), dtype=np.uint32)
    qh = np.zeros((nb, 32), dtype=np.uint32)
    for g in range(4):
        qs[:, g] = (L[:, 2 * g] & 0xF) | ((L[:, 2 * g + 1] & 0xF) << 4)
        qh |= ((L[:, 2 * g] >> 4) & 1) << (2 * g)
        qh |= ((L[:, 2 * g + 1] >> 4) & 1) << (2 * g + 1)
    if bits == 5:
        blocks[:, 16:48] = qh.astype(np.uint8)
        blocks[:, 48:176] = qs.reshape(nb, 128).astype(np.uint8)
    else:
        blocks[:, 16:144] = qs.reshape(nb, 128).astype(np.uint8)
    return blocks


# ---------------------------------------------------------------- Q6_K (r4): 16 sub-blocks of 16 weights with signed 8-bit scales
Q6K_BYTES = 210


def q6_values(blocks):
    """[nb, 210] block_q6_K { u8 ql[128]; u8 qh[64]; i8 scales[16]; half d } -> the 6-bit values minus 32, [nb, 256] int32 in element order
    (dequantize_row_q6_K's unpacking: per half n of 128 elements and l < 32, element 128 n + 32 c + l takes the low (c < 2) or high (c >= 2)
    nibble of ql[64 n + 32 (c & 1) + l] and bits 2 c, 2 c + 1 of qh[32 n + l])"""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, Q6K_BYTES)
    ql = blocks[:, 0:128].astype(np.int32).reshape(-1, 2, 2, 32)        # [nb, n, c & 1, l]
    qh = blocks[:, 128:192].astype(np.int32).reshape(-1, 2, 32)
    out = np.empty((blocks.shape[0], 2, 4, 32), dtype=np.int32)
    for c in range(4):
        nib = (ql[:, :, c & 1] & 0xF) if c < 2 else (ql[:, :, c & 1] >> 4)
        out[:, :, c] = (nib | (((qh >> (2 * c)) & 3) << 4)) - 32
    return out.reshape(-1, 256)


def dequantize_q6_K(blocks):
    """y[e] = d * scales[e / 16] * q[e], the product d * sc first (upstream: `d * sc[is] * q`, left to right), every operation a binary32 rounding"""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, Q6K_BYTES)
    d = blocks[:, 208:210].copy().view(np.float16).astype(np.float32).reshape(-1, 1)
    sc = blocks[:, 192:208].copy().view(np.int8).astype(np.float32)
    ds = (d * sc).astype(np.float32)                                     # [nb, 16]
    q = q6_values(blocks).reshape(-1, 16, 16).astype(np.float32)
    return (ds[:, :, None] * q).astype(np.float32).reshape(-1, 256)


def mul_mat_q6_K(wrows, x):
    """wrows [M, K/256*210] uint8, x [N, K] f32 -> [N, M]: ggml_vec_dot_q6_K_q8_K per element -- per super-block (d * dy) * sum_j sc_j <q_j, a_j>,
    the integer sums exact; evaluated in f64 (a checker for the path's tolerance, not a bit-level one)"""
    M = wrows.shape[0]
    N, K = x.shape
    nb = K // 256
    d8, q8, _ = quantize_q8_K(x.reshape(-1, 256))
    d8, q8 = d8.reshape(N, nb).astype(np.float64), q8.reshape(N, nb, 16, 16).astype(np.float64)
    w = wrows.reshape(M * nb, Q6K_BYTES)
    dw = w[:, 208:210].copy().view(np.float16).astype(np.float64).reshape(M, nb)
    sc = w[:, 192:208].copy().view(np.int8).astype(np.float64).reshape(M, nb, 16)
    q = q6_values(w).astype(np.float64).reshape(M, nb, 16, 16)
    dots = np.einsum("mbjl,nbjl->nmbj", q, q8)
    isum = (dots * sc[None]).sum(axis=3)
    return (dw[None] * d8[:, None, :] * isum).sum(axis=2).astype(np.float32)


def pack_q6(L):
    """L [nb, 256] values 0..63 -> (ql [nb, 128], qh [nb, 64]) uint8 (the inverse of q6_values' unpacking)"""
    L = L.astype(np.uint32).reshape(-1, 2, 4, 32)
    ql = np.zeros((L.shape[0], 2, 2, 32), dtype=np.uint32)
    qh = np.zeros((L.shape[0], 2, 32), dtype=np.uint32)
    for c in range(4):
        ql[:, :, c & 1] |= (L[:, :, c] & 0xF) << (0 if c < 2 else 4)
        qh |= (L[:, :, c] >> 4) << (2 * c)
    return ql.reshape(-1, 128).astype(np.uint8), qh.reshape(-1, 64).astype(np.uint8)


def quantize_q6_K(x):
    """quantize_row_q6_K_reference with make_qx_quants in its plain form (rmse_type 0: no least-squares refinement of the sub-block scales
    -- upstream's default searches; this is a VALID encoder of the same structure, and what kquants.hip's device quantizer follows):
    per sub-block of 16 the element of largest magnitude maps to -32 (scale = max / -32); the 16 scales as signed 8-bit multiples of
    d = (the scale of largest magnitude) / -128 (a half); the codes again under the rounded scales, l = nearest(x / (d sc)) in -32..31."""
    F = np.float32
    x = np.ascontiguousarray(x, dtype=F).reshape(-1, 16, 16)
    nb = x.shape[0]
    ax = np.abs(x)
    idx = np.argmax(ax, axis=2)
    mx = np.take_along_axis(x, idx[:, :, None], axis=2)[:, :, 0]
    amax = np.take_along_axis(ax, idx[:, :, None], axis=2)[:, :, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        iscale = np.where(amax != 0, (F(-32) / mx).astype(F), F(0)).astype(F)
        scale = np.where(amax != 0, (F(1) / iscale).astype(F), F(0)).astype(F)           # [nb, 16]
        L = np.where((amax != 0)[:, :, None], np.clip(np.rint((iscale[:, :, None] * x).astype(F)), -32, 31) + 32, 0).astype(np.int64)
        j = np.argmax(np.abs(scale), axis=1)                                            # the first scale of largest magnitude
        max_scale = scale[np.arange(nb), j]
        zero = np.abs(max_scale) == 0
        isc = np.where(zero, F(0), (F(-128) / max_scale).astype(F)).astype(F)
        d16 = np.where(zero, F(0), (F(1) / isc).astype(F)).astype(np.float16)
        sc = np.minimum(np.rint((isc[:, None] * scale).astype(F)), 127).astype(np.int64)
        dd = (d16.astype(F)[:, None] * sc.astype(F)).astype(F)
        l2 = np.clip(np.rint((x / dd[:, :, None]).astype(F)), -32, 31) + 32
    L = np.where((dd != 0)[:, :, None], np.nan_to_num(l2).astype(np.int64), L)
    L = np.where(zero[:, None, None], 0, L)                                               # (upstream: memset(y, 0) for an all-zero super-block)
    sc = np.where(zero[:, None], 0, sc)
    blocks = np.zeros((nb, Q6K_BYTES), dtype=np.uint8)
    ql, qh = pack_q6(L.reshape(nb, 256))
    blocks[:, 0:128] = ql
    blocks[:, 128:192] = qh
    blocks[:, 192:208] = sc.astype(np.int8).view(np.uint8)
    blocks[:, 208:210] = d16.reshape(-1, 1).view(np.uint8)
    return blocks
