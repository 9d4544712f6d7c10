// kquants.hip -- Q5_K (r4: and Q4_K) weights (and the Q8_K activation rule) as an UNPINNED EXTRA.
//
// The reference has no k-quants (TypeDefinitions.cs:153-169 stops at Q8_1; `grep -i q5_K` over /root/reference finds
// nothing -- SURVEY 8(a) row K), BASELINE.json's north_star and config 4 name them anyway.  What is built here follows the
// PUBLISHED upstream format (ggml k_quants, June 2023; not vendored, not referenced by any project file of the reference):
//     block_q5_K = { half d; half dmin; u8 scales[12]; u8 qh[32]; u8 qs[128] }        176 bytes per 256 weights
//     w[e] = d * sc_j * q[e] - dmin * m_j,  j = e / 32, sc_j / m_j the 6-bit entries of scales[] (get_scale_min_k4),
//     q[e] = 4 low bits from qs (element 64 g + l: low nibble of qs[32 g + l], 64 g + 32 + l: its high nibble) + bit
//     (2 g) / (2 g + 1) of qh[l] as the fifth bit
//     block_q4_K = { half d; half dmin; u8 scales[12]; u8 qs[128] }                    144 bytes: the same without the fifth bits (q in 0..15)
//     block_q8_K = { float d; i8 qs[256]; i16 bsums[16] }: iscale = -128 / (the first element of largest magnitude),
//     q = min(127, round-half-even(iscale * x)), d = 1 / iscale
//     dot per super-block: (d * dy) * sum_j sc_j * <q_j, a_j>  -  (dmin * dy) * sum_j m_j * bsum_j
// There is NO oracle in the reference for any of it: tests/np_kquants.py restates the published algorithm and is the only
// checker ("parity unpinned", stated wherever Q5_K appears).
//
// Design: a 32-element sub-block of Q5_K IS a Q5_1 block with effective scale d * sc_j (exact in f32: 11 + 6 bits) and
// effective min -(dmin * m_j) (exact) -- only the bit layout differs.  So the upload re-lays a super-block out as eight
// k-blocks of the resident planar Q5_1 form (nibble plane, fifth-bit plane, scale plane, min plane), and every Q5_1
// kernel of the path -- the mat-vec, the f16-MFMA mat-mat with its K split, the int8-MFMA mat-mat -- serves it unchanged.
// The INIT phase quantizes the activations with the Q8_K rule (one scale per 256 elements, quantize.hip K1 with K8 =
// true) into the same operand images.  The 16 header bytes of every super-block are kept beside the planes so that a
// download returns the uploaded bytes.
#include "common.h"

namespace {

__device__ __forceinline__ float h2f(uint16_t h) {          // IEEE binary16 -> binary32, exact
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    if (exp == 0) return __uint_as_float(__float_as_uint((float)man * 5.9604644775390625e-08f) | sign);
    if (exp == 31) return __uint_as_float(sign | 0x7F800000u | (man << 13));
    return __uint_as_float(sign | ((exp + 112u) << 23) | (man << 13));
}

// upstream get_scale_min_k4
__device__ __forceinline__ void scale_min_k4(int j, const uint8_t *q, uint32_t &sc, uint32_t &m) {
    if (j < 4) {
        sc = q[j] & 63u;
        m = q[j + 4] & 63u;
    } else {
        sc = (q[j + 4] & 0xFu) | ((uint32_t)(q[j - 4] >> 6) << 4);
        m = (q[j + 4] >> 4) | ((uint32_t)(q[j] >> 6) << 4);
    }
}

// one thread per (row, 32-element sub-block); rows fastest so the planar stores coalesce
// (Q5 = false: Q4_K -- 144-byte super-blocks, the nibbles straight behind the 16 header bytes, no fifth bits)
template <bool Q5>
__global__ void q5k_to_planar_kernel(const uint8_t *__restrict__ aos, uint64_t nb01, int64_t row_begin, int64_t rows, int64_t Mpad,
                                     uint8_t *__restrict__ qs, uint32_t *__restrict__ qh, float *__restrict__ d, float *__restrict__ mm,
                                     uint8_t *__restrict__ khdr) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;                           // k-block (32 elements)
    if (m >= rows) return;
    const int64_t sb = b >> 3;
    const int j = (int)(b & 7), g = j >> 1, hi = j & 1;
    const uint8_t *blk = aos + (uint64_t)(row_begin + m) * nb01 + (uint64_t)sb * (Q5 ? 176 : 144);
    const uint8_t *scales = blk + 4, *qhs = blk + 16, *ql = blk + (Q5 ? 48 : 16) + 32 * g;
    uint32_t sc, mn;
    scale_min_k4(j, scales, sc, mn);
    const float dd = h2f(*(const uint16_t *)blk), dmin = h2f(*(const uint16_t *)(blk + 2));
    const int64_t pi = b * Mpad + m;
    d[pi] = dd * (float)sc;                                 // exact: 11 + 6 significant bits
    mm[pi] = -(dmin * (float)mn);                           // the Q5_1 form adds its min: w = d q + m
    uint32_t w[4] = {0, 0, 0, 0}, hbits = 0;
#pragma unroll
    for (int l = 0; l < 32; ++l) {
        const uint32_t nib = hi ? (uint32_t)(ql[l] >> 4) : (uint32_t)(ql[l] & 15u);
        w[l >> 3] |= nib << (4 * (l & 7));                  // byte l/2 = element l | element l+1 << 4 (the Q5_1 plane's order)
        if constexpr (Q5) hbits |= (uint32_t)((qhs[l] >> j) & 1u) << l;
    }
    *(uint4 *)(qs + pi * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    qh[pi] = hbits;
    if (j == 0) *(uint4 *)(khdr + (sb * Mpad + m) * 16) = make_uint4(((const uint32_t *)blk)[0], ((const uint32_t *)blk)[1],
                                                                     ((const uint32_t *)blk)[2], ((const uint32_t *)blk)[3]);
}

// exact inverse: one thread per (row, super-block)
template <bool Q5>
__global__ void planar_to_q5k_kernel(uint8_t *__restrict__ aos, uint64_t nb01, int64_t rows, int64_t Mpad, const uint8_t *__restrict__ qs,
                                     const uint32_t *__restrict__ qh, const uint8_t *__restrict__ khdr) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t sb = blockIdx.y;
    if (m >= rows) return;
    uint8_t *blk = aos + (uint64_t)m * nb01 + (uint64_t)sb * (Q5 ? 176 : 144);
    const uint4 h = *(const uint4 *)(khdr + (sb * Mpad + m) * 16);
    // (176-byte blocks are only 16-byte aligned when nb01 is: write bytes)
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
    for (int i = 0; i < 16; ++i) blk[i] = (uint8_t)(hw[i >> 2] >> (8 * (i & 3)));
    uint8_t qhb[32];
    for (int l = 0; l < 32; ++l) qhb[l] = 0;
    for (int j = 0; j < 8; ++j) {
        const int64_t pi = (sb * 8 + j) * Mpad + m;
        const uint4 w4 = *(const uint4 *)(qs + pi * 16);
        const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
        const uint32_t hb = qh[pi];
        uint8_t *ql = blk + (Q5 ? 48 : 16) + 32 * (j >> 1);
        for (int l = 0; l < 32; ++l) {
            const uint32_t nib = (w[l >> 3] >> (4 * (l & 7))) & 15u;
            if (j & 1) ql[l] = (uint8_t)((ql[l] & 0x0Fu) | (nib << 4));
            else ql[l] = (uint8_t)nib;                      // (the even sub-block of a pair comes first: it initialises the byte)
            qhb[l] |= (uint8_t)(((hb >> l) & 1u) << j);
        }
    }
    if constexpr (Q5) for (int l = 0; l < 32; ++l) blk[16 + l] = qhb[l];
}

// dequantize_row_q5_K (Q5 = false: dequantize_row_q4_K) of the published format: one thread per (row-major) sub-block of 32 outputs
template <bool Q5>
__global__ void dequantize_q5k_kernel(const uint8_t *__restrict__ in, int64_t nsub, float *__restrict__ y) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsub) return;
    const uint8_t *blk = in + (s >> 3) * (Q5 ? 176 : 144);
    const int j = (int)(s & 7), g = j >> 1, hi = j & 1;
    uint32_t sc, mn;
    scale_min_k4(j, blk + 4, sc, mn);
    const float d1 = h2f(*(const uint16_t *)blk) * (float)sc, m1 = h2f(*(const uint16_t *)(blk + 2)) * (float)mn;
    const uint8_t *ql = blk + (Q5 ? 48 : 16) + 32 * g, *qhs = blk + 16;
    float *o = y + s * 32;
    for (int l = 0; l < 32; ++l) {
        const int q = (int)(hi ? (ql[l] >> 4) : (ql[l] & 15)) + (Q5 && ((qhs[l] >> j) & 1) ? 16 : 0);
        o[l] = d1 * (float)q - m1;                          // upstream: d1 * q - m1, one multiply then one subtract
    }
}

// ---- r4: a device quantizer for the two extension types --------------------------------------------------------------------------------
// quantize_row_q5_K_reference / _q4_K_reference of the published format (ggml k_quants, 2023-06), restated: per 32-element sub-block
// make_qkx1_quants(32, nmax, x, L, &min, 5) -- an affine code over [min(0, min x), max x], its scale refitted up to five times by least
// squares (scale = sum (x - min) l / sum l^2, min = mean(x - scale l) capped at 0) until no code changes; then the eight scales / mins of a
// super-block as 6-bit multiples of d = max scale / 63 and dmin = max min / 63 (halves), and the codes again under the ROUNDED scales:
// l = nearest((x + dmin m) / (d sc)) in 0..nmax.  Every float operation is a binary32 operation in the order written (the library is
// built with -ffp-contract=off), nearest = round half to even.  UNPINNED like the rest of the extension: tests/np_kquants.py restates the
// same steps and the device bytes are compared with its bytes; neither was run against upstream.
// Eight lanes per super-block (one sub-block each), the super-block's maxima and the bit transpositions by width-8 shuffles.
template <bool Q5>
__global__ void quantize_kq_kernel(const float *__restrict__ x, int64_t nsb, uint8_t *__restrict__ out) {
    constexpr int NMAX = Q5 ? 31 : 15, BYTES = Q5 ? 176 : 144;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t sb_raw = tid >> 3;
    const bool active = sb_raw < nsb;
    const int64_t sb = active ? sb_raw : nsb - 1;           // (idle lanes of the last group shadow the last super-block: the shuffles want every lane)
    const int j = (int)(tid & 7);
    const float *xs = x + sb * 256 + 32 * j;
    float v[32];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = ((const float4 *)xs)[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
    int L[32];
    float mn = v[0], mx = v[0];
#pragma unroll
    for (int i = 1; i < 32; ++i) { if (v[i] < mn) mn = v[i]; if (v[i] > mx) mx = v[i]; }
    float scale = 0.0f, the_min = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) L[i] = -1;                 // (upstream compares with an uninitialised L on the first try: every code "changes")
    if (mx == mn) {
#pragma unroll
        for (int i = 0; i < 32; ++i) L[i] = 0;
    } else {
        if (mn > 0.0f) mn = 0.0f;
        float iscale = (float)NMAX / (mx - mn);
        scale = 1.0f / iscale;
        for (int itry = 0; itry < 5; ++itry) {
            float sumlx = 0.0f;
            int suml2 = 0;
            bool changed = false;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                int l = (int)rintf(iscale * (v[i] - mn));
                l = l < 0 ? 0 : l > NMAX ? NMAX : l;
                if (l != L[i]) { L[i] = l; changed = true; }
                sumlx += (v[i] - mn) * (float)l;
                suml2 += l * l;
            }
            scale = sumlx / (float)suml2;
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) sum += v[i] - scale * (float)L[i];
            mn = sum / 32.0f;
            if (mn > 0.0f) mn = 0.0f;
            iscale = 1.0f / scale;
            if (!changed) break;
        }
        the_min = -mn;
    }
    // the super-block's largest scale and min (upstream: `if (scale > max_scale)` from 0)
    float max_scale = scale > 0.0f ? scale : 0.0f, max_min = the_min > 0.0f ? the_min : 0.0f;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const float a = __shfl_xor(max_scale, o, 8), b = __shfl_xor(max_min, o, 8);
        if (a > max_scale) max_scale = a;
        if (b > max_min) max_min = b;
    }
    const float inv_scale = max_scale > 0.0f ? 63.0f / max_scale : 0.0f, inv_min = max_min > 0.0f ? 63.0f / max_min : 0.0f;
    int ls = (int)rintf(inv_scale * scale), lm = (int)rintf(inv_min * the_min);
    ls = (ls & 255) > 63 ? 63 : (ls & 255);                 // (upstream stores nearest_int in a uint8_t, then MIN(63, .))
    lm = (lm & 255) > 63 ? 63 : (lm & 255);
    const _Float16 dh = (_Float16)(max_scale / 63.0f), dminh = (_Float16)(max_min / 63.0f);
    const float dd = (float)dh * (float)ls;
    if (dd != 0.0f) {
        const float dm = (float)dminh * (float)lm;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            int l = (int)rintf((v[i] + dm) / dd);
            L[i] = l < 0 ? 0 : l > NMAX ? NMAX : l;
        }
    }
    uint8_t *blk = out + sb * BYTES;
    // header: d, dmin, scales[12] (lane 0, with every lane's ls / lm)
    int lsa[8], lma[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { lsa[q] = __shfl(ls, q, 8); lma[q] = __shfl(lm, q, 8); }
    if (active && j == 0) {
        *(uint16_t *)blk = __builtin_bit_cast(uint16_t, dh);
        *(uint16_t *)(blk + 2) = __builtin_bit_cast(uint16_t, dminh);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            blk[4 + q] = (uint8_t)(lsa[q] | ((lsa[q + 4] >> 4) << 6));
            blk[8 + q] = (uint8_t)(lma[q] | ((lma[q + 4] >> 4) << 6));
            blk[12 + q] = (uint8_t)((lsa[q + 4] & 15) | ((lma[q + 4] & 15) << 4));
        }
    }
    // nibbles: sub-blocks 2g / 2g + 1 share the bytes qs[32 g ..]: the even lane writes both
    uint32_t lo[8];                                         // this lane's 32 low nibbles as bytes-to-be (four per word), and its fifth bits
    uint32_t hb = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        lo[i >> 2] |= (uint32_t)(L[i] & 15) << (8 * (i & 3));
        hb |= (uint32_t)((L[i] >> 4) & 1) << i;
    }
    uint8_t *ql = blk + (Q5 ? 48 : 16) + 32 * (j >> 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t other = __shfl_xor(lo[i], 1, 8);     // (the odd lane's nibbles go to the high halves)
        if (active && !(j & 1)) ((uint32_t *)ql)[i] = lo[i] | (other << 4);
    }
    if constexpr (Q5) {                                     // qh[l] bit q = the fifth bit of element l of sub-block q: lane j writes bytes 4 j .. 4 j + 3
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t h = __shfl(hb, q, 8);
#pragma unroll
            for (int b = 0; b < 4; ++b) word |= ((h >> (4 * j + b)) & 1u) << (8 * b + q);
        }
        if (active) ((uint32_t *)(blk + 16))[j] = word;
    }
}

// ---- r4: Q6_K ------------------------------------------------------------------------------------------------------------------------------
//     block_q6_K = { u8 ql[128]; u8 qh[64]; i8 scales[16]; half d }                    210 bytes per 256 weights
//     w[e] = d * scales[e / 16] * (q[e] - 32);  per half n of 128 elements and l < 32, element 128 n + 32 c + l has its low four bits in the
//     low (c < 2) or high (c >= 2) nibble of ql[64 n + 32 (c & 1) + l] and its two high bits in bits 2 c, 2 c + 1 of qh[32 n + l]
//     dot against Q8_K per super-block: (d * dy) * sum_j scales[j] * <q_j - 32, a_j>
// Resident form: a 32-element k-block is two 16-element sub-blocks with a scale each -- the structure of the reference's Q4_2 (two blocks per
// Q8_0 block, Ggml.cs:1217-1252) -- so a super-block becomes eight k-blocks of the planar Q4_2 form: int8 operand planes of q - 32 (the
// layout of Q8_0's planes), the first sub-block's effective scale d * sc (exact in f32: 11 + 8 bits) in the d plane, the second's in the m
// plane.  The int8 kernels that take two scales per k-block serve it (gemm_q8s.hip Q42, gemm_q.hip's int8-plane two-scale form); the
// nibble plane of the Q4_2 form stays empty.  32 header bytes per super-block (scales[16], d) are kept for the byte-exact download.
// one thread per (row, k-block)
__global__ void q6k_to_planar_kernel(const uint8_t *__restrict__ aos, uint64_t nb01, int64_t row_begin, int64_t rows, int64_t Mpad,
                                     uint8_t *__restrict__ i8p, float *__restrict__ d, float *__restrict__ mm, uint8_t *__restrict__ khdr) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (m >= rows) return;
    const int64_t sb = b >> 3;
    const int bq = (int)(b & 7), n = bq >> 2, c = bq & 3;
    const uint8_t *blk = aos + (uint64_t)(row_begin + m) * nb01 + (uint64_t)sb * 210;
    const uint8_t *ql = blk + 64 * n + 32 * (c & 1), *qh = blk + 128 + 32 * n;
    const int8_t *sc = (const int8_t *)(blk + 192);
    const float dd = h2f((uint16_t)(blk[208] | ((uint16_t)blk[209] << 8)));
    const int64_t pi = b * Mpad + m;
    d[pi] = dd * (float)sc[2 * bq];                         // exact: 11 + 8 significant bits
    mm[pi] = dd * (float)sc[2 * bq + 1];
    uint32_t ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const uint32_t nib = c < 2 ? (uint32_t)(ql[t] & 15u) : (uint32_t)(ql[t] >> 4);
        const int v = (int)(nib | (((uint32_t)(qh[t] >> (2 * c)) & 3u) << 4)) - 32;
        const uint32_t byte = (uint32_t)(uint8_t)(int8_t)v << (8 * ((t >> 1) & 3));
        if (t & 1) od[t >> 3] |= byte; else ev[t >> 3] |= byte;      // plane h byte j = element 2 j + h
    }
    *(uint4 *)(i8p + ((b * 2 + 0) * Mpad + m) * 16) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
    *(uint4 *)(i8p + ((b * 2 + 1) * Mpad + m) * 16) = make_uint4(od[0], od[1], od[2], od[3]);
    if (bq == 0) {
        uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 18; ++i) h[i >> 2] |= (uint32_t)blk[192 + i] << (8 * (i & 3));
        uint4 *o = (uint4 *)(khdr + (sb * Mpad + m) * 32);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
}

// exact inverse: one thread per (row, super-block)
__global__ void planar_to_q6k_kernel(uint8_t *__restrict__ aos, uint64_t nb01, int64_t rows, int64_t Mpad, const uint8_t *__restrict__ i8p,
                                     const uint8_t *__restrict__ khdr) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t sb = blockIdx.y;
    if (m >= rows) return;
    uint8_t *blk = aos + (uint64_t)m * nb01 + (uint64_t)sb * 210;
    for (int i = 0; i < 192; ++i) blk[i] = 0;
    for (int bq = 0; bq < 8; ++bq) {
        const int n = bq >> 2, c = bq & 3;
        uint8_t *ql = blk + 64 * n + 32 * (c & 1), *qh = blk + 128 + 32 * n;
        const int64_t b = sb * 8 + bq;
        for (int hsel = 0; hsel < 2; ++hsel) {
            const uint4 w4 = *(const uint4 *)(i8p + ((b * 2 + hsel) * Mpad + m) * 16);
            const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
            for (int j = 0; j < 16; ++j) {
                const int t = 2 * j + hsel;
                const uint32_t q = (uint32_t)((int)(int8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) + 32);
                ql[t] |= (uint8_t)(c < 2 ? (q & 15u) : ((q & 15u) << 4));
                qh[t] |= (uint8_t)((q >> 4) << (2 * c));
            }
        }
    }
    const uint8_t *h = khdr + (sb * Mpad + m) * 32;
    for (int i = 0; i < 18; ++i) blk[192 + i] = h[i];
}

// dequantize_row_q6_K of the published format: one thread per (row-major) k-block of 32 outputs
__global__ void dequantize_q6k_kernel(const uint8_t *__restrict__ in, int64_t nkb, float *__restrict__ y) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nkb) return;
    const uint8_t *blk = in + (s >> 3) * 210;
    const int bq = (int)(s & 7), n = bq >> 2, c = bq & 3;
    const uint8_t *ql = blk + 64 * n + 32 * (c & 1), *qh = blk + 128 + 32 * n;
    const int8_t *sc = (const int8_t *)(blk + 192);
    const float dd = h2f((uint16_t)(blk[208] | ((uint16_t)blk[209] << 8)));
    const float d0 = dd * (float)sc[2 * bq], d1 = dd * (float)sc[2 * bq + 1];
    float *o = y + s * 32;
    for (int t = 0; t < 32; ++t) {
        const uint32_t nib = c < 2 ? (uint32_t)(ql[t] & 15u) : (uint32_t)(ql[t] >> 4);
        const int v = (int)(nib | (((uint32_t)(qh[t] >> (2 * c)) & 3u) << 4)) - 32;
        o[t] = (t < 16 ? d0 : d1) * (float)v;               // upstream: d * sc[is] * q, left to right
    }
}

// quantize_row_q6_K_reference with make_qx_quants in its plain form (no least-squares refinement of the sub-block scales: a VALID encoder of
// the published structure, stated as such in include/ggml_hip_ext.h; tests/np_kquants.py quantize_q6_K is the same steps): sixteen lanes per
// super-block, one sub-block each.
__global__ void quantize_q6k_kernel(const float *__restrict__ x, int64_t nsb, uint8_t *__restrict__ out) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t sb_raw = tid >> 4;
    const bool active = sb_raw < nsb;
    const int64_t sb = active ? sb_raw : nsb - 1;
    const int j = (int)(tid & 15);
    const float *xs = x + sb * 256 + 16 * j;
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 t = ((const float4 *)xs)[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
    float amax = 0.0f, mx = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float ax = fabsf(v[i]); if (ax > amax) { amax = ax; mx = v[i]; } }
    const float iscale = amax != 0.0f ? -32.0f / mx : 0.0f;
    const float scale = amax != 0.0f ? 1.0f / iscale : 0.0f;
    int L[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int l = (int)rintf(iscale * v[i]);
        l = l < -32 ? -32 : l > 31 ? 31 : l;
        L[i] = amax != 0.0f ? l + 32 : 0;
    }
    // the first sub-block scale of largest magnitude
    float best = fabsf(scale);
    int bidx = j;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float ob = __shfl_xor(best, o, 16);
        const int oi = __shfl_xor(bidx, o, 16);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    const float max_scale = __shfl(scale, bidx, 16);
    const bool zero = best == 0.0f;
    const float isc = zero ? 0.0f : -128.0f / max_scale;
    const _Float16 dh = zero ? (_Float16)0.0f : (_Float16)(1.0f / isc);
    int sc = (int)rintf(isc * scale);
    sc = zero ? 0 : sc > 127 ? 127 : sc;
    const float dd = (float)dh * (float)sc;
    if (dd != 0.0f) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int l = (int)rintf(v[i] / dd);
            l = l < -32 ? -32 : l > 31 ? 31 : l;
            L[i] = l + 32;
        }
    }
    if (zero) {
#pragma unroll
        for (int i = 0; i < 16; ++i) L[i] = 0;
    }
    uint8_t *blk = out + sb * 210;
    const int n = j >> 3, c = (j >> 1) & 3, half = j & 1;   // element 16 j + i = 128 n + 32 c + (16 half + i)
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};    // sixteen bytes-to-be: low nibbles / the two high bits
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        lo[i >> 2] |= (uint32_t)(L[i] & 15) << (8 * (i & 3));
        hi[i >> 2] |= (uint32_t)(L[i] >> 4) << (8 * (i & 3));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t plo = __shfl_xor(lo[k], 4, 16);      // the lane with c ^ 2: the other nibble of the same ql bytes
        const uint32_t h2 = __shfl_xor(hi[k], 2, 16), h4 = __shfl_xor(hi[k], 4, 16), h6 = __shfl_xor(hi[k], 6, 16);   // c ^ 1, c ^ 2, c ^ 3
        if (active && c < 2) {
            const uint32_t word = lo[k] | (plo << 4);
            uint8_t *q = blk + 64 * n + 32 * (c & 1) + 16 * half + 4 * k;
            q[0] = (uint8_t)word; q[1] = (uint8_t)(word >> 8); q[2] = (uint8_t)(word >> 16); q[3] = (uint8_t)(word >> 24);
        }
        if (active && c == 0) {
            const uint32_t word = hi[k] | (h2 << 2) | (h4 << 4) | (h6 << 6);
            uint8_t *q = blk + 128 + 32 * n + 16 * half + 4 * k;
            q[0] = (uint8_t)word; q[1] = (uint8_t)(word >> 8); q[2] = (uint8_t)(word >> 16); q[3] = (uint8_t)(word >> 24);
        }
    }
    if (active) {
        blk[192 + j] = (uint8_t)(int8_t)sc;
        if (j == 0) { const uint16_t hb = __builtin_bit_cast(uint16_t, dh); blk[208] = (uint8_t)hb; blk[209] = (uint8_t)(hb >> 8); }
    }
}

}  // namespace

hipError_t launch_q5k_to_planar(int kq_type, const uint8_t *aos, uint64_t nb01, int64_t row_begin, int64_t rows, ggml_hip_weight *w, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    dim3 grid((unsigned)((rows + 127) / 128), (unsigned)w->nbk);
    if (kq_type == GGML_HIP_TYPE_Q5_K) q5k_to_planar_kernel<true><<<grid, 128, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->qs, w->qh, w->d, w->m, w->khdr);
    else q5k_to_planar_kernel<false><<<grid, 128, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->qs, w->qh, w->d, w->m, w->khdr);
    return hipGetLastError();
}

hipError_t launch_planar_to_q5k(const ggml_hip_weight *w, uint8_t *aos, hipStream_t st) {
    if (w->M <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 127) / 128), (unsigned)(w->nbk / 8));
    if (w->ext_type == GGML_HIP_TYPE_Q5_K) planar_to_q5k_kernel<true><<<grid, 128, 0, st>>>(aos, (uint64_t)(w->nbk / 8) * 176, w->M, w->Mpad, w->qs, w->qh, w->khdr);
    else planar_to_q5k_kernel<false><<<grid, 128, 0, st>>>(aos, (uint64_t)(w->nbk / 8) * 144, w->M, w->Mpad, w->qs, w->qh, w->khdr);
    return hipGetLastError();
}

hipError_t launch_dequantize_q5k(int kq_type, const void *blocks, int64_t nrows, int64_t k, float *y, hipStream_t st) {
    const int64_t nsub = nrows * (k / 32);
    if (nsub <= 0) return hipSuccess;
    if (kq_type == GGML_HIP_TYPE_Q5_K) dequantize_q5k_kernel<true><<<dim3((unsigned)((nsub + 127) / 128)), 128, 0, st>>>((const uint8_t *)blocks, nsub, y);
    else dequantize_q5k_kernel<false><<<dim3((unsigned)((nsub + 127) / 128)), 128, 0, st>>>((const uint8_t *)blocks, nsub, y);
    return hipGetLastError();
}

hipError_t launch_quantize_kq(int kq_type, const float *x, int64_t nrows, int64_t k, void *blocks, hipStream_t st) {
    const int64_t nsb = nrows * (k / 256);
    if (nsb <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((nsb * 8 + 127) / 128);
    if (kq_type == GGML_HIP_TYPE_Q5_K) quantize_kq_kernel<true><<<dim3(grid), 128, 0, st>>>(x, nsb, (uint8_t *)blocks);
    else quantize_kq_kernel<false><<<dim3(grid), 128, 0, st>>>(x, nsb, (uint8_t *)blocks);
    return hipGetLastError();
}

hipError_t launch_q6k_to_planar(const uint8_t *aos, uint64_t nb01, int64_t row_begin, int64_t rows, ggml_hip_weight *w, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    dim3 grid((unsigned)((rows + 127) / 128), (unsigned)w->nbk);
    q6k_to_planar_kernel<<<grid, 128, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->i8p, w->d, w->m, w->khdr);
    return hipGetLastError();
}

hipError_t launch_planar_to_q6k(const ggml_hip_weight *w, uint8_t *aos, hipStream_t st) {
    if (w->M <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 127) / 128), (unsigned)(w->nbk / 8));
    planar_to_q6k_kernel<<<grid, 128, 0, st>>>(aos, (uint64_t)(w->nbk / 8) * 210, w->M, w->Mpad, w->i8p, w->khdr);
    return hipGetLastError();
}

hipError_t launch_dequantize_q6k(const void *blocks, int64_t nrows, int64_t k, float *y, hipStream_t st) {
    const int64_t nkb = nrows * (k / 32);
    if (nkb <= 0) return hipSuccess;
    dequantize_q6k_kernel<<<dim3((unsigned)((nkb + 127) / 128)), 128, 0, st>>>((const uint8_t *)blocks, nkb, y);
    return hipGetLastError();
}

hipError_t launch_quantize_q6k(const float *x, int64_t nrows, int64_t k, void *blocks, hipStream_t st) {
    const int64_t nsb = nrows * (k / 256);
    if (nsb <= 0) return hipSuccess;
    quantize_q6k_kernel<<<dim3((unsigned)((nsb * 16 + 127) / 128)), 128, 0, st>>>(x, nsb, (uint8_t *)blocks);
    return hipGetLastError();
}
