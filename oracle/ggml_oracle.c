/*
 * ggml_oracle.c -- CPU oracle (test infrastructure, see ggml_oracle.h for the policy header).
 *
 * PARITY UNPINNED for the quantized functions (no runnable reference, no reference goldens);
 * pinned by hand-derived KATs, the Test3 LCG stream, Test0's layout asserts and an independent
 * numpy restatement.  Build with -ffp-contract=off: the C# JIT does not fuse a*b+c.
 *
 * All file:line citations are into /root/reference/GGMLSharp/Ggml.cs unless stated.
 */
#include "ggml_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define QK 32

/* ---- block layouts: TypeDefinitions.cs:236-290, sizes Ggml.cs:72-87 ---- */
#pragma pack(push, 1)
typedef struct { float d; uint8_t qs[QK / 2]; } block_q4_0;                 /* 20 */
typedef struct { float d; float m; uint8_t qs[QK / 2]; } block_q4_1;        /* 24 */
typedef struct { uint16_t d; uint8_t qs[8]; } block_q4_2;                   /* 10, QK4_2 = 16 */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[QK / 2]; } block_q5_0;            /* 22 */
typedef struct { uint16_t d; uint16_t m; uint8_t qh[4]; uint8_t qs[QK / 2]; } block_q5_1; /* 24 */
typedef struct { float d; int8_t qs[QK]; } block_q8_0;                      /* 36 */
typedef struct { float d; float s0; float s1; int8_t qs[QK]; } block_q8_1;  /* 44 */
#pragma pack(pop)

typedef char assert_q4_0[sizeof(block_q4_0) == 20 ? 1 : -1];
typedef char assert_q4_1[sizeof(block_q4_1) == 24 ? 1 : -1];
typedef char assert_q4_2[sizeof(block_q4_2) == 10 ? 1 : -1];
typedef char assert_q5_0[sizeof(block_q5_0) == 22 ? 1 : -1];
typedef char assert_q5_1[sizeof(block_q5_1) == 24 ? 1 : -1];
typedef char assert_q8_0[sizeof(block_q8_0) == 36 ? 1 : -1];
typedef char assert_q8_1[sizeof(block_q8_1) == 44 ? 1 : -1];

/* Ggml.cs:55-70 */
static const int BLCK_SIZE[ORACLE_TYPE_COUNT] = {1, 1, 32, 32, 16, 16, 32, 32, 32, 32, 1, 1, 1};
/* Ggml.cs:72-87 */
static const size_t TYPE_SIZE[ORACLE_TYPE_COUNT] = {4, 2, 20, 24, 10, 12, 22, 24, 36, 44, 1, 2, 4};

int oracle_blck_size(int type) { return (type >= 0 && type < ORACLE_TYPE_COUNT) ? BLCK_SIZE[type] : 0; }
size_t oracle_type_size(int type) { return (type >= 0 && type < ORACLE_TYPE_COUNT) ? TYPE_SIZE[type] : 0; }
int oracle_is_quantized(int type) { return type >= ORACLE_TYPE_Q4_0 && type <= ORACLE_TYPE_Q8_1; }

/* vec_dot_type column of quantize_fns[] (Ggml.cs:219-290); Q4_3 is `default`, Q8_1.vec_dot_q is null. */
int oracle_vec_dot_type(int type) {
    switch (type) {
    case ORACLE_TYPE_Q4_0: case ORACLE_TYPE_Q4_2: case ORACLE_TYPE_Q5_0: case ORACLE_TYPE_Q8_0:
        return ORACLE_TYPE_Q8_0;
    case ORACLE_TYPE_Q4_1: case ORACLE_TYPE_Q5_1:
        return ORACLE_TYPE_Q8_1;
    default:
        return -1;
    }
}

/* ---- half conversion (.NET explicit Half<->float casts are IEEE RNE) ---- */
static uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

uint16_t oracle_f32_to_f16(float f) {
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) { /* inf / nan */
        if (man == 0) return (uint16_t)(sign | 0x7C00u);
        return (uint16_t)(sign | 0x7C00u | 0x0200u | (man >> 13));
    }
    int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u); /* overflow -> inf */
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign; /* underflow to signed zero */
        man |= 0x800000u;                   /* implicit one */
        int shift = 14 - e;                 /* 14..24 */
        uint32_t half_man = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_man & 1u))) half_man++;
        return (uint16_t)(sign | half_man); /* may carry into the exponent: still correct */
    }
    uint32_t half = ((uint32_t)e << 10) | (man >> 13);
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++; /* carry may reach inf: correct */
    return (uint16_t)(sign | half);
}

float oracle_f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        /* subnormal: value = man * 2^-24 (exact in f32) */
        float v = (float)man * 5.9604644775390625e-08f;
        return sign ? -v : v;
    }
    if (exp == 31) return bits_f32(sign | 0x7F800000u | (man << 13));
    return bits_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* D1: Math.Round(float) -> Math.Round((double)v), half-to-even.  rint() in the default FP mode. */
static double round_half_even(float v) { return rint((double)v); }

/* (byte)x for a double that is in range by construction */
static uint8_t to_byte(double v) { return (uint8_t)(int)v; }

/* ================= quantize ================= */

/* Ggml.cs:334-377.  First max-|x| element wins (strict <, :349); d = max / -8 (:356);
 * nib = min(15, Round(x*id) + 8) (:366-367, D1); byte l/2 = nib_l | nib_{l+1} << 4 (:372). */
void oracle_quantize_row_q4_0(const float *x, void *vy, int k) {
    block_q4_0 *y = (block_q4_0 *)vy;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int l = 0; l < QK; l++) {
            const float v = x[i * QK + l];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -8;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = d;
        for (int l = 0; l < QK; l += 2) {
            const float v0 = x[i * QK + l + 0] * id;
            const float v1 = x[i * QK + l + 1] * id;
            const uint8_t vi0 = to_byte(fmin(15.0, round_half_even(v0) + 8.0));
            const uint8_t vi1 = to_byte(fmin(15.0, round_half_even(v1) + 8.0));
            y[i].qs[l / 2] = (uint8_t)(vi0 | (vi1 << 4));
        }
    }
}

/* Ggml.cs:487-528.  d = (max-min)/15, nib = (byte)Round((x-min)*id) (D1). */
void oracle_quantize_row_q4_1(const float *x, void *vy, int k) {
    block_q4_1 *y = (block_q4_1 *)vy;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float min = 3.402823466e+38f, max = -3.402823466e+38f;
        for (int l = 0; l < QK; l++) {
            const float v = x[i * QK + l];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d = (max - min) / ((1 << 4) - 1);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = d;
        y[i].m = min;
        for (int l = 0; l < QK; l += 2) {
            const float v0 = (x[i * QK + l + 0] - min) * id;
            const float v1 = (x[i * QK + l + 1] - min) * id;
            const uint8_t vi0 = to_byte(round_half_even(v0));
            const uint8_t vi1 = to_byte(round_half_even(v1));
            y[i].qs[l / 2] = (uint8_t)(vi0 | (vi1 << 4));
        }
    }
}

/* Ggml.cs:547-590, 16-element blocks; D7: the scale is stored as the IEEE half bit pattern. */
void oracle_quantize_row_q4_2(const float *x, void *vy, int k) {
    block_q4_2 *y = (block_q4_2 *)vy;
    const int nb = k / 16;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int l = 0; l < 16; l++) {
            const float v = x[i * 16 + l];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -8;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = oracle_f32_to_f16(d);
        for (int l = 0; l < 16; l += 2) {
            const float v0 = x[i * 16 + l + 0] * id;
            const float v1 = x[i * 16 + l + 1] * id;
            const uint8_t vi0 = to_byte(fmin(15.0, round_half_even(v0) + 8.0));
            const uint8_t vi1 = to_byte(fmin(15.0, round_half_even(v1) + 8.0));
            y[i].qs[l / 2] = (uint8_t)(vi0 | (vi1 << 4));
        }
    }
}

/* Ggml.cs:609-653.  d = max / -16 stored as Half (:632, a real Half field);
 * q = min(31, (int)(x*id + 16.5f)) (:641-642, no Math.Round here); 5th bit of element l -> bit l of qh. */
void oracle_quantize_row_q5_0(const float *x, void *vy, int k) {
    block_q5_0 *y = (block_q5_0 *)vy;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int l = 0; l < QK; l++) {
            const float v = x[i * QK + l];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -16;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = oracle_f32_to_f16(d);
        uint32_t qh = 0;
        for (int l = 0; l < QK; l += 2) {
            const float v0 = x[i * QK + l + 0] * id;
            const float v1 = x[i * QK + l + 1] * id;
            int t0 = (int)(v0 + 16.5f); if (t0 > 31) t0 = 31;
            int t1 = (int)(v1 + 16.5f); if (t1 > 31) t1 = 31;
            const uint32_t vi0 = (uint32_t)t0, vi1 = (uint32_t)t1;
            y[i].qs[l / 2] = (uint8_t)((vi0 & 0x0F) | ((vi1 & 0x0F) << 4));
            qh |= ((vi0 & 0x10) >> 4) << (l + 0);
            qh |= ((vi1 & 0x10) >> 4) << (l + 1);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}

/* Ggml.cs:672-714; D7: d and m stored as IEEE half bit patterns. q = (uint)(v + 0.5f). */
void oracle_quantize_row_q5_1(const float *x, void *vy, int k) {
    block_q5_1 *y = (block_q5_1 *)vy;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float min = 3.402823466e+38f, max = -3.402823466e+38f;
        for (int l = 0; l < QK; l++) {
            const float v = x[i * QK + l];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d = (max - min) / ((1 << 5) - 1);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = oracle_f32_to_f16(d);
        y[i].m = oracle_f32_to_f16(min);
        uint32_t qh = 0;
        for (int l = 0; l < QK; l += 2) {
            const float v0 = (x[i * QK + l + 0] - min) * id;
            const float v1 = (x[i * QK + l + 1] - min) * id;
            const uint32_t vi0 = (uint32_t)(v0 + 0.5f);
            const uint32_t vi1 = (uint32_t)(v1 + 0.5f);
            y[i].qs[l / 2] = (uint8_t)((vi0 & 0x0F) | ((vi1 & 0x0F) << 4));
            qh |= ((vi0 & 0x10) >> 4) << (l + 0);
            qh |= ((vi1 & 0x10) >> 4) << (l + 1);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}

/* Ggml.cs:733-762.  d = amax/127; q = Round(x*id) (D1).  D2: every l in 0..31 is written
 * (the C# loop steps l += 2 and leaves odd quants uninitialised); D4: quants are signed. */
void oracle_quantize_row_q8_0(const float *x, void *vy, int k) {
    block_q8_0 *y = (block_q8_0 *)vy;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int l = 0; l < QK; l++) {
            const float v = x[i * QK + l];
            if (amax < fabsf(v)) amax = fabsf(v);
        }
        const float d = amax / ((1 << 7) - 1);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = d;
        for (int l = 0; l < QK; ++l) {
            const float v0 = x[i * QK + l] * id;
            y[i].qs[l] = (int8_t)(int)round_half_even(v0);
        }
    }
}

/* Ggml.cs:781-823.  D3: l runs 0..15 for both halves, sums are of signed quants;
 * s0 = d * sum(qs[0..15]), s1 = d * sum(qs[16..31]) (:820-821). */
void oracle_quantize_row_q8_1(const float *x, void *vy, int k) {
    block_q8_1 *y = (block_q8_1 *)vy;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int l = 0; l < QK; l++) {
            const float v = x[i * QK + l];
            if (amax < fabsf(v)) amax = fabsf(v);
        }
        const float d = amax / ((1 << 7) - 1);
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        y[i].d = d;
        int sum0 = 0, sum1 = 0;
        for (int l = 0; l < QK / 2; ++l) {
            const float v0 = x[i * QK + l] * id;
            const float v1 = x[i * QK + QK / 2 + l] * id;
            y[i].qs[l] = (int8_t)(int)round_half_even(v0);
            y[i].qs[QK / 2 + l] = (int8_t)(int)round_half_even(v1);
            sum0 += y[i].qs[l];
            sum1 += y[i].qs[QK / 2 + l];
        }
        y[i].s0 = d * sum0;
        y[i].s1 = d * sum1;
    }
}

/* ================= dequantize ================= */

/* Ggml.cs:886-910 (scalar branch; D6) */
void oracle_dequantize_row_q4_0(const void *vx, float *y, int k) {
    const block_q4_0 *x = (const block_q4_0 *)vx;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        const float d = x[i].d;
        for (int l = 0; l < QK; l += 2) {
            const uint8_t vi = x[i].qs[l / 2];
            const int vi0 = vi & 0x0F, vi1 = vi >> 4;
            y[i * QK + l + 0] = (vi0 - 8) * d;
            y[i * QK + l + 1] = (vi1 - 8) * d;
        }
    }
}

/* Ggml.cs:962-987: v = nib * d + m, product and sum rounded separately */
void oracle_dequantize_row_q4_1(const void *vx, float *y, int k) {
    const block_q4_1 *x = (const block_q4_1 *)vx;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        const float d = x[i].d, m = x[i].m;
        for (int l = 0; l < QK; l += 2) {
            const uint8_t vi = x[i].qs[l / 2];
            const int vi0 = vi & 0x0F, vi1 = vi >> 4;
            const float p0 = vi0 * d, p1 = vi1 * d;
            y[i * QK + l + 0] = p0 + m;
            y[i * QK + l + 1] = p1 + m;
        }
    }
}

/* Ggml.cs:992-1022 (D7) */
void oracle_dequantize_row_q4_2(const void *vx, float *y, int k) {
    const block_q4_2 *x = (const block_q4_2 *)vx;
    const int nb = k / 16;
    for (int i = 0; i < nb; i++) {
        const float d = oracle_f16_to_f32(x[i].d);
        for (int l = 0; l < 16; l += 2) {
            const uint8_t vi = x[i].qs[l / 2];
            const int vi0 = vi & 0x0F, vi1 = vi >> 4;
            y[i * 16 + l + 0] = (vi0 - 8) * d;
            y[i * 16 + l + 1] = (vi1 - 8) * d;
        }
    }
}

/* Ggml.cs:1025-1061: value = ((nib | bit_l << 4) - 16) * d */
void oracle_dequantize_row_q5_0(const void *vx, float *y, int k) {
    const block_q5_0 *x = (const block_q5_0 *)vx;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        const float d = oracle_f16_to_f32(x[i].d);
        uint32_t qh; memcpy(&qh, x[i].qh, 4);
        for (int l = 0; l < QK; l += 2) {
            const uint8_t vi = x[i].qs[l / 2];
            const int vh0 = (int)((qh >> (l + 0)) & 1u) << 4;
            const int vh1 = (int)((qh >> (l + 1)) & 1u) << 4;
            const int vi0 = (vi & 0x0F) | vh0;
            const int vi1 = (vi >> 4) | vh1;
            y[i * QK + l + 0] = (vi0 - 16) * d;
            y[i * QK + l + 1] = (vi1 - 16) * d;
        }
    }
}

/* Ggml.cs:1064-1101 (D7) */
void oracle_dequantize_row_q5_1(const void *vx, float *y, int k) {
    const block_q5_1 *x = (const block_q5_1 *)vx;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        const float d = oracle_f16_to_f32(x[i].d);
        const float m = oracle_f16_to_f32(x[i].m);
        uint32_t qh; memcpy(&qh, x[i].qh, 4);
        for (int l = 0; l < QK; l += 2) {
            const uint8_t vi = x[i].qs[l / 2];
            const int vh0 = (int)((qh >> (l + 0)) & 1u) << 4;
            const int vh1 = (int)((qh >> (l + 1)) & 1u) << 4;
            const int vi0 = (vi & 0x0F) | vh0;
            const int vi1 = (vi >> 4) | vh1;
            const float p0 = vi0 * d, p1 = vi1 * d;
            y[i * QK + l + 0] = p0 + m;
            y[i * QK + l + 1] = p1 + m;
        }
    }
}

/* Ggml.cs:1104-1122, D4 signed quants */
void oracle_dequantize_row_q8_0(const void *vx, float *y, int k) {
    const block_q8_0 *x = (const block_q8_0 *)vx;
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        const float d = x[i].d;
        for (int l = 0; l < QK; ++l) y[i * QK + l] = x[i].qs[l] * d;
    }
}

/* ================= dot products ================= */

/* Ggml.cs:1125-1162: sumi int32 per block; sumf += d0*d1*sumi, evaluated (d0*d1)*(float)sumi in f32 */
void oracle_vec_dot_q4_0_q8_0(int n, float *s, const void *vx, const void *vy) {
    const int nb = n / QK;
    const block_q4_0 *x = (const block_q4_0 *)vx;
    const block_q8_0 *y = (const block_q8_0 *)vy;
    float sumf = 0.0f;
    for (int i = 0; i < nb; i++) {
        const float d0 = x[i].d, d1 = y[i].d;
        int sumi = 0;
        for (int j = 0; j < QK / 2; j++) {
            const uint8_t v0 = x[i].qs[j];
            const int i0 = (v0 & 0x0F) - 8, i1 = (v0 >> 4) - 8;
            const int i2 = y[i].qs[2 * j + 0], i3 = y[i].qs[2 * j + 1];
            sumi += i0 * i2 + i1 * i3;
        }
        const float dd = d0 * d1;
        const float t = dd * (float)sumi;
        sumf += t;
    }
    *s = sumf;
}

/* Ggml.cs:1165-1201: per element f32 arithmetic ("very slow" comment, :1185) */
void oracle_vec_dot_q4_1_q8_1(int n, float *s, const void *vx, const void *vy) {
    const int nb = n / QK;
    const block_q4_1 *x = (const block_q4_1 *)vx;
    const block_q8_1 *y = (const block_q8_1 *)vy;
    float sumf = 0.0f;
    for (int i = 0; i < nb; i++) {
        const float d0 = x[i].d, m0 = x[i].m, d1 = y[i].d;
        for (int j = 0; j < QK / 2; j++) {
            const uint8_t v0 = x[i].qs[j];
            const float a0 = d0 * (float)(v0 & 0x0F), a1 = d0 * (float)(v0 >> 4);
            const float f0 = a0 + m0, f1 = a1 + m0;
            const float f2 = d1 * (float)y[i].qs[2 * j + 0];
            const float f3 = d1 * (float)y[i].qs[2 * j + 1];
            const float p0 = f0 * f2, p1 = f1 * f3;
            const float t = p0 + p1;
            sumf += t;
        }
    }
    *s = sumf;
}

/* Ggml.cs:1204-1255: two 16-blocks of Q4_2 against one Q8_0 block */
void oracle_vec_dot_q4_2_q8_0(int n, float *s, const void *vx, const void *vy) {
    const int nb = n / QK;
    const block_q4_2 *x = (const block_q4_2 *)vx;
    const block_q8_0 *y = (const block_q8_0 *)vy;
    float sumf = 0.0f;
    for (int i = 0; i < nb; i++) {
        const uint8_t *x0 = x[2 * i + 0].qs, *x1 = x[2 * i + 1].qs;
        const int8_t *y0 = y[i].qs;
        const float d0 = oracle_f16_to_f32(x[2 * i + 0].d);
        const float d1 = oracle_f16_to_f32(x[2 * i + 1].d);
        int sumi_0 = 0, sumi_1 = 0;
        for (int j = 0; j < QK / 4; j++) {
            const uint8_t v0 = x0[j], v1 = x1[j];
            const int i0_0 = (v0 & 0x0F) - 8, i1_0 = (v0 >> 4) - 8;
            const int i0_1 = (v1 & 0x0F) - 8, i1_1 = (v1 >> 4) - 8;
            const int i2_0 = y0[2 * j + 0], i3_0 = y0[2 * j + 1];
            const int i2_1 = y0[2 * (j + QK / 4) + 0], i3_1 = y0[2 * (j + QK / 4) + 1];
            sumi_0 += i0_0 * i2_0 + i1_0 * i3_0;
            sumi_1 += i0_1 * i2_1 + i1_1 * i3_1;
        }
        const float e0 = d0 * y[i].d, e1 = d1 * y[i].d;
        const float t0 = e0 * (float)sumi_0;
        sumf += t0;
        const float t1 = e1 * (float)sumi_1;
        sumf += t1;
    }
    *s = sumf;
}

/* Ggml.cs:1258-1301: sumf += (d * sxy) * y.d */
void oracle_vec_dot_q5_0_q8_0(int n, float *s, const void *vx, const void *vy) {
    const int nb = n / QK;
    const block_q5_0 *x = (const block_q5_0 *)vx;
    const block_q8_0 *y = (const block_q8_0 *)vy;
    float sumf = 0.0f;
    for (int i = 0; i < nb; i++) {
        uint32_t qh; memcpy(&qh, x[i].qh, 4);
        const float d = oracle_f16_to_f32(x[i].d);
        int sxy = 0;
        for (int j = 0; j < QK / 2; j++) {
            const uint8_t v0 = x[i].qs[j];
            const int x0_0h = (int)((qh >> (2 * j + 0)) & 1u) << 4;
            const int x1_0h = (int)((qh >> (2 * j + 1)) & 1u) << 4;
            const int x0_0 = ((v0 & 0x0F) | x0_0h) - 16;
            const int x1_0 = ((v0 >> 4) | x1_0h) - 16;
            sxy += x0_0 * y[i].qs[2 * j + 0] + x1_0 * y[i].qs[2 * j + 1];
        }
        const float a = d * (float)sxy;
        const float t = a * y[i].d;
        sumf += t;
    }
    *s = sumf;
}

/* Ggml.cs:1304-1348: sumf += (d*sxy)*y.d + m*(y.s0 + y.s1)  (D7: d, m are IEEE halves) */
void oracle_vec_dot_q5_1_q8_1(int n, float *s, const void *vx, const void *vy) {
    const int nb = n / QK;
    const block_q5_1 *x = (const block_q5_1 *)vx;
    const block_q8_1 *y = (const block_q8_1 *)vy;
    float sumf = 0.0f;
    for (int i = 0; i < nb; i++) {
        uint32_t qh; memcpy(&qh, x[i].qh, 4);
        const float d = oracle_f16_to_f32(x[i].d);
        const float m = oracle_f16_to_f32(x[i].m);
        int sxy = 0;
        for (int j = 0; j < QK / 2; j++) {
            const uint8_t v0 = x[i].qs[j];
            const int x0_0h = (int)((qh >> (2 * j + 0)) & 1u) << 4;
            const int x1_0h = (int)((qh >> (2 * j + 1)) & 1u) << 4;
            const int x0_0 = (v0 & 0x0F) | x0_0h;
            const int x1_0 = (v0 >> 4) | x1_0h;
            sxy += x0_0 * y[i].qs[2 * j + 0] + x1_0 * y[i].qs[2 * j + 1];
        }
        const float a = d * (float)sxy;
        const float b = a * y[i].d;
        const float ss = y[i].s0 + y[i].s1;
        const float c = m * ss;
        const float t = b + c;
        sumf += t;
    }
    *s = sumf;
}

/* Ggml.cs:1351-1381: sumf += (x.d * y.d) * sumi (D4 signed) */
void oracle_vec_dot_q8_0_q8_0(int n, float *s, const void *vx, const void *vy) {
    const int nb = n / QK;
    const block_q8_0 *x = (const block_q8_0 *)vx;
    const block_q8_0 *y = (const block_q8_0 *)vy;
    float sumf = 0.0f;
    for (int i = 0; i < nb; i++) {
        int sumi = 0;
        for (int j = 0; j < QK; j++) sumi += x[i].qs[j] * y[i].qs[j];
        const float dd = x[i].d * y[i].d;
        const float t = dd * (float)sumi;
        sumf += t;
    }
    *s = sumf;
}

/* Ggml.cs:2631-2640: product in f32, running sum in f64, one cast at the end */
void oracle_vec_dot_f32(int n, float *s, const float *x, const float *y) {
    double sumf = 0.0;
    for (int i = 0; i < n; ++i) {
        const float p = x[i] * y[i];
        sumf += (double)p;
    }
    *s = (float)sumf;
}

/* Ggml.cs:2642-2651 */
void oracle_vec_dot_f16(int n, float *s, const uint16_t *x, const uint16_t *y) {
    double sumf = 0.0;
    for (int i = 0; i < n; ++i) {
        const float p = oracle_f16_to_f32(x[i]) * oracle_f16_to_f32(y[i]);
        sumf += (double)p;
    }
    *s = (float)sumf;
}

/* ================= quantize_fns[] dispatch (Ggml.cs:219-290) ================= */

int oracle_quantize_row(int type, const float *x, void *y, int k) {
    switch (type) {
    case ORACLE_TYPE_Q4_0: oracle_quantize_row_q4_0(x, y, k); return 0; /* D5 */
    case ORACLE_TYPE_Q4_1: oracle_quantize_row_q4_1(x, y, k); return 0;
    case ORACLE_TYPE_Q4_2: oracle_quantize_row_q4_2(x, y, k); return 0;
    case ORACLE_TYPE_Q5_0: oracle_quantize_row_q5_0(x, y, k); return 0;
    case ORACLE_TYPE_Q5_1: oracle_quantize_row_q5_1(x, y, k); return 0;
    case ORACLE_TYPE_Q8_0: oracle_quantize_row_q8_0(x, y, k); return 0;
    case ORACLE_TYPE_Q8_1: oracle_quantize_row_q8_1(x, y, k); return 0;
    default: return -1;
    }
}

int oracle_dequantize_row(int type, const void *x, float *y, int k) {
    switch (type) {
    case ORACLE_TYPE_Q4_0: oracle_dequantize_row_q4_0(x, y, k); return 0;
    case ORACLE_TYPE_Q4_1: oracle_dequantize_row_q4_1(x, y, k); return 0;
    case ORACLE_TYPE_Q4_2: oracle_dequantize_row_q4_2(x, y, k); return 0;
    case ORACLE_TYPE_Q5_0: oracle_dequantize_row_q5_0(x, y, k); return 0;
    case ORACLE_TYPE_Q5_1: oracle_dequantize_row_q5_1(x, y, k); return 0;
    case ORACLE_TYPE_Q8_0: oracle_dequantize_row_q8_0(x, y, k); return 0;
    default: return -1; /* Q8_1.dequantize_row_q is null (Ggml.cs:278), Q4_3 is default (:248) */
    }
}

int oracle_quantize_row_dot(int type, const float *x, void *y, int k) {
    const int vt = oracle_vec_dot_type(type);
    if (type == ORACLE_TYPE_Q8_1) { oracle_quantize_row_q8_1(x, y, k); return 0; } /* Ggml.cs:281 */
    if (vt == ORACLE_TYPE_Q8_0) { oracle_quantize_row_q8_0(x, y, k); return 0; }
    if (vt == ORACLE_TYPE_Q8_1) { oracle_quantize_row_q8_1(x, y, k); return 0; }
    return -1;
}

int oracle_vec_dot(int type, int n, float *s, const void *vx, const void *vy) {
    switch (type) {
    case ORACLE_TYPE_Q4_0: oracle_vec_dot_q4_0_q8_0(n, s, vx, vy); return 0;
    case ORACLE_TYPE_Q4_1: oracle_vec_dot_q4_1_q8_1(n, s, vx, vy); return 0;
    case ORACLE_TYPE_Q4_2: oracle_vec_dot_q4_2_q8_0(n, s, vx, vy); return 0;
    case ORACLE_TYPE_Q5_0: oracle_vec_dot_q5_0_q8_0(n, s, vx, vy); return 0;
    case ORACLE_TYPE_Q5_1: oracle_vec_dot_q5_1_q8_1(n, s, vx, vy); return 0;
    case ORACLE_TYPE_Q8_0: oracle_vec_dot_q8_0_q8_0(n, s, vx, vy); return 0;
    default: return -1; /* D8 */
    }
}

/* ================= mul_mat drivers ================= */

static int64_t nelements(const oracle_tensor *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }

/* Ggml.cs:3329-3386 */
size_t oracle_mul_mat_work_size(const oracle_tensor *src0, const oracle_tensor *src1) {
    if (src0->type == ORACLE_TYPE_F16 && src1->type == ORACLE_TYPE_F32)
        return TYPE_SIZE[ORACLE_TYPE_F16] * (size_t)nelements(src1);          /* :3356-3357 */
    if (src0->type == ORACLE_TYPE_F32 && src1->type == ORACLE_TYPE_F32) return 0; /* :3360-3364 */
    if (oracle_is_quantized(src0->type) && src1->type == ORACLE_TYPE_F32) {   /* :3365-3378 */
        const int tq = oracle_vec_dot_type(src0->type);
        if (tq < 0) return 0;
        return TYPE_SIZE[tq] * (size_t)nelements(src1) / (size_t)BLCK_SIZE[tq];
    }
    return 0;
}

typedef struct {
    const oracle_tensor *src0, *src1, *dst;
    uint8_t *wdata;
    int ith, nth;
} mm_task;

/* COMPUTE phase of Ggml.cs:6117-6163 (f32), 6381-6425 (f16), 6657-6699 (q) for thread ith of nth */
static void mm_compute(const mm_task *t) {
    const oracle_tensor *src0 = t->src0, *src1 = t->src1, *dst = t->dst;
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    const int64_t ne11 = src1->ne[1], ne12 = src1->ne[2];
    const uint64_t nb01 = src0->nb[1], nb02 = src0->nb[2], nb03 = src0->nb[3];
    const uint64_t nb11 = src1->nb[1], nb12 = src1->nb[2], nb13 = src1->nb[3];
    const uint64_t nb0 = dst->nb[0], nb1 = dst->nb[1], nb2 = dst->nb[2], nb3 = dst->nb[3];
    const uint64_t ne0 = (uint64_t)dst->ne[0];

    const uint64_t nr = (uint64_t)(ne01 * ne02 * ne03);
    const uint64_t dr = (nr + (uint64_t)t->nth - 1) / (uint64_t)t->nth;
    const uint64_t ir0 = dr * (uint64_t)t->ith;
    const uint64_t ir1 = ir0 + dr < nr ? ir0 + dr : nr;

    const int type = src0->type;
    size_t row_size = 0;
    if (oracle_is_quantized(type)) {
        const int vt = oracle_vec_dot_type(type);
        row_size = (size_t)ne00 * TYPE_SIZE[vt] / (size_t)BLCK_SIZE[vt];
    }

    for (uint64_t ir = ir0; ir < ir1; ++ir) {
        const uint64_t i03 = ir / (uint64_t)(ne02 * ne01);
        const uint64_t i02 = (ir - i03 * ne02 * ne01) / (uint64_t)ne01;
        const uint64_t i01 = ir - i03 * ne02 * ne01 - i02 * ne01;
        const uint64_t i13 = i03, i12 = i02;
        const uint8_t *src0_row = (const uint8_t *)src0->data + (i01 * nb01 + i02 * nb02 + i03 * nb03);
        if (type == ORACLE_TYPE_F32) {
            for (int64_t ic = 0; ic < ne11; ++ic) {
                float *d = (float *)((uint8_t *)dst->data + (i01 * nb0 + (uint64_t)ic * nb1 + i02 * nb2 + i03 * nb3));
                const float *y = (const float *)((const uint8_t *)src1->data + ((uint64_t)ic * nb11 + i12 * nb12 + i13 * nb13));
                oracle_vec_dot_f32((int)ne00, d, (const float *)src0_row, y);
            }
        } else if (type == ORACLE_TYPE_F16) {
            const uint16_t *src1_col = (const uint16_t *)t->wdata + (0 + i12 * ne11 + i13 * ne12 * ne11) * ne00;
            float *dst_col = (float *)((uint8_t *)dst->data + (i01 * nb0 + 0 * nb1 + i02 * nb2 + i03 * nb3));
            for (int64_t ic = 0; ic < ne11; ++ic)
                oracle_vec_dot_f16((int)ne00, &dst_col[(uint64_t)ic * ne0], (const uint16_t *)src0_row, src1_col + ic * ne00);
        } else {
            const uint8_t *src1_col = t->wdata + (0 + i12 * ne11 + i13 * ne12 * ne11) * row_size;
            float *dst_col = (float *)((uint8_t *)dst->data + (i01 * nb0 + 0 * nb1 + i02 * nb2 + i03 * nb3));
            for (int64_t ic = 0; ic < ne11; ++ic)
                oracle_vec_dot(type, (int)ne00, &dst_col[(uint64_t)ic * ne0], src0_row, src1_col + (size_t)ic * row_size);
        }
    }
}

static void *mm_thread(void *arg) { mm_compute((const mm_task *)arg); return NULL; }

int oracle_mul_mat(const oracle_tensor *src0, const oracle_tensor *src1, const oracle_tensor *dst,
                   void *wdata, size_t wsize, int nth) {
    const int type = src0->type;
    if (!(type == ORACLE_TYPE_F32 || type == ORACLE_TYPE_F16 || oracle_is_quantized(type))) return -1;
    if (oracle_is_quantized(type) && (oracle_vec_dot_type(type) < 0 || type == ORACLE_TYPE_Q8_1)) return -1; /* D8 */
    if (src1->type != ORACLE_TYPE_F32 || dst->type != ORACLE_TYPE_F32) return -2;
    /* ggml_can_mul_mat (Ggml.cs:8345-8353) + the asserts of each driver */
    if (src0->ne[0] != src1->ne[0] || src0->ne[2] != src1->ne[2] || src0->ne[3] != src1->ne[3]) return -2;
    if (dst->ne[0] != src0->ne[1] || dst->ne[1] != src1->ne[1] || dst->ne[2] != src0->ne[2] || dst->ne[3] != src0->ne[3]) return -2;
    if (src0->nb[0] != TYPE_SIZE[type] || src1->nb[0] != 4 || dst->nb[0] != 4) return -2;
    if (src0->nb[0] > src0->nb[1]) return -2; /* !ggml_is_transposed(a), Ggml.cs:8229 */
    if (!(dst->nb[0] <= dst->nb[1] && dst->nb[1] <= dst->nb[2] && dst->nb[2] <= dst->nb[3])) return -2;
    if (oracle_is_quantized(type) && src0->ne[0] % 32 != 0) return -2; /* Ggml.cs:6694 */
    if (nth < 1) nth = 1;
    const size_t need = oracle_mul_mat_work_size(src0, src1);
    if (need > 0 && (wdata == NULL || wsize < need)) return -2;

    /* INIT phase, thread 0 only (Ggml.cs:3554-3563) */
    const int64_t ne10 = src1->ne[0], ne11 = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    if (type == ORACLE_TYPE_F16) { /* Ggml.cs:6362-6379 */
        uint16_t *w = (uint16_t *)wdata;
        size_t id = 0;
        for (int64_t i13 = 0; i13 < ne13; ++i13)
            for (int64_t i12 = 0; i12 < ne12; ++i12)
                for (int64_t i11 = 0; i11 < ne11; ++i11)
                    for (int64_t i10 = 0; i10 < ne10; ++i10)
                        w[id++] = oracle_f32_to_f16(*(const float *)((const uint8_t *)src1->data +
                                    i13 * src1->nb[3] + i12 * src1->nb[2] + i11 * src1->nb[1] + i10 * src1->nb[0]));
    } else if (oracle_is_quantized(type)) { /* Ggml.cs:6641-6654 */
        const int vt = oracle_vec_dot_type(type);
        const size_t row_size = (size_t)ne10 * TYPE_SIZE[vt] / (size_t)BLCK_SIZE[vt];
        uint8_t *w = (uint8_t *)wdata;
        for (int64_t i13 = 0; i13 < ne13; ++i13)
            for (int64_t i12 = 0; i12 < ne12; ++i12)
                for (int64_t i11 = 0; i11 < ne11; ++i11) {
                    oracle_quantize_row_dot(type, (const float *)((const uint8_t *)src1->data +
                                    i13 * src1->nb[3] + i12 * src1->nb[2] + i11 * src1->nb[1]), w, (int)ne10);
                    w += row_size;
                }
    }

    /* COMPUTE phase on nth threads (Ggml.cs:3604-3605, 8487-8546); FINALIZE is a no-op */
    mm_task *tasks = (mm_task *)malloc(sizeof(mm_task) * (size_t)nth);
    pthread_t *thr = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nth);
    if (!tasks || !thr) { free(tasks); free(thr); return -2; }
    for (int i = 0; i < nth; ++i) {
        tasks[i].src0 = src0; tasks[i].src1 = src1; tasks[i].dst = dst;
        tasks[i].wdata = (uint8_t *)wdata; tasks[i].ith = i; tasks[i].nth = nth;
    }
    for (int i = 1; i < nth; ++i) pthread_create(&thr[i], NULL, mm_thread, &tasks[i]);
    mm_compute(&tasks[0]);
    for (int i = 1; i < nth; ++i) pthread_join(thr[i], NULL);
    free(tasks); free(thr);
    return 0;
}

/* ================= neighbours of the path: cpy (f32/f16 -> Q) and add_q_f32 ================= */

/* every type with both a quantize_row_q and a dequantize_row_q slot (Ggml.cs:219-290; Q4_2 / Q5_1 per D7) */
static int q_rw_ok(int t) {
    return t == ORACLE_TYPE_Q4_0 || t == ORACLE_TYPE_Q4_1 || t == ORACLE_TYPE_Q4_2 || t == ORACLE_TYPE_Q5_0 || t == ORACLE_TYPE_Q5_1 ||
           t == ORACLE_TYPE_Q8_0;
}

/* Ggml.cs:4339-4363 (dup_f32) and 3935-3966 (dup_f16): rows of src0 in (i03, i02, i01) order, dst written densely */
int oracle_cpy_to_q(const oracle_tensor *src0, const oracle_tensor *dst) {
    if ((src0->type != ORACLE_TYPE_F32 && src0->type != ORACLE_TYPE_F16) || !q_rw_ok(dst->type)) return -1;
    if (nelements(src0) != nelements(dst)) return -2;                       /* Ggml.cs:8281 */
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    if (ne00 % 32 != 0 || src0->nb[0] != TYPE_SIZE[src0->type]) return -2;
    const size_t rs = TYPE_SIZE[dst->type] * (size_t)(ne00 / BLCK_SIZE[dst->type]);   /* :4345 */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)ne00);
    if (!tmp) return -2;
    size_t id = 0;
    for (int64_t i03 = 0; i03 < ne03; i03++)
        for (int64_t i02 = 0; i02 < ne02; i02++)
            for (int64_t i01 = 0; i01 < ne01; i01++) {
                const uint8_t *row = (const uint8_t *)src0->data + i01 * src0->nb[1] + i02 * src0->nb[2] + i03 * src0->nb[3];
                const float *xf = (const float *)row;
                if (src0->type == ORACLE_TYPE_F16) {                        /* :3951-3956 */
                    for (int64_t i = 0; i < ne00; i++) tmp[i] = oracle_f16_to_f32(((const uint16_t *)row)[i]);
                    xf = tmp;
                }
                oracle_quantize_row(dst->type, xf, (uint8_t *)dst->data + id, (int)ne00);
                id += rs;
            }
    free(tmp);
    return 0;
}

int oracle_add_q_f32(const oracle_tensor *src0, const oracle_tensor *src1, const oracle_tensor *dst) {
    const int t = src0->type;
    if (!q_rw_ok(t) || dst->type != t || src1->type != ORACLE_TYPE_F32) return -1;
    for (int i = 0; i < 4; i++) if (src0->ne[i] != src1->ne[i] || src0->ne[i] != dst->ne[i]) return -2;
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    if (ne00 % 32 != 0 || src0->nb[0] != TYPE_SIZE[t] || src1->nb[0] != 4 || dst->nb[0] != TYPE_SIZE[t]) return -2;
    float *w = (float *)malloc(sizeof(float) * (size_t)ne00);
    if (!w) return -2;
    for (int64_t i03 = 0; i03 < ne03; i03++)
        for (int64_t i02 = 0; i02 < ne02; i02++)
            for (int64_t i01 = 0; i01 < ne01; i01++) {
                const uint8_t *a = (const uint8_t *)src0->data + i01 * src0->nb[1] + i02 * src0->nb[2] + i03 * src0->nb[3];
                const float *b = (const float *)((const uint8_t *)src1->data + i01 * src1->nb[1] + i02 * src1->nb[2] + i03 * src1->nb[3]);
                uint8_t *d = (uint8_t *)dst->data + i01 * dst->nb[1] + i02 * dst->nb[2] + i03 * dst->nb[3];
                oracle_dequantize_row(t, a, w, (int)ne00);                  /* :4896 */
                for (int64_t i = 0; i < ne00; i++) w[i] += b[i];           /* :4898, ggml_vec_acc_f32 */
                oracle_quantize_row(t, w, d, (int)ne00);                    /* :4900 */
            }
    free(w);
    return 0;
}

/* Ggml.cs:5705-5748, 2737-2746, 2723-2726, table 1455-1471 (A2: indexed by bit pattern) */
void oracle_silu_f32(int64_t nr, int64_t nc, const float *x, float *y) {
    for (int64_t i = 0; i < nr * nc; i++) {
        const float f = oracle_f16_to_f32(oracle_f32_to_f16(x[i]));      /* Half fp16 = (Half)x[i]; f = table_f32_f16[bits] */
        const float s = f / (1.0f + expf(-f));                          /* ggml_silu_f32 */
        y[i] = oracle_f16_to_f32(oracle_f32_to_f16(s));                  /* (float)table_silu_f16[bits] */
    }
}

/* ================= Test3 LCG (Test3/Program.cs:98-107) ================= */
static uint64_t lcg_next = 1;
void oracle_xsrand(uint64_t seed) { lcg_next = seed; }
uint32_t oracle_xrand(void) {
    lcg_next = lcg_next * 214013ULL + 2531011ULL;
    return (uint32_t)((lcg_next >> 16) & 0x7FFF);
}

/* ================= element-wise neighbours of mul_mat ================= */

/* Ggml.cs:4622-4682, contiguous branch: ggml_vec_add_f32 z[i] = x[i] + y[i] per row */
void oracle_add_f32(int64_t nr, int64_t nc, const float *x, const float *y, float *z) {
    for (int64_t j = 0; j < nr; j++)
        for (int64_t i = 0; i < nc; i++) z[j * nc + i] = x[j * nc + i] + y[j * nc + i];
}

/* Ggml.cs:5007-5035: ggml_vec_mul_f32 z[i] = x[i] * y[i] per row */
void oracle_mul_f32(int64_t nr, int64_t nc, const float *x, const float *y, float *z) {
    for (int64_t j = 0; j < nr; j++)
        for (int64_t i = 0; i < nc; i++) z[j * nc + i] = x[j * nc + i] * y[j * nc + i];
}

/* Ggml.cs:6746-6778: v = *(float *)src1->data; ggml_vec_scale_f32(nc, dst row, v) -- dst is a view of src0 (:8265) */
void oracle_scale_f32(int64_t nr, int64_t nc, float *z, float v) {
    for (int64_t j = 0; j < nr; j++)
        for (int64_t i = 0; i < nc; i++) z[j * nc + i] *= v;
}

/* Ggml.cs:5858-5920 */
void oracle_rms_norm_f32(int64_t nr, int64_t nc, const float *x, float *y) {
    const float eps = 1e-6f;                                  /* :5889 */
    for (int64_t j = 0; j < nr; j++) {
        const float *xr = x + j * nc;
        double sum = 0.0;                                     /* :5900 */
        for (int64_t i = 0; i < nc; i++) {
            const float sq = xr[i] * xr[i];                   /* float product, then widened (:5903) */
            sum += (double)sq;
        }
        const float mean = (float)(sum / (double)nc);         /* :5906 */
        const float scale = 1.0f / sqrtf(mean + eps);         /* :5915 */
        for (int64_t i = 0; i < nc; i++) y[j * nc + i] = xr[i] * scale;   /* copy + ggml_vec_scale_f32 (:5910-5917) */
    }
}
