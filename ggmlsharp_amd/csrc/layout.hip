// layout.hip -- re-layout kernels between the reference's AoS block rows and the resident planar layout.
//
// The reference stores a quantized row as consecutive AoS blocks (TypeDefinitions.cs:236-290; 20/24/22/36 bytes),
// which is awkward for 16-byte coalesced loads.  Weights are therefore uploaded once and kept block-major
// ("planar"): for k-block b, the 16 quant bytes of every row are contiguous, and the scales are a separate
// f32 plane.  planar_to_aos is the exact inverse, so a download is byte-identical to the upload.
#include "common.h"

namespace {

__device__ __forceinline__ float half_bits_to_float(uint16_t h) {
    // IEEE binary16 -> binary32, exact ((float)(Half) in Ggml.cs:1034, 1277)
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    const uint32_t man = h & 0x3FFu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) {
            out = sign;
        } else {
            const float v = (float)man * 5.9604644775390625e-08f;  // man * 2^-24, exact
            out = __float_as_uint(v) | sign;
        }
    } else if (exp == 31) {
        out = sign | 0x7F800000u | (man << 13);
    } else {
        out = sign | ((exp + 112u) << 23) | (man << 13);
    }
    return __uint_as_float(out);
}

__device__ __forceinline__ uint16_t float_to_half_bits_exact(float f) {
    // inverse of half_bits_to_float for values that came from a half (no rounding needed)
    const uint32_t x = __float_as_uint(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t exp = (x >> 23) & 0xFFu;
    const uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (man >> 13));
    if (exp == 0 && man == 0) return (uint16_t)sign;
    const int e = (int)exp - 127 + 15;
    if (e <= 0) {  // was a half subnormal: value = hm * 2^-24
        const uint32_t hm = (uint32_t)(fabsf(f) * 16777216.0f);
        return (uint16_t)(sign | hm);
    }
    return (uint16_t)(sign | ((uint32_t)e << 10) | (man >> 13));
}

// one thread per (row, k-block); rows fastest so the planar stores coalesce
template <int TYPE>
__global__ void repack_to_planar_kernel(const uint8_t *__restrict__ aos, uint64_t nb01, int64_t row_begin, int64_t rows,
                                        int64_t Mpad, int64_t nbk, uint8_t *__restrict__ qs, uint32_t *__restrict__ qh,
                                        float *__restrict__ d, float *__restrict__ mm) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (m >= rows) return;
    const uint8_t *src = aos + (uint64_t)(row_begin + m) * nb01;
    const int64_t pi = b * Mpad + m;
    if (TYPE == GGML_TYPE_Q4_0) {
        const uint32_t *s = (const uint32_t *)(src + b * 20);
        d[pi] = __uint_as_float(s[0]);
        *(uint4 *)(qs + pi * 16) = make_uint4(s[1], s[2], s[3], s[4]);
    } else if (TYPE == GGML_TYPE_Q4_1) {
        const uint32_t *s = (const uint32_t *)(src + b * 24);
        d[pi] = __uint_as_float(s[0]);
        mm[pi] = __uint_as_float(s[1]);
        *(uint4 *)(qs + pi * 16) = make_uint4(s[2], s[3], s[4], s[5]);
    } else if (TYPE == GGML_TYPE_Q5_0) {
        const uint16_t *s = (const uint16_t *)(src + b * 22);
        d[pi] = half_bits_to_float(s[0]);
        qh[pi] = (uint32_t)s[1] | ((uint32_t)s[2] << 16);
        uint32_t q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = (uint32_t)s[3 + 2 * i] | ((uint32_t)s[4 + 2 * i] << 16);
        *(uint4 *)(qs + pi * 16) = make_uint4(q[0], q[1], q[2], q[3]);
    } else if (TYPE == GGML_TYPE_Q5_1) {
        const uint16_t *s = (const uint16_t *)(src + b * 24);
        d[pi] = half_bits_to_float(s[0]);
        mm[pi] = half_bits_to_float(s[1]);
        const uint32_t *s4 = (const uint32_t *)(src + b * 24);
        qh[pi] = s4[1];
        *(uint4 *)(qs + pi * 16) = make_uint4(s4[2], s4[3], s4[4], s4[5]);
    } else if (TYPE == GGML_TYPE_Q4_2) {
        const uint16_t *s = (const uint16_t *)(src + b * 20);              // two 10-byte blocks: {d, qs[8]} {d, qs[8]}
        d[pi] = half_bits_to_float(s[0]);
        mm[pi] = half_bits_to_float(s[5]);
        uint32_t q[4];
        q[0] = (uint32_t)s[1] | ((uint32_t)s[2] << 16); q[1] = (uint32_t)s[3] | ((uint32_t)s[4] << 16);
        q[2] = (uint32_t)s[6] | ((uint32_t)s[7] << 16); q[3] = (uint32_t)s[8] | ((uint32_t)s[9] << 16);
        *(uint4 *)(qs + pi * 16) = make_uint4(q[0], q[1], q[2], q[3]);
    } else if (TYPE == GGML_TYPE_Q8_0) {
        const uint32_t *s = (const uint32_t *)(src + b * 36);
        d[pi] = __uint_as_float(s[0]);
        // split the 32 quants into even / odd elements: plane h byte j = element 2j + h
        uint32_t ev[4], od[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = s[1 + 2 * i], hi = s[2 + 2 * i];  // elements 8i..8i+3, 8i+4..8i+7
            ev[i] = (lo & 0xFFu) | ((lo >> 8) & 0xFF00u) | ((hi & 0xFFu) << 16) | ((hi << 8) & 0xFF000000u);
            od[i] = ((lo >> 8) & 0xFFu) | ((lo >> 16) & 0xFF00u) | ((hi << 8) & 0xFF0000u) | (hi & 0xFF000000u);
        }
        *(uint4 *)(qs + ((b * 2 + 0) * Mpad + m) * 16) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
        *(uint4 *)(qs + ((b * 2 + 1) * Mpad + m) * 16) = make_uint4(od[0], od[1], od[2], od[3]);
    }
}

template <int TYPE>
__global__ void planar_to_aos_kernel(uint8_t *__restrict__ aos, uint64_t nb01, int64_t rows, int64_t Mpad, int64_t nbk,
                                     const uint8_t *__restrict__ qs, const uint32_t *__restrict__ qh,
                                     const float *__restrict__ d, const float *__restrict__ mm) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (m >= rows) return;
    uint8_t *dst = aos + (uint64_t)m * nb01;
    const int64_t pi = b * Mpad + m;
    if (TYPE == GGML_TYPE_Q4_0) {
        uint32_t *s = (uint32_t *)(dst + b * 20);
        const uint4 q = *(const uint4 *)(qs + pi * 16);
        s[0] = __float_as_uint(d[pi]); s[1] = q.x; s[2] = q.y; s[3] = q.z; s[4] = q.w;
    } else if (TYPE == GGML_TYPE_Q4_1) {
        uint32_t *s = (uint32_t *)(dst + b * 24);
        const uint4 q = *(const uint4 *)(qs + pi * 16);
        s[0] = __float_as_uint(d[pi]); s[1] = __float_as_uint(mm[pi]);
        s[2] = q.x; s[3] = q.y; s[4] = q.z; s[5] = q.w;
    } else if (TYPE == GGML_TYPE_Q5_0) {
        uint16_t *s = (uint16_t *)(dst + b * 22);
        const uint4 q = *(const uint4 *)(qs + pi * 16);
        const uint32_t h = qh[pi];
        s[0] = float_to_half_bits_exact(d[pi]);
        s[1] = (uint16_t)h; s[2] = (uint16_t)(h >> 16);
        const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[3 + 2 * i] = (uint16_t)qq[i]; s[4 + 2 * i] = (uint16_t)(qq[i] >> 16); }
    } else if (TYPE == GGML_TYPE_Q5_1) {
        uint16_t *s = (uint16_t *)(dst + b * 24);
        uint32_t *s4 = (uint32_t *)(dst + b * 24);
        const uint4 q = *(const uint4 *)(qs + pi * 16);
        s[0] = float_to_half_bits_exact(d[pi]);
        s[1] = float_to_half_bits_exact(mm[pi]);
        s4[1] = qh[pi]; s4[2] = q.x; s4[3] = q.y; s4[4] = q.z; s4[5] = q.w;
    } else if (TYPE == GGML_TYPE_Q4_2) {
        uint16_t *s = (uint16_t *)(dst + b * 20);
        const uint4 q = *(const uint4 *)(qs + pi * 16);
        s[0] = float_to_half_bits_exact(d[pi]);
        s[1] = (uint16_t)q.x; s[2] = (uint16_t)(q.x >> 16); s[3] = (uint16_t)q.y; s[4] = (uint16_t)(q.y >> 16);
        s[5] = float_to_half_bits_exact(mm[pi]);
        s[6] = (uint16_t)q.z; s[7] = (uint16_t)(q.z >> 16); s[8] = (uint16_t)q.w; s[9] = (uint16_t)(q.w >> 16);
    } else if (TYPE == GGML_TYPE_Q8_0) {
        uint32_t *s = (uint32_t *)(dst + b * 36);
        const uint4 e4 = *(const uint4 *)(qs + ((b * 2 + 0) * Mpad + m) * 16);
        const uint4 o4 = *(const uint4 *)(qs + ((b * 2 + 1) * Mpad + m) * 16);
        const uint32_t ev[4] = {e4.x, e4.y, e4.z, e4.w}, od[4] = {o4.x, o4.y, o4.z, o4.w};
        s[0] = __float_as_uint(d[pi]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t e = ev[i], o = od[i];
            s[1 + 2 * i] = (e & 0xFFu) | ((o & 0xFFu) << 8) | ((e & 0xFF00u) << 8) | ((o & 0xFF00u) << 16);
            s[2 + 2 * i] = ((e >> 16) & 0xFFu) | ((o >> 8) & 0xFF00u) | ((e >> 8) & 0xFF0000u) | (o & 0xFF000000u);
        }
    }
}

// dense rows: plain strided copy (element size es bytes), rows fastest across blocks, 16 B per thread where possible
__global__ void copy_rows_kernel(const uint8_t *__restrict__ src, uint64_t src_stride, uint8_t *__restrict__ dst,
                                 uint64_t dst_stride, int64_t rows, int64_t row_bytes) {
    // rows walk a grid-stride loop in y (grid.y is clamped to 65535 by the launchers); a row of an odd number of
    // halves ends in a 2-byte tail
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c >= row_bytes) return;
    for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
        if (c + 4 <= row_bytes) *(uint32_t *)(dst + r * dst_stride + c) = *(const uint32_t *)(src + r * src_stride + c);
        else *(uint16_t *)(dst + r * dst_stride + c) = *(const uint16_t *)(src + r * src_stride + c);
    }
}

// [G][N][Ms] (rank-major all-gather result) -> [N][ldd] with column r*Ms + i
__global__ void relayout_gathered_kernel(const float *__restrict__ g, int G, int64_t N, int64_t Ms, float *__restrict__ dst,
                                         int64_t M, int64_t ldd) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // global column in [0, G*Ms)
    if (col >= M || col >= (int64_t)G * Ms) return;
    const int64_t r = col / Ms, i = col - r * Ms;
    for (int64_t n = blockIdx.y; n < N; n += gridDim.y) dst[n * ldd + col] = g[(r * N + n) * Ms + i];
}

// Direct all-gather of dst shards by the compute units: the [N][Ms] shard (row stride lds) is stored as columns
// [col0, col0 + Ms) of every peer's reference-layout [N][ldd] buffer -- peers reached over xGMI through peer-mapped /
// IPC-opened pointers.  One hop, every link of the sender busy at once (blockIdx.y = peer), already in the final layout:
// no [G][N][Ms] intermediate and no re-layout pass (SURVEY 8(e) "epilogue peer-writes").  16-byte accesses when aligned.
struct PeerPtrs { float *p[16]; };
template <bool VEC>
__global__ __launch_bounds__(256) void push_columns_kernel(const float *__restrict__ src, int64_t lds, int64_t N, int64_t Ms, PeerPtrs peers,
                                                           int64_t ldd, int64_t col0) {
    float *__restrict__ dst = peers.p[blockIdx.y];
    if (dst == nullptr) return;                     // (the sender's own slot when it computed in place)
    if (VEC) {
        const int64_t w4 = Ms / 4, total = N * w4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = i / w4, c = (i - r * w4) * 4;
            *(float4 *)(dst + r * ldd + col0 + c) = *(const float4 *)(src + r * lds + c);
        }
    } else {
        const int64_t total = N * Ms;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = i / Ms, c = i - r * Ms;
            dst[r * ldd + col0 + c] = src[r * lds + c];
        }
    }
}

// bf6 (e3m2) code of an integer |v| <= 8: 1 = 2^0, 2, 3 = 1.5 * 2, 4, 5 = 1.25 * 4, 6, 7, 8 (all exact)
__device__ __forceinline__ uint32_t bf6_code(int v) {
    const uint64_t tab = 0ull | (12ull << 5) | (16ull << 10) | (18ull << 15) | (20ull << 20) | (21ull << 25) | (22ull << 30) |
                         (23ull << 35) | (24ull << 40);
    const int a = v < 0 ? -v : v;
    return (uint32_t)((tab >> (5 * a)) & 31u) | (v < 0 ? 32u : 0u);
}

// resident quant planes -> bf6 operand planes of gemm_qmx.hip, [nbk][NF][Mpad][16 B] and [nbk][NF][Mpad][8 B]: element e of
// the block at bits [6e, 6e+5] of a 192-bit fragment.  Q4_0 / Q4_1 (NF = 1): the value nib - 8 (Q4_1's + 8 moves into
// its min term).  Q5_0 / Q8_0 (NF = 2): w = 16 * wh + wl, fragment 0 = the low digits wl in [-8, 7], fragment 1 = the
// high digits wh = floor((w + 8) / 16).
template <int TYPE>
__global__ void quants_to_bf6_kernel(const uint8_t *__restrict__ qs, const uint32_t *__restrict__ qh, int64_t rows, int64_t Mpad,
                                     uint8_t *__restrict__ q6a, uint8_t *__restrict__ q6b) {
    constexpr int NF = (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q4_1) ? 1 : 2;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (m >= rows) return;
    const int64_t pi = b * Mpad + m;
    int v[32];
    if (TYPE == GGML_TYPE_Q8_0) {      // plane h byte j = element 2j + h (signed)
        const uint4 e4 = *(const uint4 *)(qs + ((b * 2 + 0) * Mpad + m) * 16), o4 = *(const uint4 *)(qs + ((b * 2 + 1) * Mpad + m) * 16);
        const uint32_t ev[4] = {e4.x, e4.y, e4.z, e4.w}, od[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v[2 * j + 0] = (int)(int8_t)(ev[j >> 2] >> (8 * (j & 3)));
            v[2 * j + 1] = (int)(int8_t)(od[j >> 2] >> (8 * (j & 3)));
        }
    } else {
        const uint4 q = *(const uint4 *)(qs + pi * 16);
        const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
        const uint32_t hb = TYPE == GGML_TYPE_Q5_0 ? qh[pi] : 0u;
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int nib = (int)((qq[e >> 3] >> (4 * (e & 7))) & 0xFu);       // byte e/2, low nibble first (Ggml.cs:1149-1150)
            v[e] = TYPE == GGML_TYPE_Q5_0 ? (nib | (int)(((hb >> e) & 1u) << 4)) - 16 : nib - 8;   // Ggml.cs:1285-1289
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        uint32_t out[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 32; ++e) {
            const int hi = (v[e] + 8) >> 4, lo = v[e] - 16 * hi;
            const uint32_t c = bf6_code(NF == 1 ? v[e] : (f == 0 ? lo : hi));
            const int bit = 6 * e, wdx = bit >> 5, sh = bit & 31;
            out[wdx] |= c << sh;
            if (sh > 26) out[wdx + 1] |= c >> (32 - sh);
        }
        const int64_t po = (b * NF + f) * Mpad + m;
        *(uint4 *)(q6a + po * 16) = make_uint4(out[0], out[1], out[2], out[3]);
        *(uint2 *)(q6b + po * 8) = make_uint2(out[4], out[5]);
    }
}

// Q5_0 / Q5_1 / Q4_1: nibble plane (+ fifth-bit plane) -> int8 operand planes [nbk][2][Mpad][16 B] (the layout of Q8_0's qs: plane h
// byte j = element 2j + h), value (nib | bit << 4) - OFF: OFF = 16 as in ggml_vec_dot_q5_0_q8_0 (Ggml.cs:1285-1289), OFF = 0 for Q5_1
// and Q4_1, whose values stay unsigned (Ggml.cs:1330-1334, 1190-1191) and carry a min term instead.  One thread per (row, k-block).
template <int OFF, bool QH>
__global__ void q5_to_i8_kernel(const uint8_t *__restrict__ qs, const uint32_t *__restrict__ qh, int64_t rows, int64_t Mpad, uint8_t *__restrict__ i8p) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (m >= rows) return;
    const int64_t pi = b * Mpad + m;
    const uint4 q = *(const uint4 *)(qs + pi * 16);
    const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
    const uint32_t hb = QH ? qh[pi] : 0u;
    uint32_t ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 16; ++j) {                          // byte j of the block: element 2j = low nibble, 2j + 1 = high nibble
        const uint32_t byte = (qq[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const int x0 = (int)((byte & 0xFu) | (((hb >> (2 * j)) & 1u) << 4)) - OFF;
        const int x1 = (int)((byte >> 4) | (((hb >> (2 * j + 1)) & 1u) << 4)) - OFF;
        ev[j >> 2] |= (uint32_t)(uint8_t)(int8_t)x0 << (8 * (j & 3));
        od[j >> 2] |= (uint32_t)(uint8_t)(int8_t)x1 << (8 * (j & 3));
    }
    *(uint4 *)(i8p + ((b * 2 + 0) * Mpad + m) * 16) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
    *(uint4 *)(i8p + ((b * 2 + 1) * Mpad + m) * 16) = make_uint4(od[0], od[1], od[2], od[3]);
}

// Q5_1 / Q4_1 (and Q5_K in the Q5_1 form): the min plane [nbk][Mpad] f32 -> three bf16 piece planes per k-group of 8 blocks,
// [ceil(nbk / 8) * 3][Mpad][16 B] (ggml_hip_weight::mp3; split3: the pieces sum to the f32 value exactly).  One thread per (row, k-group).
__global__ void min_pieces_kernel(const float *__restrict__ mn, int64_t rows, int64_t Mpad, int64_t nbk, uint8_t *__restrict__ mp3) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = blockIdx.y;
    if (m >= rows) return;
    uint32_t w[3][4] = {};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t b = g * 8 + e;
        uint32_t q0 = 0, q1 = 0, q2 = 0;
        if (b < nbk) split3(mn[b * Mpad + m], q0, q1, q2);
        w[0][e >> 1] |= q0 << (16 * (e & 1)); w[1][e >> 1] |= q1 << (16 * (e & 1)); w[2][e >> 1] |= q2 << (16 * (e & 1));
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *(uint4 *)(mp3 + ((g * 3 + pc) * Mpad + m) * 16) = make_uint4(w[pc][0], w[pc][1], w[pc][2], w[pc][3]);
}

// d / m / qh planes -> the mat-vec's tile-major side image (common.h ggml_hip_weight::gs); one thread per (row, k-block)
__global__ void gemv_side_image_kernel(const float *__restrict__ d, const float *__restrict__ mm, const uint32_t *__restrict__ qh, int64_t Mpad,
                                       int64_t nbk, int np, uint32_t *__restrict__ gs) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mpad) return;
    for (int64_t b = blockIdx.y; b < nbk; b += gridDim.y) {
        uint32_t *o = gs + (((m >> 4) * nbk + b) * np) * 16 + (m & 15);
        int pl = 0;
        o[16 * pl++] = __float_as_uint(d[b * Mpad + m]);
        if (mm) o[16 * pl++] = __float_as_uint(mm[b * Mpad + m]);
        if (qh) o[16 * pl++] = qh[b * Mpad + m];
    }
}

// up to 32 device buffers -> (device-mapped) host tensors in one launch: blockIdx.y = copy, 16-byte pieces where both ends allow
struct scatter_table { const void *src[32]; void *dst[32]; unsigned long long bytes[32]; };
__global__ void scatter_copy_kernel(const scatter_table t) {
    const int e = blockIdx.y;
    const uint8_t *s = (const uint8_t *)t.src[e];
    uint8_t *d = (uint8_t *)t.dst[e];
    const size_t n = (size_t)t.bytes[e];
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const size_t n16 = n / 16;
        for (size_t i = i0; i < n16; i += stride) ((uint4 *)d)[i] = ((const uint4 *)s)[i];
        for (size_t i = n16 * 4 + i0; i < n / 4; i += stride) ((uint32_t *)d)[i] = ((const uint32_t *)s)[i];
    } else {
        for (size_t i = i0; i < n / 4; i += stride) ((uint32_t *)d)[i] = ((const uint32_t *)s)[i];
    }
}

}  // namespace

hipError_t launch_scatter_copy(const void *const *src, void *const *dst, const size_t *bytes, int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 32) return hipErrorInvalidValue;
    scatter_table t;
    size_t mx = 0;
    for (int i = 0; i < 32; ++i) {
        t.src[i] = i < n ? src[i] : nullptr; t.dst[i] = i < n ? dst[i] : nullptr; t.bytes[i] = i < n ? bytes[i] : 0;
        if (i < n && bytes[i] > mx) mx = bytes[i];
    }
    size_t bx = (mx / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;                               // a copy of any size is a grid-stride loop; PCIe, not CUs, bounds it
    scatter_copy_kernel<<<dim3((unsigned)bx, (unsigned)n), 256, 0, st>>>(t);
    return hipGetLastError();
}

hipError_t launch_gemv_side_image(ggml_hip_weight *w, hipStream_t st) {
    if (!w->gs || w->nbk <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->Mpad + 255) / 256), (unsigned)(w->nbk < 65535 ? w->nbk : 65535));
    gemv_side_image_kernel<<<grid, 256, 0, st>>>(w->d, w->m, w->qh, w->Mpad, w->nbk, gemv_side_planes(w->type), w->gs);
    return hipGetLastError();
}

hipError_t launch_q5_to_i8(ggml_hip_weight *w, hipStream_t st) {
    if (!w->i8p || w->M <= 0 || w->nbk <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 255) / 256), (unsigned)w->nbk);
    if (w->type == GGML_TYPE_Q5_0) q5_to_i8_kernel<16, true><<<grid, 256, 0, st>>>(w->qs, w->qh, w->M, w->Mpad, w->i8p);
    else if (w->type == GGML_TYPE_Q5_1) q5_to_i8_kernel<0, true><<<grid, 256, 0, st>>>(w->qs, w->qh, w->M, w->Mpad, w->i8p);
    else if (w->type == GGML_TYPE_Q4_2) q5_to_i8_kernel<8, false><<<grid, 256, 0, st>>>(w->qs, nullptr, w->M, w->Mpad, w->i8p);   // nib - 8 (Ggml.cs:1231-1235)
    else q5_to_i8_kernel<0, false><<<grid, 256, 0, st>>>(w->qs, nullptr, w->M, w->Mpad, w->i8p);
    return hipGetLastError();
}

hipError_t launch_min_pieces(ggml_hip_weight *w, hipStream_t st) {
    if (!w->mp3 || !w->m || w->M <= 0 || w->nbk <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 255) / 256), (unsigned)((w->nbk + 7) / 8));
    min_pieces_kernel<<<grid, 256, 0, st>>>(w->m, w->M, w->Mpad, w->nbk, w->mp3);
    return hipGetLastError();
}

hipError_t launch_nibbles_to_bf6(ggml_hip_weight *w, hipStream_t st) {
    if (!w->q6a || w->M <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 255) / 256), (unsigned)w->nbk);
    switch (w->type) {
    case GGML_TYPE_Q4_0: quants_to_bf6_kernel<GGML_TYPE_Q4_0><<<grid, 256, 0, st>>>(w->qs, w->qh, w->M, w->Mpad, w->q6a, w->q6b); break;
    case GGML_TYPE_Q4_1: quants_to_bf6_kernel<GGML_TYPE_Q4_1><<<grid, 256, 0, st>>>(w->qs, w->qh, w->M, w->Mpad, w->q6a, w->q6b); break;
    case GGML_TYPE_Q5_0: quants_to_bf6_kernel<GGML_TYPE_Q5_0><<<grid, 256, 0, st>>>(w->qs, w->qh, w->M, w->Mpad, w->q6a, w->q6b); break;
    case GGML_TYPE_Q8_0: quants_to_bf6_kernel<GGML_TYPE_Q8_0><<<grid, 256, 0, st>>>(w->qs, w->qh, w->M, w->Mpad, w->q6a, w->q6b); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_repack_to_planar(int type, const uint8_t *aos, uint64_t nb01, int64_t row_begin, int64_t rows,
                                   ggml_hip_weight *w, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    if (type == GGML_TYPE_F32 || type == GGML_TYPE_F16) {
        const int64_t row_bytes = w->K * (type == GGML_TYPE_F32 ? 4 : 2);
        dim3 grid((unsigned)(((row_bytes + 3) / 4 + 255) / 256), (unsigned)(rows < 65535 ? rows : 65535));
        copy_rows_kernel<<<grid, 256, 0, st>>>(aos + (uint64_t)row_begin * nb01, nb01, (uint8_t *)w->dense,
                                               (uint64_t)row_bytes, rows, row_bytes);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((rows + 255) / 256), (unsigned)w->nbk);
    switch (type) {
    case GGML_TYPE_Q4_0: repack_to_planar_kernel<GGML_TYPE_Q4_0><<<grid, 256, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q4_1: repack_to_planar_kernel<GGML_TYPE_Q4_1><<<grid, 256, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q5_0: repack_to_planar_kernel<GGML_TYPE_Q5_0><<<grid, 256, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q5_1: repack_to_planar_kernel<GGML_TYPE_Q5_1><<<grid, 256, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q4_2: repack_to_planar_kernel<GGML_TYPE_Q4_2><<<grid, 256, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q8_0: repack_to_planar_kernel<GGML_TYPE_Q8_0><<<grid, 256, 0, st>>>(aos, nb01, row_begin, rows, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_planar_to_aos(const ggml_hip_weight *w, uint8_t *aos, hipStream_t st) {
    if (w->M <= 0) return hipSuccess;
    if (w->type == GGML_TYPE_F32 || w->type == GGML_TYPE_F16) {
        const int64_t row_bytes = w->K * (w->type == GGML_TYPE_F32 ? 4 : 2);
        dim3 grid((unsigned)(((row_bytes + 3) / 4 + 255) / 256), (unsigned)(w->M < 65535 ? w->M : 65535));
        copy_rows_kernel<<<grid, 256, 0, st>>>((const uint8_t *)w->dense, (uint64_t)row_bytes, aos, (uint64_t)row_bytes,
                                               w->M, row_bytes);
        return hipGetLastError();
    }
    // bytes of one row of blocks; Q4_2's blocks hold 16 elements, so a 32-element k-block is two of them
    const uint64_t nb01 = (uint64_t)ggml_hip_type_size(w->type) * (uint64_t)w->nbk * (w->type == GGML_TYPE_Q4_2 ? 2 : 1);
    dim3 grid((unsigned)((w->M + 255) / 256), (unsigned)w->nbk);
    switch (w->type) {
    case GGML_TYPE_Q4_0: planar_to_aos_kernel<GGML_TYPE_Q4_0><<<grid, 256, 0, st>>>(aos, nb01, w->M, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q4_1: planar_to_aos_kernel<GGML_TYPE_Q4_1><<<grid, 256, 0, st>>>(aos, nb01, w->M, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q5_0: planar_to_aos_kernel<GGML_TYPE_Q5_0><<<grid, 256, 0, st>>>(aos, nb01, w->M, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q5_1: planar_to_aos_kernel<GGML_TYPE_Q5_1><<<grid, 256, 0, st>>>(aos, nb01, w->M, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q4_2: planar_to_aos_kernel<GGML_TYPE_Q4_2><<<grid, 256, 0, st>>>(aos, nb01, w->M, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    case GGML_TYPE_Q8_0: planar_to_aos_kernel<GGML_TYPE_Q8_0><<<grid, 256, 0, st>>>(aos, nb01, w->M, w->Mpad, w->nbk, w->qs, w->qh, w->d, w->m); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_relayout_gathered(const float *g, int G, int64_t N, int64_t Ms, float *dst, int64_t M, int64_t ldd,
                                    hipStream_t st) {
    if (N <= 0 || M <= 0) return hipSuccess;
    dim3 grid((unsigned)((M + 255) / 256), (unsigned)(N < 65535 ? N : 65535));
    relayout_gathered_kernel<<<grid, 256, 0, st>>>(g, G, N, Ms, dst, M, ldd);
    return hipGetLastError();
}

hipError_t launch_push_columns(const float *src, int64_t lds, int64_t N, int64_t Ms, float *const *peers, int G, int64_t ldd,
                               int64_t col0, hipStream_t st) {
    if (N <= 0 || Ms <= 0 || G <= 0) return hipSuccess;
    if (G > 16) return hipErrorInvalidValue;
    PeerPtrs pp;
    bool vec = Ms % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && col0 % 4 == 0 && ((uintptr_t)src & 15) == 0;
    for (int g = 0; g < 16; ++g) {
        pp.p[g] = g < G ? peers[g] : nullptr;
        if (pp.p[g] && ((uintptr_t)pp.p[g] & 15) != 0) vec = false;
    }
    const int64_t work = vec ? N * (Ms / 4) : N * Ms;
    int64_t bx = (work + 255) / 256;
    if (bx > 2048) bx = 2048;                        // grid-stride: ~8 workgroups per CU and peer keep the links busy
    dim3 grid((unsigned)bx, (unsigned)G);
    if (vec) push_columns_kernel<true><<<grid, 256, 0, st>>>(src, lds, N, Ms, pp, ldd, col0);
    else push_columns_kernel<false><<<grid, 256, 0, st>>>(src, lds, N, Ms, pp, ldd, col0);
    return hipGetLastError();
}
