// quantize.hip -- block quantize / dequantize kernels (bandwidth-bound byte work, no MFMA).
//
// K1  quantize_act: the INIT phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6641-6654): every src1 row
//     -> Q8_0 (quantize_row_q8_0_reference_impl, Ggml.cs:733-762, all 32 quants written = SURVEY D2 intent;
//     Math.Round = round-half-even = v_rndne_f32), written straight into the planar scratch the
//     dot kernels read (int8 even/odd planes + f32 scale plane + i32 block-sum plane).
// K9  quantize_rows  : f32 rows -> reference AoS blocks, bit-exact (Q4_0 334-377, Q4_1 487-528, Q5_0 609-653,
//                      Q8_0 733-762, Q8_1 781-823).
// K8  dequantize_rows: reference AoS blocks -> f32 rows, bit-exact (Q4_0 886-910, Q4_1 962-987, Q5_0 1025-1061,
//                      Q8_0 1104-1122 with signed quants).
//
// All float arithmetic here must round exactly like the C# scalar code: one IEEE operation per C# operator,
// no fused multiply-add, correctly rounded division.  Built with -ffp-contract=off.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float half_bits_to_float_q(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    const uint32_t man = h & 0x3FFu;
    if (exp == 0) {
        if (man == 0) return __uint_as_float(sign);
        const float v = (float)man * 5.9604644775390625e-08f;
        return __uint_as_float(__float_as_uint(v) | sign);
    }
    if (exp == 31) return __uint_as_float(sign | 0x7F800000u | (man << 13));
    return __uint_as_float(sign | ((exp + 112u) << 23) | (man << 13));
}

// (Half)f: IEEE round-to-nearest-even, same bit algorithm as the oracle (checked against numpy there)
__device__ __forceinline__ uint16_t float_to_half_bits_rne(float f) {
    const uint32_t x = __float_as_uint(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t exp = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (exp == 0xFF) return (uint16_t)(man == 0 ? (sign | 0x7C00u) : (sign | 0x7C00u | 0x0200u | (man >> 13)));
    const int e = (int)exp - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - e;
        uint32_t hm = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t half = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++;
    return (uint16_t)(sign | half);
}

// bf6 (e3m2) code of an integer |v| <= 8 (see layout.hip)
__device__ __forceinline__ uint32_t bf6_code_q(int v) {
    const uint64_t tab = 0ull | (12ull << 5) | (16ull << 10) | (18ull << 15) | (20ull << 20) | (21ull << 25) | (22ull << 30) |
                         (23ull << 35) | (24ull << 40);
    const int a = v < 0 ? -v : v;
    return (uint32_t)((tab >> (5 * a)) & 31u) | (v < 0 ? 32u : 0u);
}

// ---------------------------------------------------------------------------------------------------------
// K1.  Wave layout: lane = 8 * row_in_wave + t ; the 8 lanes of one row own one 32-element block, 4 floats each
// (one coalesced 128-byte line per row per block).  A workgroup = 4 waves = 32 rows x BPB consecutive k-blocks.
// ---------------------------------------------------------------------------------------------------------
#define K1_BPB 8

// IMG: 0 = int8 even/odd planes; 1 = f16 image in nibble order (Q4_0/Q4_1/Q5_0 weights); 2 = f16 image in byte-plane
// order (Q8_0 weights) -- the k-slot orders of gemm_q16.hip; 3 = bf6 digit image of gemm_qmx.hip.  The f16 images carry `as` as the float d * sum(q)
// (the Q8_1 s0 + s1 of Ggml.cs:820-821, intent D3) instead of the integer sum.
// K8 = true: the Q8_K rule of the published k-quants (kquants.hip; no counterpart in the reference): ONE scale per 256
// elements -- a workgroup column of K1_BPB = 8 k-blocks is exactly one super-block of a row, held by the row's 8 lanes --
// iscale = -128 / (first element of largest magnitude), q = min(127, round-half-even(iscale * x)), d = 1 / iscale.
template <int IMG, bool K8 = false>
__global__ __launch_bounds__(256) void quantize_act_kernel(const float *__restrict__ x, int64_t N, int64_t nbk, int64_t ld1,
                                                          int8_t *__restrict__ a8, float *__restrict__ ad,
                                                          int32_t *__restrict__ as, int64_t Npad, uint8_t *__restrict__ sp3 = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 7;
    float ds_mine = 0.0f;                                   // (IMG 0 with sp3) d * sum of k-block b0 + t of this lane's row
    const int64_t n = (int64_t)blockIdx.y * 32 + wave * 8 + (lane >> 3);
    const bool live = n < N;
    const int64_t nr = live ? n : N - 1;  // clamp loads, skip stores
    const int64_t b0 = (int64_t)blockIdx.x * K1_BPB;
    const float *row = x + nr * ld1;

    // (IMG 3) per-thread store offsets inside a k-block of the bf6 image: dword idx = 3 (t >> 2) + (t & 3) of the fragment
    const int idx6 = 3 * (t >> 2) + (t & 3);
    const uint32_t off6 = idx6 < 4 ? (uint32_t)(n * 16 + 4 * idx6) : (uint32_t)(Npad * 32 + n * 8 + 4 * (idx6 - 4));
    const uint32_t half6 = idx6 < 4 ? (uint32_t)(Npad * 16) : (uint32_t)(Npad * 8);   // half 0 (ah) -> half 1 (al)
    const uint32_t offS = (uint32_t)(n * 4);
    const uint32_t img_bytes = (uint32_t)(pad_kblocks(nbk) * Npad * 48), sc_bytes = (uint32_t)(pad_kblocks(nbk) * Npad * 4);
    const __amdgpu_buffer_rsrc_t rImg = __builtin_amdgcn_make_buffer_rsrc((void *)a8, 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAd = __builtin_amdgcn_make_buffer_rsrc((void *)ad, 0, (int)sc_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAs = __builtin_amdgcn_make_buffer_rsrc((void *)as, 0, (int)sc_bytes, 0x00020000);

    float4 v[K1_BPB];
#pragma unroll
    for (int j = 0; j < K1_BPB; ++j) {
        const int64_t b = b0 + j;
        v[j] = (b < nbk) ? *(const float4 *)(row + b * QK + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float k8_isc = 0.0f, k8_d = 0.0f;
    if constexpr (K8) {
        static_assert(K1_BPB * QK == 256 && IMG != 3, "one super-block per workgroup column; no bf6 image for k-quants");
        float am = 0.0f, mx = 0.0f;
        int ix = 0;                                         // this lane's first element of largest magnitude (element order)
#pragma unroll
        for (int j = 0; j < K1_BPB; ++j) {
            const float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (fabsf(e[c]) > am) { am = fabsf(e[c]); mx = e[c]; ix = 32 * j + 4 * t + c; }
        }
        // over the row's 8 lanes: the larger magnitude wins, equal magnitudes the earlier element
        auto join = [&](float oa, float om, int oi) {
            const bool take = oa > am || (oa == am && oi < ix);
            am = take ? oa : am; mx = take ? om : mx; ix = take ? oi : ix;
        };
        join(dpp_f<DPP_XOR1>(am), dpp_f<DPP_XOR1>(mx), dpp_i<DPP_XOR1>(ix));
        join(dpp_f<DPP_XOR2>(am), dpp_f<DPP_XOR2>(mx), dpp_i<DPP_XOR2>(ix));
        join(dpp_f<DPP_HALF_MIRROR>(am), dpp_f<DPP_HALF_MIRROR>(mx), dpp_i<DPP_HALF_MIRROR>(ix));
        k8_isc = am != 0.0f ? -128.0f / mx : 0.0f;
        k8_d = am != 0.0f ? 1.0f / k8_isc : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < K1_BPB; ++j) {
        const int64_t b = b0 + j;
        float amax = fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
        amax = group8_max(amax);
        const float d = K8 ? k8_d : amax / 127.0f;                              // Ggml.cs:751
        const float id = K8 ? k8_isc : (d != 0.0f ? 1.0f / d : 0.0f);           // Ggml.cs:752
        if constexpr (K8) {                                 // min(127, nearest_int(iscale * x)): clamp before anything reads the values
            v[j].x = fminf(127.0f, rintf(v[j].x * id)); v[j].y = fminf(127.0f, rintf(v[j].y * id));
            v[j].z = fminf(127.0f, rintf(v[j].z * id)); v[j].w = fminf(127.0f, rintf(v[j].w * id));
        }
        const float qid = K8 ? 1.0f : id;                   // (K8: v already holds the quants; x * 1 and rintf of an integer are exact)
        const int q0 = (int)rintf(v[j].x * qid);            // Ggml.cs:758-759 (all l, D2)
        const int q1 = (int)rintf(v[j].y * qid);
        const int q2 = (int)rintf(v[j].z * qid);
        const int q3 = (int)rintf(v[j].w * qid);
        const int s = group8_sum(q0 + q1 + q2 + q3);
        if (IMG == 0 && t == j) ds_mine = d * (float)s;     // (a k-block past the end of K: v = 0, so d = 0 and s = 0)
        if (IMG == 3) {
            // bf6 image of gemm_qmx.hip: a = 16*ah + al, ah = floor((a + 8) / 16) in [-8, 8], al in [-8, 7]; lane t owns
            // elements 4t..4t+3 = bits [24t, 24t+24) of both 192-bit fragments; the 4 lanes of a group (u = t & 3) hold
            // 96 bits = dwords 3g..3g+2 (g = t >> 2), and lane u < 3 assembles dword 3g + u from its own value and
            // its right neighbour's.
            // digits in float arithmetic (exact: |q| <= 127), codes from the f32 bit pattern: for an integer 1 <= |v| <= 8
            // the bf6 code (exponent bias 3, 2 mantissa bits) is (f32 bits >> 21) - ((127 - 3) << 2); 0 clamps to code 0
            const float rv[4] = {rintf(v[j].x * qid), rintf(v[j].y * qid), rintf(v[j].z * qid), rintf(v[j].w * qid)};
            uint32_t vh = 0, vl = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ah = floorf(fmaf(rv[c], 0.0625f, 0.5f));       // floor((q + 8) / 16)
                const float al = fmaf(ah, -16.0f, rv[c]);                    // q - 16 ah, in [-8, 7]
                const uint32_t bh = __float_as_uint(ah), bl = __float_as_uint(al);
                const int ch = max((int)((bh & 0x7FFFFFFFu) >> 21) - 496, 0), cl = max((int)((bl & 0x7FFFFFFFu) >> 21) - 496, 0);
                vh |= ((uint32_t)ch | ((bh >> 26) & 32u)) << (6 * c);
                vl |= ((uint32_t)cl | ((bl >> 26) & 32u)) << (6 * c);
            }
            const uint32_t nh = (uint32_t)dpp_i<DPP_NEXT>((int)vh), nl = (uint32_t)dpp_i<DPP_NEXT>((int)vl);
            const int u = t & 3, g = t >> 2, idx = 3 * g + u;
            const uint32_t dh = (vh >> (8 * u)) | (u < 3 ? nh << (24 - 8 * u) : 0u);
            const uint32_t dl = (vl >> (8 * u)) | (u < 3 ? nl << (24 - 8 * u) : 0u);
            // stores: one buffer descriptor per plane, the k-block in the uniform offset, the row / dword in a per-thread
            // offset that is the same for every k-block (no 64-bit address arithmetic per store)
            const bool pad = !(b < nbk);
            if (b < pad_kblocks(nbk)) {
                const uint32_t soff = (uint32_t)b * (uint32_t)(Npad * 48);
                if (live && u < 3) {
                    __builtin_amdgcn_raw_buffer_store_b32(pad ? 0u : dh, rImg, (int)off6, (int)soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(pad ? 0u : dl, rImg, (int)(off6 + half6), (int)soff, 0);
                }
                if (live && t == 0) {
                    __builtin_amdgcn_raw_buffer_store_b32(pad ? 0u : __float_as_uint(d), rAd, (int)offS, (int)((uint32_t)b * (uint32_t)(Npad * 4)), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(pad ? 0u : __float_as_uint(d * (float)s), rAs, (int)offS, (int)((uint32_t)b * (uint32_t)(Npad * 4)), 0);
                }
            }
            continue;
        }
        if (IMG != 0) {
            const float r0 = rintf(v[j].x * qid), r1 = rintf(v[j].y * qid), r2 = rintf(v[j].z * qid), r3 = rintf(v[j].w * qid);
            const bool odd = (t & 1) != 0;
            uint32_t o0, o1;
            int panel, half8;
            if (IMG == 1) {
                // lane t owns elements 4t..4t+3 = e_i, i = 4(t&1) + c, of (h = t>>2, kk = (t>>1)&1); k-slots are
                // [e0, e4, e1/16, e5/16, e2, e6, e3/16, e7/16]: dword c = [even lane's c-th value, odd lane's c-th value]
                const uint32_t h0 = __builtin_bit_cast(uint16_t, (_Float16)r0), h1 = __builtin_bit_cast(uint16_t, (_Float16)(r1 * 0.0625f));
                const uint32_t h2 = __builtin_bit_cast(uint16_t, (_Float16)r2), h3 = __builtin_bit_cast(uint16_t, (_Float16)(r3 * 0.0625f));
                const uint32_t send = odd ? (h0 | (h1 << 16)) : (h2 | (h3 << 16));
                const uint32_t recv = (uint32_t)dpp_i<DPP_XOR1>((int)send);
                // even lane writes dwords 0, 1; odd lane dwords 2, 3
                o0 = odd ? ((recv & 0xFFFFu) | (h2 << 16)) : (h0 | (recv << 16));
                o1 = odd ? ((recv >> 16) | (h3 << 16)) : (h1 | (recv & 0xFFFF0000u));
                panel = 2 * ((t >> 1) & 1) + (t >> 2);
                half8 = t & 1;
            } else {
                // plane h holds elements 2jj + h; lane t owns jj = 2(t&3), 2(t&3)+1 of kk = t>>2: x, z -> plane 0, y, w ->
                // plane 1; k-slots [jj0, jj2, jj1, jj3, jj4, jj6, jj5, jj7]: dword pair u = (t&3)>>1 of the panel row is
                // [even.x, odd.x], [even.z, odd.z] (plane 0) or the same with y, w (plane 1)
                const uint32_t h0 = __builtin_bit_cast(uint16_t, (_Float16)r0), h1 = __builtin_bit_cast(uint16_t, (_Float16)r1);
                const uint32_t h2 = __builtin_bit_cast(uint16_t, (_Float16)r2), h3 = __builtin_bit_cast(uint16_t, (_Float16)r3);
                const uint32_t send = odd ? (h0 | (h2 << 16)) : (h1 | (h3 << 16));
                const uint32_t recv = (uint32_t)dpp_i<DPP_XOR1>((int)send);
                // even lane writes plane 0 (x, z of both lanes), odd lane plane 1 (y, w of both lanes)
                o0 = odd ? ((recv & 0xFFFFu) | (h1 << 16)) : (h0 | (recv << 16));
                o1 = odd ? ((recv >> 16) | (h3 << 16)) : (h2 | (recv & 0xFFFF0000u));
                panel = 2 * (t >> 2) + (t & 1);
                half8 = (t & 3) >> 1;
            }
            if (live && b < nbk) {
                *(uint2 *)(a8 + ((b * 4 + panel) * Npad + n) * 16 + 8 * half8) = make_uint2(o0, o1);
                if (t == 0) {
                    ad[b * Npad + n] = d;
                    as[b * Npad + n] = (int32_t)__float_as_uint(d * (float)s);
                }
            } else if (live && b < pad_kblocks(nbk)) {     // zero pad blocks up to a whole stage (finite * 0 = 0)
                *(uint2 *)(a8 + ((b * 4 + panel) * Npad + n) * 16 + 8 * half8) = make_uint2(0u, 0u);
                if (t == 0) {
                    ad[b * Npad + n] = 0.0f;
                    as[b * Npad + n] = 0;
                }
            }
            continue;
        }
        // int8 image: elements 4t, 4t+2 are even (plane 0 bytes 2t, 2t+1); 4t+1, 4t+3 odd (plane 1 bytes 2t, 2t+1)
        const uint32_t e16 = ((uint32_t)q0 & 0xFFu) | (((uint32_t)q2 & 0xFFu) << 8);
        const uint32_t o16 = ((uint32_t)q1 & 0xFFu) | (((uint32_t)q3 & 0xFFu) << 8);
        const bool even_lane = (t & 1) == 0;
        const uint32_t recv = (uint32_t)dpp_i<DPP_XOR1>((int)(even_lane ? o16 : e16));
        // even lane t: plane 0 bytes [2t, 2t+4) = own e16 | partner e16 << 16
        // odd  lane t: plane 1 bytes [2t-2, 2t+2) = partner o16 | own o16 << 16
        const uint32_t word = even_lane ? (e16 | (recv << 16)) : (recv | (o16 << 16));
        if (live && b < nbk) {
            const int h = even_lane ? 0 : 1;
            const int byte_off = even_lane ? 2 * t : 2 * t - 2;
            *(uint32_t *)(a8 + ((b * 2 + h) * Npad + n) * 16 + byte_off) = word;
            if (t == 0) {
                ad[b * Npad + n] = d;
                as[b * Npad + n] = s;
            }
        }
    }
    // r4: the min-term operand of K3p-int8 (gemm_qmp.hip) -- d * (float)sum, the Q8_1 s0 + s1 of Ggml.cs:820-821, as three bf16 pieces that
    // sum to it exactly; the workgroup column is one k-group of 8 blocks, the row's 8 lanes write its 16 bytes of each piece plane
    if (IMG == 0 && sp3 != nullptr && live) {
        uint32_t q0, q1, q2;
        split3(ds_mine, q0, q1, q2);
        uint8_t *o = sp3 + (((int64_t)blockIdx.x * 3) * Npad + n) * 16 + 2 * t;
        *(uint16_t *)o = (uint16_t)q0; *(uint16_t *)(o + Npad * 16) = (uint16_t)q1; *(uint16_t *)(o + 2 * Npad * 16) = (uint16_t)q2;
        // (the product takes k-groups in pairs: behind an odd count stands a k-group of zeros)
        if ((gridDim.x & 1) && blockIdx.x == gridDim.x - 1) {
            o += 3 * Npad * 16;
            *(uint16_t *)o = 0; *(uint16_t *)(o + Npad * 16) = 0; *(uint16_t *)(o + 2 * Npad * 16) = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K1, bf6 image (gemm_qmx.hip), lane-per-block form.  One wave = 64 src1 rows x one k-block:
//   global -> LDS by DMA, eight 1-KiB pieces (8 rows x 128 B each, coalesced 128-byte lines), laid out so that the
//   transposed read -- lane = row, eight ds_read_b128 = the row's 32 floats -- is conflict-free (piece stride 1152 B);
//   one division pair per BLOCK instead of per 8 lanes, 5 VALU per element for the two digits, and the 64 bf6 codes of a
//   lane from two v_cvt_scalef32_2xpk16_bf6_f32 (it packs a[i] at position 2i, b[i] at 2i+1: tools/cvt_probe.hip);
//   stores are lane-contiguous (1 KiB / 512 B runs per wave).  Same arithmetic as quantize_act_kernel<3>, bit for bit
//   (Ggml.cs:751-759), ~1/4 of its VALU work.  Waves are independent (each reads only what it loaded): no barrier.
// ---------------------------------------------------------------------------------------------------------
#define K1V_PIECE 1152      // 8 rows x 128 B + 128 B: consecutive row groups land in opposite halves of the 64 LDS banks

using f32x16q = __attribute__((ext_vector_type(16))) float;
using u32x6q = __attribute__((ext_vector_type(6))) uint32_t;
using u32x4q = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2q = __attribute__((ext_vector_type(2))) uint32_t;
typedef __attribute__((address_space(3))) void lds_void_q;

__global__ __launch_bounds__(256) void quantize_act_bf6_kernel(const float *__restrict__ x, int N, int nbk, uint32_t ld1_bytes,
                                                               uint32_t x_bytes, int8_t *__restrict__ a8, float *__restrict__ ad,
                                                               int32_t *__restrict__ as, int Npad) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * 8 * K1V_PIECE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * 4 + wave;                     // this wave's k-block (< pad_kblocks(nbk) by the grid)
    const int n0 = blockIdx.y * 64;
    uint8_t *wl = lds + wave * (8 * K1V_PIECE);

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    // piece g = rows n0 + 8g .. +7: lane L fetches 16 B chunk L / 8 of row 8g + L % 8 -> LDS [chunk][row][16 B].
    // A ragged last row group re-reads row N - 1 for the rows past N (their lanes store nothing); pad k-blocks load nothing.
    if (b < nbk) {
        if (n0 + 64 <= N) {                                  // uniform
            const uint32_t voff = (uint32_t)(n0 + (lane & 7)) * ld1_bytes + (uint32_t)(lane >> 3) * 16u;
#pragma unroll
            for (int g = 0; g < 8; ++g)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_q *)(wl + g * K1V_PIECE), 16, (int)voff,
                                                         (int)((uint32_t)(8 * g) * ld1_bytes + (uint32_t)b * 128u), 0, 0);
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int row = min(n0 + 8 * g + (lane & 7), N - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void_q *)(wl + g * K1V_PIECE), 16,
                                                         (int)((uint32_t)row * ld1_bytes + (uint32_t)(lane >> 3) * 16u), (int)((uint32_t)b * 128u), 0, 0);
            }
        }
    }
    // The DMA's LDS write is only complete for a reader after vmcnt(0) AND lgkmcnt(0) (with vmcnt(0) alone, now and then the
    // upper 64 bytes of every 128-byte line of a piece were still the previous contents -- found with tools/stress_k1b.py).
    // The barrier is the customary third part of this wait; it costs nothing here.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float v[32];
    {
        const uint8_t *rp = wl + (lane >> 3) * K1V_PIECE + (lane & 7) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t = *(const float4 *)(rp + c * 128);
            v[4 * c + 0] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
        }
    }
    const bool pad = !(b < nbk);
    float amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 32; ++e) amax = fmaxf(amax, fabsf(v[e]));
    const float d = amax / 127.0f;                          // Ggml.cs:751
    const float id = d != 0.0f ? 1.0f / d : 0.0f;           // Ggml.cs:752
    float sum = 0.0f;                                       // exact: |sum| <= 32 * 127
    f32x16q he, ho, le, lo;                                  // digits of the even / odd elements
#pragma unroll
    for (int e = 0; e < 32; ++e) {
        const float q = rintf(v[e] * id);                   // Ggml.cs:758-759 (D1, D2)
        sum += q;
        const float ah = floorf(fmaf(q, 0.0625f, 0.5f));    // a = 16 ah + al, ah = floor((a + 8) / 16) in [-8, 8]
        const float al = fmaf(ah, -16.0f, q);               // in [-8, 7]
        if (e & 1) { ho[e >> 1] = ah; lo[e >> 1] = al; } else { he[e >> 1] = ah; le[e >> 1] = al; }
    }
    // Through asm with an early-clobber result: hipcc's builtin lets the 6-register result overlap the 32 source registers,
    // and the instruction writes its result before it has read all of them (seen as garbage digits at elements 12..16).
    // Both converts in ONE statement, so that no result shares a register with ANY of the 64 sources.  As two statements
    // hipcc gave the second result the first convert's source registers and, in 22 of 100 launches, some rows came out with
    // wrong quants 0..4 (tools/dbg_k1b.py; 8 wait states between the two did not help, distinct registers did: 0 of 250).
    // The instruction evidently reads its sources for many cycles after issue with no interlock; the trailing nops keep
    // whatever the compiler puts into the source registers next at a distance as well.
    u32x6q fh, fl;
    asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %2, %3, 1.0\n\ts_nop 7\n\tv_cvt_scalef32_2xpk16_bf6_f32 %1, %4, %5, 1.0\n\t"
                 "s_nop 15\n\ts_nop 15"
                 : "=&v"(fh), "=&v"(fl) : "v"(he), "v"(ho), "v"(le), "v"(lo));
    float dv = d, sv = d * sum;                             // the Q8_1 s0 + s1 of Ggml.cs:820-821 (intent D3)
    if (pad) {                                              // uniform: pad k-blocks are zero quants with zero scales
        fh = (u32x6q){0, 0, 0, 0, 0, 0}; fl = fh; dv = 0.0f; sv = 0.0f;
    }
    const int n = n0 + lane;
    if (n >= N) return;
    const uint32_t img_bytes = (uint32_t)pad_kblocks(nbk) * (uint32_t)Npad * 48u, sc_bytes = (uint32_t)pad_kblocks(nbk) * (uint32_t)Npad * 4u;
    const __amdgpu_buffer_rsrc_t rImg = __builtin_amdgcn_make_buffer_rsrc((void *)a8, 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAd = __builtin_amdgcn_make_buffer_rsrc((void *)ad, 0, (int)sc_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rAs = __builtin_amdgcn_make_buffer_rsrc((void *)as, 0, (int)sc_bytes, 0x00020000);
    const uint32_t blk = (uint32_t)b * (uint32_t)Npad * 48u;
    // image per k-block: [half][Npad][16 B] then [half][Npad][8 B] (half 0 = ah, half 1 = al)
    __builtin_amdgcn_raw_buffer_store_b128((u32x4q){fh[0], fh[1], fh[2], fh[3]}, rImg, n * 16, (int)blk, 0);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4q){fl[0], fl[1], fl[2], fl[3]}, rImg, n * 16, (int)(blk + (uint32_t)Npad * 16u), 0);
    __builtin_amdgcn_raw_buffer_store_b64((u32x2q){fh[4], fh[5]}, rImg, n * 8, (int)(blk + (uint32_t)Npad * 32u), 0);
    __builtin_amdgcn_raw_buffer_store_b64((u32x2q){fl[4], fl[5]}, rImg, n * 8, (int)(blk + (uint32_t)Npad * 40u), 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rAd, n * 4, (int)((uint32_t)b * (uint32_t)Npad * 4u), 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv), rAs, n * 4, (int)((uint32_t)b * (uint32_t)Npad * 4u), 0);
}

// ---------------------------------------------------------------------------------------------------------
// K9 / K8: one thread per block, rows contiguous.  These serve the Seam-2 row functions and ggml_cpy-style
// callers; they are checked bit for bit against the oracle.
// ---------------------------------------------------------------------------------------------------------
// ---- one block: 32 floats <-> one reference-format block (bit-exact statements of the reference's row functions) ----
template <int TYPE>
__device__ __forceinline__ void quant_block(const float (&v)[QK], uint8_t *__restrict__ out) {
    if (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q5_0) {
        float amax = 0.0f, mx = 0.0f;                       // Ggml.cs:343-354 / 616-627
#pragma unroll
        for (int l = 0; l < QK; ++l) {
            const float a = fabsf(v[l]);
            if (amax < a) { amax = a; mx = v[l]; }
        }
        if (TYPE == GGML_TYPE_Q4_0) {
            const float d = mx / -8.0f;                     // Ggml.cs:356
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            uint8_t *o = out;
            *(uint32_t *)o = __float_as_uint(d);
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < QK; l += 2) {
                // (byte)Math.Min(15, Math.Round(v) + 8), Ggml.cs:366-367; rintf = half-to-even (D1)
                const int q0 = (int)fminf(15.0f, rintf(v[l + 0] * id) + 8.0f);
                const int q1 = (int)fminf(15.0f, rintf(v[l + 1] * id) + 8.0f);
                w[l / 8] |= (uint32_t)((q0 & 0xFF) | ((q1 << 4) & 0xFF)) << (8 * ((l / 2) & 3));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) *(uint32_t *)(o + 4 + 4 * k) = w[k];
        } else {
            const float d = mx / -16.0f;                    // Ggml.cs:629
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            uint16_t *o = (uint16_t *)(out);
            o[0] = float_to_half_bits_rne(d);               // (Half)d, Ggml.cs:632
            uint32_t qh = 0;
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int l = 0; l < QK; l += 2) {
                int t0 = (int)(v[l + 0] * id + 16.5f); t0 = t0 > 31 ? 31 : t0;   // Ggml.cs:641-642
                int t1 = (int)(v[l + 1] * id + 16.5f); t1 = t1 > 31 ? 31 : t1;
                w[l / 8] |= (uint32_t)((t0 & 0x0F) | ((t1 & 0x0F) << 4)) << (8 * ((l / 2) & 3));
                qh |= (uint32_t)((t0 & 0x10) >> 4) << (l + 0);
                qh |= (uint32_t)((t1 & 0x10) >> 4) << (l + 1);
            }
            o[1] = (uint16_t)qh; o[2] = (uint16_t)(qh >> 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[3 + 2 * k] = (uint16_t)w[k]; o[4 + 2 * k] = (uint16_t)(w[k] >> 16); }
        }
    } else if (TYPE == GGML_TYPE_Q4_1) {
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;  // Ggml.cs:496-504
#pragma unroll
        for (int l = 0; l < QK; ++l) {
            if (v[l] < mn) mn = v[l];
            if (v[l] > mx) mx = v[l];
        }
        const float d = (mx - mn) / 15.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        uint8_t *o = out;
        *(uint32_t *)o = __float_as_uint(d);
        *(uint32_t *)(o + 4) = __float_as_uint(mn);
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int l = 0; l < QK; l += 2) {
            const int q0 = (int)rintf((v[l + 0] - mn) * id);  // Ggml.cs:514-518
            const int q1 = (int)rintf((v[l + 1] - mn) * id);
            w[l / 8] |= (uint32_t)((q0 & 0xFF) | ((q1 << 4) & 0xFF)) << (8 * ((l / 2) & 3));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint32_t *)(o + 8 + 4 * k) = w[k];
    } else if (TYPE == GGML_TYPE_Q4_2) {
        // two 16-element blocks {Half d; qs[8]} (Ggml.cs:547-590); the half is an IEEE bit pattern (intent)
        uint16_t *o = (uint16_t *)(out);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            float amax = 0.0f, mx = 0.0f;                   // Ggml.cs:556-568
#pragma unroll
            for (int l = 0; l < 16; ++l) {
                const float a = fabsf(v[16 * hb + l]);
                if (amax < a) { amax = a; mx = v[16 * hb + l]; }
            }
            const float d = mx / -8.0f;                     // Ggml.cs:570-571
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            o[5 * hb] = float_to_half_bits_rne(d);          // (Half)d, Ggml.cs:573
            uint32_t w[2] = {0, 0};
#pragma unroll
            for (int l = 0; l < 16; l += 2) {               // Ggml.cs:575-586 (D1 half-even)
                const int q0 = (int)fminf(15.0f, rintf(v[16 * hb + l + 0] * id) + 8.0f);
                const int q1 = (int)fminf(15.0f, rintf(v[16 * hb + l + 1] * id) + 8.0f);
                w[l / 8] |= (uint32_t)((q0 & 0xFF) | ((q1 << 4) & 0xFF)) << (8 * ((l / 2) & 3));
            }
            o[5 * hb + 1] = (uint16_t)w[0]; o[5 * hb + 2] = (uint16_t)(w[0] >> 16);
            o[5 * hb + 3] = (uint16_t)w[1]; o[5 * hb + 4] = (uint16_t)(w[1] >> 16);
        }
    } else if (TYPE == GGML_TYPE_Q5_1) {
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;  // Ggml.cs:679-687
#pragma unroll
        for (int l = 0; l < QK; ++l) {
            if (v[l] < mn) mn = v[l];
            if (v[l] > mx) mx = v[l];
        }
        const float d = (mx - mn) / 31.0f;                  // Ggml.cs:689-690
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        uint16_t *o = (uint16_t *)(out);
        o[0] = float_to_half_bits_rne(d);                   // Ggml.cs:692-693 (D7, intent)
        o[1] = float_to_half_bits_rne(mn);
        uint32_t qh = 0;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int l = 0; l < QK; l += 2) {                   // Ggml.cs:697-710: (uint)(v + 0.5f), low nibble + 5th bit
            const uint32_t t0 = (uint32_t)((v[l + 0] - mn) * id + 0.5f);
            const uint32_t t1 = (uint32_t)((v[l + 1] - mn) * id + 0.5f);
            w[l / 8] |= ((t0 & 0x0Fu) | ((t1 & 0x0Fu) << 4)) << (8 * ((l / 2) & 3));
            qh |= ((t0 & 0x10u) >> 4) << (l + 0);
            qh |= ((t1 & 0x10u) >> 4) << (l + 1);
        }
        uint32_t *o4 = (uint32_t *)(out);
        o4[1] = qh;
#pragma unroll
        for (int k = 0; k < 4; ++k) o4[2 + k] = w[k];
    } else {  // Q8_0 / Q8_1
        float amax = 0.0f;
#pragma unroll
        for (int l = 0; l < QK; ++l) amax = fmaxf(amax, fabsf(v[l]));
        const float d = amax / 127.0f;
        const float id = d != 0.0f ? 1.0f / d : 0.0f;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int sum0 = 0, sum1 = 0;
#pragma unroll
        for (int l = 0; l < QK; ++l) {
            const int q = (int)rintf(v[l] * id);
            w[l / 4] |= ((uint32_t)q & 0xFFu) << (8 * (l & 3));
            if (l < 16) sum0 += q; else sum1 += q;
        }
        if (TYPE == GGML_TYPE_Q8_0) {
            uint32_t *o = (uint32_t *)(out);
            o[0] = __float_as_uint(d);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[1 + k] = w[k];
        } else {
            uint32_t *o = (uint32_t *)(out);
            o[0] = __float_as_uint(d);
            o[1] = __float_as_uint(d * (float)sum0);         // Ggml.cs:820-821 (D3 signed sums)
            o[2] = __float_as_uint(d * (float)sum1);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[3 + k] = w[k];
        }
    }
}

template <int TYPE>
__device__ __forceinline__ void dequant_block(const uint8_t *__restrict__ in, float (&v)[QK]) {
    if (TYPE == GGML_TYPE_Q4_0) {
        const uint32_t *s = (const uint32_t *)(in);
        const float d = __uint_as_float(s[0]);
#pragma unroll
        for (int l = 0; l < QK; l += 2) {
            const uint32_t byte = (s[1 + l / 8] >> (8 * ((l / 2) & 3))) & 0xFFu;
            v[l + 0] = (float)((int)(byte & 0x0F) - 8) * d;   // Ggml.cs:899-900
            v[l + 1] = (float)((int)(byte >> 4) - 8) * d;
        }
    } else if (TYPE == GGML_TYPE_Q4_1) {
        const uint32_t *s = (const uint32_t *)(in);
        const float d = __uint_as_float(s[0]), m = __uint_as_float(s[1]);
#pragma unroll
        for (int l = 0; l < QK; l += 2) {
            const uint32_t byte = (s[2 + l / 8] >> (8 * ((l / 2) & 3))) & 0xFFu;
            const float p0 = (float)(byte & 0x0F) * d, p1 = (float)(byte >> 4) * d;  // Ggml.cs:976-977, mul then add
            v[l + 0] = p0 + m;
            v[l + 1] = p1 + m;
        }
    } else if (TYPE == GGML_TYPE_Q5_0) {
        const uint16_t *s = (const uint16_t *)(in);
        const float d = half_bits_to_float_q(s[0]);
        const uint32_t qh = (uint32_t)s[1] | ((uint32_t)s[2] << 16);
#pragma unroll
        for (int l = 0; l < QK; l += 2) {
            const uint32_t byte = (s[3 + l / 4] >> (8 * ((l / 2) & 1))) & 0xFFu;
            const int vi0 = (int)((byte & 0x0F) | (((qh >> (l + 0)) & 1u) << 4));
            const int vi1 = (int)((byte >> 4) | (((qh >> (l + 1)) & 1u) << 4));
            v[l + 0] = (float)(vi0 - 16) * d;                  // Ggml.cs:1051-1052
            v[l + 1] = (float)(vi1 - 16) * d;
        }
    } else if (TYPE == GGML_TYPE_Q4_2) {
        const uint16_t *s = (const uint16_t *)(in);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const float d = half_bits_to_float_q(s[5 * hb]);
#pragma unroll
            for (int l = 0; l < 16; l += 2) {
                const uint32_t byte = (s[5 * hb + 1 + l / 4] >> (8 * ((l / 2) & 1))) & 0xFFu;
                v[16 * hb + l + 0] = (float)((int)(byte & 0x0F) - 8) * d;   // Ggml.cs:1011-1012
                v[16 * hb + l + 1] = (float)((int)(byte >> 4) - 8) * d;
            }
        }
    } else if (TYPE == GGML_TYPE_Q5_1) {
        const uint16_t *s = (const uint16_t *)(in);
        const uint32_t *s4 = (const uint32_t *)(in);
        const float d = half_bits_to_float_q(s[0]), m = half_bits_to_float_q(s[1]);
        const uint32_t qh = s4[1];
#pragma unroll
        for (int l = 0; l < QK; l += 2) {
            const uint32_t byte = (s4[2 + l / 8] >> (8 * ((l / 2) & 3))) & 0xFFu;
            const int vi0 = (int)((byte & 0x0F) | (((qh >> (l + 0)) & 1u) << 4));
            const int vi1 = (int)((byte >> 4) | (((qh >> (l + 1)) & 1u) << 4));
            const float p0 = (float)vi0 * d, p1 = (float)vi1 * d;    // Ggml.cs:1091-1092, mul then add
            v[l + 0] = p0 + m;
            v[l + 1] = p1 + m;
        }
    } else {  // Q8_0
        const uint32_t *s = (const uint32_t *)(in);
        const float d = __uint_as_float(s[0]);
#pragma unroll
        for (int l = 0; l < QK; ++l) {
            const int q = (int)(int8_t)((s[1 + l / 4] >> (8 * (l & 3))) & 0xFFu);
            v[l] = (float)q * d;                               // Ggml.cs:1119 (signed, D4)
        }
    }
}

// bytes per 32 elements (Q4_2: two of its 10-byte blocks)
template <int TYPE> struct BlockBytes { static constexpr int value = TYPE == GGML_TYPE_Q4_0 ? 20 : TYPE == GGML_TYPE_Q4_1 ? 24 : TYPE == GGML_TYPE_Q4_2 ? 20 : TYPE == GGML_TYPE_Q5_0 ? 22 : TYPE == GGML_TYPE_Q5_1 ? 24 : TYPE == GGML_TYPE_Q8_0 ? 36 : 44; };

// K9: rows of f32 (SRC_F16 = false) or f16 (true; widened exactly first, Ggml.cs:3951-3956) -> blocks.
// Row r of the source starts at x + r * ld elements; blocks of a row are contiguous, rows of blocks are contiguous.
template <int TYPE, bool SRC_F16>
__global__ void quantize_rows_kernel(const void *__restrict__ x, int64_t ld, int64_t nbr, int64_t nblocks, uint8_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const int64_t r = i / nbr, bi = i - r * nbr;
    float v[QK];
    if (SRC_F16) {
        const uint4 *p = (const uint4 *)((const uint16_t *)x + r * ld + bi * QK);
#pragma unroll
        for (int l = 0; l < QK / 8; ++l) {
            const uint4 h = p[l];
            const uint32_t w[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[8 * l + 2 * k + 0] = half_bits_to_float_q((uint16_t)(w[k] & 0xFFFFu));
                v[8 * l + 2 * k + 1] = half_bits_to_float_q((uint16_t)(w[k] >> 16));
            }
        }
    } else {
        const float4 *p = (const float4 *)((const float *)x + r * ld + bi * QK);
#pragma unroll
        for (int l = 0; l < QK / 4; ++l) {
            const float4 f = p[l];
            v[4 * l + 0] = f.x; v[4 * l + 1] = f.y; v[4 * l + 2] = f.z; v[4 * l + 3] = f.w;
        }
    }
    quant_block<TYPE>(v, out + i * BlockBytes<TYPE>::value);
}

// One thread dequantizes one block (the arithmetic of dequant_block, bit for bit); the 128 blocks of a workgroup go out through LDS so
// that the stores are the 16 KB they cover in order -- a thread writing its own 128 bytes puts 64 scattered 16-byte pieces into
// every store instruction (3.2 TB/s of traffic before, see DESIGN 5 for the figure after).
template <int TYPE>
__global__ __launch_bounds__(128) void dequantize_rows_kernel(const uint8_t *__restrict__ in, int64_t nblocks, float *__restrict__ y) {
    constexpr int ROW = QK + 4;                              // floats per LDS row: 16-byte aligned, rows 4 banks apart
    __shared__ __attribute__((aligned(16))) float sm[128 * ROW];
    const int t = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 128, i = b0 + t;
    if (i < nblocks) {
        float v[QK];
        dequant_block<TYPE>(in + i * BlockBytes<TYPE>::value, v);
#pragma unroll
        for (int l = 0; l < QK / 4; ++l) *(float4 *)(sm + t * ROW + 4 * l) = make_float4(v[4 * l + 0], v[4 * l + 1], v[4 * l + 2], v[4 * l + 3]);
    }
    __syncthreads();
    const int64_t nleft = nblocks - b0;
    const int nb = nleft < 128 ? (int)nleft : 128;           // blocks of this workgroup
    float4 *o = (float4 *)(y + b0 * QK);
#pragma unroll
    for (int j = 0; j < QK / 4; ++j) {
        const int idx = t + 128 * j, row = idx >> 3, c4 = idx & 7;
        if (row < nb) o[idx] = *(const float4 *)(sm + row * ROW + 4 * c4);
    }
}

// ggml_compute_forward_add_q_f32 (Ggml.cs:4797-4906) per block: dequantize_row_q -> ggml_vec_acc_f32 (y += x) ->
// quantize_row_q, all in registers: 0.625 + 4 B read, 0.625 B written per element (Q4_0).  Bit-exact composition.
template <int TYPE>
__global__ __launch_bounds__(128) void add_q_f32_kernel(const uint8_t *__restrict__ in, const float *__restrict__ x, int64_t nblocks,
                                                        uint8_t *__restrict__ out) {
    const int t = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 128, i = b0 + t;
    float v[QK];
    // The types with the lighter block arithmetic take x the way dequantize_rows_kernel's result goes out: the 16 KB of the
    // workgroup's 128 blocks in address order, handed to the thread that owns the block through LDS (4096 x 4096: Q4_0 17.6 ->
    // 15.5 us, Q4_1 17.9 -> 16.3, Q8_0 17.6 -> 16.3; the 5-bit types and Q4_2 lose 10-15 % that way -- their requantization
    // is what bounds them -- and K9, which only reads x, measures level either way)
    if constexpr (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q8_0) {
        constexpr int ROW = QK + 4;
        __shared__ __attribute__((aligned(16))) float sm[128 * ROW];
        const int64_t nleft = nblocks - b0;
        const int nb = nleft < 128 ? (int)nleft : 128;       // blocks of this workgroup
        const float4 *p = (const float4 *)(x + b0 * QK);
        float4 f[QK / 4];
#pragma unroll
        for (int j = 0; j < QK / 4; ++j) {
            const int idx = t + 128 * j;
            f[j] = (idx >> 3) < nb ? p[idx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (i < nblocks) dequant_block<TYPE>(in + i * BlockBytes<TYPE>::value, v);
#pragma unroll
        for (int j = 0; j < QK / 4; ++j) {
            const int idx = t + 128 * j;
            *(float4 *)(sm + (idx >> 3) * ROW + 4 * (idx & 7)) = f[j];
        }
        __syncthreads();
        if (i >= nblocks) return;
#pragma unroll
        for (int l = 0; l < QK / 4; ++l) {
            const float4 g = *(const float4 *)(sm + t * ROW + 4 * l);
            v[4 * l + 0] += g.x; v[4 * l + 1] += g.y; v[4 * l + 2] += g.z; v[4 * l + 3] += g.w;
        }
    } else {
        if (i >= nblocks) return;
        dequant_block<TYPE>(in + i * BlockBytes<TYPE>::value, v);
        const float4 *p = (const float4 *)(x + i * QK);
#pragma unroll
        for (int l = 0; l < QK / 4; ++l) {
            const float4 f = p[l];
            v[4 * l + 0] += f.x; v[4 * l + 1] += f.y; v[4 * l + 2] += f.z; v[4 * l + 3] += f.w;
        }
    }
    quant_block<TYPE>(v, out + i * BlockBytes<TYPE>::value);
}

// reference-format Q8_0 / Q8_1 rows (block_q8_0 36 B, block_q8_1 44 B) -> the planar scratch the dot kernels read.
// Serves ggml_hip_vec_dot (Seam 2), whose vy argument is already-quantized blocks.
template <int TYPE>
__global__ void q8_aos_to_planes_kernel(const uint8_t *__restrict__ in, int64_t N, int64_t nbk, int8_t *__restrict__ a8,
                                        float *__restrict__ ad, int32_t *__restrict__ as, int64_t Npad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * nbk) return;
    const int64_t n = i / nbk, b = i % nbk;
    constexpr int BS = TYPE == GGML_TYPE_Q8_0 ? 36 : 44;
    constexpr int QOFF = TYPE == GGML_TYPE_Q8_0 ? 1 : 3;
    const uint32_t *s = (const uint32_t *)(in + i * BS);
    uint32_t ev[4], od[4];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = s[QOFF + 2 * k], hi = s[QOFF + 2 * k + 1];
        ev[k] = (lo & 0xFFu) | ((lo >> 8) & 0xFF00u) | ((hi & 0xFFu) << 16) | ((hi << 8) & 0xFF000000u);
        od[k] = ((lo >> 8) & 0xFFu) | ((lo >> 16) & 0xFF00u) | ((hi << 8) & 0xFF0000u) | (hi & 0xFF000000u);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += (int)(int8_t)(lo >> (8 * e)) + (int)(int8_t)(hi >> (8 * e));
    }
    *(uint4 *)(a8 + ((b * 2 + 0) * Npad + n) * 16) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
    *(uint4 *)(a8 + ((b * 2 + 1) * Npad + n) * 16) = make_uint4(od[0], od[1], od[2], od[3]);
    ad[b * Npad + n] = __uint_as_float(s[0]);
    as[b * Npad + n] = sum;
}

}  // namespace

hipError_t launch_quantize_act(const float *x, int64_t N, int64_t K, int64_t ld1, act_planes p, int image, hipStream_t st, bool q8k) {
    if (N <= 0) return hipSuccess;
    const int64_t nbk = K / QK;
    uint8_t *sp3 = nullptr;                                 // image 0 + the min-term piece planes (they live in the half of the a8 region image 0 leaves free)
    if (image & ACT_IMAGE_MIN_PIECES) {
        image &= ~ACT_IMAGE_MIN_PIECES;
        if (image != 0 || nbk < 8 || !p.sp3) return hipErrorInvalidValue;   // (room: (nbk / 8 + 2) * 3 piece planes within the 2 nbk planes image 0 leaves free)
        sp3 = p.sp3;
    }
    dim3 grid((unsigned)((nbk + K1_BPB - 1) / K1_BPB), (unsigned)((N + 31) / 32));
    if (q8k) {                                              // Q8_K rule (kquants.hip): K is whole super-blocks, no bf6 image
        if (K % 256 != 0 || image == 3) return hipErrorInvalidValue;
        if (image == 1) quantize_act_kernel<1, true><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad);
        else if (image == 2) quantize_act_kernel<2, true><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad);
        else quantize_act_kernel<0, true><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad, sp3);
        return hipGetLastError();
    }
    if (image == 1)
        quantize_act_kernel<1><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad);
    else if (image == 2)
        quantize_act_kernel<2><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad);
    else if (image == 3) {
        // lane-per-block form when the 32-bit buffer offsets reach (they do for every shape whose image does); rows must
        // be 16-byte aligned for the DMA (ld1 % 4, checked by the callers for every device entry)
        static const bool old_k1 = dev_env_set("GGML_HIP_K1_OLD");   // developer A/B switch
        const uint64_t xb = ((uint64_t)(N - 1) * (uint64_t)ld1 + (uint64_t)K) * 4;
        if (!old_k1 && xb <= 0xFFFFFFFFull && ld1 % 4 == 0 && ((uintptr_t)x & 15) == 0) {
            dim3 g2((unsigned)(pad_kblocks(nbk) / 4), (unsigned)((N + 63) / 64));
            quantize_act_bf6_kernel<<<g2, 256, 0, st>>>(x, (int)N, (int)nbk, (uint32_t)(ld1 * 4), (uint32_t)xb, p.a8, p.ad, p.as, (int)p.Npad);
        } else {
            quantize_act_kernel<3><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad);
        }
    }
    else
        quantize_act_kernel<0><<<grid, 256, 0, st>>>(x, N, nbk, ld1, p.a8, p.ad, p.as, p.Npad, sp3);
    return hipGetLastError();
}

hipError_t launch_quantize_rows(int type, int src_type, const void *x, int64_t ld, int64_t nrows, int64_t k, void *blocks,
                                hipStream_t st) {
    const int64_t nbr = k / QK, nblocks = nrows * nbr;
    if (nblocks <= 0) return hipSuccess;
    dim3 grid((unsigned)((nblocks + 127) / 128));
    uint8_t *o = (uint8_t *)blocks;
#define QR(T) do { if (src_type == GGML_TYPE_F16) quantize_rows_kernel<T, true><<<grid, 128, 0, st>>>(x, ld, nbr, nblocks, o); \
                   else quantize_rows_kernel<T, false><<<grid, 128, 0, st>>>(x, ld, nbr, nblocks, o); } while (0)
    switch (type) {
    case GGML_TYPE_Q4_0: QR(GGML_TYPE_Q4_0); break;
    case GGML_TYPE_Q4_1: QR(GGML_TYPE_Q4_1); break;
    case GGML_TYPE_Q5_0: QR(GGML_TYPE_Q5_0); break;
    case GGML_TYPE_Q4_2: QR(GGML_TYPE_Q4_2); break;
    case GGML_TYPE_Q5_1: QR(GGML_TYPE_Q5_1); break;
    case GGML_TYPE_Q8_0: QR(GGML_TYPE_Q8_0); break;
    case GGML_TYPE_Q8_1: QR(GGML_TYPE_Q8_1); break;
    default: return hipErrorInvalidValue;
    }
#undef QR
    return hipGetLastError();
}

hipError_t launch_add_q_f32(int type, const void *blocks_in, const float *x, int64_t nrows, int64_t k, void *blocks_out,
                            hipStream_t st) {
    const int64_t nblocks = nrows * (k / QK);
    if (nblocks <= 0) return hipSuccess;
    dim3 grid((unsigned)((nblocks + 127) / 128));
    const uint8_t *in = (const uint8_t *)blocks_in;
    uint8_t *o = (uint8_t *)blocks_out;
    switch (type) {
    case GGML_TYPE_Q4_0: add_q_f32_kernel<GGML_TYPE_Q4_0><<<grid, 128, 0, st>>>(in, x, nblocks, o); break;
    case GGML_TYPE_Q4_1: add_q_f32_kernel<GGML_TYPE_Q4_1><<<grid, 128, 0, st>>>(in, x, nblocks, o); break;
    case GGML_TYPE_Q5_0: add_q_f32_kernel<GGML_TYPE_Q5_0><<<grid, 128, 0, st>>>(in, x, nblocks, o); break;
    case GGML_TYPE_Q4_2: add_q_f32_kernel<GGML_TYPE_Q4_2><<<grid, 128, 0, st>>>(in, x, nblocks, o); break;
    case GGML_TYPE_Q5_1: add_q_f32_kernel<GGML_TYPE_Q5_1><<<grid, 128, 0, st>>>(in, x, nblocks, o); break;
    case GGML_TYPE_Q8_0: add_q_f32_kernel<GGML_TYPE_Q8_0><<<grid, 128, 0, st>>>(in, x, nblocks, o); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_dequantize_rows(int type, const void *blocks, int64_t nrows, int64_t k, float *y, hipStream_t st) {
    const int64_t nblocks = nrows * (k / QK);
    if (nblocks <= 0) return hipSuccess;
    dim3 grid((unsigned)((nblocks + 127) / 128));
    const uint8_t *in = (const uint8_t *)blocks;
    switch (type) {
    case GGML_TYPE_Q4_0: dequantize_rows_kernel<GGML_TYPE_Q4_0><<<grid, 128, 0, st>>>(in, nblocks, y); break;
    case GGML_TYPE_Q4_1: dequantize_rows_kernel<GGML_TYPE_Q4_1><<<grid, 128, 0, st>>>(in, nblocks, y); break;
    case GGML_TYPE_Q5_0: dequantize_rows_kernel<GGML_TYPE_Q5_0><<<grid, 128, 0, st>>>(in, nblocks, y); break;
    case GGML_TYPE_Q4_2: dequantize_rows_kernel<GGML_TYPE_Q4_2><<<grid, 128, 0, st>>>(in, nblocks, y); break;
    case GGML_TYPE_Q5_1: dequantize_rows_kernel<GGML_TYPE_Q5_1><<<grid, 128, 0, st>>>(in, nblocks, y); break;
    case GGML_TYPE_Q8_0: dequantize_rows_kernel<GGML_TYPE_Q8_0><<<grid, 128, 0, st>>>(in, nblocks, y); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_q8_aos_to_planes(int q8type, const void *blocks, int64_t N, int64_t K, act_planes p, hipStream_t st) {
    const int64_t nbk = K / QK, total = N * nbk;
    if (total <= 0) return hipSuccess;
    dim3 grid((unsigned)((total + 127) / 128));
    if (q8type == GGML_TYPE_Q8_0)
        q8_aos_to_planes_kernel<GGML_TYPE_Q8_0><<<grid, 128, 0, st>>>((const uint8_t *)blocks, N, nbk, p.a8, p.ad, p.as, p.Npad);
    else if (q8type == GGML_TYPE_Q8_1)
        q8_aos_to_planes_kernel<GGML_TYPE_Q8_1><<<grid, 128, 0, st>>>((const uint8_t *)blocks, N, nbk, p.a8, p.ad, p.as, p.Npad);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
