// gemm_q8s.hip -- K3s-i8: Q8_0 (r4: and Q5_0) batches of 5 .. 64 rows on the int8 matrix cores, the stage-free form of gemm_qmx.hip K3s.
// (Its own translation unit for its compile flags: with SLP vectorization the packed f32 forms of the scale-accumulate want
// aligned register tuples across the loop over K and the looped kernels spill -- 256 registers + 0.1 .. 1.4 KB of scratch against
// 119 registers without it; Makefile FLAGS_gemm_q8s.hip.)
#include "common.h"
#include "plan.h"
#include <utility>

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---- K3s for Q8_0: batches of 5 .. 64 rows, K >= 2048 (the stage-free form of gemm_qmx.hip on the int8 matrix cores) --------------------
// Q8_0's resident planes ARE int8 MFMA operands -- [k-block][half][row][16 B], the halves holding the even and the odd elements, the way K1
// writes the activations (image 0) -- so a k-block is one 16-byte load per operand and lane and one v_mfma_i32_32x32x32_i8, no digits and
// nothing to expand: 36 B per 32 weights through the CU's memory path where the bf6 planes of Q4_0 carry 28 and its activation image 48.
// Structure as K3s: a workgroup = WMT 32-row weight tiles x one 32-column slice of src1 x KS waves with a contiguous range of k-blocks each;
// a wave requests its first NB blocks before anything else (K <= 4096: all of them; longer K: round after round), keeps its
// slice of the row scales in its own LDS slice; the waves' sums are added in wave order, every wave taking its share of the rows.
// Arithmetic per block as in the kernel above: acc = fma((float)sumi, d1 * d0, acc) (Ggml.cs:1377-1378).
template <int KS, int NB, bool ROT, int WMT, bool Q5 = false, int MINT = 0, bool Q42 = false>     // MINT: bf16 pieces of the weight's min (0: no min term; 2: Q5_1; 3: Q5_K); Q42: two 16-element blocks per k-block
__device__ __forceinline__
void gemm_q8_small_body(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8, const float *__restrict__ ad,
                        float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbk, int nloc, int64_t ldd, const mm_epilogue &ep, int ntw,
                        uint32_t w_bytes, uint32_t a_bytes, int wg, const uint8_t *__restrict__ mp3 = nullptr, const uint8_t *__restrict__ sp3 = nullptr,
                        const float *__restrict__ wm = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    int rt, ct;
    k3s_tile_of(wg, ntw, (N + 31) / 32, rt, ct);
    const int m0 = rt * 32 * WMT, n0 = ct * 32;
    const int kb0 = wave * nloc;
    // a slot past the wave's range or past the end of K (K1's image 0 does not write the k-blocks K is padded to) repeats a valid block:
    // its table row is zero, so it adds (sumi * 0) = +0
    auto blk = [&](int i) { const int kb = kb0 + (i < nloc ? i : nloc - 1); return kb < nbk ? kb : nbk - 1; };

    struct WB { i32x4 q[WMT]; float d[WMT]; float m[Q42 ? WMT : 1]; };   // (Q4_2: m = the scale of the k-block's second 16-element block)
    WB wb[NB];
    i32x4 ab[NB];
    // raw buffer addressing: one 32-bit offset per lane and plane, the k-block in the scalar offset (planes past their end read 0)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(qs), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wd), 0, (int)(w_bytes / 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(a8), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q42 ? wm : wd), 0, (int)(w_bytes / 8), 0x00020000);
    const uint32_t offW = (uint32_t)((hh * Mpad + m0 + l31) * 16), offD = (uint32_t)((m0 + l31) * 4), offA = (uint32_t)((hh * Npad + n0 + l31) * 16);
    const uint32_t w_blk = (uint32_t)(2 * Mpad * 16), d_blk = (uint32_t)(Mpad * 4), a_blk = (uint32_t)(2 * Npad * 16);
    auto load_blk = [&](WB &f, i32x4 &a, int i) {
        const uint32_t kb = (uint32_t)blk(i);
        a = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)offA, (int)(kb * a_blk), 0));
#pragma unroll
        for (int t = 0; t < WMT; ++t) {
            f.q[t] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(offW + 512u * t), (int)(kb * w_blk), 0));
            f.d[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)(offD + 128u * t), (int)(kb * d_blk), 0));
            if constexpr (Q42) f.m[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, (int)(offD + 128u * t), (int)(kb * d_blk), 0));
        }
    };

    // ---- this wave's slice of the row scales: rows x 32 floats, straight into its own LDS slice (rows past its range: zero) ----
    const int trows = (nloc + NB - 1) / NB * NB;            // whole rounds of NB slots: rows past the wave's range are zero
    float *const tabD = (float *)smem8 + (size_t)wave * trows * 32;
    constexpr int TP = 8;                                   // float4 pieces per lane and round: 64 table rows
    f32x4 td[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int idx = lane + 64 * j, b = idx >> 3, c4 = idx & 7;
        const bool ok = b < nloc && kb0 + b < nbk;
        td[j] = ok ? *(const f32x4 *)(ad + (size_t)(kb0 + b) * Npad + n0 + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    // (min-term types) the wave's FIRST chunk of min-term pieces is requested here, in front of the blocks' loads: loads return in order, so the
    // chunk's wait below is then for the chunk alone (r5: issued behind the blocks it waited for all of them, and every further chunk exposed a
    // memory round trip with nothing else in flight -- Q5_1 4096 x 11008 x 16 29.4 us where Q5_0 took 13.6)
    using f32x16m = __attribute__((ext_vector_type(16))) float;
    using bf16x8m = __attribute__((ext_vector_type(8))) __bf16;
    constexpr int MP = MINT ? MINT : 1;
    const int nchunks = (nbk + 15) / 16;
    const __amdgpu_buffer_rsrc_t rMP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(MINT ? mp3 : qs), 0, MINT ? (int)((uint32_t)(nchunks * 6) * (uint32_t)Mpad * 16u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rSP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(MINT ? sp3 : qs), 0, MINT ? (int)((uint32_t)(nchunks * 6) * (uint32_t)Npad * 16u) : 0, 0x00020000);
    const uint32_t vM = (uint32_t)((3 * hh) * Mpad + m0 + l31) * 16u, vS = (uint32_t)((3 * hh) * Npad + n0 + l31) * 16u;
    auto chunk_load = [&](int c, i32x4 (&sa_)[3], i32x4 (&mb_)[WMT][MP]) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            if (pc < MINT)
#pragma unroll
                for (int t = 0; t < WMT; ++t)
                    mb_[t][pc] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rMP, (int)(vM + 512u * t), (int)((uint32_t)(6 * c + pc) * (uint32_t)Mpad * 16u), 0));
            sa_[pc] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rSP, (int)vS, (int)((uint32_t)(6 * c + pc) * (uint32_t)Npad * 16u), 0));
        }
    };
    i32x4 sa0[3], mb0[WMT][MP];
    if constexpr (MINT) { if (wave < nchunks) chunk_load(wave, sa0, mb0); }
    static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; load_blk(wb[u], ab[u], u); });
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int idx = lane + 64 * j;
        if (idx < trows * 8) *(f32x4 *)(tabD + 4 * idx) = td[j];
    }
    if (trows > 64) {                                       // (r4: a second round of pieces -- 65 .. 128 k-blocks per wave, K up to 32768; it was K <= 16384)
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = lane + 64 * (j + TP), b = idx >> 3, c4 = idx & 7;
            const bool ok = b < nloc && kb0 + b < nbk;
            td[j] = ok ? *(const f32x4 *)(ad + (size_t)(kb0 + b) * Npad + n0 + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = lane + 64 * (j + TP);
            if (idx < trows * 8) *(f32x4 *)(tabD + 4 * idx) = td[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float acc[WMT][16];
#pragma unroll
    for (int t = 0; t < WMT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (r4, Q5_1) the min terms m * (s0 + s1) of sixteen k-blocks of a tile as one K = 16 product on the bf16 cores, in front of the wave's
    // blocks -- gemm_qmp.hip's min-term product on this kernel's tiles: both operands as bf16 pieces that sum to the f32 value exactly
    // (ggml_hip_weight::mp3 from the upload, act_planes::sp3 = d1 * (float)sum(a) from K1), a Q5_1 min is an f16 value (two pieces, five
    // products); chunk c (k-blocks 16c .. 16c + 15, k-group 2c + lane half) is wave c % KS's, by K alone.
    if constexpr (MINT) {
        // chunk c's products while chunk c + KS's pieces travel (two register sets, the second copied down): the ORDER of an element's additions is
        // unchanged -- the wave's chunks in ascending order, six / five products each, smallest first, then its blocks
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest products first (activation piece x weight piece)
        if (wave < nchunks) {
            for (int c = wave;;) {
                const int cn = c + KS;
                i32x4 sa1[3], mb1[WMT][MP];
                if (cn < nchunks) chunk_load(cn, sa1, mb1);
#pragma unroll
                for (int t = 0; t < WMT; ++t) {
                    f32x16m a;
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = acc[t][r];
                    static_for<6>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if constexpr (PB[k] < MINT)             // (a Q5_1 min is an f16 value: two pieces, five products; Q5_K's is an f32 product: three, six)
                            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8m, sa0[PA[k]]), __builtin_bit_cast(bf16x8m, mb0[t][PB[k]]), a, 0, 0, 0);
                    });
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = a[r];
                }
                if (cn >= nchunks) break;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    sa0[pc] = sa1[pc];
#pragma unroll
                    for (int t = 0; t < WMT; ++t) if (pc < MINT) mb0[t][pc] = mb1[t][pc];
                }
                c = cn;
            }
        }
    }

    auto block = [&](int i, auto uc) {                      // block i of the wave out of slot u
        constexpr int u = decltype(uc)::value;
        WB &w = wb[u];
        i32x16 t[WMT];
        float dw[WMT];
        const float *dp = tabD + i * 32 + 4 * hh;
        if constexpr (Q42) {
            // Q4_2 (r4): a k-block is two 16-element blocks with a scale each (Ggml.cs:1217-1252): bytes 0..7 of both operand planes are the
            // first block's elements, bytes 8..15 the second's -- one v_mfma_i32_32x32x16_i8 per half gives sumi_0 and sumi_1 apart;
            // sumf += (d0 * yd) * sumi_0, then += (d1 * yd) * sumi_1 (its second scale rides in the m plane)
            i32x16 t2[WMT];
            float dm[WMT];
            const long a_lo = (long)(uint32_t)ab[u][0] | ((long)ab[u][1] << 32), a_hi = (long)(uint32_t)ab[u][2] | ((long)ab[u][3] << 32);
#pragma unroll
            for (int k = 0; k < WMT; ++k) {
                const long w_lo = (long)(uint32_t)w.q[k][0] | ((long)w.q[k][1] << 32), w_hi = (long)(uint32_t)w.q[k][2] | ((long)w.q[k][3] << 32);
                t[k] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a_lo, w_lo, zero, 0, 0, 0);
                t2[k] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a_hi, w_hi, zero, 0, 0, 0);
                dw[k] = w.d[k]; dm[k] = w.m[k];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 da = *(const f32x4 *)(dp + 8 * q);
#pragma unroll
                for (int k = 0; k < WMT; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[k][4 * q + e] = fmaf(dw[k] * da[e], (float)t[k][4 * q + e], acc[k][4 * q + e]);
                        acc[k][4 * q + e] = fmaf(dm[k] * da[e], (float)t2[k][4 * q + e], acc[k][4 * q + e]);
                    }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < WMT; ++k) { t[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ab[u], w.q[k], zero, 0, 0, 0); dw[k] = w.d[k]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 da = *(const f32x4 *)(dp + 8 * q);
#pragma unroll
            for (int k = 0; k < WMT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if constexpr (Q5) acc[k][4 * q + e] = fmaf(dw[k] * (float)t[k][4 * q + e], da[e], acc[k][4 * q + e]);      // (d * sxy) * y.d, Ggml.cs:1296-1298
                    else acc[k][4 * q + e] = fmaf((float)t[k][4 * q + e], da[e] * dw[k], acc[k][4 * q + e]);                    // Ggml.cs:1377-1378
        }
    };
    if constexpr (!ROT) {
        static_for<NB>([&](auto uc) { block(decltype(uc)::value, uc); });
    } else {
        // longer K: rounds of NB blocks, a round's loads issued together behind the previous round's arithmetic (refilling a slot as soon
        // as its own MFMA has issued -- the form of gemm_qmx.hip -- spills here: 256 registers + 0.3 KB of scratch)
        static_for<NB>([&](auto uc) { block(decltype(uc)::value, uc); });
        for (int base = NB; base < nloc; base += NB) {
            __builtin_amdgcn_sched_barrier(0);              // (hoisted above the arithmetic, the loads would double the live slots)
            static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; load_blk(wb[u], ab[u], base + u); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<NB>([&](auto uc) { block(base + decltype(uc)::value, uc); });
        }
    }

    // ---- the waves' sums in wave order, every wave its share of the rows ----
    __syncthreads();
    float *xch = (float *)smem8 + lane;
#pragma unroll
    for (int t = 0; t < WMT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(size_t)((wave * WMT + t) * 16 + r) * 64] = acc[t][r];
    __syncthreads();
    constexpr int RW = 16 * WMT / KS;
    static_assert((16 * WMT) % KS == 0, "rows per wave");
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int rr = wave * RW + k, t = rr / 16, r = rr % 16;
        float v = xch[(size_t)(t * 16 + r) * 64];
#pragma unroll
        for (int g = 1; g < KS; ++g) v += xch[(size_t)((g * WMT + t) * 16 + r) * 64];
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * hh, m = m0 + 32 * t + l31;
        if (n < N && m < M) {
            if (ep.mode == 3) {                             // the row split's exchange: the same position in this rank's and every peer's [N][M] buffer (r5: the batched-decode forms too)
                dst[(size_t)n * ldd + m] = v;
                for (int pk = 0; pk < ep.npush; ++pk) ep.push[pk][(size_t)n * ldd + m] = v;
            } else if (ep.mode == 2) {
                dst[(size_t)n * ldd + m] = v * ep.scale;
            } else {
                dst[(size_t)n * ldd + m] = v;
                if (ep.mode == 1) ep.dst2[(size_t)n * ep.ld2 + m] = v + ep.addend[(size_t)n * ep.ld_add + m];
            }
        }
    }
}

template <int KS, int NB, bool ROT, int WMT, int TY>       // TY: 0 Q8_0, 1 Q5_0, 2 Q5_1 (Q5_0's block term + the min-term product), 3 Q5_K in the Q5_1 form (three min pieces), 4 Q4_2 (two 16-element blocks per k-block)
__global__ __launch_bounds__(KS * 64, 1)
void gemm_q8_small_kernel(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8, const float *__restrict__ ad,
                          float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbk, int nloc, int64_t ldd, const mm_epilogue ep, int ntw,
                          uint32_t w_bytes, uint32_t a_bytes, const uint8_t *__restrict__ mp3, const uint8_t *__restrict__ sp3, const float *__restrict__ wm) {
    gemm_q8_small_body<KS, NB, ROT, WMT, TY == 1 || TY == 2 || TY == 3, (TY == 2 || TY == 3 ? TY : 0), TY == 4>(qs, wd, a8, ad, dst, M, N, Mpad, Npad, nbk, nloc, ldd, ep, ntw,
                                                                                                     w_bytes, a_bytes, (int)blockIdx.x, mp3, sp3, wm);
}

// ---- K3s-i8 on 16-ROW tiles (r5): the same form for matrices whose 32-row tiles do not fill the chip --------------------------------------
// 4096 rows at up to 32 src1 rows are 128 workgroups of the form above on 256 CUs: half of the chip pulls the whole matrix through its
// memory path, and a CU's path sustains about 10 B per cycle from HBM -- 4096 x 4096 x 32 measured a flat 8-10 us for every type where the
// 19 MB of Q8_0 planes are 3 us of HBM time.  Here a workgroup owns a 16-row weight tile x NCT 16-column slices of src1 on
// v_mfma_i32_16x16x32_i8 (K = 32: one quant block, so the block's integer sum comes out apart as the reference's sumi does): twice the
// workgroups, every CU streaming (NCT = 1 or 2: up to 32 src1 rows; four slices at 33..64 rows measured slower than two 32-row workgroups).
// THE SUMMATION TREE IS THE 32-ROW FORM'S: the same KS = 8 contiguous ranges of nloc k-blocks (wave w: w * nloc ..), a range's blocks in
// ascending order with the same f32 statement per block, the eight partial sums added in wave order -- so the geometry may follow M and a
// row shard computes the bits of the unsplit matrix whichever of the two forms either of them runs (tests/test_gpu_parity.py
// test_k3s_16_row_tiles_are_bitwise_the_32_row_form).
// Operands: the resident int8 planes [k-block][half h][row][16 B] (plane h byte j = element 2 j + h) and K1's image 0 of the same layout;
// lane (row or column l % 16, group g = l / 16) takes bytes 8 (g & 1) .. + 7 of plane g >> 1 of its row -- 8-byte loads, the same element
// permutation on both operands, 512 contiguous bytes per operand and k-block per wave instruction pair of planes.
// D[i = src1 column][j = weight row]: lane holds weight row l % 16 and the columns 4 (l / 16) + 0..3 of the slice.
using i32x4v = __attribute__((ext_vector_type(4))) int;
template <int KS, int NB, bool ROT, int NCT, bool Q5, bool Q42 = false, int MINT = 0>     // Q42: two 16-element blocks per k-block, a scale each (Q4_2, and Q6_K in its form); MINT: bf16 pieces of the weight's min (2: Q5_1; 3: Q5_K / Q4_K)
__device__ __forceinline__
void gemm_q8_small16_body(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8, const float *__restrict__ ad,
                          float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbk, int nloc, int64_t ldd, const mm_epilogue &ep, int ntw,
                          uint32_t w_bytes, uint32_t a_bytes, int wg, const float *__restrict__ wm = nullptr,
                          const uint8_t *__restrict__ mp3 = nullptr, const uint8_t *__restrict__ sp3 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    constexpr int TN = 16 * NCT;                            // src1 columns of a workgroup
    int rt, ct;
    k3s_tile_of(wg, ntw, (N + TN - 1) / TN, rt, ct);
    const int m0 = rt * 16, n0 = ct * TN;
    const int kb0 = wave * nloc;
    // (a slot past the wave's range or past the end of K repeats a valid block: its table row is zero, so it adds (sumi * 0) = +0)
    auto blk = [&](int i) { const int kb = kb0 + (i < nloc ? i : nloc - 1); return kb < nbk ? kb : nbk - 1; };

    struct WB { long q; float d; float m; };               // (Q4_2: m = the scale of the k-block's second 16-element block)
    WB wb[NB];
    long ab[NB][NCT];
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(qs), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wd), 0, (int)(w_bytes / 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(a8), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Q42 ? wm : wd), 0, (int)(w_bytes / 8), 0x00020000);
    const int hpl = g4 >> 1, sub = g4 & 1;                  // plane and 8-byte piece of this lane's group
    const uint32_t offW = (uint32_t)((hpl * Mpad + m0 + l15) * 16 + 8 * sub), offD = (uint32_t)((m0 + l15) * 4);
    const uint32_t offA = (uint32_t)((hpl * Npad + n0 + l15) * 16 + 8 * sub);
    const uint32_t w_blk = (uint32_t)(2 * Mpad * 16), d_blk = (uint32_t)(Mpad * 4), a_blk = (uint32_t)(2 * Npad * 16);
    auto load_blk = [&](WB &f, long (&a)[NCT], int i) {
        const uint32_t kb = (uint32_t)blk(i);
#pragma unroll
        for (int c = 0; c < NCT; ++c)
            a[c] = __builtin_bit_cast(long, __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(offA + 256u * c), (int)(kb * a_blk), 0));
        f.q = __builtin_bit_cast(long, __builtin_amdgcn_raw_buffer_load_b64(rW, (int)offW, (int)(kb * w_blk), 0));
        f.d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)offD, (int)(kb * d_blk), 0));
        if constexpr (Q42) f.m = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, (int)offD, (int)(kb * d_blk), 0));
    };

    // (min-term types, r5) the min terms m * (s0 + s1) of sixteen k-blocks as ONE K = 16 product on the bf16 cores in front of the wave's blocks -- the 32-row
    // form's product, instruction for instruction (v_mfma_f32_32x32x16_bf16 on the same piece planes, chunk c = wave, wave + KS, .. in ascending order, five / six
    // piece products each, smallest first), so an element's sum has the same bits: the weight operand carries this workgroup's 16 rows in lanes 0..15 of each
    // half and zeros elsewhere, the activation operand the 32 columns at n0; an element of the product depends on its own row and column alone.  The result
    // (lane = weight row, 16 registers x 2 halves = 32 columns) is then handed to the 16 x 16 tiles' lanes, once per wave.
    using f32x16m = __attribute__((ext_vector_type(16))) float;
    using bf16x8m = __attribute__((ext_vector_type(8))) __bf16;
    constexpr int MP = MINT ? MINT : 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int nchunks = (nbk + 15) / 16;
    const __amdgpu_buffer_rsrc_t rMP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(MINT ? mp3 : qs), 0, MINT ? (int)((uint32_t)(nchunks * 6) * (uint32_t)Mpad * 16u) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rSP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(MINT ? sp3 : qs), 0, MINT ? (int)((uint32_t)(nchunks * 6) * (uint32_t)Npad * 16u) : 0, 0x00020000);
    const uint32_t vM = (uint32_t)((3 * hh) * Mpad + m0 + (l31 & 15)) * 16u, vS = (uint32_t)((3 * hh) * Npad + n0 + (TN == 16 ? (l31 & 15) : l31)) * 16u;
    const bool row_lane = l31 < 16, col_lane = TN == 32 || l31 < 16;
    auto chunk_load = [&](int c, i32x4 (&sa_)[3], i32x4 (&mb_)[MP]) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            if (pc < MINT) {
                const i32x4 v = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rMP, (int)vM, (int)((uint32_t)(6 * c + pc) * (uint32_t)Mpad * 16u), 0));
                mb_[pc] = row_lane ? v : i32x4{0, 0, 0, 0};
            }
            const i32x4 u = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rSP, (int)vS, (int)((uint32_t)(6 * c + pc) * (uint32_t)Npad * 16u), 0));
            sa_[pc] = col_lane ? u : i32x4{0, 0, 0, 0};
        }
    };
    i32x4 sa0[3], mb0[MP];
    if constexpr (MINT) { if (wave < nchunks) chunk_load(wave, sa0, mb0); }   // (in front of the blocks' loads: its wait is then for the chunk alone)

    // ---- this wave's slice of the row scales: rows x TN floats into its own LDS slice (rows past its range: zero) ----
    const int trows = (nloc + NB - 1) / NB * NB;
    float *const tabD = (float *)smem8 + (size_t)wave * trows * TN;
    constexpr int PPR = TN / 4;                             // float4 pieces per table row
    constexpr int TP = 8;                                   // pieces per lane and round: 64 * TP / PPR rows (>= 64)
    for (int base = 0; base < trows * PPR; base += 64 * TP) {
        f32x4 td[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = base + lane + 64 * j, b = idx / PPR, c4 = idx % PPR;
            const bool ok = b < nloc && kb0 + b < nbk;
            td[j] = ok ? *(const f32x4 *)(ad + (size_t)(kb0 + b) * Npad + n0 + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        if (base == 0) static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; load_blk(wb[u], ab[u], u); });
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = base + lane + 64 * j;
            if (idx < trows * PPR) *(f32x4 *)(tabD + 4 * idx) = td[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    float acc[NCT][4];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.0f;
    if constexpr (MINT) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest products first (activation piece x weight piece)
        if (wave < nchunks) {
            f32x16m a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int c = wave;;) {
                const int cn = c + KS;
                i32x4 sa1[3], mb1[MP];
                if (cn < nchunks) chunk_load(cn, sa1, mb1);
                static_for<6>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (PB[k] < MINT)
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8m, sa0[PA[k]]), __builtin_bit_cast(bf16x8m, mb0[PB[k]]), a, 0, 0, 0);
                });
                if (cn >= nchunks) break;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) { sa0[pc] = sa1[pc]; if (pc < MINT) mb0[pc] = mb1[pc]; }
                c = cn;
            }
            // D[column (r & 3) + 8 (r >> 2) + 4 hh][weight row l31] -> the 16 x 16 tiles' lanes (row l15, group g4): column 16 c + 4 g4 + e of slice c sits in
            // lane l15 + 32 (g4 & 1), register 4 (2 c + (g4 >> 1)) + e
            const int src = 4 * (l15 + 32 * (g4 & 1));
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (through float temporaries: __builtin_bit_cast applied to an element of an ext_vector reads element 0 whatever the index -- hipcc 7.2)
                    const float f0 = a[8 * c + e], f1 = a[8 * c + 4 + e];
                    const int v0 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, f0));
                    const int v1 = __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, f1));
                    acc[c][e] = __builtin_bit_cast(float, (g4 >> 1) ? v1 : v0);
                }
        }
    }
    const i32x4v zero = {0, 0, 0, 0};
    auto block = [&](int i, auto uc) {                      // block i of the wave out of slot u
        constexpr int u = decltype(uc)::value;
        const float dw = wb[u].d;
        const float *dp = tabD + i * TN + 4 * g4;
        if constexpr (Q42) {
            // Q4_2: the lane groups with piece 0 (bytes 0..7 of a plane) hold the k-block's first 16-element block, those with piece 1 its second
            // (Ggml.cs:1217-1252): one product per half with the weight operand zero in the other half's lanes -- sumi_0 and sumi_1 apart, each
            // under its own scale; the 32-row form's two statements in its order
            const long q0 = sub ? 0l : wb[u].q, q1 = sub ? wb[u].q : 0l;
            const float dm = wb[u].m;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const i32x4v t = __builtin_amdgcn_mfma_i32_16x16x32_i8(ab[u][c], q0, zero, 0, 0, 0);
                const i32x4v t2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(ab[u][c], q1, zero, 0, 0, 0);
                const f32x4 da = *(const f32x4 *)(dp + 16 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[c][e] = fmaf(dw * da[e], (float)t[e], acc[c][e]);
                    acc[c][e] = fmaf(dm * da[e], (float)t2[e], acc[c][e]);
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const i32x4v t = __builtin_amdgcn_mfma_i32_16x16x32_i8(ab[u][c], wb[u].q, zero, 0, 0, 0);
            const f32x4 da = *(const f32x4 *)(dp + 16 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if constexpr (Q5) acc[c][e] = fmaf(dw * (float)t[e], da[e], acc[c][e]);      // (d * sxy) * y.d, Ggml.cs:1296-1298
                else acc[c][e] = fmaf((float)t[e], da[e] * dw, acc[c][e]);                    // Ggml.cs:1377-1378
        }
    };
    static_for<NB>([&](auto uc) { block(decltype(uc)::value, uc); });
    if constexpr (ROT) {
        // longer K: rounds of NB blocks, a round's loads issued together behind the previous round's arithmetic (the 32-row form's structure).
        // (r5, measured: refilling a slot as soon as its own product has issued, blocks pinned in program order -- NB - 1 blocks in flight across
        // the whole range, what the MX form does -- was SLOWER here: Q8_0 4096 x 11008 x 16 13.5 -> 15.4 us, x 32 19.1 -> 19.4.)
        for (int base = NB; base < nloc; base += NB) {
            __builtin_amdgcn_sched_barrier(0);              // (hoisted above the arithmetic, the loads would double the live slots)
            static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; load_blk(wb[u], ab[u], base + u); });
            __builtin_amdgcn_sched_barrier(0);
            static_for<NB>([&](auto uc) { block(base + decltype(uc)::value, uc); });
        }
    }

    // ---- the waves' sums in wave order (the 32-row form's tree), the NCT * 4 result registers dealt over the waves ----
    __syncthreads();
    float *xch = (float *)smem8 + lane;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[(size_t)((wave * NCT + c) * 4 + r) * 64] = acc[c][r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < NCT * 4; ++rr) {
        if (rr % KS != wave) continue;                      // (uniform)
        const int c = rr / 4, r = rr % 4;
        float v = xch[(size_t)(c * 4 + r) * 64];
#pragma unroll
        for (int g = 1; g < KS; ++g) v += xch[(size_t)((g * NCT + c) * 4 + r) * 64];
        const int n = n0 + 16 * c + 4 * g4 + r, m = m0 + l15;
        if (n < N && m < M) {
            if (ep.mode == 3) {                             // the row split's exchange: the same position in this rank's and every peer's [N][M] buffer (r5: the batched-decode forms too)
                dst[(size_t)n * ldd + m] = v;
                for (int pk = 0; pk < ep.npush; ++pk) ep.push[pk][(size_t)n * ldd + m] = v;
            } else if (ep.mode == 2) {
                dst[(size_t)n * ldd + m] = v * ep.scale;
            } else {
                dst[(size_t)n * ldd + m] = v;
                if (ep.mode == 1) ep.dst2[(size_t)n * ep.ld2 + m] = v + ep.addend[(size_t)n * ep.ld_add + m];
            }
        }
    }
}

template <int KS, int NB, bool ROT, int NCT, int TY>         // TY: 0 Q8_0, 1 Q5_0, 2 Q5_1, 3 Q5_K / Q4_K in the Q5_1 form, 4 Q4_2 (the 32-row kernel's numbering)
__global__ __launch_bounds__(KS * 64, 1)
void gemm_q8_small16_kernel(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8, const float *__restrict__ ad,
                            float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbk, int nloc, int64_t ldd, const mm_epilogue ep, int ntw,
                            uint32_t w_bytes, uint32_t a_bytes, const float *__restrict__ wm, const uint8_t *__restrict__ mp3, const uint8_t *__restrict__ sp3) {
    gemm_q8_small16_body<KS, NB, ROT, NCT, TY == 1 || TY == 2 || TY == 3, TY == 4, (TY == 2 || TY == 3 ? TY : 0)>(qs, wd, a8, ad, dst, M, N, Mpad, Npad, nbk, nloc, ldd, ep, ntw, w_bytes,
                                                                                                      a_bytes, (int)blockIdx.x, wm, mp3, sp3);
}

// several Q8_0 matrices behind one activation image in one launch (gemm_qmx.hip gemm_qmx_small_multi_kernel has the story)
struct q8s_set {
    int n; int wg_end[4];
    const uint8_t *qs[4]; const float *d[4]; float *dst[4];
    int M[4], Mpad[4]; int64_t ldd[4]; uint32_t w_bytes[4];
};
template <int KS, int NB, bool ROT, int WMT>
__global__ __launch_bounds__(KS * 64, 1)
void gemm_q8_small_multi_kernel(const q8s_set ws, const int8_t *__restrict__ a8, const float *__restrict__ ad, int N, int Npad, int nbk, int nloc,
                                uint32_t a_bytes, int ncol) {
    const int b = (int)blockIdx.x;
    const int k = (b >= ws.wg_end[0]) + (b >= ws.wg_end[1]) + (b >= ws.wg_end[2]);
    const int first = k == 0 ? 0 : k == 1 ? ws.wg_end[0] : k == 2 ? ws.wg_end[1] : ws.wg_end[2];
#define Q8S(f) (k == 0 ? ws.f[0] : k == 1 ? ws.f[1] : k == 2 ? ws.f[2] : ws.f[3])
    const mm_epilogue ep{0, nullptr, 0, nullptr, 0, 1.0f};
    gemm_q8_small_body<KS, NB, ROT, WMT>(Q8S(qs), Q8S(d), a8, ad, Q8S(dst), Q8S(M), N, Q8S(Mpad), Npad, nbk, nloc, Q8S(ldd), ep, (Q8S(wg_end) - first) / ncol,
                                         Q8S(w_bytes), a_bytes, b - first);
#undef Q8S
}

hipError_t launch_q8_small_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st) {
    constexpr int KS = 8;
    const int nbkp = (int)pad_kblocks(w[0]->nbk);
    int nloc = (nbkp + KS - 1) / KS;
    nloc += nloc & 1;                                       // (plan.cpp k3p_i8_nloc: the single-matrix form's ranges -- a group computes its members' bits)
    const int ncol = (int)((N + 31) / 32);
    if (nloc > 128 || p.Npad < 32 * ncol) return hipErrorNotSupported;
    int64_t t32 = 0;
    for (int i = 0; i < n_w; ++i) t32 += (w[i]->M + 31) / 32 * ncol;
    // r5: three / four tiles per workgroup as well -- the fewest that keep the group inside one round of the chip (gate / up of a 7B model, 2 x 11008 rows: 230 workgroups of 96
    // rows; it was 344 of 64, a second round a third full).  A 96-row tile may overhang the padded rows: it reads the neighbouring plane there and stores nothing.
    auto groups = [&](int t) { int64_t g = 0; for (int i = 0; i < n_w; ++i) g += (w[i]->M + 32 * t - 1) / (32 * t) * ncol; return g; };
    const int wmt = t32 <= 256 ? 1 : t32 <= 512 ? 2 : groups(3) <= 256 ? 3 : 4;
    const uint64_t aq_bytes = (uint64_t)nbkp * 2 * (uint64_t)p.Npad * 16;
    if (aq_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
    q8s_set ws = {};
    ws.n = n_w;
    int wgs = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < n_w) {
            const ggml_hip_weight *x = w[i];
            const uint64_t wq_bytes = (uint64_t)nbkp * 2 * (uint64_t)x->Mpad * 16;
            if (x->type != GGML_TYPE_Q8_0 || !x->qs || !x->d || x->nbk != w[0]->nbk || x->Mpad % (wmt == 3 ? 32 : 32 * wmt) != 0 || wq_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
            wgs += (int)((x->M + 32 * wmt - 1) / (32 * wmt)) * ncol;
            ws.qs[i] = x->qs; ws.d[i] = x->d; ws.dst[i] = dst[i]; ws.M[i] = (int)x->M; ws.Mpad[i] = (int)x->Mpad; ws.ldd[i] = ldd[i]; ws.w_bytes[i] = (uint32_t)wq_bytes;
        }
        ws.wg_end[i] = wgs;
    }
    const int nb = wmt >= 3 ? 4 : wmt == 2 || nloc <= 8 ? 8 : 16, rows = (nloc + nb - 1) / nb * nb;   // (table rows: whole rounds of the form's slots)
    const int tab = KS * rows * 32 * 4, xch = KS * wmt * 16 * 64 * 4;
    const int lds = tab > xch ? tab : xch;
    dim3 grid((unsigned)wgs);
    (void)hipGetLastError();
#define Q8M_GO(NB, ROT, WMT) do { \
        auto kern = gemm_q8_small_multi_kernel<KS, NB, ROT, WMT>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid, KS * 64, lds, st>>>(ws, p.a8, p.ad, (int)N, (int)p.Npad, (int)w[0]->nbk, nloc, (uint32_t)aq_bytes, ncol); } while (0)
    if (wmt == 4) { if (nloc <= 4) Q8M_GO(4, false, 4); else Q8M_GO(4, true, 4); }
    else if (wmt == 3) { if (nloc <= 4) Q8M_GO(4, false, 3); else Q8M_GO(4, true, 3); }
    else if (wmt == 2) { if (nloc <= 8) Q8M_GO(8, false, 2); else Q8M_GO(8, true, 2); }
    else if (nloc <= 8) Q8M_GO(8, false, 1);
    else if (nloc <= 16) Q8M_GO(16, false, 1);
    else Q8M_GO(16, true, 1);
#undef Q8M_GO
    return hipGetLastError();
}

hipError_t launch_q8_small(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue &ep) {
    constexpr int KS = 8;
    // k-blocks per wave and tiles per workgroup (one up to 256 tile groups, two beyond: same bits) from plan.cpp (plan_k3s_i8)
    const int nbkp = (int)pad_kblocks(w->nbk);
    const int nloc = pl.nloc, wmt = pl.wmt;
    const int ncol = (int)((N + 31) / 32);
    // r4: Q5_0 too -- its int8 operand planes (ggml_hip_weight::i8p, built at upload for K3p) have Q8_0's layout; only the order of the two
    // scale multiplications differs.  (It ran the staged f16 forms here: 4096 x 4096 x 64 16.3 us against Q8_0's 7.8.)
    // ... and Q5_1 (9 .. 64 rows): Q5_0's block term on planes of the unsigned values + the min-term product in front of the blocks
    // (4096 x 4096 x 64 17.3 us on the staged forms, 4096 x 11008 x 32 36.4).
    // ... and Q4_2 (9 .. 256 rows -- it was 17 .. 64; plan.cpp q8_small_serves has the measurements): int8 planes of nib - 8 (built at upload, r4), two K = 16 MFMAs per block
    // (4096 x 4096 x 32 40.3 us on the staged int8 kernel).
    const bool q42 = w->type == GGML_TYPE_Q4_2;
    const bool q41 = w->type == GGML_TYPE_Q4_1;             // (r5: from 65 rows -- unsigned values 0..15 on the int8 planes, an f32 min: Q5_K's instantiation, as in K3p)
    const bool q51 = w->type == GGML_TYPE_Q5_1 || q41, q5 = w->type == GGML_TYPE_Q5_0 || q51;
    const uint8_t *planes = q5 || q42 ? w->i8p : w->qs;
    if (q42 && !w->m) return hipErrorInvalidValue;
    if (pl.family != MMF_K3S_I8 || !(q5 || q42 || w->type == GGML_TYPE_Q8_0) || !planes || !w->d || nloc > 128 || KS * nloc < nbkp || p.Npad < 32 * ncol) return hipErrorInvalidValue;
    if (q51 && (!w->mp3 || !p.sp3)) return hipErrorInvalidValue;
    const bool q5k = (q51 && w->ext_type != 0) || q41;      // (the Q5_K extension: activations by the Q8_K rule, three min pieces; Q4_1: three pieces too)
    if (w->Mpad % (32 * wmt) != 0) return hipErrorInvalidValue;
    const uint64_t wq_bytes = (uint64_t)nbkp * 2 * (uint64_t)w->Mpad * 16, aq_bytes = (uint64_t)nbkp * 2 * (uint64_t)p.Npad * 16;
    if (wq_bytes > 0xFFFFFFFFull || aq_bytes > 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (pl.tile_m == 16) {
        // r5: 16-row tiles (plan_k3s_i8: Q8_0 / Q5_0 where the 32-row tiles leave CUs idle) -- the same tree, NCT 16-column slices per workgroup
        if (!(w->type == GGML_TYPE_Q8_0 || q5 || q42)) return hipErrorInvalidValue;
        const int nct = pl.tile_n / 16;
        const int ncg = (int)((N + pl.tile_n - 1) / pl.tile_n);
        if (p.Npad < (int64_t)pl.tile_n * ncg || w->Mpad % 16 != 0) return hipErrorInvalidValue;
        const int ntw16 = (int)((w->M + 15) / 16);
        if (nct != 1 && nct != 2) return hipErrorInvalidValue;       // (four slices per workgroup were built and measured: slower than two 32-row workgroups, plan.cpp)
        const int nb16 = nloc <= 8 ? 8 : 16, rows16 = (nloc + nb16 - 1) / nb16 * nb16;
        const int tab16 = KS * rows16 * pl.tile_n * 4, xch16 = KS * nct * 4 * 64 * 4;
        const int lds16 = tab16 > xch16 ? tab16 : xch16;
        if (lds16 > 160 * 1024) return hipErrorInvalidValue;
        dim3 grid16((unsigned)(ntw16 * ncg));
        (void)hipGetLastError();
#define Q8S16_GO1(NB, ROT, NCT, TY) do { \
        auto kern = gemm_q8_small16_kernel<KS, NB, ROT, NCT, TY>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid16, KS * 64, lds16, st>>>(planes, w->d, p.a8, p.ad, dst, (int)w->M, (int)N, (int)w->Mpad, (int)p.Npad, (int)w->nbk, nloc, ldd, ep, ntw16, \
                                          (uint32_t)wq_bytes, (uint32_t)aq_bytes, w->m, w->mp3, p.sp3); } while (0)
#define Q8S16_GO(NB, ROT, NCT) do { if (q42) Q8S16_GO1(NB, ROT, NCT, 4); else if (q5k) Q8S16_GO1(NB, ROT, NCT, 3); else if (q51) Q8S16_GO1(NB, ROT, NCT, 2); \
                                    else if (q5) Q8S16_GO1(NB, ROT, NCT, 1); else Q8S16_GO1(NB, ROT, NCT, 0); } while (0)
        if (nct == 2) { if (nloc <= 8) Q8S16_GO(8, false, 2); else if (nloc <= 16) Q8S16_GO(16, false, 2); else Q8S16_GO(16, true, 2); }
        else { if (nloc <= 8) Q8S16_GO(8, false, 1); else if (nloc <= 16) Q8S16_GO(16, false, 1); else Q8S16_GO(16, true, 1); }
#undef Q8S16_GO
#undef Q8S16_GO1
        return hipGetLastError();
    }
    const int ntw = (int)((w->M + 32 * wmt - 1) / (32 * wmt));
    const int nb = wmt == 4 ? 4 : wmt == 2 || nloc <= 8 ? 8 : 16, rows = (nloc + nb - 1) / nb * nb;   // (table rows: whole rounds of the form's slots)
    const int tab = KS * rows * 32 * 4, xch = KS * wmt * 16 * 64 * 4;
    const int lds = tab > xch ? tab : xch;
    dim3 grid((unsigned)(ntw * ncol));
    (void)hipGetLastError();                                // (the value returned below is this launch's, not an earlier call's)
#define Q8S_GO1(NB, ROT, WMT, TY) do { \
        auto kern = gemm_q8_small_kernel<KS, NB, ROT, WMT, TY>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid, KS * 64, lds, st>>>(planes, w->d, p.a8, p.ad, dst, (int)w->M, (int)N, (int)w->Mpad, (int)p.Npad, (int)w->nbk, nloc, ldd, ep, ntw, \
                                      (uint32_t)wq_bytes, (uint32_t)aq_bytes, w->mp3, p.sp3, w->m); } while (0)
#define Q8S_GO(NB, ROT, WMT) do { if (q42) Q8S_GO1(NB, ROT, WMT, 4); else if (q5k) Q8S_GO1(NB, ROT, WMT, 3); else if (q51) Q8S_GO1(NB, ROT, WMT, 2); else if (q5) Q8S_GO1(NB, ROT, WMT, 1); else Q8S_GO1(NB, ROT, WMT, 0); } while (0)
    // one tile per workgroup: a wave's range in 8 / 16 slots, longer K in rounds of 16; two tiles (more than 256 tile groups): 8 slots,
    // in rounds beyond K = 2048
    if (wmt == 4) {                                         // (r5: four tiles per workgroup, more than 512 tile groups -- Q8_0 / Q5_0: four slots in turn)
        if (q42 || q51) return hipErrorInvalidValue;
        if (q5) { if (nloc <= 4) Q8S_GO1(4, false, 4, 1); else Q8S_GO1(4, true, 4, 1); }
        else { if (nloc <= 4) Q8S_GO1(4, false, 4, 0); else Q8S_GO1(4, true, 4, 0); }
    }
    else if (wmt == 2) { if (nloc <= 8) Q8S_GO(8, false, 2); else Q8S_GO(8, true, 2); }
    else if (nloc <= 8) Q8S_GO(8, false, 1);
    else if (nloc <= 16) Q8S_GO(16, false, 1);
    else Q8S_GO(16, true, 1);
#undef Q8S_GO
#undef Q8S_GO1
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_q8_small(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue *ep) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    const mm_epilogue none{0, nullptr, 0, nullptr, 0, 1.0f};
    return launch_q8_small(w, pl, p, N, dst, ldd, st, ep ? *ep : none);
}

hipError_t launch_gemm_q8_small_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st) {
    if (n_w < 2 || n_w > 4 || N < 5 || N > 64) return hipErrorNotSupported;
    for (int i = 0; i < n_w; ++i)
        if (!w[i] || w[i]->type != GGML_TYPE_Q8_0 || w[i]->M <= 0) return hipErrorNotSupported;
    return launch_q8_small_multi(w, n_w, p, N, dst, ldd, st);
}

