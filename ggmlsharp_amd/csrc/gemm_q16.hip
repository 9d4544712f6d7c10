// gemm_q16.hip -- K3: quantized mat-mat (N > GEMV_MAX_N), block-scaled, on the f16 matrix cores.
//
// COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698):
//   dst[n*ldd + m] = sum_b (dw[m,b] * da[n,b]) * sumi_b(m,n),   sumi_b = the integer dot of one 32-element block
// (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1136-1159; _q5_0_q8_0 1270-1298; _q8_0_q8_0 1362-1378; _q4_1_q8_1 1176-1198).
//
// Why f16 operands for an integer dot: every operand is a small integer (weights in [-128,127], Q8 activations in
// [-127,127]) and is exact in f16; every product (< 2^14) and every 32-term block sum (< 2^19) is exact in the f32
// accumulator of v_mfma_f32_16x16x32_f16.  So one MFMA (K = 32 = one quant block) returns sumi_b for a 16x16 tile
// bit-exactly, ALREADY IN F32.  The int8 MFMA returns int32, and the 16 v_cvt_f32_i32 per 32x32 tile that follow
// are half-rate VALU ops which, measured, made the scale-accumulate epilogue the bottleneck (profiles/).
// Here the epilogue per 16x16 tile is 4 multiplies (da*dw) + 4 fmas, exactly the f32 work of Ggml.cs:1158.
//
// Orientation: MFMA rows = src1 rows n (A = activations), MFMA cols = weight rows m (B = weights): a lane owns one
// m, so dst stores run along m (dst is [n][m], m fastest, Ggml.cs:6692-6697).
//
// Element order inside a block: lane group g = lane>>4 supplies k-slots 8g..8g+7 of A and of B, and the MFMA pairs
// slot with slot, so any fixed permutation of the 32 elements is fine as long as A and B use the same one.
// "Panel" p (= lane group g) holds the 8 elements e = p, p+4, ..., p+28 (e mod 4 == p): that is what the nibble
// unpack ((q >> 4p) & 0x000F000F) yields for free, and K1 writes the activations in the same order.
// LDS images are [k-block][panel][row][16 B]; a fragment is one conflict-free ds_read_b128.
//
// Workgroup = 256 threads = 2x2 waves, wave tile 64x64 = 4x4 MFMA tiles, 2 workgroups per CU.  K streams through
// a double-buffered LDS ring, BKB k-blocks per stage: activations by global_load_lds (f16 planes from K1),
// weights through registers (nibble/byte -> f16 with the 0x6400 magic: as_f16(0x6400 | q) = 1024 + q).
#include "common.h"
#include <cstdlib>
#include <utility>

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define G16_BKB 2   // k-blocks per stage
#define G16_T 128   // workgroup tile edge (n and m)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, 0);
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// two small unsigned integers packed in the low bits of each 16-bit half -> two f16 of (value - off)
__device__ __forceinline__ uint32_t u16pair_to_f16(uint32_t pair, float off_plus_1024) {
    const uint32_t bits = pair | 0x64006400u;          // as f16: 1024 + value (ulp is 1 in [1024, 2048))
    const f16x2 v = __builtin_bit_cast(f16x2, bits);
    const _Float16 o = (_Float16)off_plus_1024;
    const f16x2 r = v - (f16x2){o, o};                 // exact
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ uint32_t q5_hi16(uint32_t qh, int i, int p) {
    // bits of elements 8i+p and 8i+4+p -> bit 4 of the low / high 16-bit half
    const uint32_t t = qh >> (8 * i + p);
    return ((t & 1u) << 4) | ((t & 0x10u) << 16);
}

template <int TYPE>
struct Tile16 {
    static constexpr int A_BYTES = G16_BKB * 4 * G16_T * 16;
    static constexpr int W_BYTES = G16_BKB * 4 * G16_T * 16;
    static constexpr int SC_BYTES = G16_BKB * G16_T * 4;   // one f32 plane
    static constexpr int NSC = (TYPE == GGML_TYPE_Q4_1) ? 4 : 2;  // da, dw (+ mw, sa)
    static constexpr int STAGE = A_BYTES + W_BYTES + NSC * SC_BYTES;
    static constexpr int LDS = 2 * STAGE;
};

template <int TYPE>
__global__ __launch_bounds__(256, 2) void gemm_q16_kernel(const uint8_t *__restrict__ wqs, const uint32_t *__restrict__ wqh,
                                                         const float *__restrict__ wd, const float *__restrict__ wm,
                                                         const uint8_t *__restrict__ a16, const float *__restrict__ ad,
                                                         const int32_t *__restrict__ as, float *__restrict__ dst,
                                                         int64_t M, int64_t N, int64_t Mpad, int64_t Npad, int64_t nbk,
                                                         int64_t ldd, int dbg) {
    using T = Tile16<TYPE>;
    constexpr int TN = G16_T, TM = G16_T, BKB = G16_BKB;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wm_ = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int64_t n0 = (int64_t)blockIdx.y * TN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nstages = (int)((nbk + BKB - 1) / BKB);
    auto stage_ptr = [&](int s) { return smem + (s & 1) * T::STAGE; };

    // ---- global -> registers / LDS-DMA for one stage ----
    constexpr int W_CHUNKS = BKB * TM;            // 16-byte weight chunks per stage (Q8_0: two per row-block)
    constexpr int WPT = (TYPE == GGML_TYPE_Q8_0 ? 2 : 1) * W_CHUNKS / 256;
    uint4 wreg[WPT];
    uint32_t hreg[WPT];
    constexpr int SPT = BKB * TN / 256;           // scale values per thread per plane
    float dareg[SPT], dwreg[SPT], mwreg[SPT], sareg[SPT];

    auto issue_loads = [&](int s) {
        uint8_t *sA = stage_ptr(s);
        const int64_t kb0 = (int64_t)s * BKB;
        constexpr int A_CHUNKS = BKB * 4 * TN;
#pragma unroll
        for (int i = 0; i < A_CHUNKS / 256; ++i) {
            const int c = tid + 256 * i;
            const int bp = c / TN, row = c % TN;   // bp = bb * 4 + panel
            int64_t b = kb0 + (bp >> 2);
            if (b >= nbk) b = nbk - 1;             // tail stage: harmless re-read, its scales are zero
            const uint8_t *gp = a16 + (((b * 4 + (bp & 3)) * Npad) + n0 + row) * 16;
            glds16(gp, sA + (size_t)(c - lane) * 16);
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int c = tid + 256 * i;
            if (TYPE == GGML_TYPE_Q8_0) {
                const int bh = c / TM, row = c % TM;   // bh = bb * 2 + plane (even / odd elements)
                int64_t b = kb0 + (bh >> 1);
                if (b >= nbk) b = nbk - 1;
                wreg[i] = *(const uint4 *)(wqs + (((b * 2 + (bh & 1)) * Mpad) + m0 + row) * 16);
            } else {
                const int bb = c / TM, row = c % TM;
                int64_t b = kb0 + bb;
                if (b >= nbk) b = nbk - 1;
                wreg[i] = *(const uint4 *)(wqs + ((b * Mpad) + m0 + row) * 16);
                if (TYPE == GGML_TYPE_Q5_0) hreg[i] = wqh[b * Mpad + m0 + row];
            }
        }
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int i = tid + 256 * k;
            const int bb = i / TN, row = i % TN;
            const int64_t b = kb0 + bb;
            const bool ok = b < nbk;
            dareg[k] = ok ? ad[b * Npad + n0 + row] : 0.0f;
            dwreg[k] = ok ? wd[b * Mpad + m0 + row] : 0.0f;
            if (TYPE == GGML_TYPE_Q4_1) {
                sareg[k] = ok ? (float)as[b * Npad + n0 + row] : 0.0f;
                mwreg[k] = ok ? wm[b * Mpad + m0 + row] : 0.0f;
            }
        }
    };

    // ---- registers -> LDS (weights converted to f16 panels, scales) ----
    auto store_stage = [&](int s) {
        uint8_t *sp = stage_ptr(s);
        uint8_t *sW = sp + T::A_BYTES;
        float *sDa = (float *)(sp + T::A_BYTES + T::W_BYTES);
        float *sDw = sDa + BKB * TN;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int c = tid + 256 * i;
            const uint32_t q[4] = {wreg[i].x, wreg[i].y, wreg[i].z, wreg[i].w};
            if (TYPE == GGML_TYPE_Q8_0) {
                // plane h byte j = element 2j + h (signed).  word k = elements 8k+h, 8k+2+h, 8k+4+h, 8k+6+h.
                // panel p = h + 2*odd_byte: bytes (0,2) of every word -> panel h, bytes (1,3) -> panel h + 2.
                const int bh = c / TM, row = c % TM, bb = bh >> 1, h = bh & 1;
                uint32_t pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t x = q[k] ^ 0x80808080u;             // signed byte -> biased unsigned
                    pa[k] = u16pair_to_f16(x & 0x00FF00FFu, 1024.0f + 128.0f);                       // bytes 0, 2
                    pb[k] = u16pair_to_f16((x >> 8) & 0x00FF00FFu, 1024.0f + 128.0f);                // bytes 1, 3
                }
                *(uint4 *)(sW + ((size_t)((bb * 4 + h) * TM + row)) * 16) = make_uint4(pa[0], pa[1], pa[2], pa[3]);
                *(uint4 *)(sW + ((size_t)((bb * 4 + h + 2) * TM + row)) * 16) = make_uint4(pb[0], pb[1], pb[2], pb[3]);
            } else {
                const int bb = c / TM, row = c % TM;
                constexpr float OFF = TYPE == GGML_TYPE_Q4_0 ? 8.0f : (TYPE == GGML_TYPE_Q5_0 ? 16.0f : 0.0f);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    uint32_t w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t pair = (q[k] >> (4 * p)) & 0x000F000Fu;   // elements 8k+p, 8k+4+p (Ggml.cs:1149-1150)
                        if (TYPE == GGML_TYPE_Q5_0) pair |= q5_hi16(hreg[i], k, p);   // Ggml.cs:1285-1289
                        w[k] = u16pair_to_f16(pair, 1024.0f + OFF);
                    }
                    *(uint4 *)(sW + ((size_t)((bb * 4 + p) * TM + row)) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int i = tid + 256 * k;
            sDa[i] = dareg[k];
            sDw[i] = dwreg[k];
            if (TYPE == GGML_TYPE_Q4_1) {
                sDw[BKB * TM + i] = mwreg[k];
                sDw[2 * BKB * TM + i] = sareg[k];
            }
        }
    };

    // ---- one stage of MFMAs + block-scale epilogues ----
    auto compute = [&](int s) {
        const uint8_t *sp = stage_ptr(s);
        const uint8_t *sA = sp;
        const uint8_t *sW = sp + T::A_BYTES;
        const float *sDa = (const float *)(sp + T::A_BYTES + T::W_BYTES);
        const float *sDw = sDa + BKB * TN;
        const float *sMw = sDw + BKB * TM;
        const float *sSa = sMw + BKB * TM;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

        // Software pipeline over the stage's NT = BKB*16 tiles (order: k-block, i, j): the MFMA of tile t+DEPTH is
        // issued before the scale-accumulate of tile t, so the matrix core's latency (and the next k-block's
        // fragment reads, issued at the first tile of the current block) hide behind the VALU work.
        constexpr int NT = BKB * 16, DEPTH = 2, RING = DEPTH + 1;
        f16x8 af[2][4], bf[2][4];
        f32x4 da[2][4], sa[2][4];
        float dw[2][4], mw[2][4];
        f32x4 tacc[RING];

        auto load_block = [&](auto bc) {
            constexpr int bb = decltype(bc)::value, p = bb & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + 16 * i;
                af[p][i] = *(const f16x8 *)(sA + ((size_t)((bb * 4 + g) * TN + row + l15)) * 16);
                da[p][i] = *(const f32x4 *)(sDa + bb * TN + row + 4 * g);
                if (TYPE == GGML_TYPE_Q4_1) sa[p][i] = *(const f32x4 *)(sSa + bb * TN + row + 4 * g);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = wm_ * 64 + 16 * j + l15;
                bf[p][j] = *(const f16x8 *)(sW + ((size_t)((bb * 4 + g) * TM + col)) * 16);
                dw[p][j] = sDw[bb * TM + col];
                if (TYPE == GGML_TYPE_Q4_1) mw[p][j] = sMw[bb * TM + col];
            }
        };
        auto mfma_tile = [&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / 16, i = (t % 16) / 4, j = t % 4;
            tacc[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[bb & 1][i], bf[bb & 1][j], zero, 0, 0, 0);
        };

        load_block(std::integral_constant<int, 0>{});
        static_for<DEPTH>([&](auto tc) { mfma_tile(tc); });
        static_for<NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / 16, i = (t % 16) / 4, j = t % 4, p = bb & 1;
            if constexpr (t % 16 == 0 && bb + 1 < BKB) load_block(std::integral_constant<int, bb + 1>{});
            if constexpr (t + DEPTH < NT) mfma_tile(std::integral_constant<int, t + DEPTH>{});
            const f32x4 sc = da[p][i] * dw[p][j];                        // d1 * d0, Ggml.cs:1158
            acc[i][j] = __builtin_elementwise_fma(tacc[t % RING], sc, acc[i][j]);
            if (TYPE == GGML_TYPE_Q4_1)                                  // + m0 * d1 * sum(a)  (Ggml.cs:1190-1196 factorised)
                acc[i][j] = __builtin_elementwise_fma((f32x4){mw[p][j], mw[p][j], mw[p][j], mw[p][j]}, da[p][i] * sa[p][i], acc[i][j]);
            // pin: keeps the optimiser from sinking the scale-accumulates below the stage's last MFMA
            asm volatile("" : "+v"(acc[i][j]));
        });
    };

    // ---- main loop: double-buffered, one barrier per stage ----
    issue_loads(0);
    store_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const bool more = (s + 1) < nstages && !(dbg & 1);   // dbg bit 0: timing ablation, no staging after stage 0
        if (more) issue_loads(s + 1);
        if (!(dbg & 2)) compute(s);                          // dbg bit 1: timing ablation, staging only
        if (more) store_stage(s + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- dst[n][m]: D[row = 4*(lane>>4) + r][col = lane & 15] ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + wm_ * 64 + 16 * j + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t n = n0 + wn * 64 + 16 * i + 4 * g + r;
                if (n < N && m < M) dst[n * ldd + m] = acc[i][j][r];
            }
        }
}

template <int TYPE>
hipError_t launch_typed(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    using T = Tile16<TYPE>;
    static bool attr_set = false;
    auto kern = gemm_q16_kernel<TYPE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((unsigned)((w->M + G16_T - 1) / G16_T), (unsigned)((N + G16_T - 1) / G16_T));
    static const int dbg = [] { const char *e = getenv("GGML_HIP_GEMM_DBG"); return e ? atoi(e) : 0; }();  // developer ablations
    kern<<<grid, 256, T::LDS, st>>>(w->qs, w->qh, w->d, w->m, (const uint8_t *)p.a8, p.ad, p.as, dst, w->M, N, w->Mpad,
                                    p.Npad, w->nbk, ldd, dbg);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_q16(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0>(w, p, N, dst, ldd, st);
    default: return hipErrorInvalidValue;
    }
}
