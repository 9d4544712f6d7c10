// gemm_q.hip -- K3: quantized mat-mat (N > GEMV_MAX_N) on the int8 matrix cores, block-scaled.
//
// COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698):
//   dst[n*ldd + m] = sum_b (dw[m,b] * da[n,b]) * sumi_b(m,n),   sumi_b = exact int32 dot of one 32-element block
// (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1136-1159; _q5_0_q8_0 1270-1298; _q8_0_q8_0 1362-1378; _q4_1_q8_1 1176-1198;
// _q5_1_q8_1 1318-1344: unsigned 5-bit values + the min term; _q4_2_q8_0 1216-1252: a k-block is two 16-element blocks
// with their own scales, so its tile takes two v_mfma_i32_32x32x16_i8 -- bytes 0..7 of both operand halves are elements
// 0..15, bytes 8..15 elements 16..31 -- and two scale-accumulates).
// The reference's K-block of 32 is exactly one v_mfma_i32_32x32x32_i8 step, so each MFMA yields the 32x32 tile of
// sumi_b for one k-block; the f32 scale-accumulate of Ggml.cs:1158 is then applied per tile on the VALU:
//   acc[r] = fma((float)sumi[r], da[row(r)] * dw[col], acc[r]).
// Arithmetic is the reference's (integer block dot, two f32 scales per block); only the order of the f32 adds over
// blocks is the same here too (ascending b), with fma instead of mul+add.
//
// Orientation: MFMA rows = src1 rows n (A operand = Q8 activations), MFMA cols = weight rows m (B operand), so a
// lane owns one m and the 16 accumulator registers walk n: the final stores are 128-byte segments of dst rows
// (dst is [n][m] with m fastest, Ggml.cs:6692-6697).
//
// Operand images in LDS are [k-block][half h][row][16 B]: lane (row, h) reads its 16 int8 with one conflict-free
// ds_read_b128.  Half h = 0 holds the even elements of the block, h = 1 the odd ones -- the order the nibble
// unpack (q & 0x0F0F0F0F, (q >> 4) & 0x0F0F0F0F) produces for free; K1 writes the activations in the same order,
// and the MFMA pairs element j of half h in A with element j of half h in B, so the block dot is unchanged.
//
// Workgroup = 256 threads = 2x2 waves, wave tile (32*IT) x (32*JT), 2 workgroups per CU (2 waves per SIMD so the
// VALU epilogue issues at full rate).  K is streamed in stages of BKB k-blocks through a double-buffered LDS ring:
// activations by global_load_lds (16 B/lane, image is lane-linear), weights through registers (nibble -> int8).
#include "common.h"
#include <cstdlib>
#include <utility>

namespace {

using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

#define BKB 4  // k-blocks per stage

// compile-time loop: f(std::integral_constant<int, 0>{}), f(<1>), ... -- indices are constants inside f, so register
// arrays indexed with them never become runtime-indexed (which would push them to scratch)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ void glds16(const void *g, void *l) {
    __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, 0);
}

// nibble word (elements interleaved) -> int8 words of (value - off), bytewise, no cross-byte borrow
__device__ __forceinline__ uint32_t sub_bytes(uint32_t x, uint32_t off4) {
    return ((x | 0x80808080u) - off4) ^ 0x80808080u;
}

__device__ __forceinline__ uint32_t q5_hi(uint32_t qh, int i, int sel) {
    const uint32_t t = ((qh >> (8 * i + sel)) & 0x55u);
    return (t * 0x00410410u) & 0x10101010u;
}

template <int TYPE> struct Traits {
    static constexpr bool HAS_M = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1 || TYPE == GGML_TYPE_Q4_2;   // second weight plane
    static constexpr bool MIN = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1;                              // ... that is a min
    static constexpr bool HAS_H = TYPE == GGML_TYPE_Q5_0 || TYPE == GGML_TYPE_Q5_1;
};

template <int TYPE, int IT, int JT>
struct Tile {
    static constexpr int TN = 64 * IT;  // src1 rows per workgroup
    static constexpr int TM = 64 * JT;  // weight rows per workgroup
    static constexpr int A_BYTES = BKB * 2 * TN * 16;
    static constexpr int W_BYTES = BKB * 2 * TM * 16;
    static constexpr int DA_BYTES = BKB * TN * 4;
    static constexpr int DW_BYTES = BKB * TM * 4;
    static constexpr int MW_BYTES = Traits<TYPE>::HAS_M ? BKB * TM * 4 : 0;
    static constexpr int SA_BYTES = Traits<TYPE>::MIN ? BKB * TN * 4 : 0;
    static constexpr int STAGE = A_BYTES + W_BYTES + DA_BYTES + DW_BYTES + MW_BYTES + SA_BYTES;
    static constexpr int LDS = 2 * STAGE;
};

template <int TYPE, int IT, int JT>
__global__ __launch_bounds__(256, 2) void gemm_q_kernel(const uint8_t *__restrict__ wqs, const uint32_t *__restrict__ wqh,
                                                       const float *__restrict__ wd, const float *__restrict__ wm,
                                                       const int8_t *__restrict__ a8, const float *__restrict__ ad,
                                                       const int32_t *__restrict__ as, float *__restrict__ dst, int64_t M,
                                                       int64_t N, int64_t Mpad, int64_t Npad, int64_t nbk, int64_t ldd,
                                                       int tiles_m, int tiles_n) {
    using T = Tile<TYPE, IT, JT>;
    constexpr int TN = T::TN, TM = T::TM;
    constexpr bool HAS_M = Traits<TYPE>::HAS_M, MIN = Traits<TYPE>::MIN, HAS_H = Traits<TYPE>::HAS_H;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wm_ = wave & 1;
    // XCD-aware tile order (speed only): workgroups b, b+8, b+16, ... share an XCD and its L2; each XCD gets a
    // contiguous run of the tile list ordered "m fastest", so its resident workgroups share activation panels.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    int tm_i, tn_i;
    if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0) {
        // 2 x 4 blocks of the tile grid per XCD (gemm_qmx.hip): its resident workgroups share half of the weight panels and a
        // quarter of the activation panels
        const int hm = tiles_m >> 1, l = bid >> 3;
        tm_i = (xcd & 1) * hm + l % hm;
        tn_i = (xcd >> 1) * (tiles_n >> 2) + l / hm;
    } else {
        const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        tm_i = t_lin % tiles_m;
        tn_i = t_lin / tiles_m;
    }
    const int64_t m0 = (int64_t)tm_i * TM;
    const int64_t n0 = (int64_t)tn_i * TN;

    f32x4 acc[IT][JT][4];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][j][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nstages = (int)((nbk + BKB - 1) / BKB);

    // ---- staging helpers ----
    constexpr int W_CHUNKS = BKB * TM;                               // 16-byte nibble chunks (Q4/Q5) per stage
    constexpr int W_PER_THREAD = (W_CHUNKS + 255) / 256;
    uint4 wreg[W_PER_THREAD];
    uint32_t hreg[W_PER_THREAD];

    auto stage_ptr = [&](int s) { return smem + (s & 1) * T::STAGE; };

    // Every global address = uniform 64-bit base advancing by a constant per stage + per-thread 32-bit offset fixed
    // for the whole kernel: no 64-bit or integer-multiply VALU work (both quarter-rate) inside the loop.
    constexpr int A_PER_THREAD = BKB * 2 * TN / 256;
    constexpr int W8_PER_THREAD = BKB * 2 * TM / 256;                   // Q8_0: int8 planes go straight to LDS
    constexpr int SA_PER_THREAD = (BKB * TN + 255) / 256;
    constexpr int SW_PER_THREAD = (BKB * TM + 255) / 256;
    const uint32_t a_blk = (uint32_t)(2 * Npad * 16);
    const uint32_t w_blk = (uint32_t)(Mpad * (TYPE == GGML_TYPE_Q8_0 ? 32 : 16));
    uint32_t offA[A_PER_THREAD], offW[TYPE == GGML_TYPE_Q8_0 ? W8_PER_THREAD : W_PER_THREAD], offH[W_PER_THREAD];
    uint32_t offDa[SA_PER_THREAD], offDw[SW_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int c = tid + 256 * i, bh = c / TN, row = c % TN;          // bh = bb * 2 + half
        offA[i] = (uint32_t)(bh >> 1) * a_blk + (uint32_t)(((bh & 1) * Npad + n0 + row) * 16);
    }
    if (TYPE == GGML_TYPE_Q8_0) {
#pragma unroll
        for (int i = 0; i < W8_PER_THREAD; ++i) {
            const int c = tid + 256 * i, bh = c / TM, row = c % TM;
            offW[i] = (uint32_t)(bh >> 1) * w_blk + (uint32_t)(((bh & 1) * Mpad + m0 + row) * 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < W_PER_THREAD; ++i) {
            const int c = tid + 256 * i, bb = c / TM, row = c % TM;
            offW[i] = (uint32_t)bb * w_blk + (uint32_t)((m0 + row) * 16);
            offH[i] = (uint32_t)((bb * Mpad + m0 + row) * 4);
        }
    }
#pragma unroll
    for (int k = 0; k < SA_PER_THREAD; ++k) {
        const int i = tid + 256 * k, bb = i / TN, row = i % TN;
        offDa[k] = (uint32_t)((bb * Npad + n0 + row) * 4);
    }
#pragma unroll
    for (int k = 0; k < SW_PER_THREAD; ++k) {
        const int i = tid + 256 * k, bb = i / TM, row = i % TM;
        offDw[k] = (uint32_t)((bb * Mpad + m0 + row) * 4);
    }

    auto issue_loads = [&](int s) {
        uint8_t *sp = stage_ptr(s);
        const int64_t kb0 = (int64_t)s * BKB;
        const int64_t left = nbk - kb0;                                  // k-blocks this stage really has (uniform)
        const uint8_t *gA = (const uint8_t *)a8 + kb0 * a_blk;
        const uint8_t *gW = wqs + kb0 * w_blk;
        // activations: [bb][h][TN rows][16 B], one 1-KiB wave instruction per 64 rows
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) {
            const int c = tid + 256 * i, bb = (c / TN) >> 1;
            // a k-block past the end re-reads block 0 of the stage (harmless: its scales are zeroed below)
            const uint32_t off = offA[i] - (bb >= left ? (uint32_t)bb * a_blk : 0u);
            glds16(gA + off, sp + (size_t)(c - lane) * 16);              // wave-uniform LDS base, hardware adds lane*16
        }
        if (TYPE == GGML_TYPE_Q8_0) {
            uint8_t *sW = sp + T::A_BYTES;
#pragma unroll
            for (int i = 0; i < W8_PER_THREAD; ++i) {
                const int c = tid + 256 * i, bb = (c / TM) >> 1;
                const uint32_t off = offW[i] - (bb >= left ? (uint32_t)bb * w_blk : 0u);
                glds16(gW + off, sW + (size_t)(c - lane) * 16);
            }
        } else {
#pragma unroll
            for (int i = 0; i < W_PER_THREAD; ++i) {
                const int c = tid + 256 * i, bb = c / TM;
                const bool dead = bb >= left;
                wreg[i] = *(const uint4 *)(gW + (offW[i] - (dead ? (uint32_t)bb * w_blk : 0u)));
                if (HAS_H)
                    hreg[i] = *(const uint32_t *)((const uint8_t *)(wqh + kb0 * Mpad) + (offH[i] - (dead ? (uint32_t)(bb * Mpad * 4) : 0u)));
            }
        }
    };

    // scales: loaded to registers with the stage's other loads, written to LDS after the compute phase
    // (an LDS store right behind the load would stall on vmcnt(0) and drain the in-flight LDS-DMA).
    // A block index past nbk gets scale 0, which turns the whole tail block into +0.
    float dareg[SA_PER_THREAD], sareg[SA_PER_THREAD], dwreg[SW_PER_THREAD], mwreg[SW_PER_THREAD];
    auto load_scales = [&](int s) {
        const int64_t kb0 = (int64_t)s * BKB;
        const int64_t left = nbk - kb0;
        const uint8_t *gDa = (const uint8_t *)(ad + kb0 * Npad), *gSa = (const uint8_t *)(as + kb0 * Npad);
        const uint8_t *gDw = (const uint8_t *)(wd + kb0 * Mpad), *gMw = (const uint8_t *)(wm + kb0 * Mpad);
#pragma unroll
        for (int k = 0; k < SA_PER_THREAD; ++k) {
            const int i = tid + 256 * k, bb = i / TN;
            const bool ok = (i < BKB * TN) && (bb < left);
            const uint32_t off = offDa[k] - (ok ? 0u : (uint32_t)(bb * Npad * 4));   // dead block: re-read block 0 of the stage
            const float v = *(const float *)(gDa + off);
            dareg[k] = ok ? v : 0.0f;
            if (MIN) { const int sv = *(const int32_t *)(gSa + off); sareg[k] = ok ? (float)sv : 0.0f; }
        }
#pragma unroll
        for (int k = 0; k < SW_PER_THREAD; ++k) {
            const int i = tid + 256 * k, bb = i / TM;
            const bool ok = (i < BKB * TM) && (bb < left);
            const uint32_t off = offDw[k] - (ok ? 0u : (uint32_t)(bb * Mpad * 4));
            const float v = *(const float *)(gDw + off);
            dwreg[k] = ok ? v : 0.0f;
            if (HAS_M) { const float v2 = *(const float *)(gMw + off); mwreg[k] = ok ? v2 : 0.0f; }
        }
    };
    auto store_scales = [&](int s) {
        uint8_t *sp = stage_ptr(s);
        float *sDa = (float *)(sp + T::A_BYTES + T::W_BYTES);
        float *sDw = (float *)(sp + T::A_BYTES + T::W_BYTES + T::DA_BYTES);
        float *sMw = (float *)(sp + T::A_BYTES + T::W_BYTES + T::DA_BYTES + T::DW_BYTES);
        float *sSa = (float *)(sp + T::A_BYTES + T::W_BYTES + T::DA_BYTES + T::DW_BYTES + T::MW_BYTES);
#pragma unroll
        for (int k = 0; k < SA_PER_THREAD; ++k) {
            const int i = tid + 256 * k;
            if (i < BKB * TN) {
                sDa[i] = dareg[k];
                if (MIN) sSa[i] = sareg[k];
            }
        }
#pragma unroll
        for (int k = 0; k < SW_PER_THREAD; ++k) {
            const int i = tid + 256 * k;
            if (i < BKB * TM) {
                sDw[i] = dwreg[k];
                if (HAS_M) sMw[i] = mwreg[k];
            }
        }
    };

    auto store_weights = [&](int s) {
        if (TYPE == GGML_TYPE_Q8_0) return;
        uint8_t *sW = stage_ptr(s) + T::A_BYTES;
#pragma unroll
        for (int i = 0; i < W_PER_THREAD; ++i) {
            const int c = tid + 256 * i;
            const int bb = c / TM, row = c % TM;
            const uint32_t q[4] = {wreg[i].x, wreg[i].y, wreg[i].z, wreg[i].w};
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo[k] = q[k] & 0x0F0F0F0Fu;         // elements 8k+0,2,4,6
                hi[k] = (q[k] >> 4) & 0x0F0F0F0Fu;  // elements 8k+1,3,5,7
                if (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q4_2) {       // (nib - 8), Ggml.cs:1149-1150 / 1231-1235
                    lo[k] = sub_bytes(lo[k], 0x08080808u);
                    hi[k] = sub_bytes(hi[k], 0x08080808u);
                } else if (TYPE == GGML_TYPE_Q5_1) {  // (nib | bit << 4), unsigned 0..31, Ggml.cs:1330-1334
                    lo[k] |= q5_hi(hreg[i], k, 0);
                    hi[k] |= q5_hi(hreg[i], k, 1);
                } else if (TYPE == GGML_TYPE_Q5_0) {  // ((nib | bit << 4) - 16), Ggml.cs:1285-1289
                    lo[k] = sub_bytes(lo[k] | q5_hi(hreg[i], k, 0), 0x10101010u);
                    hi[k] = sub_bytes(hi[k] | q5_hi(hreg[i], k, 1), 0x10101010u);
                }  // Q4_1: unsigned nibbles 0..15 as they are (Ggml.cs:1190-1191)
            }
            *(uint4 *)(sW + ((size_t)((bb * 2 + 0) * TM + row)) * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            *(uint4 *)(sW + ((size_t)((bb * 2 + 1) * TM + row)) * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        }
    };

    // One stage = BKB k-blocks x (IT x JT) MFMA tiles per wave, fully unrolled and software-pipelined by hand:
    // while the VALU applies the block scales to tile t (cvt + mul + fma per element), the matrix core already
    // runs tile t+1 into the other accumulator, and the LDS reads for the next k-block / next row group are in
    // flight.  sched_barrier(0) between tiles keeps the compiler from hoisting all MFMAs (and their 16-register
    // results) to the top, which is what it does otherwise and what makes the 2x2 wave tile spill.
    auto compute = [&](int s) {
        const uint8_t *sp = stage_ptr(s);
        const uint8_t *sA = sp;
        const uint8_t *sW = sp + T::A_BYTES;
        const float *sDa = (const float *)(sp + T::A_BYTES + T::W_BYTES);
        const float *sDw = (const float *)(sp + T::A_BYTES + T::W_BYTES + T::DA_BYTES);
        const float *sMw = (const float *)(sp + T::A_BYTES + T::W_BYTES + T::DA_BYTES + T::DW_BYTES);
        const float *sSa = (const float *)(sp + T::A_BYTES + T::W_BYTES + T::DA_BYTES + T::DW_BYTES + T::MW_BYTES);
        constexpr int NT = BKB * IT * JT;   // tiles per stage, order: bb slowest, then i, then j
        constexpr int NG = BKB * IT;        // (bb, i) groups: one set of 16 row scales each

        // fragment ring: a k-block's operands are fetched AHEAD tiles before its first MFMA issues
        constexpr int TPB = IT * JT;                 // tiles per k-block
        constexpr int AHEAD = TPB == 1 ? 2 : 1;      // k-blocks of look-ahead for the fragment loads
        constexpr int RING = AHEAD + 1;
        i32x4 af[RING][IT], bf[RING][JT];
        float dw[RING][JT], mw[RING][JT];
        f32x4 da[2][4], sa[2][4];
        i32x16 tacc[2], tacc2[TYPE == GGML_TYPE_Q4_2 ? 2 : 1];   // Q4_2: the second 16-element block's sums
        const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

        auto load_block = [&](auto bbc) {   // operand fragments + per-lane weight scales of k-block bb
            constexpr int bb = decltype(bbc)::value, p = bb % RING;
#pragma unroll
            for (int i = 0; i < IT; ++i)
                af[p][i] = *(const i32x4 *)(sA + ((size_t)((bb * 2 + hh) * TN + wn * 32 * IT + 32 * i + l31)) * 16);
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int col = wm_ * 32 * JT + 32 * j + l31;
                bf[p][j] = *(const i32x4 *)(sW + ((size_t)((bb * 2 + hh) * TM + col)) * 16);
                dw[p][j] = sDw[bb * TM + col];
                if (HAS_M) mw[p][j] = sMw[bb * TM + col];
            }
        };
        auto load_group = [&](auto gc) {    // the 16 per-register row scales of group g = bb * IT + i
            constexpr int g = decltype(gc)::value, bb = g / IT, i = g % IT, p = g & 1;
            const int row = wn * 32 * IT + 32 * i;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                da[p][k] = *(const f32x4 *)(sDa + bb * TN + row + 8 * k + 4 * hh);
                if (MIN) sa[p][k] = *(const f32x4 *)(sSa + bb * TN + row + 8 * k + 4 * hh);
            }
        };
        auto mfma_tile = [&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / (IT * JT), i = (t / JT) % IT, j = t % JT;
            if constexpr (TYPE == GGML_TYPE_Q4_2) {
                const i32x4 a = af[bb % RING][i], b = bf[bb % RING][j];
                const long a0 = (long)(((uint64_t)(uint32_t)a[1] << 32) | (uint32_t)a[0]), a1 = (long)(((uint64_t)(uint32_t)a[3] << 32) | (uint32_t)a[2]);
                const long b0 = (long)(((uint64_t)(uint32_t)b[1] << 32) | (uint32_t)b[0]), b1 = (long)(((uint64_t)(uint32_t)b[3] << 32) | (uint32_t)b[2]);
                tacc[t & 1] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a0, b0, zero, 0, 0, 0);
                tacc2[t & 1] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a1, b1, zero, 0, 0, 0);
            } else {
                tacc[t & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[bb % RING][i], bf[bb % RING][j], zero, 0, 0, 0);
            }
        };

        static_for<AHEAD>([&](auto bc) { load_block(bc); });
        load_group(std::integral_constant<int, 0>{});
        mfma_tile(std::integral_constant<int, 0>{});
        static_for<NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int bb = t / (IT * JT), i = (t / JT) % IT, j = t % JT, g = t / JT;
            if constexpr (t + 1 < NT) mfma_tile(std::integral_constant<int, t + 1>{});
            // prefetch: a later k-block's fragments at the first tile of this block, next group's row scales at the
            // first tile of this group (the ring slots they overwrite were last read by an earlier tile's MFMA)
            if constexpr (i == 0 && j == 0 && bb + AHEAD < BKB) load_block(std::integral_constant<int, bb + AHEAD>{});
            if constexpr (j == 0 && g + 1 < NG) load_group(std::integral_constant<int, g + 1>{});
            const i32x16 tt = tacc[t & 1];
            const float dwj = dw[bb % RING][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = da[g & 1][r >> 2][r & 3] * dwj;
                acc[i][j][r >> 2][r & 3] = fmaf((float)tt[r], sc, acc[i][j][r >> 2][r & 3]);
            }
            if constexpr (TYPE == GGML_TYPE_Q4_2) {   // sumf += (d1 * yd) * sumi_1 (Ggml.cs:1250): the second block's scale rides in the m plane
                const i32x16 t2 = tacc2[t & 1];
                const float dw2 = mw[bb % RING][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sc = da[g & 1][r >> 2][r & 3] * dw2;
                    acc[i][j][r >> 2][r & 3] = fmaf((float)t2[r], sc, acc[i][j][r >> 2][r & 3]);
                }
            }
            if (MIN) {  // + m0 * d1 * sum(a)  (Ggml.cs:1190-1196 factorised; Q5_1: m * (s0 + s1), Ggml.cs:1344)
                const float mwj = mw[bb % RING][j];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][k][e] = fmaf(mwj, da[g & 1][k][e] * sa[g & 1][k][e], acc[i][j][k][e]);
            }
            // pin: the optimiser otherwise sinks every tile's scale-accumulate below the last MFMA of the stage
            // (the sums are only needed at the end), which serialises matrix core and VALU and blows the registers
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(acc[i][j][k]));
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- main loop: double-buffered, one barrier per stage ----
    issue_loads(0);
    load_scales(0);
    store_weights(0);
    store_scales(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA of stage 0 landed (this wave's part)
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const bool more = (s + 1) < nstages;
        if (more) {
            issue_loads(s + 1);
            load_scales(s + 1);
        }
        compute(s);
        if (more) {
            store_weights(s + 1);
            store_scales(s + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: dst[n][m], lanes along m ----
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            const int64_t m = m0 + wm_ * 32 * JT + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t n = n0 + wn * 32 * IT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (n < N && m < M) dst[n * ldd + m] = acc[i][j][r >> 2][r & 3];
            }
        }
}

template <int TYPE, int IT, int JT>
hipError_t launch_cfg(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    using T = Tile<TYPE, IT, JT>;
    auto kern = gemm_q_kernel<TYPE, IT, JT>;
        static PerDeviceOnce once;   // per kernel instantiation; the attribute is set once per device
    const hipError_t attr = once.max_dynamic_lds((const void *)kern, T::LDS);
    if (attr != hipSuccess) return attr;
    const int tiles_m = (int)((w->M + T::TM - 1) / T::TM), tiles_n = (int)((N + T::TN - 1) / T::TN);
    dim3 grid((unsigned)(tiles_m * tiles_n));
    kern<<<grid, 256, T::LDS, st>>>(w->qs, w->qh, w->d, w->m, p.a8, p.ad, p.as, dst, w->M, N, w->Mpad, p.Npad, w->nbk, ldd,
                                    tiles_m, tiles_n);
    return hipGetLastError();
}

template <int TYPE>
hipError_t launch_typed(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    // 128x128 tiles when they already give >= 2 workgroups per CU, else 64x64 to fill the chip
    const int64_t big = ((w->M + 127) / 128) * ((N + 127) / 128);
    static const char *force = dev_env_str("GGML_HIP_GEMM_TILE");  // developer override: "1" = 64x64, "2" = 128x128
    if (force && force[0] == '1') return launch_cfg<TYPE, 1, 1>(w, p, N, dst, ldd, st);
    if (force && force[0] == '2') return launch_cfg<TYPE, 2, 2>(w, p, N, dst, ldd, st);
    if (big >= 512) return launch_cfg<TYPE, 2, 2>(w, p, N, dst, ldd, st);
    return launch_cfg<TYPE, 1, 1>(w, p, N, dst, ldd, st);
}

// ---- K3s for Q8_0: batches of 5 .. 64 rows, K >= 2048 (the stage-free form of gemm_qmx.hip on the int8 matrix cores) --------------------
// Q8_0's resident planes ARE int8 MFMA operands -- [k-block][half][row][16 B], the halves holding the even and the odd elements, the way K1
// writes the activations (image 0) -- so a k-block is one 16-byte load per operand and lane and one v_mfma_i32_32x32x32_i8, no digits and
// nothing to expand: 36 B per 32 weights through the CU's memory path where the bf6 planes of Q4_0 carry 28 and its activation image 48.
// Structure as K3s: a workgroup = WMT 32-row weight tiles x one 32-column slice of src1 x KS waves with a contiguous range of k-blocks each;
// a wave requests its first NB blocks before anything else (K <= 4096: all of them; longer K: round after round), keeps its
// slice of the row scales in its own LDS slice; the waves' sums are added in wave order, every wave taking its share of the rows.
// Arithmetic per block as in the kernel above: acc = fma((float)sumi, d1 * d0, acc) (Ggml.cs:1377-1378).
template <int KS, int NB, bool ROT, int WMT>
__device__ __forceinline__
void gemm_q8_small_body(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8, const float *__restrict__ ad,
                        float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbkp, int nloc, int64_t ldd, const mm_epilogue &ep, int ntw,
                        uint32_t w_bytes, uint32_t a_bytes, int wg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int m0 = (wg % ntw) * 32 * WMT, n0 = (wg / ntw) * 32;
    const int kb0 = wave * nloc;
    // a slot past the wave's range (or past the end of K) repeats a valid block: its table row is zero, so it adds (sumi * 0) = +0
    auto blk = [&](int i) { const int kb = kb0 + (i < nloc ? i : nloc - 1); return kb < nbkp ? kb : nbkp - 1; };

    struct WB { i32x4 q[WMT]; float d[WMT]; };
    WB wb[NB];
    i32x4 ab[NB];
    // raw buffer addressing: one 32-bit offset per lane and plane, the k-block in the scalar offset (planes past their end read 0)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(qs), 0, (int)w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wd), 0, (int)(w_bytes / 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t *>(a8), 0, (int)a_bytes, 0x00020000);
    const uint32_t offW = (uint32_t)((hh * Mpad + m0 + l31) * 16), offD = (uint32_t)((m0 + l31) * 4), offA = (uint32_t)((hh * Npad + n0 + l31) * 16);
    const uint32_t w_blk = (uint32_t)(2 * Mpad * 16), d_blk = (uint32_t)(Mpad * 4), a_blk = (uint32_t)(2 * Npad * 16);
    auto load_blk = [&](WB &f, i32x4 &a, int i) {
        const uint32_t kb = (uint32_t)blk(i);
        a = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)offA, (int)(kb * a_blk), 0));
#pragma unroll
        for (int t = 0; t < WMT; ++t) {
            f.q[t] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(offW + 512u * t), (int)(kb * w_blk), 0));
            f.d[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)(offD + 128u * t), (int)(kb * d_blk), 0));
        }
    };

    // ---- this wave's slice of the row scales: rows x 32 floats, straight into its own LDS slice (rows past its range: zero) ----
    const int trows = (nloc + NB - 1) / NB * NB;            // whole rounds of NB slots: rows past the wave's range are zero
    float *const tabD = (float *)smem8 + (size_t)wave * trows * 32;
    constexpr int TP = 8;                                   // float4 pieces per lane (nloc <= 64)
    f32x4 td[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int idx = lane + 64 * j, b = idx >> 3, c4 = idx & 7;
        const bool ok = b < nloc && kb0 + b < nbkp;
        td[j] = ok ? *(const f32x4 *)(ad + (size_t)(kb0 + b) * Npad + n0 + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    static_for<NB>([&](auto uc) { constexpr int u = decltype(uc)::value; load_blk(wb[u], ab[u], u); });
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int idx = lane + 64 * j;
        if (idx < trows * 8) *(f32x4 *)(tabD + 4 * idx) = td[j];
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

    auto block = [&](int i, auto uc) {                      // block i of the wave out of slot u
        constexpr int u = decltype(uc)::value;
        WB &w = wb[u];
        i32x16 t[WMT];
        float dw[WMT];
#pragma unroll
        for (int k = 0; k < WMT; ++k) { t[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ab[u], w.q[k], zero, 0, 0, 0); dw[k] = w.d[k]; }
        const float *dp = tabD + i * 32 + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 da = *(const f32x4 *)(dp + 8 * q);
#pragma unroll
            for (int k = 0; k < WMT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[k][4 * q + e] = fmaf((float)t[k][4 * q + e], da[e] * dw[k], acc[k][4 * q + e]);
        }
    };
    if constexpr (!ROT) {
        static_for<NB>([&](auto uc) { block(decltype(uc)::value, uc); });
    } else {
        // longer K: rounds of NB blocks, a round's loads issued together behind the previous round's last MFMA (a slot refilled as soon
        // as its own MFMA had issued -- the form of gemm_qmx.hip -- spilled here: 256 registers + 200 B of scratch)
        static_for<NB>([&](auto uc) { block(decltype(uc)::value, uc); });
        for (int base = NB; base < nloc; base += NB) {
            __builtin_amdgcn_sched_barrier(0);              // (a round's loads stay behind the previous round's arithmetic: hoisted, they double the live slots)
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
            if (ep.mode == 2) {
                dst[(size_t)n * ldd + m] = v * ep.scale;
            } else {
                dst[(size_t)n * ldd + m] = v;
                if (ep.mode == 1) ep.dst2[(size_t)n * ep.ld2 + m] = v + ep.addend[(size_t)n * ep.ld_add + m];
            }
        }
    }
}

template <int KS, int NB, bool ROT, int WMT>
__global__ __launch_bounds__(KS * 64, 1)
void gemm_q8_small_kernel(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8, const float *__restrict__ ad,
                          float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbkp, int nloc, int64_t ldd, const mm_epilogue ep, int ntw,
                          uint32_t w_bytes, uint32_t a_bytes) {
    gemm_q8_small_body<KS, NB, ROT, WMT>(qs, wd, a8, ad, dst, M, N, Mpad, Npad, nbkp, nloc, ldd, ep, ntw, w_bytes, a_bytes, (int)blockIdx.x);
}

// several Q8_0 matrices behind one activation image in one launch (gemm_qmx.hip gemm_qmx_small_multi_kernel has the story)
struct q8s_set {
    int n; int wg_end[4];
    const uint8_t *qs[4]; const float *d[4]; float *dst[4];
    int M[4], Mpad[4]; int64_t ldd[4]; uint32_t w_bytes[4];
};
template <int KS, int NB, bool ROT, int WMT>
__global__ __launch_bounds__(KS * 64, 1)
void gemm_q8_small_multi_kernel(const q8s_set ws, const int8_t *__restrict__ a8, const float *__restrict__ ad, int N, int Npad, int nbkp, int nloc,
                                uint32_t a_bytes, int ncol) {
    const int b = (int)blockIdx.x;
    const int k = (b >= ws.wg_end[0]) + (b >= ws.wg_end[1]) + (b >= ws.wg_end[2]);
    const int first = k == 0 ? 0 : k == 1 ? ws.wg_end[0] : k == 2 ? ws.wg_end[1] : ws.wg_end[2];
#define Q8S(f) (k == 0 ? ws.f[0] : k == 1 ? ws.f[1] : k == 2 ? ws.f[2] : ws.f[3])
    const mm_epilogue ep{0, nullptr, 0, nullptr, 0, 1.0f};
    gemm_q8_small_body<KS, NB, ROT, WMT>(Q8S(qs), Q8S(d), a8, ad, Q8S(dst), Q8S(M), N, Q8S(Mpad), Npad, nbkp, nloc, Q8S(ldd), ep, (Q8S(wg_end) - first) / ncol,
                                         Q8S(w_bytes), a_bytes, b - first);
#undef Q8S
}

hipError_t launch_q8_small_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st) {
    constexpr int KS = 8;
    const int nbkp = (int)pad_kblocks(w[0]->nbk);
    const int nloc = (nbkp + KS - 1) / KS;
    const int ncol = (int)((N + 31) / 32);
    if (nloc > 16 || p.Npad < 32 * ncol) return hipErrorNotSupported;
    int64_t t32 = 0;
    for (int i = 0; i < n_w; ++i) t32 += (w[i]->M + 31) / 32 * ncol;
    const int wmt = t32 <= 256 || nloc > 8 ? 1 : 2;
    const uint64_t aq_bytes = (uint64_t)nbkp * 2 * (uint64_t)p.Npad * 16;
    if (aq_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
    q8s_set ws = {};
    ws.n = n_w;
    int wgs = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < n_w) {
            const ggml_hip_weight *x = w[i];
            const uint64_t wq_bytes = (uint64_t)nbkp * 2 * (uint64_t)x->Mpad * 16;
            if (x->type != GGML_TYPE_Q8_0 || !x->qs || !x->d || x->nbk != w[0]->nbk || x->Mpad % (32 * wmt) != 0 || wq_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
            wgs += (int)((x->M + 32 * wmt - 1) / (32 * wmt)) * ncol;
            ws.qs[i] = x->qs; ws.d[i] = x->d; ws.dst[i] = dst[i]; ws.M[i] = (int)x->M; ws.Mpad[i] = (int)x->Mpad; ws.ldd[i] = ldd[i]; ws.w_bytes[i] = (uint32_t)wq_bytes;
        }
        ws.wg_end[i] = wgs;
    }
    const int nb = wmt == 2 || nloc <= 8 ? 8 : 16, rows = (nloc + nb - 1) / nb * nb;
    const int tab = KS * rows * 32 * 4, xch = KS * wmt * 16 * 64 * 4;
    const int lds = tab > xch ? tab : xch;
    dim3 grid((unsigned)wgs);
#define Q8M_GO(NB, WMT) do { \
        auto kern = gemm_q8_small_multi_kernel<KS, NB, false, WMT>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid, KS * 64, lds, st>>>(ws, p.a8, p.ad, (int)N, (int)p.Npad, nbkp, nloc, (uint32_t)aq_bytes, ncol); } while (0)
    if (nloc <= 8) { if (wmt == 2) Q8M_GO(8, 2); else Q8M_GO(8, 1); }
    else Q8M_GO(16, 1);
#undef Q8M_GO
    return hipGetLastError();
}

hipError_t launch_q8_small(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue &ep) {
    constexpr int KS = 8;
    const int nbkp = (int)pad_kblocks(w->nbk);
    const int nloc = (nbkp + KS - 1) / KS;
    const int ncol = (int)((N + 31) / 32);
    if (w->type != GGML_TYPE_Q8_0 || !w->qs || !w->d || nloc > 64 || p.Npad < 32 * ncol) return hipErrorNotSupported;
    const int64_t t32 = (w->M + 31) / 32 * ncol;
    if (nloc > 16) return hipErrorNotSupported;
    const int wmt = t32 <= 256 || nloc > 8 ? 1 : 2;
    if (w->Mpad % (32 * wmt) != 0) return hipErrorNotSupported;
    const uint64_t wq_bytes = (uint64_t)nbkp * 2 * (uint64_t)w->Mpad * 16, aq_bytes = (uint64_t)nbkp * 2 * (uint64_t)p.Npad * 16;
    if (wq_bytes > 0xFFFFFFFFull || aq_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
    const int ntw = (int)((w->M + 32 * wmt - 1) / (32 * wmt));
    const int nb = wmt == 2 || nloc <= 8 ? 8 : 16, rows = (nloc + nb - 1) / nb * nb;
    const int tab = KS * rows * 32 * 4, xch = KS * wmt * 16 * 64 * 4;
    const int lds = tab > xch ? tab : xch;
    dim3 grid((unsigned)(ntw * ncol));
#define Q8S_GO(NB, ROT, WMT) do { \
        auto kern = gemm_q8_small_kernel<KS, NB, ROT, WMT>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid, KS * 64, lds, st>>>(w->qs, w->d, p.a8, p.ad, dst, (int)w->M, (int)N, (int)w->Mpad, (int)p.Npad, nbkp, nloc, ldd, ep, ntw, \
                                      (uint32_t)wq_bytes, (uint32_t)aq_bytes); } while (0)
    // (K <= 4096 only: the looped forms for longer K spill -- 256 registers + 0.5..1.4 KB of scratch whatever the refill order; such
    // shapes stay on the staged f16 form, see api.cpp act_image_kind)
    if (nloc <= 8) { if (wmt == 2) Q8S_GO(8, false, 2); else Q8S_GO(8, false, 1); }
    else Q8S_GO(16, false, 1);
#undef Q8S_GO
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_q8_small(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue *ep) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    const mm_epilogue none{0, nullptr, 0, nullptr, 0, 1.0f};
    return launch_q8_small(w, p, N, dst, ldd, st, ep ? *ep : none);
}

hipError_t launch_gemm_q8_small_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st) {
    if (n_w < 2 || n_w > 4 || N < 5 || N > 64) return hipErrorNotSupported;
    for (int i = 0; i < n_w; ++i)
        if (!w[i] || w[i]->type != GGML_TYPE_Q8_0 || w[i]->M <= 0) return hipErrorNotSupported;
    return launch_q8_small_multi(w, n_w, p, N, dst, ldd, st);
}

hipError_t launch_gemm_q(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_2: return launch_typed<GGML_TYPE_Q4_2>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_1: return launch_typed<GGML_TYPE_Q5_1>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0>(w, p, N, dst, ldd, st);
    default: return hipErrorInvalidValue;
    }
}
