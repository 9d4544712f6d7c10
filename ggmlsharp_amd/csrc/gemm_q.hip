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
#include "plan.h"
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

// r4: a pseudo-type for weights that live in the planar Q4_2 form ON INT8 PLANES (the Q6_K extension, kquants.hip): Q8_0's operand path
// (the planes go to LDS as they are) with Q4_2's arithmetic (two 16-element sub-blocks per k-block, the second's scale in the m plane)
constexpr int GQ_TYPE_I8X2 = 100;
template <int TYPE> constexpr bool W_I8 = TYPE == GGML_TYPE_Q8_0 || TYPE == GQ_TYPE_I8X2;      // weights: int8 planes, no unpacking
template <int TYPE> constexpr bool TWO_SC = TYPE == GGML_TYPE_Q4_2 || TYPE == GQ_TYPE_I8X2;    // two sub-block sums and scales per k-block

template <int TYPE> struct Traits {
    static constexpr bool HAS_M = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1 || TWO_SC<TYPE>;   // second weight plane
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
    const uint32_t w_blk = (uint32_t)(Mpad * (W_I8<TYPE> ? 32 : 16));
    uint32_t offA[A_PER_THREAD], offW[W_I8<TYPE> ? W8_PER_THREAD : W_PER_THREAD], offH[W_PER_THREAD];
    uint32_t offDa[SA_PER_THREAD], offDw[SW_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int c = tid + 256 * i, bh = c / TN, row = c % TN;          // bh = bb * 2 + half
        offA[i] = (uint32_t)(bh >> 1) * a_blk + (uint32_t)(((bh & 1) * Npad + n0 + row) * 16);
    }
    if (W_I8<TYPE>) {
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
        if (W_I8<TYPE>) {
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
        if (W_I8<TYPE>) return;
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
        i32x16 tacc[2], tacc2[TWO_SC<TYPE> ? 2 : 1];   // Q4_2: the second 16-element block's sums
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
            if constexpr (TWO_SC<TYPE>) {
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
            if constexpr (TWO_SC<TYPE>) {   // sumf += (d1 * yd) * sumi_1 (Ggml.cs:1250): the second block's scale rides in the m plane
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
    // r4 (see gemm_qmx.hip): on one-round grids the younger wave of a SIMD (the second workgroup on the CU) takes priority on two stages of three
#ifndef GQ_PRIO
#define GQ_PRIO 3
#endif
    bool younger = false;
    if constexpr (GQ_PRIO != 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        younger = (hwid & 1u) != 0 && (int64_t)tiles_m * tiles_n <= 512;
    }
    for (int s = 0; s < nstages; ++s) {
        if constexpr (GQ_PRIO != 0) { if (younger) { if (s % GQ_PRIO != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } }
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
    kern<<<grid, 256, T::LDS, st>>>(TYPE == GQ_TYPE_I8X2 ? w->i8p : w->qs, w->qh, w->d, w->m, p.a8, p.ad, p.as, dst, w->M, N, w->Mpad, p.Npad, w->nbk, ldd,
                                    tiles_m, tiles_n);
    return hipGetLastError();
}

// 128 x 128 tiles when they already give >= 2 workgroups per CU, else 64 x 64 to fill the chip (plan.cpp plan_i8; one chain over K either way)
template <int TYPE>
hipError_t launch_typed(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (pl.family != MMF_I8) return hipErrorInvalidValue;
    if (pl.form == I8F_128x128) return launch_cfg<TYPE, 2, 2>(w, p, N, dst, ldd, st);
    return launch_cfg<TYPE, 1, 1>(w, p, N, dst, ldd, st);
}

}  // namespace

hipError_t launch_gemm_q(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_2:
        if (w->ext_type != 0) return w->i8p ? launch_typed<GQ_TYPE_I8X2>(w, pl, p, N, dst, ldd, st) : hipErrorInvalidValue;   // (Q6_K: the form's values live on the int8 planes only)
        return launch_typed<GGML_TYPE_Q4_2>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_1: return launch_typed<GGML_TYPE_Q5_1>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0>(w, pl, p, N, dst, ldd, st);
    default: return hipErrorInvalidValue;
    }
}
