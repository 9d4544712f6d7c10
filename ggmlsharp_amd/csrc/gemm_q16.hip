// gemm_q16.hip -- K3: quantized mat-mat for large N, block-scaled, on the f16 matrix cores.
//
// COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698):
//   dst[n*ldd + m] = sum_b (dw[m,b] * da[n,b]) * sumi_b(m,n),   sumi_b = the integer dot of one 32-element block
// (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1136-1159; _q5_0_q8_0 1270-1298; _q8_0_q8_0 1362-1378; _q4_1_q8_1 1176-1198).
//
// Why f16 operands for an integer dot: every operand is a small integer (weights in [-128,127], Q8 activations in
// [-127,127]) and is exact in f16; every product (< 2^14) and every 32-term block sum (< 2^19) is exact in the f32
// accumulator of v_mfma_f32_32x32x16_f16.  Two MFMAs (K = 2 x 16 = one quant block) return sumi_b for a 32x32 tile
// bit-exactly and ALREADY IN F32.  The int8 MFMA (gemm_q.hip) returns int32, and its 16 v_cvt_f32_i32 per tile are
// half-rate VALU ops: measured (tools/valu_ubench.hip) they are ~45 % of that kernel's VALU time, and
// the VALU scale-accumulate -- the reference's own f32 work per block, Ggml.cs:1158 -- is what bounds this path.
//
// Orientation: MFMA rows = src1 rows n (A = activations), MFMA cols = weight rows m (B = weights): a lane owns one
// m, so dst stores are 128-byte segments along m (dst is [n][m], m fastest, Ggml.cs:6692-6697).
//
// Element order inside a block: MFMA kk (0/1) lane half h supplies k-slots 8h..8h+7, and the MFMA pairs slot with
// slot, so any fixed permutation of the 32 elements is fine as long as A and B use the same one.  "Panel"
// p = 2*kk + h holds the 8 elements e = p, p+4, ..., p+28 -- what the nibble unpack ((q >> 4p) & 0x000F000F)
// yields for free; K1 writes the activations in the same order.  LDS images are [k-block][panel][row][16 B]; a
// fragment is one conflict-free ds_read_b128.
//
// Structure: workgroup = 512 threads = 8 waves as 4(n) x 2(m), wave tile 32(n) x 64(m), workgroup tile 128 x 128,
// two workgroups per CU = 4 waves per SIMD (<= 128 VGPRs): the VALU scale-accumulate, which bounds this kernel,
// always has several waves to issue from, and one workgroup's barrier / staging bubbles are filled by the other.
// K streams in stages of 2 k-blocks through a double-buffered LDS ring: activations by global_load_lds (f16
// planes from K1), weights through registers (nibble/byte -> f16: as_f16(0x6400 | q) = 1024 + q).
// Row scales: lane 16q+k keeps da of MFMA row R(k, q>>1) in ONE register and every product da*dw is a
// v_mul_f32_dpp row_share:k, instead of 16 broadcast ds_read_b128 per tile row -- LDS traffic is the second
// resource this kernel is short of.
#include "common.h"
#include <cstdlib>
#include <utility>

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TN = 128, TM = 128, BKB = 2;

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

// two small unsigned integers in the low bits of each 16-bit half -> two f16 of (value - off)
__device__ __forceinline__ uint32_t u16pair_to_f16(uint32_t pair, float off_plus_1024) {
    const uint32_t bits = pair | 0x64006400u;          // as f16: 1024 + value (ulp is 1 in [1024, 2048))
    const f16x2 v = __builtin_bit_cast(f16x2, bits);
    const _Float16 o = (_Float16)off_plus_1024;
    const f16x2 r = v - (f16x2){o, o};                 // exact
    return __builtin_bit_cast(uint32_t, r);
}

__device__ __forceinline__ uint32_t q5_hi16(uint32_t qh, int i, int p) {
    const uint32_t t = qh >> (8 * i + p);              // bits of elements 8i+p (-> bit 4) and 8i+4+p (-> bit 20)
    return ((t & 1u) << 4) | ((t & 0x10u) << 16);
}

// sc[r] = da_row[r] * dw for the 16 accumulator registers of a 32x32 tile: lane 16q+k of `vda` holds the scale of
// MFMA row R(k, q>>1), and row_share:r hands every lane the value of lane r of its own 16-lane row.
__device__ __forceinline__ void scale16(float (&sc)[16], float vda, float dw) {
    asm("s_nop 1\n\t"
        "v_mul_f32_dpp %0, %16, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %1, %16, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %2, %16, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %3, %16, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %4, %16, %17 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %5, %16, %17 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %6, %16, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %7, %16, %17 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %8, %16, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %9, %16, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %10, %16, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %11, %16, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %12, %16, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %13, %16, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %14, %16, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %15, %16, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "=&v"(sc[0]), "=&v"(sc[1]), "=&v"(sc[2]), "=&v"(sc[3]), "=&v"(sc[4]), "=&v"(sc[5]), "=&v"(sc[6]), "=&v"(sc[7]),
          "=&v"(sc[8]), "=&v"(sc[9]), "=&v"(sc[10]), "=&v"(sc[11]), "=&v"(sc[12]), "=&v"(sc[13]), "=&v"(sc[14]), "=&v"(sc[15])
        : "v"(vda), "v"(dw));
}

template <int TYPE>
struct Lds {
    static constexpr int A_BYTES = BKB * 4 * TN * 16;      // 16 KB f16 activation image
    static constexpr int W_BYTES = BKB * 4 * TM * 16;      // 16 KB f16 weight image
    static constexpr int SC_BYTES = BKB * TN * 4;          // one f32 plane (TN == TM)
    static constexpr int NSC = (TYPE == GGML_TYPE_Q4_1) ? 4 : 2;  // da, dw (+ mw, sa)
    static constexpr int STAGE = A_BYTES + W_BYTES + NSC * SC_BYTES;
    static constexpr int TOTAL = 2 * STAGE;
};

template <int TYPE>
__global__ __launch_bounds__(512, 4) void gemm_q16_kernel(const uint8_t *__restrict__ wqs, const uint32_t *__restrict__ wqh,
                                                         const float *__restrict__ wd, const float *__restrict__ wm,
                                                         const uint8_t *__restrict__ a16, const float *__restrict__ ad,
                                                         const int32_t *__restrict__ as, float *__restrict__ dst, int64_t M,
                                                         int64_t N, int64_t Mpad, int64_t Npad, int64_t nbk, int64_t ldd,
                                                         int tiles_m, int tiles_n, int dbg) {
    using L = Lds<TYPE>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wm_ = wave & 1;

    // XCD-aware tile order (speed only): workgroups b, b+8, b+16, ... share an XCD and its L2.  Give each XCD a
    // contiguous run of the tile list ordered "m fastest", so concurrently resident workgroups of one XCD share their
    // activation panel (same n tile) and walk neighbouring weight panels.
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t m0 = (int64_t)(t_lin % tiles_m) * TM;
    const int64_t n0 = (int64_t)(t_lin / tiles_m) * TN;
    const int nstages = (int)((nbk + BKB - 1) / BKB);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    auto stage_ptr = [&](int s) { return smem + (s & 1) * L::STAGE; };

    // ---- global -> LDS-DMA (activations) / registers (weights, scales) for one stage ----
    // Every address is "uniform 64-bit base that advances by a constant per stage" + "per-thread 32-bit offset fixed
    // for the whole kernel", so the loop has no 64-bit or integer-multiply VALU work (both are quarter-rate).
    // thread t expands half of weight chunk t>>1: panels 2*(t&1), 2*(t&1)+1 (Q8_0: chunk = (k-block, plane, row))
    uint4 wreg;
    uint32_t hreg = 0;
    float screg, sc2reg = 0.0f;

    const uint32_t a_blk = (uint32_t)(4 * Npad * 16);                 // bytes of one k-block of the f16 image
    const uint32_t w_blk = (uint32_t)(Mpad * (TYPE == GGML_TYPE_Q8_0 ? 32 : 16));
    uint32_t offA[2], offW, offS, offH;
    int bbA[2], bbW, bbS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                     // 1024 16-byte chunks: [bb][p][128 rows]
        const int c = tid + 512 * i, bp = c >> 7, row = c & 127;      // bp = bb * 4 + panel
        bbA[i] = bp >> 2;
        offA[i] = (uint32_t)bbA[i] * a_blk + (uint32_t)(((bp & 3) * Npad + n0 + row) * 16);
    }
    if (TYPE == GGML_TYPE_Q8_0) {
        const int bh = tid >> 7, row = tid & 127;                     // bh = bb * 2 + plane: 512 chunks, one per thread
        bbW = bh >> 1;
        offW = (uint32_t)bbW * w_blk + (uint32_t)(((bh & 1) * Mpad + m0 + row) * 16);
        offH = 0;
    } else {
        const int c = tid >> 1, row = c & 127;
        bbW = c >> 7;
        offW = (uint32_t)bbW * w_blk + (uint32_t)((m0 + row) * 16);
        offH = (uint32_t)((bbW * Mpad + m0 + row) * 4);
    }
    {
        const int i = tid & 255, row = i & 127;
        bbS = i >> 7;
        offS = (uint32_t)((bbS * (tid < 256 ? Npad : Mpad) + (tid < 256 ? n0 : m0) + row) * 4);
    }
    const bool odd_tail = (nbk & 1) != 0;                             // last stage holds one real k-block

    auto issue_loads = [&](int s) {
        uint8_t *sA = stage_ptr(s);
        const int64_t kb0 = (int64_t)s * BKB;
        const bool tail = odd_tail && s == nstages - 1;               // uniform
        const uint8_t *gA = a16 + kb0 * a_blk;
        const uint8_t *gW = wqs + kb0 * w_blk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // tail stage: the missing k-block re-reads the previous one (harmless, its weight scales are zero)
            const uint32_t off = offA[i] - ((tail && bbA[i]) ? a_blk : 0u);
            glds16(gA + off, sA + (size_t)(tid + 512 * i - lane) * 16);
        }
        wreg = *(const uint4 *)(gW + (offW - ((tail && bbW) ? w_blk : 0u)));
        if (TYPE == GGML_TYPE_Q5_0) hreg = *(const uint32_t *)((const uint8_t *)(wqh + kb0 * Mpad) + (offH - ((tail && bbW) ? (uint32_t)(Mpad * 4) : 0u)));
        {   // scales: threads 0..255 -> da[bb][row], 256..511 -> dw[bb][row]
            const bool dead = tail && bbS;                            // this thread's k-block is past the end
            if (tid < 256) {
                const uint32_t off = offS - (dead ? (uint32_t)(Npad * 4) : 0u);
                screg = *(const float *)((const uint8_t *)(ad + kb0 * Npad) + off);
                if (TYPE == GGML_TYPE_Q4_1) sc2reg = (float)*(const int32_t *)((const uint8_t *)(as + kb0 * Npad) + off);
            } else {
                const uint32_t off = offS - (dead ? (uint32_t)(Mpad * 4) : 0u);
                const float v = *(const float *)((const uint8_t *)(wd + kb0 * Mpad) + off);
                screg = dead ? 0.0f : v;                              // scale 0 turns a k-block past the end into +0
                if (TYPE == GGML_TYPE_Q4_1) {
                    const float v2 = *(const float *)((const uint8_t *)(wm + kb0 * Mpad) + off);
                    sc2reg = dead ? 0.0f : v2;
                }
            }
        }
    };

    auto store_stage = [&](int s) {
        uint8_t *sp = stage_ptr(s);
        uint8_t *sW = sp + L::A_BYTES;
        float *sSc = (float *)(sp + L::A_BYTES + L::W_BYTES);   // [da | dw | sa | mw], BKB*128 floats each
        const uint32_t q[4] = {wreg.x, wreg.y, wreg.z, wreg.w};
        if (TYPE == GGML_TYPE_Q8_0) {
            // plane h byte j = element 2j + h (signed): bytes (0,2) of every word -> panel h, bytes (1,3) -> panel h + 2
            const int bh = tid >> 7, row = tid & 127, bb = bh >> 1, h = bh & 1;
            uint32_t pa[4], pb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t x = q[k] ^ 0x80808080u;   // signed byte -> biased unsigned
                pa[k] = u16pair_to_f16(x & 0x00FF00FFu, 1024.0f + 128.0f);
                pb[k] = u16pair_to_f16((x >> 8) & 0x00FF00FFu, 1024.0f + 128.0f);
            }
            *(uint4 *)(sW + ((size_t)((bb * 4 + h) * TM + row)) * 16) = make_uint4(pa[0], pa[1], pa[2], pa[3]);
            *(uint4 *)(sW + ((size_t)((bb * 4 + h + 2) * TM + row)) * 16) = make_uint4(pb[0], pb[1], pb[2], pb[3]);
        } else {
            const int c = tid >> 1, bb = c >> 7, row = c & 127;
            constexpr float OFF = TYPE == GGML_TYPE_Q4_0 ? 8.0f : (TYPE == GGML_TYPE_Q5_0 ? 16.0f : 0.0f);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = 2 * (tid & 1) + pp;
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t pair = (q[k] >> (4 * p)) & 0x000F000Fu;           // elements 8k+p, 8k+4+p (Ggml.cs:1149-1150)
                    if (TYPE == GGML_TYPE_Q5_0) pair |= q5_hi16(hreg, k, p);   // Ggml.cs:1285-1289
                    w[k] = u16pair_to_f16(pair, 1024.0f + OFF);
                }
                *(uint4 *)(sW + ((size_t)((bb * 4 + p) * TM + row)) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        sSc[tid] = screg;                                  // tid < 256: da, else dw (contiguous planes)
        if (TYPE == GGML_TYPE_Q4_1) sSc[512 + tid] = sc2reg;
    };

    // ---- one stage: BKB k-blocks x 2 tiles (32 x 32) x 2 MFMAs per wave ----
    auto compute = [&](int s) {
        const uint8_t *sp = stage_ptr(s);
        const uint8_t *sA = sp;
        const uint8_t *sW = sp + L::A_BYTES;
        const float *sDa = (const float *)(sp + L::A_BYTES + L::W_BYTES);
        const float *sDw = sDa + BKB * TN;
        const float *sSa = sDw + BKB * TM;                 // Q4_1 only
        const float *sMw = sSa + BKB * TN;                 // Q4_1 only
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int row0 = wn * 32;
        // lane 16q + k owns the scale of MFMA row R(k, q>>1) = (k&3) + 8*(k>>2) + 4*(q>>1)
        const int myrow = row0 + (lane & 3) + 8 * ((lane >> 2) & 3) + 4 * hh;
        constexpr int NT = BKB * 2;

        f16x8 af[2], bf[2][2];
        float vda, vsa = 0.0f, dw[2], mw[2];
        f32x16 tacc[2];

        auto load_block = [&](auto bc) {
            constexpr int bb = decltype(bc)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                af[kk] = *(const f16x8 *)(sA + ((size_t)((bb * 4 + 2 * kk + hh) * TN + row0 + l31)) * 16);
            vda = sDa[bb * TN + myrow];
            if (TYPE == GGML_TYPE_Q4_1) vsa = sSa[bb * TN + myrow];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wm_ * 64 + 32 * j + l31;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    bf[j][kk] = *(const f16x8 *)(sW + ((size_t)((bb * 4 + 2 * kk + hh) * TM + col)) * 16);
                dw[j] = sDw[bb * TM + col];
                if (TYPE == GGML_TYPE_Q4_1) mw[j] = sMw[bb * TM + col];
            }
        };
        auto mfma_tile = [&](auto tc) {
            constexpr int t = decltype(tc)::value, j = t & 1;
            f32x16 x = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[j][0], zero, 0, 0, 0);
            tacc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[j][1], x, 0, 0, 0);
        };

        load_block(std::integral_constant<int, 0>{});
        mfma_tile(std::integral_constant<int, 0>{});
        static_for<NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t >> 1, j = t & 1;
            // tile t+1's MFMAs go out before tile t's scale-accumulate (matrix-core latency hidden behind the VALU)
            if constexpr (j == 0) mfma_tile(std::integral_constant<int, t + 1>{});
            float sc[16];
            scale16(sc, vda, dw[j]);                                   // d1 * d0, Ggml.cs:1158
            const f32x16 tt = tacc[t & 1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = fmaf(tt[r], sc[r], acc[j][r]);
            if (TYPE == GGML_TYPE_Q4_1) {                              // + m0 * d1 * sum(a) (Ggml.cs:1190-1196 factorised)
                float ds[16];
                scale16(ds, vda * vsa, mw[j]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += ds[r];
            }
            // pin: keeps the optimiser from sinking the scale-accumulates below the stage's last MFMA
            asm volatile("" : "+v"(acc[j]));
            if constexpr (j == 1 && bb + 1 < BKB) {
                load_block(std::integral_constant<int, bb + 1>{});
                mfma_tile(std::integral_constant<int, t + 1>{});
            }
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

    // ---- dst[n][m]: D[row = (r&3) + 8*(r>>2) + 4*hh][col = lane & 31] ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t m = m0 + wm_ * 64 + 32 * j + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (n < N && m < M) dst[n * ldd + m] = acc[j][r];
        }
    }
}

template <int TYPE>
hipError_t launch_typed(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    using L = Lds<TYPE>;
    static bool attr_set = false;
    auto kern = gemm_q16_kernel<TYPE>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L::TOTAL);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static const int dbg = [] { const char *e = getenv("GGML_HIP_GEMM_DBG"); return e ? atoi(e) : 0; }();  // developer ablations
    const int tiles_m = (int)((w->M + TM - 1) / TM), tiles_n = (int)((N + TN - 1) / TN);
    dim3 grid((unsigned)(tiles_m * tiles_n));
    kern<<<grid, 512, L::TOTAL, st>>>(w->qs, w->qh, w->d, w->m, (const uint8_t *)p.a8, p.ad, p.as, dst, w->M, N, w->Mpad, p.Npad,
                                      w->nbk, ldd, tiles_m, tiles_n, dbg);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_q16(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    if (p.Npad % TN != 0) return hipErrorInvalidValue;
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0>(w, p, N, dst, ldd, st);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0>(w, p, N, dst, ldd, st);
    default: return hipErrorInvalidValue;
    }
}
