// gemm_qmx.hip -- K3m: quantized mat-mat for large N (Q4_0, Q4_1; Q5_0, Q8_0 with two weight digits), block-scaled, on the MX matrix
// path of gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4 with bf6 = e3m2 operands).
//
// COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698):
//   dst[n*ldd + m] = sum_b (dw[m,b] * da[n,b]) * sumi_b(m,n),   sumi_b = the integer dot of one 32-element block
// (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1136-1159; _q4_1_q8_1 1176-1198).
//
// One MFMA per 32x32 tile and quant block, exact:  the instruction multiplies 32x64 by 64x32 with one power-of-two
// (E8M0) scale per operand row and 32-element K group.  Every 4-bit weight (nib - 8, in [-8, 7]) is exact in bf6
// (e3m2 holds every integer up to 8).  A Q8 activation a in [-127, 127] is split as a = 16*ah + al with ah in [-8, 8],
// al in [-8, 7], both exact in bf6: K group 0 carries ah with block scale 2^4, K group 1 carries al with scale 2^0,
// and the weights of the block sit in both groups.  So D = 16*sum(w*ah) + sum(w*al) = sumi_b, every partial sum an
// integer below 2^24, i.e. bit-exact in the f32 accumulator -- in 32 matrix-pipe cycles instead of the 64 of two
// f16 MFMAs (gemm_q16.hip), with NO weight expansion on the VALU: the resident bf6 planes (layout.hip, built once at
// upload) are loaded straight into the B operand registers.  Measured (tools/tile_ubench.hip): 54 ns per tile and
// k-block per SIMD against 78 ns for the f16 form, next to a floor of 36 ns for the 32 scale-accumulate VALU
// instructions alone -- the reference's own per-block f32 work (Ggml.cs:1158), which is what bounds this path.
// Operand layout (probed with exact integer data, tools/mx_probe.hip): lane l holds row/col l & 31, K elements
// 32*(l >> 5) .. +31, element e at bits [6e, 6e+5] of a 192-bit fragment (6 VGPRs); the scale byte of a lane applies
// to that lane's row and K group.
//
// Wider weights (Q5_0 in [-16, 15], Q8_0 in [-128, 127]) get the same two-digit split as the activations, w = 16*wh + wl:
// two fragments per block and two chained MFMAs, B = wh with block scale 2^4 and B = wl with 2^0, against the SAME A
// operand: 256*sum(wh*ah) + 16*(sum(wh*al) + sum(wl*ah)) + sum(wl*al) = sumi_b, |sumi_b| < 2^19, still bit-exact.
//
// Everything else follows gemm_q16.hip: MFMA rows = src1 rows n, cols = weight rows m (dst stores are 128-byte
// segments along m); activations reach LDS by DMA (K1 writes the bf6 image), row scales by broadcast ds_read_b128;
// raw buffer addressing; inline-asm scale-accumulate; K zero-padded to whole stages by the host side.
#include "common.h"
#include "plan.h"
#include <cstdlib>
#include <utility>

// Developer timing ablations (never in the product build): -DGGML_MX_DBG=<bits>  1 no barrier / DMA, 2 no LDS fragment
// or scale reads after the first, 4 no weight reloads, 8 no MFMA after the first, 16 no scale-accumulate, 32 no row-scale reads, 64 no fragment reads.
#if !defined(GGML_HIP_DEV)      // the product build has no ablation paths: the switches exist only under -DGGML_HIP_DEV
#undef GGML_MX_DBG
#undef GGML_MX_XCD1D
#endif
#ifndef GGML_MX_DBG
#define GGML_MX_DBG 0
#endif
#ifndef GGML_MX_XCD1D
#define GGML_MX_XCD1D 0        // 1: the former XCD order everywhere (A/B)
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using i32x8 = __attribute__((ext_vector_type(8))) int;

typedef __attribute__((address_space(3))) void lds_void;

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void blds16(rsrc_t r, void *l, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)l, 16, (int)voff, (int)soff, 0, 0);
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#ifdef K3M_CLOCK            // tools/k3m_clock.hip: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) every wave spends in the K loop
__device__ unsigned long long k3m_clock_buf[8192 * 2];
__device__ unsigned int k3m_where_buf[8192 * 2];             // HW_ID and XCC_ID of the wave: which CU / SIMD / wave slot / XCD ran it
#endif

// WMT x WNT 32x32 tiles per wave, WGM x WGN waves per workgroup, KB k-blocks per LDS stage
#ifndef GGML_MX_FULLDRAIN    // developer A/B: 1 = vmcnt(0) at the drain point and at the barrier, as through round 2
#define GGML_MX_FULLDRAIN 0
#endif
template <int TYPE, int WMT, int WNT, int WGM, int WGN, int KB>
struct Cfg {
    static constexpr int TM = WGM * WMT * 32, TN = WGN * WNT * 32, NT = WGM * WGN * 64;
    static constexpr int A16_BYTES = KB * 2 * TN * 16;     // first 16 bytes of every 24-byte activation fragment
    static constexpr int A8_BYTES = KB * 2 * TN * 8;       // last 8 bytes
    static constexpr int NSC = (TYPE == GGML_TYPE_Q4_1) ? 2 : 1;
    static constexpr int SC_BYTES = KB * TN * 4;           // one f32 plane of row scales
    static constexpr int STAGE = A16_BYTES + A8_BYTES + NSC * SC_BYTES;
    static constexpr int TOTAL = 2 * STAGE;
    static constexpr int P = NT / TN;
    static_assert(NT % TN == 0 && (P == 1 || P == 2 || P == 4), "chunk decomposition");
    static_assert((KB * 2 * TN) % NT == 0 && (KB * TN) % NT == 0, "DMA rounds");
    static constexpr int A16_ROUNDS = KB * 2 * TN / NT, A8_ROUNDS = KB * TN / NT;
    static constexpr int SC_CHUNKS = KB * TN / 4;          // 16-byte pieces per plane
    static_assert((SC_CHUNKS % 64 == 0 || SC_CHUNKS == 32) && SC_CHUNKS <= NT, "scale DMA is whole waves (or the first half of one: 32-column tiles)");
    static constexpr int NPIECE = A16_ROUNDS + A8_ROUNDS + 1;
    // DMA pieces per tile: one where the stage has enough tiles (the pieces must all be issued before the drain point in the
    // middle of the stage's last k-block), two for the small wave tiles
    static constexpr int DRAIN_T = (KB - 1) * WMT * WNT + WMT * WNT / 2;
    static constexpr int PP = (NPIECE + DRAIN_T - 1) / DRAIN_T;
    static_assert(PP <= 3, "at most three DMA pieces per tile");
};

// FB = weight fragment buffers: 2 = fragments of k-block kb+2 are loaded into the buffer k-block kb has just released;
// 1 = in place, for kb+1 (enough look-ahead when a wave has >= 4 m-tiles between two uses of a fragment)
// KSP = 1 | 2 | 4: with 2 (4) the workgroup holds two (four) wave groups that take alternate LDS stages of K (each with its own stage
// buffers) and add their accumulators through LDS at the end -- twice the waves for grids too small to fill the chip.
// VS = 1 | 2 | 4 "virtual" groups per wave group: the wave group runs the stage sets of VS groups one after the other, keeping each
// set's sum apart, and the sums are added in group order -- the summation tree of a KSP * VS-way split, bit for bit, without
// the extra waves.  What fixes an element's bits is KSP * VS (chosen from N and K); how it is spread over waves may follow M.
template <int TYPE, int WMT, int WNT, int WGM, int WGN, int KB, int FB, int KSP, int VS>
__global__ __launch_bounds__(WGM * WGN * 64 * KSP, (VS > 1 && WGM * WGN * KSP == 4 && WMT * WNT == 2 ? 3 : 2))   // (three 4-wave workgroups per CU: 168 registers)
void gemm_qmx_kernel(const uint8_t *__restrict__ w6a, const uint8_t *__restrict__ w6b, const float *__restrict__ wd,
                     const float *__restrict__ wm, const uint8_t *__restrict__ a6, const float *__restrict__ ad,
                     const float *__restrict__ asd, float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nstages,
                     int ldd, int tiles_m, int tiles_n, uint32_t w6a_bytes, uint32_t wd_bytes, uint32_t a_bytes,
                     uint32_t ad_bytes, uint32_t dst_bytes, const mm_epilogue ep) {
    using C = Cfg<TYPE, WMT, WNT, WGM, WGN, KB>;
    constexpr int NTILE = WMT * WNT, P = C::P;
    constexpr int NF = (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q4_1) ? 1 : 2;   // weight digits (fragments) per block
    static_assert(KB % 2 == 0, "fragment buffers alternate by k-block parity");
    static_assert(VS == 1 || (KSP == 1 && (VS == 2 || VS == 4)) || (KSP == 2 && VS == 2), "virtual split forms");
    constexpr int KV = KSP * VS;                            // width of the summation tree
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave_all / (WGM * WGN);                 // K-split group (0 when KSP == 1)
    const int wave = wave_all % (WGM * WGN);                // wave inside its group
    const int tid = wave * 64 + lane;                       // thread inside its group
    const int l31 = lane & 31, hh = lane >> 5;
    const int wn = wave / WGM, wm_ = wave % WGM;           // waves of one group: m fastest
    uint8_t *const gsm = smem + grp * C::TOTAL;             // this group's two stage buffers

    // XCD-aware tile order (speed only): workgroups b, b+8, b+16, ... share an XCD and its L2.  Each XCD gets a
    // contiguous run of the tile list ordered "m fastest".
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    int tm_i, tn_i;
    if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0 && !GGML_MX_XCD1D) {
        // 2 x 4 blocks of the tile grid per XCD: an XCD's 64 resident workgroups share half of the weight panels and a quarter of
        // the activation panels (4096^3: 14 MB through each L2 instead of 18 MB with a run of whole n-columns)
        const int hm = tiles_m >> 1, l = bid >> 3;
        tm_i = (xcd & 1) * hm + l % hm;
        tn_i = (xcd >> 1) * (tiles_n >> 2) + l / hm;
    } else {
        const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        tm_i = t_lin % tiles_m;
        tn_i = t_lin / tiles_m;
    }
    const int m0 = tm_i * C::TM;
    const int n0 = tn_i * C::TN;

    f32x16 acc[WMT][WNT];                                   // (register tuples: the min-term MFMA of Q4_1 accumulates into them in place)
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- activation image (K1, kind 3) + row scales: global -> LDS by DMA, one piece (1 KiB per wave) per tile ----
    // image per k-block: [half][Npad][16 B] then [half][Npad][8 B]  (half 0 = the 16*ah group, half 1 = the al group);
    // LDS stage: [bb][half][TN][16 B] then [bb][half][TN][8 B] then the scale planes.  Chunk c = tid + NT*i splits into a
    // per-thread part (one 32-bit offset per region for the whole kernel) and a part that is uniform in i.
    const uint32_t a_blk = (uint32_t)Npad * 48u;
    const int t16 = tid / C::TN, t8 = tid / (C::TN / 2);
    const uint32_t voff16 = (uint32_t)(t16 >> 1) * a_blk + (uint32_t)(((t16 & 1) * Npad + n0 + tid % C::TN) * 16);
    const uint32_t voff8 = (uint32_t)(t8 >> 1) * a_blk + (uint32_t)(32 * Npad + (t8 & 1) * Npad * 8 + n0 * 8 + (tid % (C::TN / 2)) * 16);
    const uint32_t voffS = (uint32_t)(((tid / (C::TN / 4)) * Npad + n0 + 4 * (tid % (C::TN / 4))) * 4);
    const rsrc_t rA = make_rsrc(a6, a_bytes), rAd = make_rsrc(ad, ad_bytes), rAs = make_rsrc(asd, ad_bytes);
    // Pieces past the end of K are dropped by the descriptor's range check (they would fill a buffer nobody reads).
    auto dma_piece = [&](int s, int buf, auto pc) {             // stage s of K -> stage buffer buf of this group
        constexpr int i = decltype(pc)::value;
        uint8_t *sp = gsm + (buf & 1) * C::STAGE;
        const uint32_t s0 = (uint32_t)s * KB * a_blk;
        if constexpr (i < C::A16_ROUNDS) {
            constexpr int u = P * i;
            blds16(rA, sp + (size_t)(wave * 64 + C::NT * i) * 16, voff16, s0 + (u >> 1) * a_blk + (u & 1) * (uint32_t)(Npad * 16));
        } else if constexpr (i < C::A16_ROUNDS + C::A8_ROUNDS) {
            constexpr int i8 = i - C::A16_ROUNDS, u = 2 * P * i8;
            blds16(rA, sp + C::A16_BYTES + (size_t)(wave * 64 + C::NT * i8) * 16, voff8, s0 + (u >> 1) * a_blk);
        } else if (wave * 64 < C::SC_CHUNKS) {                           // uniform per wave
            const uint32_t sS0 = (uint32_t)s * KB * (uint32_t)(Npad * 4);
            if (C::SC_CHUNKS % 64 == 0 || lane < C::SC_CHUNKS % 64) {   // (32-column tiles: half a wave of 16-byte pieces; the DMA honours exec)
                blds16(rAd, sp + C::A16_BYTES + C::A8_BYTES + (size_t)(wave * 64) * 16, voffS, sS0);
                if (TYPE == GGML_TYPE_Q4_1)
                    blds16(rAs, sp + C::A16_BYTES + C::A8_BYTES + C::SC_BYTES + (size_t)(wave * 64) * 16, voffS, sS0);
            }
        }
    };

    // ---- weights: the bf6 planes go straight into the B operand registers (both lane halves hold the same block) ----
    struct Frag { u32x4 lo[WMT][NF]; u32x2 hi[WMT][NF]; float d[WMT]; float mn[WMT]; };   // [.][0] = low digit, [.][1] = high digit
    // one per-thread offset per plane; the m-tiles of a wave are 32 rows apart = a constant the instruction's immediate
    // offset field takes
    const int mrow = m0 + wm_ * WMT * 32 + l31;
    const uint32_t offA = (uint32_t)(mrow * 16), offB = (uint32_t)(mrow * 8), offD = (uint32_t)(mrow * 4);
    const uint32_t wa_frag = (uint32_t)(Mpad * 16), wb_frag = (uint32_t)(Mpad * 8);                 // planes are [nbk][NF][Mpad][16 | 8]
    const uint32_t wa_blk = NF * wa_frag, wb_blk = NF * wb_frag, d_blk = (uint32_t)(Mpad * 4);
    const rsrc_t rWa = make_rsrc(w6a, w6a_bytes), rWb = make_rsrc(w6b, w6a_bytes / 2), rWd = make_rsrc(wd, wd_bytes);
    const rsrc_t rWm = make_rsrc(TYPE == GGML_TYPE_Q4_1 ? (const void *)wm : (const void *)wd, wd_bytes);
    // the planes carry spare (zero) k-blocks past the padded end, so the look-ahead never needs a bounds check
    auto load_frag_one = [&](int kb, Frag &f, auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int g = 0; g < NF; ++g) {
            f.lo[i][g] = __builtin_amdgcn_raw_buffer_load_b128(rWa, (int)(offA + 512u * i), (int)((uint32_t)kb * wa_blk + g * wa_frag), 0);
            f.hi[i][g] = __builtin_amdgcn_raw_buffer_load_b64(rWb, (int)(offB + 256u * i), (int)((uint32_t)kb * wb_blk + g * wb_frag), 0);
        }
        f.d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWd, (int)(offD + 128u * i), (int)((uint32_t)kb * d_blk), 0));
        if constexpr (TYPE == GGML_TYPE_Q4_1)
            f.mn[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWm, (int)(offD + 128u * i), (int)((uint32_t)kb * d_blk), 0));
    };

    Frag frag[FB];                                          // k-block kb uses frag[kb % FB]
    const int scale_a = hh ? 127 : 131;                     // E8M0: K group 0 (the ah digits) carries 2^4

    auto compute = [&](int s, int sn, int buf) {                 // stage s of K, resident in stage buffer buf; sn = this wave group's next stage
        const uint8_t *sp = gsm + (buf & 1) * C::STAGE;
        const uint8_t *sA16 = sp + ((size_t)(hh * C::TN + wn * WNT * 32 + l31)) * 16;
        const uint8_t *sA8 = sp + C::A16_BYTES + ((size_t)(hh * C::TN + wn * WNT * 32 + l31)) * 8;
        const float *sDa = (const float *)(sp + C::A16_BYTES + C::A8_BYTES) + wn * WNT * 32 + 4 * hh;
        // Q4_1 only: d1 * sum(a), read as the A operand of a 32x32x2 f32 MFMA -- lane half hh takes the odd k-block of a pair
        const float *sSp = (const float *)(sp + C::A16_BYTES + C::A8_BYTES + C::SC_BYTES) + hh * C::TN + wn * WNT * 32 + l31;
        const int kb0 = s * KB;
        constexpr int LAST = KB * NTILE - 1, DRAIN = (KB - 1) * NTILE + NTILE / 2;
        static_assert(DRAIN == C::DRAIN_T && C::NPIECE <= C::PP * DRAIN, "all DMA pieces are issued before the drain point");

        u32x4 af_lo;
        u32x2 af_hi;
        float af_s = 0.0f;                                              // Q4_1: d1 * sum(a) of row l31, k-block (bb - 1) + hh
        f32x4 da[4];
        f32x16 tacc[2];
        float dcur[WMT], mcur[WMT];

        auto fetch_af = [&](auto nc) {                                   // nc = n-tile index within the stage
            constexpr int g = decltype(nc)::value, bb = g / WNT, j = g % WNT;
            af_lo = *(const u32x4 *)(sA16 + (bb * 2 * C::TN + 32 * j) * 16);
            af_hi = *(const u32x2 *)(sA8 + (bb * 2 * C::TN + 32 * j) * 8);
            if constexpr (TYPE == GGML_TYPE_Q4_1 && (bb & 1)) af_s = sSp[(bb - 1) * C::TN + 32 * j];
        };
        auto fetch_da = [&](auto nc, auto gc) {
            constexpr int g = decltype(nc)::value, bb = g / WNT, j = g % WNT, q = decltype(gc)::value;
            da[q] = *(const f32x4 *)(sDa + bb * C::TN + 32 * j + 8 * q);
        };
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // the MFMA of tile t, then everything that becomes possible once it is issued
        auto issue = [&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / NTILE, j = (t % NTILE) / WMT, i = t % WMT;
            Frag &f = frag[bb % FB];
            const i32x8 a = {(int)af_lo[0], (int)af_lo[1], (int)af_lo[2], (int)af_lo[3], (int)af_hi[0], (int)af_hi[1], 0, 0};
            const i32x8 b = {(int)f.lo[i][0][0], (int)f.lo[i][0][1], (int)f.lo[i][0][2], (int)f.lo[i][0][3], (int)f.hi[i][0][0], (int)f.hi[i][0][1], 0, 0};
            if constexpr (GGML_MX_DBG & 8 && t != 0) {
                asm volatile("" : "+v"(tacc[t & 1]));
            } else if constexpr (NF == 1) {
                tacc[t & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, zero, 3, 3, 0, scale_a, 0, 127);
            } else {      // high weight digit (x 2^4) first, then the low digit on top
                const i32x8 bh = {(int)f.lo[i][1][0], (int)f.lo[i][1][1], (int)f.lo[i][1][2], (int)f.lo[i][1][3], (int)f.hi[i][1][0], (int)f.hi[i][1][1], 0, 0};
                const f32x16 x = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bh, zero, 3, 3, 0, scale_a, 0, 131);
                tacc[t & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, x, 3, 3, 0, scale_a, 0, 127);
            }
            if constexpr (j == 0) {                                       // first use of this block's scales: keep them
                dcur[i] = f.d[i];                                         // (the buffer is reloaded before the last use)
                if constexpr (TYPE == GGML_TYPE_Q4_1) {                   // nib = (nib - 8) + 8: the min term's weight factor is m0 + 8 d0
                    const float m8 = fmaf(8.0f, f.d[i], f.mn[i]);
                    // B operand of the pair's MFMA: k-blocks bb - 1 | bb in the two lane halves
                    mcur[i] = (bb & 1) == 0 ? m8 : (hh ? m8 : mcur[i]);
                }
            }
            // Q4_1 min term, + (m0 + 8 d0) * (d1 * sum(a)) per block (Ggml.cs:1190-1196 factorised): an outer product per block, i.e.
            // a K = 2 matrix product per pair of blocks -- one v_mfma_f32_32x32x2_f32 straight into the tile's accumulators,
            // issued a whole tile step (32 VALU instructions) before and after the scale-accumulates that touch them.  A BUILTIN, so
            // that hipcc pads its own reads of the result (it splits the tuple with v_mov); the inline-asm readers are kept a
            // step away by an empty volatile statement on the tuple ahead of the previous tile's scale-accumulates (below).
            if constexpr (TYPE == GGML_TYPE_Q4_1 && (bb & 1)) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_s, mcur[i], acc[i][j], 0, 0, 0);
                // one tile per k-block (32-column forms): the SAME accumulators take the previous tile's scale-accumulates right
                // behind this MFMA: its 16 passes + 4 wait states are spent here (tests/test_isa_audit.py found the asm 6 slots
                // behind it; twenty idle cycles per pair of k-blocks)
                if constexpr (NTILE == 1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][j]));
            }
            if constexpr (i == WMT - 1 && bb * WNT + j + 1 < KB * WNT && !(GGML_MX_DBG & (2 | 64))) fetch_af(std::integral_constant<int, bb * WNT + j + 1>{});
            // the fragment of m-tile i is dead once the MFMA of the block's last n-tile is issued: reload it for kb + 2
            // (this group's k-blocks: the stage after s is sn)
            if constexpr (j == WNT - 1 && !(GGML_MX_DBG & 4))
                load_frag_one(bb + FB < KB ? kb0 + bb + FB : sn * KB + (bb + FB - KB), f, std::integral_constant<int, i>{});
        };

        fetch_af(std::integral_constant<int, 0>{});
        static_for<4>([&](auto gc) { fetch_da(std::integral_constant<int, 0>{}, gc); });
        issue(std::integral_constant<int, 0>{});

        static_for<KB * NTILE>([&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / NTILE, tl = t % NTILE, j = tl / WMT, i = tl % WMT;
            // this tile's weight scales, read before issue(t + 1) may replace them with the next block's (WMT == 1)
            const float dw = dcur[i];
            if constexpr (!(GGML_MX_DBG & 1)) {
                static_for<C::PP>([&](auto uc) {
                    constexpr int pc = C::PP * t + decltype(uc)::value;
                    if constexpr (pc < C::NPIECE) dma_piece(sn, buf + 1, std::integral_constant<int, pc>{});
                });
            }
            // The DMA pieces of the next stage must have landed before this wave arrives at the stage's barrier.  Waiting
            // for them HERE, in the middle of the stage's last k-block, costs nothing (every vector-memory operation
            // issued so far is at least half a k-block old) and leaves the weight loads issued after this point in
            // flight across the barrier.
            // (Vector-memory operations complete in issue order: where no wave group skips a stage, waiting until only the weight
            // loads issued BEHIND the last DMA piece are outstanding is enough -- fragment of tile u, the last n-tile of its block,
            // is requested in step u - 1; the last piece in step LASTP.)
            if constexpr (t == DRAIN) {
                constexpr int LASTP = (C::NPIECE - 1) / C::PP;
                constexpr int FRAG_LOADS = NF * 2 + 1 + (TYPE == GGML_TYPE_Q4_1 ? 1 : 0);
                constexpr int YOUNGER = [] { int n = 0; for (int u = LASTP + 1; u <= DRAIN; ++u) n += ((u % NTILE) / WMT == WNT - 1); return n; }() * FRAG_LOADS;
                if constexpr (KSP == 1 && VS == 1 && YOUNGER < 64 && !GGML_MX_FULLDRAIN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (t < LAST) issue(std::integral_constant<int, t + 1>{});
            f32x16 &ac = acc[i][j];
            static_for<4>([&](auto gc) {
                constexpr int q = decltype(gc)::value;
                float t0, t1, t2, t3;
                // Hazard note (inline asm is not padded by hipcc): an MFMA result needs 12 wait states before a VALU
                // reads it.  Tile t's MFMA is pinned above tile t-1's 32 VALU instructions, so only the first tile of
                // a stage can be short of them: it carries its own s_nop.
                if constexpr (t == 0 && q == 0) asm volatile("s_nop 11" : "+v"(tacc[0]));
                // Q4_1: the next tile's accumulators were just written by its min-term MFMA (issue(t + 1) above); this empty statement
                // keeps that MFMA above this tile's 32 VALU instructions (volatile statements keep their order)
                if constexpr (q == 0 && t < LAST && TYPE == GGML_TYPE_Q4_1 && (((t + 1) / NTILE) & 1)) {
                    constexpr int tn = t + 1, j1 = (tn % NTILE) / WMT, i1 = tn % WMT;
                    asm volatile("" : "+v"(acc[i1][j1]));
                }
                // acc += (sumi * d1) * d0, Ggml.cs:1158.  The other MFMA result rides along as a dummy operand of
                // group 0 so that its MFMA stays above this tile's VALU work (which is what it overlaps with).
                if constexpr ((GGML_MX_DBG & 16) != 0) {
                    asm volatile("" : "+v"(ac[4 * q]) : "v"(tacc[t & 1]), "v"(da[q]), "v"(dw), "v"(tacc[(t + 1) & 1]));
                } else if constexpr (q == 0 && t < LAST) {
                    asm volatile("v_mul_f32 %4, %8, %12\n\tv_mul_f32 %5, %9, %13\n\tv_mul_f32 %6, %10, %14\n\tv_mul_f32 %7, %11, %15\n\t"
                                 "v_fmac_f32 %0, %4, %16\n\tv_fmac_f32 %1, %5, %16\n\tv_fmac_f32 %2, %6, %16\n\tv_fmac_f32 %3, %7, %16"
                                 : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                                 : "v"(tacc[t & 1][0]), "v"(tacc[t & 1][1]), "v"(tacc[t & 1][2]), "v"(tacc[t & 1][3]), "v"(da[0][0]),
                                   "v"(da[0][1]), "v"(da[0][2]), "v"(da[0][3]), "v"(dw), "v"(tacc[(t + 1) & 1]));
                } else {
                    asm volatile("v_mul_f32 %4, %8, %12\n\tv_mul_f32 %5, %9, %13\n\tv_mul_f32 %6, %10, %14\n\tv_mul_f32 %7, %11, %15\n\t"
                                 "v_fmac_f32 %0, %4, %16\n\tv_fmac_f32 %1, %5, %16\n\tv_fmac_f32 %2, %6, %16\n\tv_fmac_f32 %3, %7, %16"
                                 : "+v"(ac[4 * q + 0]), "+v"(ac[4 * q + 1]), "+v"(ac[4 * q + 2]), "+v"(ac[4 * q + 3]), "=&v"(t0), "=&v"(t1),
                                   "=&v"(t2), "=&v"(t3)
                                 : "v"(tacc[t & 1][4 * q + 0]), "v"(tacc[t & 1][4 * q + 1]), "v"(tacc[t & 1][4 * q + 2]),
                                   "v"(tacc[t & 1][4 * q + 3]), "v"(da[q][0]), "v"(da[q][1]), "v"(da[q][2]), "v"(da[q][3]), "v"(dw));
                }
                // this group's row scales are dead after the n-tile's last m-tile: fetch the next n-tile's into their place
                if constexpr (i == WMT - 1 && bb * WNT + j + 1 < KB * WNT && !(GGML_MX_DBG & (2 | 32)))
                    fetch_da(std::integral_constant<int, bb * WNT + j + 1>{}, gc);
            });
        });
    };

    // ---- main loop: double-buffered LDS, one barrier per stage (vmcnt(0): see gemm_q16.hip).  Group grp takes stages
    //      grp, grp + KSP, ...; every wave passes the same number of barriers. ----
    static_assert(FB <= KB, "fragment look-ahead stays within two stages");
    const int s_first = grp * VS;                           // first stage of this wave group's first (virtual) group
    static_for<C::NPIECE>([&](auto pc) { dma_piece(s_first, 0, pc); });
    static_for<WMT>([&](auto ic) {
        static_for<FB>([&](auto gc) { load_frag_one(s_first * KB + decltype(gc)::value, frag[decltype(gc)::value], ic); });
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                          // no LDS read may move above the barrier
    // Virtual group v = grp * VS + pass takes stages v, v + KV, v + 2 KV, ...; a wave group runs its passes back to back (the
    // prefetch of a pass's last stage fetches the next pass's first), then idles through the barriers it has left.
    f32x16 sav[VS > 1 ? WMT : 1][VS > 1 ? WNT : 1];         // sum of this wave group's finished passes
    const int niter = VS * ((nstages + KV - 1) / KV);
    int sc = s_first, pass = 0;
#ifdef K3M_CLOCK
    const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
#endif
    // r4: two 4-wave workgroups share a CU, one wave of each per SIMD, and the arbiter serves the OLDER wave first: on real data the older
    // workgroup's waves finished their K loops at ~123 us and the younger one's at ~147 us (tools/k3m_clock.hip: per-wave loop times
    // bimodal), i.e. for the last 24 us every SIMD ran a lone wave at half the issue rate.  As in K3p (gemm_qmp.hip), the younger wave of
    // a SIMD -- the one in the odd wave slot -- takes priority on two stages of three, so both finish together.  Speed only: real data,
    // back to back, 4096^3 146.7 -> 144.8 us, 4096 x 11008 x 2048 211.8 -> 208.0, Q4_1 4096^3 208.0 -> 202.9 (same bits).  Only for grids
    // of one round (at most two workgroups per CU): with more rounds the early finisher's place is taken by the next workgroup at once,
    // and its store tail overlaps that workgroup's start (32000 x 4096 x 2048: 586 us without, 589 with).
#ifndef GGML_MX_PRIO
#define GGML_MX_PRIO 15
#endif
#ifndef GGML_MX_PRIO_ON
#define GGML_MX_PRIO_ON 8
#endif
    bool younger = false;
    if constexpr (WGM * WGN * KSP == 4 && GGML_MX_PRIO != 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        younger = (hwid & 1u) != 0 && nwg <= 512;            // WAVE_ID: the slot inside the SIMD
    }
    for (int it = 0; it < niter; ++it) {
        if constexpr (WGM * WGN * KSP == 4 && GGML_MX_PRIO != 0) {
            // (priority on GGML_MX_PRIO_ON stages of every GGML_MX_PRIO.  With 2 of 3 the younger workgroup finished 6.5 us BEFORE the older one,
            // with 1 of 2 4.7 us behind it; 8 of 15: 138.1 | 139.2 us (tools/k3m_clock.hip, K loop median by wave slot).  What is left of
            // the spread is between XCDs -- the odd ones run ~5 % slower on the boxes looked at, in-kernel clock 1.71 against 1.82 GHz.)
            if (younger) { if (it % GGML_MX_PRIO < GGML_MX_PRIO_ON) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        }
        int sn = sc + KV, pn = pass;
        if (VS > 1 && sn >= nstages) { pn = pass + 1; sn = pn < VS ? grp * VS + pn : nstages; }
        if ((KSP == 1 && VS == 1) || sc < nstages) compute(sc, sn, it);
        // (the drain inside compute, again for a skipped stage; without K split every stage runs and the weight loads in flight stay in flight)
        if constexpr (KSP == 1 && VS == 1 && !GGML_MX_FULLDRAIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (!(GGML_MX_DBG & 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (VS > 1) {
            if (pn != pass && pn < VS) {                    // pass boundary (uniform): bank this pass's sum, start the next from zero
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j) {
                        sav[i][j] = pass == 0 ? acc[i][j] : sav[i][j] + acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
                    }
            }
        }
        sc = sn; pass = pn;
    }
#ifdef K3M_CLOCK
    {
        asm volatile("" : "+v"(acc[0][0]));
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
        const size_t slot = (size_t)blockIdx.x * (WGM * WGN * KSP) + wave_all;
        if (lane == 0 && slot < 8192) { k3m_clock_buf[2 * slot] = c1 - clk_c0; k3m_clock_buf[2 * slot + 1] = r1 - clk_r0; }
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        if (lane == 0 && slot < 8192) { k3m_where_buf[2 * slot] = hw; k3m_where_buf[2 * slot + 1] = xcc; }
    }
#endif

    // ---- K split: groups 1 .. KSP-1 hand their accumulators to group 0 through LDS (the stage buffers are free now); group 0
    //      adds them in group order, so the summation tree is fixed ----
    if constexpr (KSP > 1) {
        constexpr int GRP_FLOATS = WGM * WGN * NTILE * 16 * 64;           // one (virtual) group's accumulators
        static_assert((KSP - 1) * VS * GRP_FLOATS * 4 <= KSP * C::TOTAL, "K-split exchange fits the stage buffers");
        float *xch = (float *)smem + (size_t)wave * (NTILE * 16 * 64) + lane;
        if (grp != 0) {
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (VS > 1) xch[(size_t)((grp - 1) * VS) * GRP_FLOATS + ((i * WNT + j) * 16 + r) * 64] = sav[i][j][r];
                        xch[(size_t)((grp - 1) * VS + VS - 1) * GRP_FLOATS + ((i * WNT + j) * 16 + r) * 64] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (grp != 0) return;
    }
    if constexpr (VS > 1) {                                 // this wave group's own passes first, in order
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[i][j] = sav[i][j] + acc[i][j];
    }
    if constexpr (KSP > 1) {
        constexpr int GRP_FLOATS = WGM * WGN * NTILE * 16 * 64;
        const float *xch = (const float *)smem + (size_t)wave * (NTILE * 16 * 64) + lane;
#pragma unroll
        for (int g = VS; g < KV; ++g)
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[(size_t)(g - VS) * GRP_FLOATS + ((i * WNT + j) * 16 + r) * 64];
    }

    // ---- dst[n][m]: D[row = (r&3) + 8*(r>>2) + 4*hh][col = lane & 31]; descriptor + per-lane offset + uniform row offset ----
    // descriptor based at the workgroup's tile origin: offsets below stay inside 32 bits for any dst size
    (void)dst_bytes;
    const rsrc_t rD = make_rsrc(dst + (size_t)n0 * ldd + m0, 0xFFFFFFFFu);
    const bool full = n0 + C::TN <= N && m0 + C::TM <= M;                    // uniform
    const uint32_t lane_off = (uint32_t)((4 * hh * ldd + l31) * 4);
    if (ep.mode == 3) {
        // the exchange of a row split, fused into the store phase (common.h mm_epilogue mode 3): every element goes to the same position of
        // this rank's own [N][M] buffer (dst) and of every peer's -- one descriptor per destination, the tile's offsets shared
        rsrc_t rP[MM_PUSH_MAX];
#pragma unroll
        for (int k = 0; k < MM_PUSH_MAX; ++k) rP[k] = make_rsrc((k < ep.npush ? ep.push[k] : dst) + (size_t)n0 * ldd + m0, 0xFFFFFFFFu);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int mb = (wm_ * WMT + i) * 32, nb = (wn * WNT + j) * 32;        // relative to (n0, m0)
                if (m0 + mb >= M || n0 + nb >= N) continue;                    // uniform
                const bool mok = m0 + mb + l31 < M;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nr = nb + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][j][r];
                    if (!(mok && n0 + nr + 4 * hh < N)) continue;
                    const uint32_t so = (uint32_t)(nr * ldd + mb) * 4u, vb = __builtin_bit_cast(uint32_t, v);
                    __builtin_amdgcn_raw_buffer_store_b32(vb, rD, (int)lane_off, (int)so, 0);
#pragma unroll
                    for (int k = 0; k < MM_PUSH_MAX; ++k)
                        if (k < ep.npush) __builtin_amdgcn_raw_buffer_store_b32(vb, rP[k], (int)lane_off, (int)so, 0);
                }
            }
        return;
    }
    if (ep.mode != 0) {
        // the add / scale node that follows this mul_mat, applied as the accumulators are stored (common.h mm_epilogue): uniform
        // branch, taken only by the fused seams; one f32 operation per element, the separate kernel's result bit for bit
        const int lda = (int)ep.ld_add, ld2 = (int)ep.ld2;
        const rsrc_t rS = make_rsrc(ep.mode == 1 ? ep.addend + (size_t)n0 * lda + m0 : dst, 0xFFFFFFFFu);
        const rsrc_t rD2 = make_rsrc(ep.mode == 1 ? ep.dst2 + (size_t)n0 * ld2 + m0 : dst, 0xFFFFFFFFu);
        const uint32_t lane_s = (uint32_t)((4 * hh * lda + l31) * 4), lane_2 = (uint32_t)((4 * hh * ld2 + l31) * 4);
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int mb = (wm_ * WMT + i) * 32, nb = (wn * WNT + j) * 32;        // relative to (n0, m0)
                if (m0 + mb >= M || n0 + nb >= N) continue;                    // uniform
                const bool mok = m0 + mb + l31 < M;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nr = nb + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][j][r];
                    if (!(mok && n0 + nr + 4 * hh < N)) continue;
                    if (ep.mode == 2) {
                        const float o = v * ep.scale;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), rD, (int)lane_off, (int)((uint32_t)(nr * ldd + mb) * 4u), 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rD, (int)lane_off, (int)((uint32_t)(nr * ldd + mb) * 4u), 0);
                        const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, (int)lane_s, (int)((uint32_t)(nr * lda + mb) * 4u), 0));
                        const float o = v + a;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), rD2, (int)lane_2, (int)((uint32_t)(nr * ld2 + mb) * 4u), 0);
                    }
                }
            }
        return;
    }
    if (full) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nr = (wn * WNT + j) * 32 + (r & 3) + 8 * (r >> 2), mb = (wm_ * WMT + i) * 32;   // relative to (n0, m0)
                    const float v = acc[i][j][r];           // (a named float: __builtin_bit_cast of a vector ELEMENT reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rD, (int)lane_off,
                                                          (int)((uint32_t)(nr * ldd + mb) * 4u), 0);
                }
    } else {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int mb = (wm_ * WMT + i) * 32, nb = (wn * WNT + j) * 32;        // relative to (n0, m0)
                if (m0 + mb >= M || n0 + nb >= N) continue;                    // uniform
                const bool mok = m0 + mb + l31 < M;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nr = nb + (r & 3) + 8 * (r >> 2);
                    const float v = acc[i][j][r];
                    if (mok && n0 + nr + 4 * hh < N)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rD, (int)lane_off,
                                                              (int)((uint32_t)(nr * ldd + mb) * 4u), 0);
                }
            }
    }
}

// ---- K3s: batches of up to 32 rows (a batched decoder's step), Q4_0 / Q4_1 -------------------------------------------------
// The staged kernel above is a latency chain here: one 32 x 32 tile per wave and k-block, a stage of four k-blocks between two
// barriers, and at every stage's drain point the wave waits for everything it has in flight -- one memory round trip per stage,
// eight of them for K = 4096.  With the weights cold in HBM (a decoder walks through gigabytes of them) that is 17 us for
// 4096 x 4096 x 32 where the 15 MB of bf6 planes are 3 us of HBM time.
// This form has no stages: a workgroup is one 32-row weight tile and KS waves that each take a CONTIGUOUS range of k-blocks,
// in PAIRS.  A wave requests everything its first NP pairs need before anything else -- for K <= 4096 that is its whole range:
// all of the matrix is in flight at once across the grid; longer K refills a pair's slots as soon as its MFMAs have issued --
// in the order it will consume it (loads return in order: the wait in front of a pair is exactly for that pair's data).
//   * weights: the MX operand wants a block's fragment in BOTH lane halves (the halves are the two digit groups of the
//     activations).  Loading it twice doubles the traffic through the texture path, so lanes 0..31 fetch block i and lanes
//     32..63 block i + 1 of the same rows in ONE instruction, and v_permlane32_swap hands each half the other's copy: half the
//     load instructions, half the registers that wait for data.
//   * activations: straight from the K1 image in L2 into registers (no reuse between waves that LDS would serve);
//   * row scales: the wave's own slice, loaded by itself into its own LDS slice -- no barrier before the end;
//   * the waves' sums are added in wave order through LDS: a KS-way tree, fixed by N and K like every other form's.
// Arithmetic per block as above: one MX MFMA, then acc += (sumi * d1) * d0 (Ggml.cs:1158); Q4_1's min term per pair of blocks as
// a 32x32x2 f32 MFMA in front of the pair's scale-accumulates.  Everything is builtins and plain C: the compiler counts the
// loads and pads the MFMA hazards.
#ifdef K3S_TRACE            // tools/k3s_trace.hip: time stamps (100 MHz) of wave 0 / wave KS-1 of every workgroup at the phase ends
__device__ unsigned long long k3s_trace_buf[2048 * 2 * 8];
#define K3S_STAMP(k) do { if (lane == 0 && (wave == 0 || wave == KS - 1) && blockIdx.x < 2048) \
        k3s_trace_buf[((size_t)blockIdx.x * 2 + (wave != 0)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define K3S_STAMP(k) do { } while (0)
#endif
// (wg = the workgroup's index inside its matrix: blockIdx.x for one matrix, see the multi-matrix entry below)
template <int TYPE, int KS, int NP, bool ROT, int WMT>
__device__ __forceinline__
void gemm_qmx_small_body(const uint8_t *__restrict__ w6a, const uint8_t *__restrict__ w6b, const float *__restrict__ wd,
                         const float *__restrict__ wm, const uint8_t *__restrict__ a6, const float *__restrict__ ad,
                         const float *__restrict__ asd, float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbkp,
                         int nloc, int ldd, uint32_t w6a_bytes, uint32_t wd_bytes, uint32_t a_bytes, uint32_t ad_bytes,
                         const mm_epilogue &ep, int wg, int ntw) {
    static_assert(TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q4_1, "one weight digit per block");
    constexpr bool Q41 = TYPE == GGML_TYPE_Q4_1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    K3S_STAMP(0);
    int rt, ct;
    k3s_tile_of(wg, ntw, (N + 31) / 32, rt, ct);           // (XCD-aware: the column tiles of a row tile behind one L2, common.h)
    const int m0 = rt * 32 * WMT;                           // WMT 32-row weight tiles per workgroup: one activation fragment serves them all
    const int n0 = ct * 32;                                 // ... and one 32-column slice of src1 (33..64 rows: two workgroups per tile group)
    const int kb0 = wave * nloc;                            // this wave's k-blocks: kb0 .. kb0 + nloc - 1 (nloc even; the planes are zero
    const int npair = nloc >> 1;                            // past the end of K, the descriptors' range check covers the rest)
    // ---- descriptors and per-lane offsets (planes [nbk][Mpad][16 | 8 | 4]; image per k-block [half][Npad][16] then [half][Npad][8]) ----
    const rsrc_t rWa = make_rsrc(w6a, w6a_bytes), rWb = make_rsrc(w6b, w6a_bytes / 2), rWd = make_rsrc(wd, wd_bytes);
    const rsrc_t rWm = make_rsrc(Q41 ? (const void *)wm : (const void *)wd, wd_bytes);
    const rsrc_t rA = make_rsrc(a6, a_bytes), rAs = make_rsrc(Q41 ? asd : ad, ad_bytes);
    const uint32_t voffS = (uint32_t)((hh * Npad + n0 + l31) * 4);
    const int mrow = m0 + l31;
    const uint32_t wa_blk = (uint32_t)(Mpad * 16), wb_blk = (uint32_t)(Mpad * 8), d_blk = (uint32_t)(Mpad * 4);
    // lane half hh takes block (pair's first) + hh
    const uint32_t offA = (uint32_t)(mrow * 16) + hh * wa_blk, offB = (uint32_t)(mrow * 8) + hh * wb_blk, offD = (uint32_t)(mrow * 4) + hh * d_blk;
    const uint32_t a_blk = (uint32_t)Npad * 48u;
    const uint32_t voff16 = (uint32_t)((hh * Npad + n0 + l31) * 16), voff8 = (uint32_t)(32 * Npad + (hh * Npad + n0 + l31) * 8);

    struct WP { u32x4 lo[WMT]; u32x2 hi[WMT]; float d[WMT]; float mn[WMT]; float s; };   // lanes 0..31: the pair's first block, lanes 32..63: its second
    struct AF { u32x4 lo; u32x2 hi; };
    WP wp[NP];
    AF af[2 * NP];
    auto load_pair = [&](WP &f, AF &a0, AF &a1, int pr) {   // pair pr of this wave (clamped to its last: never another wave's blocks)
        const int kb = kb0 + 2 * (pr < npair ? pr : npair - 1);
        a0.lo = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voff16, (int)((uint32_t)kb * a_blk), 0);
        a0.hi = __builtin_amdgcn_raw_buffer_load_b64(rA, (int)voff8, (int)((uint32_t)kb * a_blk), 0);
        a1.lo = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voff16, (int)((uint32_t)(kb + 1) * a_blk), 0);
        a1.hi = __builtin_amdgcn_raw_buffer_load_b64(rA, (int)voff8, (int)((uint32_t)(kb + 1) * a_blk), 0);
#pragma unroll
        for (int t = 0; t < WMT; ++t) {                      // (tiles are 32 rows apart: a constant the instruction's offset field takes)
            f.lo[t] = __builtin_amdgcn_raw_buffer_load_b128(rWa, (int)(offA + 512u * t), (int)((uint32_t)kb * wa_blk), 0);
            f.hi[t] = __builtin_amdgcn_raw_buffer_load_b64(rWb, (int)(offB + 256u * t), (int)((uint32_t)kb * wb_blk), 0);
            f.d[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWd, (int)(offD + 128u * t), (int)((uint32_t)kb * d_blk), 0));
            if constexpr (Q41) f.mn[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWm, (int)(offD + 128u * t), (int)((uint32_t)kb * d_blk), 0));
        }
        if constexpr (Q41) {
            // d1 * sum(a) of column l31, block kb + hh (K1's second plane): the A operand of the pair's min-term MFMA.  A slot that
            // repeats the wave's last pair must add nothing: its offset points past the plane, the range check returns 0.
            f.s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rAs, (int)voffS, pr < npair ? (int)((uint32_t)kb * (uint32_t)(Npad * 4)) : 0x7FFFFFF0, 0));
        }
    };

    // ---- this wave's slice of the row scales: rows x 32 floats, straight into its own LDS slice.
    //      Rows past the wave's range are ZERO: a slot that holds a clamped (repeated) pair then adds (sumi * 0) * d0 = +0. ----
    const int trows = ROT ? nloc : 2 * NP;
    float *const tabD = (float *)smem + (size_t)wave * trows * 32;
    constexpr int TP = 8;                                   // float4 pieces per lane and round: 64 table rows
    f32x4 td[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int idx = lane + 64 * j, b = idx >> 3, c4 = idx & 7;
        const bool ok = b < nloc && kb0 + b < nbkp;
        const size_t e = (size_t)(kb0 + (ok ? b : 0)) * Npad + n0 + 4 * c4;
        td[j] = ok ? *(const f32x4 *)(ad + e) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    static_for<NP>([&](auto uc) { constexpr int u = decltype(uc)::value; load_pair(wp[u], af[2 * u], af[2 * u + 1], u); });
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int idx = lane + 64 * j;
        if (idx < trows * 8) {
            *(f32x4 *)(tabD + 4 * idx) = td[j];
        }
    }
    if (trows > 64) {                                       // (r4: a second round of pieces -- 65 .. 128 k-blocks per wave, K up to 32768; it was K <= 16384)
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = lane + 64 * (j + TP), b = idx >> 3, c4 = idx & 7;
            const bool ok = b < nloc && kb0 + b < nbkp;
            const size_t e = (size_t)(kb0 + (ok ? b : 0)) * Npad + n0 + 4 * c4;
            td[j] = ok ? *(const f32x4 *)(ad + e) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = lane + 64 * (j + TP);
            if (idx < trows * 8) *(f32x4 *)(tabD + 4 * idx) = td[j];
        }
    }
    // (same wave wrote and reads: LDS operations of one wave complete in order; the compiler needs the fence)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    K3S_STAMP(1);
    f32x16 acc[WMT];
#pragma unroll
    for (int t = 0; t < WMT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int scale_a = hh ? 127 : 131;                     // E8M0: K group 0 (the ah digits) carries 2^4

    // v_permlane32_swap: x in both operands -> {x's lower half in both halves, x's upper half in both halves}
    auto both = [](uint32_t x, uint32_t &b0, uint32_t &b1) {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        b0 = r[0]; b1 = r[1];
    };
    // pair pr of the wave (blocks i = 2 pr, i + 1) out of slot u: [min term of the pair,] block i, block i + 1
    auto pair = [&](int pr, auto uc) {
        constexpr int u = decltype(uc)::value;
        WP &w = wp[u];
        AF &a0 = af[2 * u], &a1 = af[2 * u + 1];
        const i32x8 A0 = {(int)a0.lo[0], (int)a0.lo[1], (int)a0.lo[2], (int)a0.lo[3], (int)a0.hi[0], (int)a0.hi[1], 0, 0};
        const i32x8 A1 = {(int)a1.lo[0], (int)a1.lo[1], (int)a1.lo[2], (int)a1.lo[3], (int)a1.hi[0], (int)a1.hi[1], 0, 0};
        const int i = 2 * pr;
        if constexpr (WMT >= 4) {
            // four tiles behind one activation fragment (more than 512 tiles): a tile's two MFMAs, then its scale-accumulates, tile by
            // tile -- the products of all four at once would be 128 registers; the row scales of the pair stay in registers instead
            const float *dp = tabD + i * 32 + 4 * hh;
            f32x4 da0[4], da1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { da0[q] = *(const f32x4 *)(dp + 8 * q); da1[q] = *(const f32x4 *)(dp + 32 + 8 * q); }
#pragma unroll
            for (int t = 0; t < WMT; ++t) {
                uint32_t b0[6], b1[6];
                both(w.lo[t][0], b0[0], b1[0]); both(w.lo[t][1], b0[1], b1[1]); both(w.lo[t][2], b0[2], b1[2]); both(w.lo[t][3], b0[3], b1[3]);
                both(w.hi[t][0], b0[4], b1[4]); both(w.hi[t][1], b0[5], b1[5]);
                uint32_t d0u, d1u;
                both(__builtin_bit_cast(uint32_t, w.d[t]), d0u, d1u);
                const float e0 = __builtin_bit_cast(float, d0u), e1 = __builtin_bit_cast(float, d1u);
                const i32x8 B0 = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b0[4], (int)b0[5], 0, 0};
                const i32x8 B1 = {(int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3], (int)b1[4], (int)b1[5], 0, 0};
                const f32x16 x0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, B0, zero, 3, 3, 0, scale_a, 0, 127);
                const f32x16 x1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, B1, zero, 3, 3, 0, scale_a, 0, 127);
                if constexpr (Q41) {
                    const float m8 = fmaf(8.0f, w.d[t], w.mn[t]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.s, m8, acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = x0[4 * q + e] * da0[q][e];
                        acc[t][4 * q + e] = __builtin_fmaf(x, e0, acc[t][4 * q + e]);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = x1[4 * q + e] * da1[q][e];
                        acc[t][4 * q + e] = __builtin_fmaf(x, e1, acc[t][4 * q + e]);
                    }
            }
            if constexpr (ROT) load_pair(w, a0, a1, pr + NP);
            return;
        }
        f32x16 t0[WMT], t1[WMT];
        float d0[WMT], d1[WMT];
#pragma unroll
        for (int t = 0; t < WMT; ++t) {
            uint32_t b0[6], b1[6];
            both(w.lo[t][0], b0[0], b1[0]); both(w.lo[t][1], b0[1], b1[1]); both(w.lo[t][2], b0[2], b1[2]); both(w.lo[t][3], b0[3], b1[3]);
            both(w.hi[t][0], b0[4], b1[4]); both(w.hi[t][1], b0[5], b1[5]);
            uint32_t d0u, d1u;
            both(__builtin_bit_cast(uint32_t, w.d[t]), d0u, d1u);
            d0[t] = __builtin_bit_cast(float, d0u); d1[t] = __builtin_bit_cast(float, d1u);
            const i32x8 B0 = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b0[4], (int)b0[5], 0, 0};
            const i32x8 B1 = {(int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3], (int)b1[4], (int)b1[5], 0, 0};
            t0[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, B0, zero, 3, 3, 0, scale_a, 0, 127);
            t1[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, B1, zero, 3, 3, 0, scale_a, 0, 127);
            if constexpr (Q41) {
                // + (m0 + 8 d0) * (d1 * sum(a)) per block (Ggml.cs:1190-1196 factorised; nib = (nib - 8) + 8): the two blocks of the
                // pair are the K = 2 of one f32 MFMA -- block i in lane half 0, block i + 1 in lane half 1, which is how they were loaded
                const float m8 = fmaf(8.0f, w.d[t], w.mn[t]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.s, m8, acc[t], 0, 0, 0);
            }
        }
        if constexpr (ROT) load_pair(w, a0, a1, pr + NP);   // the fragments are in the MFMAs' hands: the slot takes the pair NP further on
        const float *dp0 = tabD + i * 32 + 4 * hh, *dp1 = dp0 + 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 da = *(const f32x4 *)(dp0 + 8 * q);
#pragma unroll
            for (int t = 0; t < WMT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = t0[t][4 * q + e] * da[e];
                    acc[t][4 * q + e] = __builtin_fmaf(x, d0[t], acc[t][4 * q + e]);
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 da = *(const f32x4 *)(dp1 + 8 * q);
#pragma unroll
            for (int t = 0; t < WMT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = t1[t][4 * q + e] * da[e];
                    acc[t][4 * q + e] = __builtin_fmaf(x, d1[t], acc[t][4 * q + e]);
                }
        }
    };
    if constexpr (!ROT) {                                    // the whole range sits in the slots (npair <= NP; slots past it: zero table rows)
        static_for<NP>([&](auto pc) { pair(decltype(pc)::value, pc); });
    } else {
        // whole rounds of NP pairs, then the rest (a guard around every pair would cost the exact waits: the compiler merges the
        // counters of the two paths at every join)
        int base = 0;
        for (; base + NP <= npair; base += NP)
            static_for<NP>([&](auto pc) { pair(base + decltype(pc)::value, pc); });
        static_for<NP>([&](auto pc) { if (base + decltype(pc)::value < npair) pair(base + decltype(pc)::value, pc); });
    }

    // ---- the waves' sums, added in wave order (the tables are dead: every wave is past its loop at the first barrier).  Every wave
    //      takes its share of the result rows a lane holds: the additions of one element are the same, in the same order, whoever
    //      makes them (one wave adding all of it: 1.4 us after the barrier; this way 0.5) ----
#ifdef K3S_TRACE
    asm volatile("" : "+v"(acc[0]));
    K3S_STAMP(2);
#endif
    __syncthreads();
    K3S_STAMP(3);
    float *xch = (float *)smem + lane;
#pragma unroll
    for (int t = 0; t < WMT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(size_t)((wave * WMT + t) * 16 + r) * 64] = acc[t][r];
    __syncthreads();
    K3S_STAMP(4);
    static_assert((16 * WMT) % KS == 0, "rows per wave");   // (r5: three tiles per workgroup too -- six result rows per wave, spanning two tiles)
    constexpr int RW = 16 * WMT / KS;
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int rr = wave * RW + k, t = rr / 16, r = rr % 16;   // (uniform)
        float v = xch[(size_t)(t * 16 + r) * 64];
#pragma unroll
        for (int g = 1; g < KS; ++g) v += xch[(size_t)((g * WMT + t) * 16 + r) * 64];
        // ---- dst[n][m]: D[row = (r&3) + 8*(r>>2) + 4*hh][col = lane & 31] ----
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
    K3S_STAMP(6);
}

template <int TYPE, int KS, int NP, bool ROT, int WMT>
__global__ __launch_bounds__(KS * 64, 1)
void gemm_qmx_small_kernel(const uint8_t *__restrict__ w6a, const uint8_t *__restrict__ w6b, const float *__restrict__ wd,
                           const float *__restrict__ wm, const uint8_t *__restrict__ a6, const float *__restrict__ ad,
                           const float *__restrict__ asd, float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nbkp,
                           int nloc, int ldd, uint32_t w6a_bytes, uint32_t wd_bytes, uint32_t a_bytes, uint32_t ad_bytes,
                           const mm_epilogue ep, int ntw) {
    gemm_qmx_small_body<TYPE, KS, NP, ROT, WMT>(w6a, w6b, wd, wm, a6, ad, asd, dst, M, N, Mpad, Npad, nbkp, nloc, ldd, w6a_bytes, wd_bytes, a_bytes,
                                                ad_bytes, ep, (int)blockIdx.x, ntw);
}

// ---- K3s on 16-ROW tiles (r5, Q4_0): the form above for matrices whose 32-row tiles do not fill the chip ------------------------------------
// 4096 rows at up to 32 src1 rows are 128 workgroups on 256 CUs: half of the chip pulls the whole matrix through its memory path (a flat
// 8 us at 4096 x 4096 for 1 .. 32 rows where the planes are 2.5 us of HBM time).  Here a workgroup owns a 16-row weight tile x NCT 16-column
// slices of src1 on v_mfma_scale_f32_16x16x128_f8f6f4.  That instruction sums over FOUR 32-element K groups (lane l: row / column l & 15,
// K group l >> 4) where a quant block fills two (the ah digits with E8M0 scale 2^4, the al digits), and the reference wants every block's
// integer sum apart -- so a PAIR of blocks rides in one operand set: K groups 0 / 1 = block i, 2 / 3 = block i + 1, and the pair takes two
// MFMAs whose WEIGHT operand is zero in the other block's lanes (masked once per pair, whatever the number of column slices): exact zeros,
// exact integer sums, the statement of the 32-row form per block.
// THE SUMMATION TREE IS THE 32-ROW FORM'S (KS = 8 contiguous ranges of nloc k-blocks, pairs in ascending order, block i before block i + 1,
// acc += (sumi * d1) * d0 per block, the eight partial sums added in wave order): geometry that may follow M
// (tests/test_gpu_fullsize.py test_k3s_16_row_tiles_are_bitwise_the_32_row_form).
template <int KS, int NP, bool ROT, int NCT, bool Q41>
__device__ __forceinline__
void gemm_qmx_small16_body(const uint8_t *__restrict__ w6a, const uint8_t *__restrict__ w6b, const float *__restrict__ wd, const float *__restrict__ wm,
                           const uint8_t *__restrict__ a6, const float *__restrict__ ad, const float *__restrict__ asd, float *__restrict__ dst, int M, int N, int Mpad, int Npad,
                           int nbkp, int nloc, int ldd, uint32_t w6a_bytes, uint32_t wd_bytes, uint32_t a_bytes, const mm_epilogue &ep, int wg, int ntw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, kg = lane >> 4;              // K group: 0 / 1 = the pair's first block (ah / al digits), 2 / 3 = its second
    constexpr int TN = 16 * NCT;
    int rt, ct;
    k3s_tile_of(wg, ntw, (N + TN - 1) / TN, rt, ct);
    const int m0 = rt * 16, n0 = ct * TN;
    const int kb0 = wave * nloc, npair = nloc >> 1;
    const rsrc_t rWa = make_rsrc(w6a, w6a_bytes), rWb = make_rsrc(w6b, w6a_bytes / 2), rWd = make_rsrc(wd, wd_bytes);
    const rsrc_t rA = make_rsrc(a6, a_bytes);
    const rsrc_t rWm = make_rsrc(Q41 ? (const void *)wm : (const void *)wd, wd_bytes), rAs = make_rsrc(Q41 ? asd : ad, (uint32_t)nbkp * (uint32_t)Npad * 4u);
    const uint32_t wa_blk = (uint32_t)(Mpad * 16), wb_blk = (uint32_t)(Mpad * 8), d_blk = (uint32_t)(Mpad * 4), a_blk = (uint32_t)Npad * 48u;
    const uint32_t second = (uint32_t)(kg >> 1), half = (uint32_t)(kg & 1);
    const int mrow = m0 + l15;
    const uint32_t offA = (uint32_t)(mrow * 16) + second * wa_blk, offB = (uint32_t)(mrow * 8) + second * wb_blk, offD = (uint32_t)(mrow * 4);
    const uint32_t voff16 = (uint32_t)((half * Npad + n0 + l15) * 16) + second * a_blk;
    const uint32_t voff8 = (uint32_t)(32 * Npad + (half * Npad + n0 + l15) * 8) + second * a_blk;
    const uint32_t voffS = (uint32_t)((n0 + l15) * 4) + second * (uint32_t)(Npad * 4), offDm = offD + second * d_blk;   // (Q4_1: the lane's own block of the pair)

    struct WP { u32x4 lo; u32x2 hi; float d0, d1; float mn; };   // lanes of K groups 0 / 1: the pair's first block, 2 / 3: its second; both blocks' row scales in every lane (Q4_1: + the min of the lane's own block)
    struct AF { u32x4 lo; u32x2 hi; float s; };             // (Q4_1: d1 * sum(a) of the lane's column and block -- K1's second plane)
    WP wp[NP];
    AF af[NP][NCT];
    auto load_pair = [&](WP &f, AF (&a)[NCT], int pr) {     // pair pr of this wave (clamped to its last: its table rows past the range are zero)
        const int kb = kb0 + 2 * (pr < npair ? pr : npair - 1);
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            a[c].lo = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(voff16 + 256u * c), (int)((uint32_t)kb * a_blk), 0);
            a[c].hi = __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(voff8 + 128u * c), (int)((uint32_t)kb * a_blk), 0);
            // (Q4_1) a slot that repeats the wave's last pair must add nothing: its offset points past the plane, the range check returns 0
            if constexpr (Q41) a[c].s = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rAs, (int)(voffS + 64u * c), pr < npair ? (int)((uint32_t)kb * (uint32_t)(Npad * 4)) : 0x7FFFFFF0, 0));
        }
        f.lo = __builtin_amdgcn_raw_buffer_load_b128(rWa, (int)offA, (int)((uint32_t)kb * wa_blk), 0);
        f.hi = __builtin_amdgcn_raw_buffer_load_b64(rWb, (int)offB, (int)((uint32_t)kb * wb_blk), 0);
        // (every lane needs the row scale of BOTH blocks: two 4-byte loads.  One load per lane group + v_permlane32_swap, the 32-row form's idiom,
        // came out of hipcc with the swap's second result dropped and block i's scale on both blocks: tools/experiments/dbg_tile16.py)
        f.d0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWd, (int)offD, (int)((uint32_t)kb * d_blk), 0));
        f.d1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWd, (int)offD, (int)((uint32_t)(kb + 1) * d_blk), 0));
        if constexpr (Q41) f.mn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWm, (int)offDm, (int)((uint32_t)kb * d_blk), 0));
    };

    // ---- this wave's slice of the row scales: rows x TN floats into its own LDS slice; rows past the wave's range or past K are ZERO ----
    const int trows = ROT ? nloc : 2 * NP;
    float *const tabD = (float *)smem + (size_t)wave * trows * TN;
    constexpr int PPR = TN / 4, TP = 8;
    for (int base = 0; base < trows * PPR; base += 64 * TP) {
        f32x4 td[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = base + lane + 64 * j, b = idx / PPR, c4 = idx % PPR;
            const bool ok = b < nloc && kb0 + b < nbkp;
            td[j] = ok ? *(const f32x4 *)(ad + (size_t)(kb0 + b) * Npad + n0 + 4 * c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        if (base == 0) static_for<NP>([&](auto uc) { constexpr int u = decltype(uc)::value; load_pair(wp[u], af[u], u); });
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int idx = base + lane + 64 * j;
            if (idx < trows * PPR) *(f32x4 *)(tabD + 4 * idx) = td[j];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    f32x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int scale_a = half ? 127 : 131;                   // E8M0: the ah digits carry 2^4
    const bool first_blk = kg < 2;
    auto pair = [&](int pr, auto uc) {
        constexpr int u = decltype(uc)::value;
        WP &w = wp[u];
        const uint32_t wr[6] = {w.lo[0], w.lo[1], w.lo[2], w.lo[3], w.hi[0], w.hi[1]};
        i32x8 B0, B1;
#pragma unroll
        for (int k = 0; k < 6; ++k) { B0[k] = first_blk ? (int)wr[k] : 0; B1[k] = first_blk ? 0 : (int)wr[k]; }
        B0[6] = B0[7] = B1[6] = B1[7] = 0;
        const float d0 = w.d0, d1 = w.d1;
        const int i = 2 * pr;
        const float *dp0 = tabD + i * TN + 4 * kg, *dp1 = dp0 + TN;
        f32x4 t0[NCT], t1[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const i32x8 A = {(int)af[u][c].lo[0], (int)af[u][c].lo[1], (int)af[u][c].lo[2], (int)af[u][c].lo[3], (int)af[u][c].hi[0], (int)af[u][c].hi[1], 0, 0};
            t0[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B0, zero4, 3, 3, 0, scale_a, 0, 127);
            t1[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B1, zero4, 3, 3, 0, scale_a, 0, 127);
        }
        if constexpr (Q41) {
            // + (m0 + 8 d0) * (d1 * sum(a)) per block (Ggml.cs:1190-1196 factorised; nib = (nib - 8) + 8), IN FRONT of the pair's block terms as in
            // the 32-row form: there the pair's two blocks are the K = 2 of one v_mfma_f32_32x32x2_f32, here k = 0 and k = 2 of one
            // v_mfma_f32_16x16x4_f32 (lane group kg supplies k = kg: the ah-digit groups 0 / 2 carry the blocks, the al-digit groups 1 / 3 zeros).
            // The instruction is a k-ordered fmaf chain (cdna_hip_programming.md), and fma(0, 0, x) = x for every x the accumulators can hold
            // (they start at +0 and never become -0): the same two fmaf in the same order, the same bits.
            const float m8 = half ? 0.0f : fmaf(8.0f, second ? d1 : d0, w.mn);
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(half ? 0.0f : af[u][c].s, m8, acc[c], 0, 0, 0);
        }
        if constexpr (ROT) load_pair(w, af[u], pr + NP);     // the fragments are in the MFMAs' hands: the slot takes the pair NP further on
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const f32x4 da = *(const f32x4 *)(dp0 + 16 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = t0[c][e] * da[e]; acc[c][e] = __builtin_fmaf(x, d0, acc[c][e]); }   // (sumi * d1) * d0, Ggml.cs:1158
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const f32x4 da = *(const f32x4 *)(dp1 + 16 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = t1[c][e] * da[e]; acc[c][e] = __builtin_fmaf(x, d1, acc[c][e]); }
        }
        // Pairs in program order (longer K only): left alone hipcc gathers a round's sixteen MFMAs, then its eight refills, and waits for nearly
        // all of them at the top of the next round (s_waitcnt vmcnt(9) five times: ONE pair in flight) -- with one 16-column slice a pair's
        // arithmetic is too little to hold the schedule apart.  Pinned, the wait in front of a pair is for that pair alone and the seven
        // others stay in flight (4096 x 11008 x 16: 17.9 -> see DESIGN.md section 11).
        if constexpr (ROT) __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (!ROT) {
        static_for<NP>([&](auto pc) { pair(decltype(pc)::value, pc); });
    } else {
        int base = 0;
        for (; base + NP <= npair; base += NP)
            static_for<NP>([&](auto pc) { pair(base + decltype(pc)::value, pc); });
        static_for<NP>([&](auto pc) { if (base + decltype(pc)::value < npair) pair(base + decltype(pc)::value, pc); });
    }

    // ---- the waves' sums in wave order (the 32-row form's tree), the NCT * 4 result registers dealt over the waves ----
    __syncthreads();
    float *xch = (float *)smem + lane;
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
        const int n = n0 + 16 * c + 4 * kg + r, m = m0 + l15;
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

template <int KS, int NP, bool ROT, int NCT, bool Q41>
__global__ __launch_bounds__(KS * 64, 1)
void gemm_qmx_small16_kernel(const uint8_t *__restrict__ w6a, const uint8_t *__restrict__ w6b, const float *__restrict__ wd, const float *__restrict__ wm,
                             const uint8_t *__restrict__ a6, const float *__restrict__ ad, const float *__restrict__ asd, float *__restrict__ dst, int M, int N,
                             int Mpad, int Npad, int nbkp, int nloc, int ldd, uint32_t w6a_bytes, uint32_t wd_bytes, uint32_t a_bytes, const mm_epilogue ep, int ntw) {
    gemm_qmx_small16_body<KS, NP, ROT, NCT, Q41>(w6a, w6b, wd, wm, a6, ad, asd, dst, M, N, Mpad, Npad, nbkp, nloc, ldd, w6a_bytes, wd_bytes, a_bytes, ep,
                                                 (int)blockIdx.x, ntw);
}

// Several weight matrices behind ONE activation image (q / k / v, gate / up of a batched decoder's step): the workgroups of all of
// them in one launch -- 4096 rows are 128 tiles, half of the chip; three such matrices fill it.  A workgroup picks its matrix from
// its index (uniform, once); everything after that is the single-matrix kernel, so every matrix gets the bits of its own call.
struct mxs_set {
    int n; int wg_end[4];                                   // workgroups [wg_end[i - 1], wg_end[i]) belong to matrix i
    const uint8_t *a[4], *b[4]; const float *d[4], *m[4]; float *dst[4];
    int M[4], Mpad[4], ldd[4]; uint32_t wa_bytes[4], wd_bytes[4];
};
template <int TYPE, int KS, int NP, bool ROT, int WMT>
__global__ __launch_bounds__(KS * 64, 1)
void gemm_qmx_small_multi_kernel(const mxs_set ws, const uint8_t *__restrict__ a6, const float *__restrict__ ad, const float *__restrict__ asd,
                                 int N, int Npad, int nbkp, int nloc, uint32_t a_bytes, uint32_t ad_bytes, int ncol) {
    const int b = (int)blockIdx.x;
    const int k = (b >= ws.wg_end[0]) + (b >= ws.wg_end[1]) + (b >= ws.wg_end[2]);
    const int first = k == 0 ? 0 : k == 1 ? ws.wg_end[0] : k == 2 ? ws.wg_end[1] : ws.wg_end[2];
#define MXS(f) (k == 0 ? ws.f[0] : k == 1 ? ws.f[1] : k == 2 ? ws.f[2] : ws.f[3])
    const mm_epilogue ep{0, nullptr, 0, nullptr, 0, 1.0f};
    gemm_qmx_small_body<TYPE, KS, NP, ROT, WMT>(MXS(a), MXS(b), MXS(d), MXS(m), a6, ad, asd, MXS(dst), MXS(M), N, MXS(Mpad), Npad, nbkp, nloc, MXS(ldd),
                                                MXS(wa_bytes), MXS(wd_bytes), a_bytes, ad_bytes, ep, b - first, ((k == 0 ? ws.wg_end[0] : k == 1 ? ws.wg_end[1] : k == 2 ? ws.wg_end[2] : ws.wg_end[3]) - first) / ncol);
#undef MXS
}

// the epilogue of the call in flight on this host thread (set by launch_gemm_qmx around launch_typed: the tile-form selection
// below has a dozen call sites, the epilogue concerns none of them)
thread_local mm_epilogue t_epilogue = {0, nullptr, 0, nullptr, 0, 1.0f};

template <int TYPE, int WMT, int WNT, int WGM, int WGN, int KB, int FB, int KSP = 1, int VS = 1>
hipError_t launch_cfg(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    using C = Cfg<TYPE, WMT, WNT, WGM, WGN, KB>;
    auto kern = gemm_qmx_kernel<TYPE, WMT, WNT, WGM, WGN, KB, FB, KSP, VS>;
        static PerDeviceOnce once;   // per kernel instantiation; the attribute is set once per device
    const hipError_t attr = once.max_dynamic_lds((const void *)kern, C::TOTAL * KSP);
    if (attr != hipSuccess) return attr;
    if (w->Mpad % C::TM != 0 || p.Npad % C::TN != 0 || !w->q6a || !w->q6b) return hipErrorInvalidValue;
    const int tiles_m = (int)((w->M + C::TM - 1) / C::TM), tiles_n = (int)((N + C::TN - 1) / C::TN);
    dim3 grid((unsigned)(tiles_m * tiles_n));
    const int nstages = (int)((w->nbk + KB - 1) / KB);      // planes and image are zero-padded to whole stages (KB | K_STAGE_PAD)
    static_assert(K_STAGE_PAD % KB == 0, "stage padding");
    const uint64_t nba = (uint64_t)pad_kblocks(w->nbk);
    constexpr int NF = (TYPE == GGML_TYPE_Q4_0 || TYPE == GGML_TYPE_Q4_1) ? 1 : 2;
    const uint64_t wq_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 16 * NF;
    const uint64_t wd_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 4, a_bytes = nba * 48 * (uint64_t)p.Npad;
    const uint64_t ad_bytes = nba * (uint64_t)p.Npad * 4, dst_bytes = ((uint64_t)(N - 1) * (uint64_t)ldd + (uint64_t)w->M) * 4;
    constexpr uint64_t LIM = 0xFFFFFFFFull;                  // 32-bit buffer offsets
    if (wq_bytes > LIM || a_bytes > LIM || (uint64_t)C::TN * (uint64_t)ldd * 4 > LIM) return hipErrorNotSupported;   // api.cpp routes such shapes to gemm_q.hip
    kern<<<grid, C::NT * KSP, C::TOTAL * KSP, st>>>(w->q6a, w->q6b, w->d, w->m, (const uint8_t *)p.a8, p.ad, (const float *)p.as, dst, (int)w->M,
                                        (int)N, (int)w->Mpad, (int)p.Npad, nstages, (int)ldd, tiles_m, tiles_n, (uint32_t)wq_bytes,
                                        (uint32_t)wd_bytes, (uint32_t)a_bytes, (uint32_t)ad_bytes, (uint32_t)dst_bytes, t_epilogue);
    return hipGetLastError();
}

// K3s launch: KS = 8 waves per 32-row tile, each a contiguous eighth of K in pairs of blocks; the slots hold a wave's whole range
// for K <= 4096 (8 pairs) and K <= 2048 (4 pairs), longer K refills them in turn.  Chosen by N and K alone.
template <int TYPE>
hipError_t launch_small(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    constexpr int KS = 8;
    if (!w->q6a || !w->q6b) return hipErrorInvalidValue;
    // geometry and applicability were decided by plan.cpp (plan_k3s_mx): k-blocks per wave, 32-row tiles per workgroup (more tiles than CUs:
    // two of them per workgroup, four beyond 512 tiles -- every workgroup pulls the whole activation image through its CU's vector
    // memory path, and that path is what bounds the form; same blocks in the same order per element: the same bits)
    const int nbkp = (int)pad_kblocks(w->nbk);
    const int nloc = pl.nloc, wmt = pl.wmt;
    const uint64_t nba = (uint64_t)nbkp;
    const uint64_t wq_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 16, wd_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 4;
    const uint64_t a_bytes = nba * 48 * (uint64_t)p.Npad, ad_bytes = nba * (uint64_t)p.Npad * 4;
    if (nloc > 128 || wq_bytes > 0xFFFFFFFFull || a_bytes > 0xFFFFFFFFull) return hipErrorInvalidValue;    // (the plan never sends such a shape here)
    if (pl.tile_m == 16) {
        // r5: 16-row tiles (plan_k3s_mx: where the 32-row tiles leave CUs idle) -- the same tree, NCT 16-column slices per workgroup
        {
            constexpr bool Q41 = TYPE == GGML_TYPE_Q4_1;
            const int nct = pl.tile_n / 16, ncg = (int)((N + pl.tile_n - 1) / pl.tile_n);
            if (p.Npad < (int64_t)pl.tile_n * ncg || w->Mpad % 16 != 0 || (nloc & 1)) return hipErrorInvalidValue;
            if (nct != 1 && nct != 2) return hipErrorInvalidValue;
            if (Q41 && (!w->m || !p.as)) return hipErrorInvalidValue;
            const int np16 = nloc <= 8 ? 4 : 8;                          // pairs in flight per wave (the slots; beyond 16 k-blocks per wave: in turn)
            const bool rot16 = nloc > 2 * np16;
            const int rows16 = rot16 ? nloc : 2 * np16;
            const int tab16 = KS * rows16 * pl.tile_n * 4, xch16 = KS * nct * 4 * 64 * 4;
            const int lds16 = tab16 > xch16 ? tab16 : xch16;
            if (lds16 > 160 * 1024) return hipErrorInvalidValue;
            const int ntw16 = (int)((w->M + 15) / 16);
            dim3 grid16((unsigned)(ntw16 * ncg));
#define K3S16_GO(NP, ROT, NCT) do { \
            auto kern = gemm_qmx_small16_kernel<KS, NP, ROT, NCT, Q41>; \
            static PerDeviceOnce once; \
            const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
            if (attr != hipSuccess) return attr; \
            kern<<<grid16, KS * 64, lds16, st>>>(w->q6a, w->q6b, w->d, w->m, (const uint8_t *)p.a8, p.ad, (const float *)p.as, dst, (int)w->M, (int)N, (int)w->Mpad, \
                                              (int)p.Npad, nbkp, nloc, (int)ldd, (uint32_t)wq_bytes, (uint32_t)wd_bytes, (uint32_t)a_bytes, t_epilogue, ntw16); } while (0)
            if (nct == 2) {
                if (nloc <= 8) K3S16_GO(4, false, 2); else if (nloc <= 16) K3S16_GO(8, false, 2);
                else if constexpr (Q41) K3S16_GO(6, true, 2);            // (Q4_1 carries the min and d1 * sum(a) per slot: eight slots of two slices spill)
                else K3S16_GO(8, true, 2);
            }
            else { if (nloc <= 8) K3S16_GO(4, false, 1); else if (nloc <= 16) K3S16_GO(8, false, 1); else K3S16_GO(8, true, 1); }
#undef K3S16_GO
            return hipGetLastError();
        }
    }
    const int rows = nloc <= 8 ? 8 : nloc <= 16 ? 16 : nloc;
    const int ncol = (int)((N + 31) / 32);                  // 32-column slices of src1: one workgroup per tile group and slice
    if (p.Npad < 32 * ncol) return hipErrorInvalidValue;
    const bool two = wmt == 2;
    const int tab = KS * rows * 32 * 4, xch = KS * wmt * 16 * 64 * 4;
    const int lds = tab > xch ? tab : xch;
    const int ntw = (int)((w->M + 32 * wmt - 1) / (32 * wmt));
    dim3 grid((unsigned)(ntw * ncol));
    if (w->Mpad % (32 * wmt) != 0) return hipErrorInvalidValue;
#define K3S_GO(NP, ROT, WMT) do { \
        auto kern = gemm_qmx_small_kernel<TYPE, KS, NP, ROT, WMT>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid, KS * 64, lds, st>>>(w->q6a, w->q6b, w->d, w->m, (const uint8_t *)p.a8, p.ad, (const float *)p.as, dst, (int)w->M, (int)N, \
                                      (int)w->Mpad, (int)p.Npad, nbkp, nloc, (int)ldd, (uint32_t)wq_bytes, (uint32_t)wd_bytes, (uint32_t)a_bytes, \
                                      (uint32_t)ad_bytes, t_epilogue, ntw); } while (0)
    if (wmt == 4) { if constexpr (TYPE == GGML_TYPE_Q4_0) K3S_GO(2, true, 4); }
    else if (two) { if (nloc <= 8) K3S_GO(4, false, 2); else if constexpr (TYPE == GGML_TYPE_Q4_1) K3S_GO(3, true, 2); else K3S_GO(4, true, 2); }
    else if (nloc <= 8) K3S_GO(4, false, 1);
    else if (nloc <= 16) K3S_GO(8, false, 1);
    else { if constexpr (TYPE == GGML_TYPE_Q4_1) K3S_GO(4, true, 1); else K3S_GO(8, true, 1); }   // (Q4_1 carries three more registers per pair)
#undef K3S_GO
    return hipGetLastError();
}

// K3s for several matrices of one type and K behind one activation image: hipErrorNotSupported where the single-matrix form would
// not run either (the caller then computes them one after the other).  Tiles per workgroup by the tiles of all of them together.
template <int TYPE>
hipError_t launch_small_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st) {
    constexpr int KS = 8;
    const int nbkp = (int)pad_kblocks(w[0]->nbk);
    int nloc = (nbkp + KS - 1) / KS;
    nloc += nloc & 1;
    if (w[0]->nbk < 64 || nloc > 128) return hipErrorNotSupported;
    const uint64_t nba = (uint64_t)nbkp;
    const uint64_t a_bytes = nba * 48 * (uint64_t)p.Npad, ad_bytes = nba * (uint64_t)p.Npad * 4;
    if (a_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
    const int ncol = (int)((N + 31) / 32);
    if (p.Npad < 32 * ncol) return hipErrorNotSupported;
    int64_t t32 = 0;
    for (int i = 0; i < n_w; ++i) t32 += (w[i]->M + 31) / 32 * ncol;
    // r5: THREE tiles per workgroup where that is the fewest that keeps the group inside one round of the chip (gate / up of a 7B model: 2 x 11008 rows are 230 workgroups of 96 rows,
    // 172 of 128): geometry on the form's tree, like the other tile counts.  A 96-row tile may overhang the padded rows: it reads the neighbouring plane's bytes there (inside the
    // buffer, or zeros past it) and stores nothing.
    auto groups = [&](int t) { int64_t g = 0; for (int i = 0; i < n_w; ++i) g += (w[i]->M + 32 * t - 1) / (32 * t) * ncol; return g; };
    const int wmt = t32 <= 256 ? 1 : t32 <= 512 || TYPE == GGML_TYPE_Q4_1 ? 2 : groups(3) <= 256 ? 3 : 4;
    mxs_set ws = {};
    ws.n = n_w;
    int wgs = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < n_w) {
            const ggml_hip_weight *x = w[i];
            const uint64_t wq_bytes = (nba + K_LOOKAHEAD) * (uint64_t)x->Mpad * 16, wd_bytes = (nba + K_LOOKAHEAD) * (uint64_t)x->Mpad * 4;
            if (!x->q6a || !x->q6b || x->nbk != w[0]->nbk || x->Mpad % (wmt == 3 ? 32 : 32 * wmt) != 0 || wq_bytes > 0xFFFFFFFFull) return hipErrorNotSupported;
            wgs += (int)((x->M + 32 * wmt - 1) / (32 * wmt)) * ncol;
            ws.a[i] = x->q6a; ws.b[i] = x->q6b; ws.d[i] = x->d; ws.m[i] = x->m; ws.dst[i] = dst[i];
            ws.M[i] = (int)x->M; ws.Mpad[i] = (int)x->Mpad; ws.ldd[i] = (int)ldd[i]; ws.wa_bytes[i] = (uint32_t)wq_bytes; ws.wd_bytes[i] = (uint32_t)wd_bytes;
        }
        ws.wg_end[i] = wgs;
    }
    const int rows = nloc <= 8 ? 8 : nloc <= 16 ? 16 : nloc;
    const int tab = KS * rows * 32 * 4, xch = KS * wmt * 16 * 64 * 4;
    const int lds = tab > xch ? tab : xch;
    dim3 grid((unsigned)wgs);
#define K3M_GO(NP, ROT, WMT) do { \
        auto kern = gemm_qmx_small_multi_kernel<TYPE, KS, NP, ROT, WMT>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<grid, KS * 64, lds, st>>>(ws, (const uint8_t *)p.a8, p.ad, (const float *)p.as, (int)N, (int)p.Npad, nbkp, nloc, (uint32_t)a_bytes, \
                                      (uint32_t)ad_bytes, ncol); } while (0)
    if (wmt == 4) { if constexpr (TYPE == GGML_TYPE_Q4_0) K3M_GO(2, true, 4); }
    else if (wmt == 3) { if constexpr (TYPE == GGML_TYPE_Q4_0) K3M_GO(2, true, 3); }
    else if (wmt == 2) { if (nloc <= 8) K3M_GO(4, false, 2); else if constexpr (TYPE == GGML_TYPE_Q4_1) K3M_GO(3, true, 2); else K3M_GO(4, true, 2); }
    else if (nloc <= 8) K3M_GO(4, false, 1);
    else if (nloc <= 16) K3M_GO(8, false, 1);
    else { if constexpr (TYPE == GGML_TYPE_Q4_1) K3M_GO(4, true, 1); else K3M_GO(8, true, 1); }
#undef K3M_GO
    return hipGetLastError();
}

#ifndef MX_SMALL_FB
#define MX_SMALL_FB 2                  // weight fragment look-ahead of the small-batch forms (A/B 4: 4096 x 4096 x {16, 64} 13.2 us either way,
                                       // 32000 x 4096 x 32 32 -> 66 us: these forms are bound by their latency chain, not by the prefetch depth)
#endif
// The form was chosen by plan.cpp (plan_mx: by type, N and K -- the tile height by the tile count); the measurements behind the choice:
//   * 256 x 128 (8 tiles per wave) only where the registers allow it (Q4_0) and only above 512 rows; 256 x 128 with waves of 128 x 64
//     (half the LDS reads, weights single-buffered) measured 10 % slower: A/B only;
//   * prompt-sized batches (N <= 512) rarely fill the chip with one wave group per tile: two groups per 128 x 64 tile split K between them
//     (4096 x 4096 x 512: 45.8 -> 31.9 us; 4096 x 11008 x 512: 107 -> 75 us; N = 1024 is 10 % slower that way);
//   * batches up to 128 rows: 32-row weight tiles with K split four ways -- M / 32 x N / 64 workgroups of 4 waves instead of M / 128 x N / 64
//     of 8 (4096 x 4096 x 64 covered 32 CUs); the same tree on taller tiles where those cover the chip (compute us, 32-row x 4 waves | 64-row
//     x 8 | 128-row x 16, N = 64: M = 4096 13 | 14 | 23, M = 11008 29 | 17 | 27, M = 32000 57 | 34 | 32); the 16-wave form only for Q4_0;
//   * up to 64 rows, K >= 2048: the stage-free form K3s (weights cold, staged | K3s: 4096 x 4096 x 32 16.8 | 12.3 us, 4096 x 11008 36.7 | 24.8);
//   * 129 .. 256 rows (Q4_0): the four-way tree really split while the tiles are few (4096 x 4096 x 256 28.5 -> 21 us), else two wave groups
//     running two stage sets each (11008 x 4096 x 256 44 us, 32000 x 4096 x 256 112 against 107 us: the banked sums cost registers);
//   * a short, wide product (a row shard): 64 x 64 tiles of four 1-tile waves wherever the 128 x 64 grid leaves CUs idle
//     (1024 x 4096 x 768 37.0 -> 26.3 us, 512 x 4096 x 1024 36.8 -> 24.1, 512 x 11008 x 2048 94.5 -> 71.8).
template <int TYPE>
hipError_t launch_typed(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    constexpr bool Q40 = TYPE == GGML_TYPE_Q4_0, Q4 = Q40 || TYPE == GGML_TYPE_Q4_1;
    if (pl.family == MMF_K3S_MX) {
        if constexpr (Q4) return launch_small<TYPE>(w, pl, p, N, dst, ldd, st);
        return hipErrorInvalidValue;
    }
    if (pl.family == MMF_K3P_MX) return launch_gemm_qmx_mid(w, pl, p, N, dst, ldd, st, t_epilogue);
    if (pl.family != MMF_MX) return hipErrorInvalidValue;
    switch (pl.form) {
    case MXF_256x128:     if constexpr (Q40) return launch_cfg<TYPE, 2, 4, 4, 1, 4, 2>(w, p, N, dst, ldd, st); break;
    case MXF_256x128_ALT: if constexpr (Q40) return launch_cfg<TYPE, 4, 2, 2, 2, 4, 1>(w, p, N, dst, ldd, st); break;
    case MXF_N32_H64:     return launch_cfg<TYPE, 1, 1, 2, 1, 4, 2, 4>(w, p, N, dst, ldd, st);
    case MXF_N32_H32:     return launch_cfg<TYPE, 1, 1, 1, 1, 4, 2, 4>(w, p, N, dst, ldd, st);
    case MXF_S4_H128:     if constexpr (Q40) return launch_cfg<TYPE, 1, 2, 4, 1, 4, MX_SMALL_FB, 4>(w, p, N, dst, ldd, st); break;
    case MXF_S4_H64:      return launch_cfg<TYPE, 1, 2, 2, 1, 4, MX_SMALL_FB, 4>(w, p, N, dst, ldd, st);
    case MXF_S4_H32:      return launch_cfg<TYPE, 1, 2, 1, 1, 4, MX_SMALL_FB, 4>(w, p, N, dst, ldd, st);
    case MXF_S2V2_H64:    if constexpr (Q40) return launch_cfg<TYPE, 1, 2, 2, 1, 4, 2, 2, 2>(w, p, N, dst, ldd, st); break;
    case MXF_S2_H128:     return launch_cfg<TYPE, 1, 2, 4, 1, 4, 2, 2>(w, p, N, dst, ldd, st);
    case MXF_S2_H64:      return launch_cfg<TYPE, 1, 2, 2, 1, 4, 2, 2>(w, p, N, dst, ldd, st);
    case MXF_128x128:     return launch_cfg<TYPE, 2, 2, 2, 2, 4, 2>(w, p, N, dst, ldd, st);
    case MXF_64x64:       if constexpr (Q40) return launch_cfg<TYPE, 1, 1, 2, 2, 4, 2>(w, p, N, dst, ldd, st); break;
    case MXF_128x64:      return launch_cfg<TYPE, 1, 2, 4, 1, 4, 2>(w, p, N, dst, ldd, st);
    default: break;
    }
    return hipErrorInvalidValue;                            // (a form the plan never gives this type)
}

}  // namespace

hipError_t launch_gemm_qmx_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st) {
    if (n_w < 2 || n_w > 4 || N < 5 || N > 64) return hipErrorNotSupported;
    for (int i = 0; i < n_w; ++i)
        if (!w[i] || w[i]->type != w[0]->type || w[i]->M <= 0 || (uint64_t)32 * (uint64_t)ldd[i] * 4 > 0xFFFFFFFFull) return hipErrorNotSupported;
    switch (w[0]->type) {
    case GGML_TYPE_Q4_0: return launch_small_multi<GGML_TYPE_Q4_0>(w, n_w, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_small_multi<GGML_TYPE_Q4_1>(w, n_w, p, N, dst, ldd, st);
    default: return hipErrorNotSupported;
    }
}

hipError_t launch_gemm_qmx(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st,
                           const mm_epilogue *ep) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    struct Scope {                                          // the epilogue lives for this launch only
        explicit Scope(const mm_epilogue *e) { if (e) t_epilogue = *e; }
        ~Scope() { t_epilogue = mm_epilogue{0, nullptr, 0, nullptr, 0, 1.0f}; }
    } scope(ep);
    if (ep && ep->mode == 1 && ((uint64_t)64 * (uint64_t)ep->ld_add * 4 > 0xFFFFFFFFull || (uint64_t)256 * (uint64_t)ep->ld2 * 4 > 0xFFFFFFFFull))
        return hipErrorNotSupported;                        // (32-bit offsets inside a tile, as for dst)
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0>(w, pl, p, N, dst, ldd, st);
    default: return hipErrorInvalidValue;
    }
}
