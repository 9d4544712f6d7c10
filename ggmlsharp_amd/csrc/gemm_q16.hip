// gemm_q16.hip -- K3: quantized mat-mat for large N, block-scaled, on the f16 matrix cores (v5, register-tile design).
//
// COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698):
//   dst[n*ldd + m] = sum_b (dw[m,b] * da[n,b]) * sumi_b(m,n),   sumi_b = the integer dot of one 32-element block
// (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1136-1159; _q5_0_q8_0 1270-1298; _q8_0_q8_0 1362-1378; _q4_1_q8_1 1176-1198).
//
// Why f16 operands for an integer dot: every operand is a small integer (weights in [-128,127], Q8 activations in
// [-127,127]) and is exact in f16; every product and every 32-term block sum (< 2^19) is exact in the f32
// accumulator of v_mfma_f32_32x32x16_f16.  Two MFMAs (K = 2 x 16 = one quant block) return sumi_b for a 32x32 tile
// bit-exactly and ALREADY IN F32: no v_cvt_f32_i32 (half rate on gfx950).  What bounds this path is the reference's
// own per-block f32 work (Ggml.cs:1158): one multiply and one fma per output per block, i.e. 32 VALU instructions
// next to 2 MFMAs (64 matrix-pipe cycles) per 32x32 tile and k-block.  Everything else is designed to stay off the
// VALU:
//   * activations: K1 writes them as f16, already in MFMA fragment order; they reach LDS by global_load_lds (DMA, no
//     VALU) and a fragment is one conflict-free ds_read_b128;
//   * activation scales da: a broadcast ds_read_b128 hands every lane the 4 row scales of 4 accumulator registers;
//   * weights: each lane loads ITS OWN 8 bytes (16 nibbles) of the resident 4-bit planes straight from L2 into
//     registers and expands them there, once per k-block for a 64 x 128 (m x n) wave tile: 9 VALU per 8 nibbles
//     (see unpack), amortised over 4 n-tiles;
//   * no weight staging through LDS, no 64-bit or integer-multiply address arithmetic in the loop.
//
// Orientation: MFMA rows = src1 rows n (A = activations), MFMA cols = weight rows m (B = weights): a lane owns one
// m, so dst stores are 128-byte segments along m (dst is [n][m], m fastest, Ggml.cs:6692-6697).
//
// Element order inside a block (the MFMA pairs k-slot with k-slot, so any permutation is fine as long as A and B use
// the same one): lane half h and MFMA kk cover the 8 elements e_i = 16h + 8kk + i, i.e. dword kk of the lane's 8
// bytes; k-slots hold [e0, e4, e1, e5, e2, e6, e3, e7] -- what (x & 0x000F000F), (x & 0x00F000F0) yield without a
// shift.  The second mask leaves the nibble 4 bits up, i.e. the weight times 16; K1 stores the matching activations
// divided by 16 (exact in f16), so the products are unchanged.  "Panel" p = 2*kk + h of the activation image holds
// those 8 f16 for every row: image [k-block][panel][row][16 B].
#include "common.h"
#include "plan.h"
#include <cstdlib>
#include <utility>

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// Raw buffer addressing everywhere: a 128-bit descriptor in SGPRs + a per-thread 32-bit offset fixed for the whole kernel
// + a uniform 32-bit offset that advances per k-block / stage.  No 64-bit VGPR addresses, no address VALU in the loop.
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

// (bits & mask) | 0x6400 in each half is the f16 1024 + v (ulp 1 in [1024, 2048)); subtracting 1024 + off is exact.
__device__ __forceinline__ uint32_t magic_sub(uint32_t bits, float off_plus_1024) {
    const f16x2 v = __builtin_bit_cast(f16x2, bits);
    const _Float16 o = (_Float16)off_plus_1024;
    const f16x2 r = v - (f16x2){o, o};
    return __builtin_bit_cast(uint32_t, r);
}

// 8 nibbles (one dword of the reference's qs bytes, Ggml.cs:1149-1150) -> 8 f16 in k-slot order
// [e0, e4, 16*e1, 16*e5, e2, e6, 16*e3, 16*e7], each minus OFF (Q4_0: 8, Q4_1: 0).  9 VALU.
template <int OFF>
__device__ __forceinline__ f16x8 unpack_nib8(uint32_t x) {
    const uint32_t y = x >> 8;
    u32x4 o;
    o[0] = magic_sub((x & 0x000F000Fu) | 0x64006400u, 1024.0f + OFF);
    o[1] = magic_sub((x & 0x00F000F0u) | 0x64006400u, 1024.0f + 16.0f * OFF);
    o[2] = magic_sub((y & 0x000F000Fu) | 0x64006400u, 1024.0f + OFF);
    o[3] = magic_sub((y & 0x00F000F0u) | 0x64006400u, 1024.0f + 16.0f * OFF);
    return __builtin_bit_cast(f16x8, o);
}

// Q5_0 (Ggml.cs:1285-1289): t8 = the 8 high bits of elements e0..e7 (bit i of t8 = element e_i); value = (nib | hb << 4) - OFF
// (OFF = 16; Q5_1, Ggml.cs:1330-1334: the unsigned value, OFF = 0, its min term goes the way of Q4_1's)
template <int OFF>
__device__ __forceinline__ f16x8 unpack_q5(uint32_t x, uint32_t t8) {
    const uint32_t y = x >> 8;
    // spread: bit i of t8 -> bit 4 (+16 for i >= 4) in the plain slots, bit 8 (+16) in the x16 slots.  One multiply puts a copy of
    // t8 at bits 4..11 and 16..23 (t8 < 256: no overlap); each slot then takes its two bits with one shift and one mask.
    const uint32_t p = t8 * 0x10010u;
    const uint32_t h0 = p & 0x00100010u;
    const uint32_t h1 = (p << 3) & 0x01000100u;
    const uint32_t h2 = (p >> 2) & 0x00100010u;
    const uint32_t h3 = (p << 1) & 0x01000100u;
    u32x4 o;
    o[0] = magic_sub((x & 0x000F000Fu) | h0 | 0x64006400u, 1024.0f + OFF);
    o[1] = magic_sub((x & 0x00F000F0u) | h1 | 0x64006400u, 1024.0f + 16.0f * OFF);
    o[2] = magic_sub((y & 0x000F000Fu) | h2 | 0x64006400u, 1024.0f + OFF);
    o[3] = magic_sub((y & 0x00F000F0u) | h3 | 0x64006400u, 1024.0f + 16.0f * OFF);
    return __builtin_bit_cast(f16x8, o);
}

// Q8_0: 8 signed bytes (two dwords of one plane) -> 8 f16 in k-slot order [b0, b2, b1, b3, b4, b6, b5, b7]
__device__ __forceinline__ f16x8 unpack_i8x8(uint32_t x0, uint32_t x1) {
    const uint32_t u0 = x0 ^ 0x80808080u, u1 = x1 ^ 0x80808080u;   // signed byte -> biased unsigned
    u32x4 o;
    o[0] = magic_sub((u0 & 0x00FF00FFu) | 0x64006400u, 1024.0f + 128.0f);
    o[1] = magic_sub(((u0 >> 8) & 0x00FF00FFu) | 0x64006400u, 1024.0f + 128.0f);
    o[2] = magic_sub((u1 & 0x00FF00FFu) | 0x64006400u, 1024.0f + 128.0f);
    o[3] = magic_sub(((u1 >> 8) & 0x00FF00FFu) | 0x64006400u, 1024.0f + 128.0f);
    return __builtin_bit_cast(f16x8, o);
}

template <int TYPE> struct WT {
    static constexpr int QW = (TYPE == GGML_TYPE_Q8_0) ? 4 : 2;     // raw dwords per lane, m-tile and k-block
    static constexpr bool MIN = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1;   // unsigned values + a min term
    static constexpr bool QH = TYPE == GGML_TYPE_Q5_0 || TYPE == GGML_TYPE_Q5_1;    // fifth bits
};

// WMT x WNT 32x32 tiles per wave, WGM x WGN waves per workgroup, KB k-blocks per LDS stage
template <int TYPE, int WMT, int WNT, int WGM, int WGN, int KB>
struct Cfg {
    static constexpr int TM = WGM * WMT * 32, TN = WGN * WNT * 32, NT = WGM * WGN * 64;
    static constexpr int A_BYTES = KB * 4 * TN * 16;       // f16 activation image of one stage
    static constexpr int NSC = WT<TYPE>::MIN ? 2 : 1;
    static constexpr int SC_BYTES = KB * TN * 4;           // one f32 plane of row scales
    static constexpr int STAGE = A_BYTES + NSC * SC_BYTES;
    static constexpr int TOTAL = 2 * STAGE;
    static constexpr int A_CHUNKS = KB * 4 * TN;           // 16-byte pieces
    static_assert(A_CHUNKS % NT == 0, "DMA rounds");
    static constexpr int A_ROUNDS = A_CHUNKS / NT;
    static constexpr int SC_CHUNKS = KB * TN / 4;          // 16-byte pieces per plane
    static_assert((SC_CHUNKS % 64 == 0 || SC_CHUNKS == 32) && SC_CHUNKS <= NT, "scale DMA is whole waves (or the first half of one: 32-column tiles)");
    static constexpr int WPS = 2;                          // waves per SIMD aimed at (256 registers each)
};

// KSP = 1 | 2 | 4: with 2 (4) the workgroup holds two (four) wave groups that take alternate LDS stages of K (each with its own stage
// buffers) and add their accumulators through LDS at the end -- twice the waves for grids too small to fill the chip.
template <int TYPE, int WMT, int WNT, int WGM, int WGN, int KB, int KSP>
__global__ __launch_bounds__(WGM * WGN * 64 * KSP, (Cfg<TYPE, WMT, WNT, WGM, WGN, KB>::WPS))
void gemm_q16_kernel(const uint8_t *__restrict__ wqs, const uint32_t *__restrict__ wqh, const float *__restrict__ wd,
                     const float *__restrict__ wm, const uint8_t *__restrict__ a16, const float *__restrict__ ad,
                     const float *__restrict__ asd, float *__restrict__ dst, int M, int N, int Mpad, int Npad, int nstages,
                     int ldd, int tiles_m, int tiles_n, uint32_t wq_bytes, uint32_t wd_bytes, uint32_t a_bytes,
                     uint32_t ad_bytes, uint32_t dst_bytes) {
    using C = Cfg<TYPE, WMT, WNT, WGM, WGN, KB>;
    constexpr int QW = WT<TYPE>::QW;
    constexpr int NTILE = WMT * WNT;
    static_assert(KB % 2 == 0, "fragment buffers alternate by k-block parity");
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
    // contiguous run of the tile list ordered "m fastest": its resident workgroups share few activation panels and
    // walk all weight panels together.
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
    const int m0 = tm_i * C::TM;
    const int n0 = tn_i * C::TN;

    f32x16 acc[WMT][WNT];                                   // (register tuples: the min-term MFMA accumulates into them in place)
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- activation image + row scales: global -> LDS by DMA, one stage (KB k-blocks) at a time ----
    // Chunk c = tid + NT*i of a stage is [bb][panel][TN rows] x 16 B.  NT is a multiple of TN and NT/TN divides 4, so
    // the row and the low bits of the panel are per-thread constants and everything that depends on i is uniform:
    // ONE per-thread 32-bit offset serves every DMA of the kernel.  K (weights and image) is zero-padded to whole
    // stages by the host side, so there is no tail logic anywhere in the loop.
    constexpr int P = C::NT / C::TN;
    static_assert(C::NT % C::TN == 0 && (P == 1 || P == 2 || P == 4), "chunk decomposition");
    const uint32_t a_pan = (uint32_t)(Npad * 16);                       // bytes of one panel of the f16 image
    const uint32_t a_blk = 4 * a_pan;                                   // bytes of one k-block
    const uint32_t voffA = (uint32_t)(((tid / C::TN) * Npad + n0 + (tid % C::TN)) * 16);
    const uint32_t voffS = (uint32_t)(((tid / (C::TN / 4)) * Npad + n0 + 4 * (tid % (C::TN / 4))) * 4);
    const rsrc_t rA = make_rsrc(a16, a_bytes), rAd = make_rsrc(ad, ad_bytes), rAs = make_rsrc(asd, ad_bytes);

    // One DMA piece = one wave-instruction = 1 KiB.  A piece costs its wave ~100 issue cycles, so the pieces of stage
    // s+1 are spread over the first tiles of stage s (one per tile) instead of being issued as a burst.  Pieces past
    // the end of K are dropped by the descriptor's range check (they would only fill a buffer nobody reads).
    constexpr int NPIECE = C::A_ROUNDS + 1;
    auto dma_piece = [&](int s, int buf, auto pc) {             // stage s of K -> stage buffer buf of this group
        constexpr int i = decltype(pc)::value;
        uint8_t *sp = gsm + (buf & 1) * C::STAGE;
        if constexpr (i < C::A_ROUNDS) {
            constexpr int bb = (P * i) >> 2, pan = (P * i) & 3;
            blds16(rA, sp + (size_t)(wave * 64 + C::NT * i) * 16, voffA, (uint32_t)s * KB * a_blk + bb * a_blk + pan * a_pan);
        } else if (wave * 64 < C::SC_CHUNKS) {                           // uniform per wave
            const uint32_t sS0 = (uint32_t)s * KB * (uint32_t)(Npad * 4);
            if (C::SC_CHUNKS % 64 == 0 || lane < C::SC_CHUNKS % 64) {   // (32-column tiles: half a wave of 16-byte pieces; the DMA honours exec)
                blds16(rAd, sp + C::A_BYTES + (size_t)(wave * 64) * 16, voffS, sS0);
                if (WT<TYPE>::MIN) blds16(rAs, sp + C::A_BYTES + C::SC_BYTES + (size_t)(wave * 64) * 16, voffS, sS0);
            }
        }
    };

    // ---- weights: registers only.  Lane (l31, hh) of m-tile i owns row m0 + (wm_*WMT + i)*32 + l31, byte half hh ----
    struct Raw { uint32_t q[WMT][QW]; uint32_t qh[WMT]; float d[WMT]; float mn[WMT]; };
    struct Frag { f16x8 b[WMT][2]; float d[WMT]; float mn[WMT]; };
    uint32_t offW[WMT], offD[WMT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
        const int m = m0 + (wm_ * WMT + i) * 32 + l31;
        offW[i] = (TYPE == GGML_TYPE_Q8_0) ? (uint32_t)((hh * Mpad + m) * 16) : (uint32_t)(m * 16 + 8 * hh);
        offD[i] = (uint32_t)(m * 4);
    }
    const uint32_t w_blk = (uint32_t)(Mpad * (TYPE == GGML_TYPE_Q8_0 ? 32 : 16));
    const uint32_t d_blk = (uint32_t)(Mpad * 4);

    // the planes carry two spare (zero) k-blocks past the padded end, so the look-ahead never needs a bounds check
    const rsrc_t rWq = make_rsrc(wqs, wq_bytes), rWd = make_rsrc(wd, wd_bytes);
    const rsrc_t rWh = make_rsrc(WT<TYPE>::QH ? (const void *)wqh : (const void *)wd, wd_bytes);
    const rsrc_t rWm = make_rsrc(WT<TYPE>::MIN ? (const void *)wm : (const void *)wd, wd_bytes);
    auto load_raw_one = [&](int kb, Raw &r, auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t sq = (uint32_t)kb * w_blk, sd = (uint32_t)kb * d_blk;
        if constexpr (QW == 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rWq, (int)offW[i], (int)sq, 0);
            r.q[i][0] = v[0]; r.q[i][1] = v[1]; r.q[i][2] = v[2]; r.q[i][3] = v[3];
        } else {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rWq, (int)offW[i], (int)sq, 0);
            r.q[i][0] = v[0]; r.q[i][1] = v[1];
        }
        r.d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWd, (int)offD[i], (int)sd, 0));
        if constexpr (WT<TYPE>::QH) r.qh[i] = __builtin_amdgcn_raw_buffer_load_b32(rWh, (int)offD[i], (int)sd, 0);
        if constexpr (WT<TYPE>::MIN) r.mn[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWm, (int)offD[i], (int)sd, 0));
    };
    auto unpack_one = [&](const Raw &r, Frag &f, auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (TYPE == GGML_TYPE_Q4_0) {
            f.b[i][0] = unpack_nib8<8>(r.q[i][0]);
            f.b[i][1] = unpack_nib8<8>(r.q[i][1]);
        } else if constexpr (TYPE == GGML_TYPE_Q4_1) {
            f.b[i][0] = unpack_nib8<0>(r.q[i][0]);
            f.b[i][1] = unpack_nib8<0>(r.q[i][1]);
        } else if constexpr (WT<TYPE>::QH) {
            constexpr int OFF = TYPE == GGML_TYPE_Q5_0 ? 16 : 0;
            const uint32_t t16 = r.qh[i] >> (16 * hh);
            f.b[i][0] = unpack_q5<OFF>(r.q[i][0], t16 & 0xFFu);
            f.b[i][1] = unpack_q5<OFF>(r.q[i][1], (t16 >> 8) & 0xFFu);
        } else {
            f.b[i][0] = unpack_i8x8(r.q[i][0], r.q[i][1]);
            f.b[i][1] = unpack_i8x8(r.q[i][QW - 2], r.q[i][QW - 1]);
        }
        f.d[i] = r.d[i];
        if constexpr (WT<TYPE>::MIN) f.mn[i] = r.mn[i];
    };

    Raw raw;
    Frag frag[2];                                           // k-block kb uses frag[kb & 1]

    // ---- one stage: KB k-blocks x NTILE tiles, software-pipelined: MFMAs of tile t+1 are issued before the
    //      scale-accumulate of tile t.  Register budget (256 at two waves per SIMD) is what shapes the order:
    //      * activation fragments of the next n-tile are fetched right after the last MFMA that reads the current ones,
    //      * the row scales of the next n-tile replace the current ones group by group as the last scale-accumulate
    //        of the current n-tile retires them,
    //      * the weights of k-block kb+1 are expanded, m-tile by m-tile, right after the last MFMA that reads the
    //        current fragments of that m-tile, and the raw loads of kb+2 start right after.
    //      The scale-accumulate itself is inline asm, 4 accumulator registers at a time (mul into a temp, fmac): left
    //      to the scheduler, the 16 products of a tile are hoisted far above their fmacs and the kernel spills. ----
    auto compute = [&](int s, int buf) {                         // stage s of K, resident in stage buffer buf
        const uint8_t *sp = gsm + (buf & 1) * C::STAGE;
        const uint8_t *sA = sp + ((size_t)(hh * C::TN + wn * WNT * 32 + l31)) * 16;
        const float *sDa = (const float *)(sp + C::A_BYTES) + wn * WNT * 32 + 4 * hh;
        // Min term: on the VALU (16 v_fmac per tile and block) for the 8-tile waves, whose registers are full; otherwise as a K = 2
        // f32 MFMA per pair of k-blocks (issue_b) -- d1 * sum(a) is then read as its A operand, lane half hh = the odd block of the pair
        // (r3: by the SPLIT, not by the tile count -- the two forms add the same terms in a different order, and above 512 src1 rows the
        // 8-tile form and the 2-tile form serve the same N: a 26000-row Q5_1 matrix and its 4000-row shard differed in the last bits
        // (tools/sweep_parity.py with row shards).  Unsplit forms: VALU; forms that split K (up to 512 rows): MFMA.)
        static_assert(KSP == 1 || NTILE < 8, "the split forms have room for the min-term MFMA");
        constexpr bool MIN_MFMA = WT<TYPE>::MIN && KSP > 1, MIN_VALU = WT<TYPE>::MIN && !MIN_MFMA;
        const float *sSa = sDa + C::SC_BYTES / 4;                        // d1 * sum(a) (Q5_1: d1 * (s0 + s1))
        const float *sSp = (const float *)(sp + C::A_BYTES + C::SC_BYTES) + hh * C::TN + wn * WNT * 32 + l31;
        const int kb0 = s * KB;
        constexpr int LAST = KB * NTILE - 1, DRAIN = (KB - 1) * NTILE + NTILE / 2;
        constexpr int PP = (NPIECE + DRAIN - 1) / DRAIN;                // DMA pieces per tile (1 unless a stage has few tiles)
        static_assert(PP <= 3 && NPIECE <= PP * DRAIN, "all DMA pieces are issued before the drain point");

        f16x8 af[2];
        f32x4 da[4], sa[MIN_VALU ? 4 : 1];
        f32x16 tacc[2];
        float af_s = 0.0f;                                               // min term: A operand (row l31, k-block (bb - 1) + hh)

        auto fetch_af = [&](auto nc) {                                   // nc = n-tile index within the stage
            constexpr int g = decltype(nc)::value, bb = g / WNT, j = g % WNT;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) af[kk] = *(const f16x8 *)(sA + ((bb * 4 + 2 * kk) * C::TN + 32 * j) * 16);
            if constexpr (MIN_MFMA && (bb & 1)) af_s = sSp[(bb - 1) * C::TN + 32 * j];
        };
        auto fetch_da = [&](auto nc, auto gc) {
            constexpr int g = decltype(nc)::value, bb = g / WNT, j = g % WNT, q = decltype(gc)::value;
            da[q] = *(const f32x4 *)(sDa + bb * C::TN + 32 * j + 8 * q);
            if constexpr (MIN_VALU) sa[q] = *(const f32x4 *)(sSa + bb * C::TN + 32 * j + 8 * q);
        };
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // MFMAs of tile t in two halves (one wave cannot issue the second before the first has left the matrix pipe, so
        // VALU work goes between them), then everything that becomes possible once both are issued
        f32x16 thalf;
        auto issue_a = [&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / NTILE, i = t % WMT;
            thalf = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], frag[bb & 1].b[i][0], zero, 0, 0, 0);
        };
        auto issue_b = [&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / NTILE, j = (t % NTILE) / WMT, i = t % WMT;
            tacc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], frag[bb & 1].b[i][1], thalf, 0, 0, 0);
            // Min term, + m0 * (d1 * sum(a)) per block (Ggml.cs:1190-1196 factorised; Q5_1: m * (s0 + s1), :1344): an outer product per
            // block, i.e. a K = 2 matrix product per pair of blocks -- one v_mfma_f32_32x32x2_f32 straight into the tile's
            // accumulators, a whole tile step (32 VALU instructions) away from the scale-accumulates that touch them.
            if constexpr (MIN_MFMA && (bb & 1)) {
                const float mpair = hh ? frag[1].mn[i] : frag[0].mn[i];                   // k-blocks bb - 1 | bb in the two lane halves
                // (a BUILTIN, so that hipcc pads its own reads of the result -- it splits the tuple with v_mov; the inline-asm readers
                // are a tile step away: the dummy operand of this step's third scale-accumulate group keeps the MFMA above it)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af_s, mpair, acc[i][j], 0, 0, 0);
                // one tile per k-block (32-column forms): the SAME accumulators take the previous tile's scale-accumulates right
                // behind this MFMA: its 16 passes + 4 wait states are spent here (tests/test_isa_audit.py)
                if constexpr (WMT * WNT == 1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc[i][j]));
            }
            if constexpr (i == WMT - 1 && bb * WNT + j + 1 < KB * WNT) fetch_af(std::integral_constant<int, bb * WNT + j + 1>{});
            if constexpr (j == WNT - 1) {
                unpack_one(raw, frag[(bb + 1) & 1], std::integral_constant<int, i>{});
                // (this group's k-blocks: the stage after s is s + KSP)
                load_raw_one(bb + 2 < KB ? kb0 + bb + 2 : kb0 + KSP * KB + (bb + 2 - KB), raw, std::integral_constant<int, i>{});
            }
        };

        fetch_af(std::integral_constant<int, 0>{});
        static_for<4>([&](auto gc) { fetch_da(std::integral_constant<int, 0>{}, gc); });
        issue_a(std::integral_constant<int, 0>{});
        issue_b(std::integral_constant<int, 0>{});

        static_for<KB * NTILE>([&](auto tc) {
            constexpr int t = decltype(tc)::value, bb = t / NTILE, tl = t % NTILE, j = tl / WMT, i = tl % WMT;
            static_for<PP>([&](auto uc) {
                constexpr int pc = PP * t + decltype(uc)::value;
                if constexpr (pc < NPIECE) dma_piece(s + KSP, buf + 1, std::integral_constant<int, pc>{});
            });
            // The DMA pieces of the next stage must have landed before this wave arrives at the stage's barrier.  Waiting
            // for them HERE, in the middle of the stage's last k-block, costs nothing (every vector-memory operation
            // issued so far is at least half a k-block old) and leaves the weight loads issued after this point in
            // flight across the barrier.
            if constexpr (t == DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (t < LAST) issue_a(std::integral_constant<int, t + 1>{});
            const float dw = frag[bb & 1].d[i];
            // (read here, like dw: with one tile per k-block the unpack behind the next tile's second MFMA -- issued from inside this tile's
            // scale-accumulate groups -- refills this very fragment buffer)
            float mw_blk = 0.0f;
            if constexpr (MIN_VALU) mw_blk = frag[bb & 1].mn[i];
            f32x16 &ac = acc[i][j];
            static_for<4>([&](auto gc) {
                constexpr int q = decltype(gc)::value;
                float t0, t1, t2, t3;
                // acc += (sumi * d1) * d0, Ggml.cs:1158.  The other MFMA result rides along as a dummy operand of
                // group 0 so that its MFMAs stay above this tile's VALU work (they are what it overlaps with).
                // Hazard note (inline asm is not padded by hipcc): an MFMA result needs 12 wait states before a VALU
                // reads it.  Tile t's MFMAs are pinned above tile t-1's 32 VALU instructions, so only the first tile of
                // a stage can be short of them: it carries its own s_nop.
                if constexpr (t == 0 && q == 0) asm volatile("s_nop 11" : "+v"(tacc[0]));
                if constexpr (q == 2 && t < LAST && MIN_MFMA && (((t + 1) / NTILE) & 1)) {
                    // the next tile's accumulators were just written by its min-term MFMA (issue_b above): this empty statement keeps
                    // that MFMA above the two scale-accumulate groups that follow (volatile statements keep their order)
                    constexpr int tn = t + 1, j1 = (tn % NTILE) / WMT, i1 = tn % WMT;
                    asm volatile("" : "+v"(acc[i1][j1]));
                }
                if constexpr (q == 0 && t < LAST) {
                    asm volatile("v_mul_f32 %4, %8, %12\n\tv_mul_f32 %5, %9, %13\n\tv_mul_f32 %6, %10, %14\n\tv_mul_f32 %7, %11, %15\n\t"
                                 "v_fmac_f32 %0, %4, %16\n\tv_fmac_f32 %1, %5, %16\n\tv_fmac_f32 %2, %6, %16\n\tv_fmac_f32 %3, %7, %16"
                                 : "+v"(ac[0]), "+v"(ac[1]), "+v"(ac[2]), "+v"(ac[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                                 : "v"(tacc[t & 1][0]), "v"(tacc[t & 1][1]), "v"(tacc[t & 1][2]), "v"(tacc[t & 1][3]), "v"(da[0][0]),
                                   "v"(da[0][1]), "v"(da[0][2]), "v"(da[0][3]), "v"(dw), "v"(thalf));
                } else {
                    asm volatile("v_mul_f32 %4, %8, %12\n\tv_mul_f32 %5, %9, %13\n\tv_mul_f32 %6, %10, %14\n\tv_mul_f32 %7, %11, %15\n\t"
                                 "v_fmac_f32 %0, %4, %16\n\tv_fmac_f32 %1, %5, %16\n\tv_fmac_f32 %2, %6, %16\n\tv_fmac_f32 %3, %7, %16"
                                 : "+v"(ac[4 * q + 0]), "+v"(ac[4 * q + 1]), "+v"(ac[4 * q + 2]), "+v"(ac[4 * q + 3]), "=&v"(t0), "=&v"(t1),
                                   "=&v"(t2), "=&v"(t3)
                                 : "v"(tacc[t & 1][4 * q + 0]), "v"(tacc[t & 1][4 * q + 1]), "v"(tacc[t & 1][4 * q + 2]),
                                   "v"(tacc[t & 1][4 * q + 3]), "v"(da[q][0]), "v"(da[q][1]), "v"(da[q][2]), "v"(da[q][3]), "v"(dw));
                }
                if constexpr (MIN_VALU) {
                    const float mw = mw_blk;
                    asm volatile("v_fmac_f32 %0, %4, %8\n\tv_fmac_f32 %1, %5, %8\n\tv_fmac_f32 %2, %6, %8\n\tv_fmac_f32 %3, %7, %8"
                                 : "+v"(ac[4 * q + 0]), "+v"(ac[4 * q + 1]), "+v"(ac[4 * q + 2]), "+v"(ac[4 * q + 3])
                                 : "v"(sa[q][0]), "v"(sa[q][1]), "v"(sa[q][2]), "v"(sa[q][3]), "v"(mw));
                }
                if constexpr (q == 1 && t < LAST) {
                    issue_b(std::integral_constant<int, t + 1>{});
                    asm volatile("" : "+v"(tacc[(t + 1) & 1]));              // the second MFMA stays above groups 2, 3
                }
                // this group's row scales are dead after the n-tile's last m-tile: fetch the next n-tile's into their place
                if constexpr (i == WMT - 1 && bb * WNT + j + 1 < KB * WNT)
                    fetch_da(std::integral_constant<int, bb * WNT + j + 1>{}, gc);
            });
        });
    };

    // ---- main loop: double-buffered LDS, one barrier per stage.  The DMA drain is a vmcnt(0) placed where it is free
    //      (see DRAIN in compute), not a counted wait at the barrier: hipcc may sink the weight loads (read-only buffer
    //      loads) below an asm wait, so "all but the youngest N" would not be a statement about the DMA pieces. ----
    static_for<NPIECE>([&](auto pc) { dma_piece(grp, 0, pc); });
    static_for<WMT>([&](auto ic) { load_raw_one(grp * KB, raw, ic); });
    static_for<WMT>([&](auto ic) { unpack_one(raw, frag[0], ic); load_raw_one(grp * KB + 1, raw, ic); });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                          // no LDS read may move above the barrier
    const int niter = (nstages + KSP - 1) / KSP;            // group grp takes stages grp, grp + KSP, ...; same barrier count for all
    // r4 (see gemm_qmx.hip): on one-round grids of 4-wave workgroups the younger wave of a SIMD takes priority on two stages of three
#ifndef Q16_PRIO
#define Q16_PRIO 3
#endif
    bool younger = false;
    if constexpr (WGM * WGN * KSP == 4 && Q16_PRIO != 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        younger = (hwid & 1u) != 0 && tiles_m * tiles_n <= 512;
    }
    for (int it = 0; it < niter; ++it) {
        if constexpr (WGM * WGN * KSP == 4 && Q16_PRIO != 0) {
            if (younger) { if (it % Q16_PRIO != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        }
        const int s = it * KSP + grp;
        if (KSP == 1 || s < nstages) compute(s, it);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the drain inside compute, again for a skipped stage)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- K split: groups 1 .. KSP-1 hand their accumulators to group 0 through LDS (the stage buffers are free now); group 0
    //      adds them in group order, so the summation tree is fixed ----
    if constexpr (KSP > 1) {
        constexpr int GRP_FLOATS = WGM * WGN * NTILE * 16 * 64;           // one group's accumulators
        static_assert((KSP - 1) * GRP_FLOATS * 4 <= KSP * C::TOTAL, "K-split exchange fits the stage buffers");
        float *xch = (float *)smem + (size_t)wave * (NTILE * 16 * 64) + lane;
        if (grp != 0) {
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[(size_t)(grp - 1) * GRP_FLOATS + ((i * WNT + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int g = 1; g < KSP; ++g)
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[(size_t)(g - 1) * GRP_FLOATS + ((i * WNT + j) * 16 + r) * 64];
    }

    // ---- dst[n][m]: D[row = (r&3) + 8*(r>>2) + 4*hh][col = lane & 31]; descriptor + per-lane offset + uniform row offset ----
    // descriptor based at the workgroup's tile origin: offsets below stay inside 32 bits for any dst size
    (void)dst_bytes;
    const rsrc_t rD = make_rsrc(dst + (size_t)n0 * ldd + m0, 0xFFFFFFFFu);
    const bool full = n0 + C::TN <= N && m0 + C::TM <= M;                    // uniform
    const uint32_t lane_off = (uint32_t)((4 * hh * ldd + l31) * 4);
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

template <int TYPE, int WMT, int WNT, int WGM, int WGN, int KB, int KSP = 1>
hipError_t launch_cfg(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    using C = Cfg<TYPE, WMT, WNT, WGM, WGN, KB>;
    auto kern = gemm_q16_kernel<TYPE, WMT, WNT, WGM, WGN, KB, KSP>;
        static PerDeviceOnce once;   // per kernel instantiation; the attribute is set once per device
    const hipError_t attr = once.max_dynamic_lds((const void *)kern, C::TOTAL * KSP);
    if (attr != hipSuccess) return attr;
    if (w->Mpad % C::TM != 0 || p.Npad % C::TN != 0) return hipErrorInvalidValue;
    const int tiles_m = (int)((w->M + C::TM - 1) / C::TM), tiles_n = (int)((N + C::TN - 1) / C::TN);
    dim3 grid((unsigned)(tiles_m * tiles_n));
    const int nstages = (int)((w->nbk + KB - 1) / KB);      // planes and image are zero-padded to whole stages (KB | K_STAGE_PAD)
    static_assert(K_STAGE_PAD % KB == 0, "stage padding");
    const uint64_t nba = (uint64_t)pad_kblocks(w->nbk);
    const uint64_t wq_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * (TYPE == GGML_TYPE_Q8_0 ? 32 : 16);
    const uint64_t wd_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 4, a_bytes = nba * 4 * (uint64_t)p.Npad * 16;
    const uint64_t ad_bytes = nba * (uint64_t)p.Npad * 4, dst_bytes = ((uint64_t)(N - 1) * (uint64_t)ldd + (uint64_t)w->M) * 4;
    constexpr uint64_t LIM = 0xFFFFFFFFull;                  // 32-bit buffer offsets
    if (wq_bytes > LIM || a_bytes > LIM || (uint64_t)C::TN * (uint64_t)ldd * 4 > LIM) return hipErrorNotSupported;   // api.cpp routes such shapes to gemm_q.hip
    kern<<<grid, C::NT * KSP, C::TOTAL * KSP, st>>>(w->qs, w->qh, w->d, w->m, (const uint8_t *)p.a8, p.ad, (const float *)p.as, dst, (int)w->M,
                                        (int)N, (int)w->Mpad, (int)p.Npad, nstages, (int)ldd, tiles_m, tiles_n, (uint32_t)wq_bytes,
                                        (uint32_t)wd_bytes, (uint32_t)a_bytes, (uint32_t)ad_bytes, (uint32_t)dst_bytes);
    return hipGetLastError();
}

// The form was chosen by plan.cpp (plan_f16: the K split by N and K, the tile height by the tile count).  The measurements behind it:
//   * batches up to 256 rows: 32-row weight tiles, K split four ways inside the workgroup (129 .. 256 rows, four-way | two-way us: Q8_0
//     4096 x 4096 x 256 27 | 37, 4096 x 11008 x 256 58 | 81, Q5_0 4096 x 4096 x 256 25 | 39; 11008 x 4096 x 256 63 | 58); the same split on
//     64-row tiles of 8 waves / 128-row tiles of 16 waves where those cover the chip (the 16-wave form not for the min-term types: registers);
//   * prompt-sized batches: 128 x 64 tiles, two wave groups splitting K;
//   * the big tile (256 x 128, 8 tiles per wave) when it still fills the chip with >= 2 workgroups per CU;
//   * short or narrow products (row shards): 64 x 64 tiles of four 1-tile waves wherever the 128 x 64 grid leaves CUs idle -- the same
//     unsplit K loop per element, the same bits (Q8_0 / Q5_0: 256 x 4096 x 2048 48.7 -> 36.3 us, 512 x 11008 x 2048 125.9 -> 93.2; not the
//     min-term types: their 48 VALU instructions per tile make the one-tile wave slower -- Q5_1 512 x 4096 x 4096 61.0 -> 66.0 us);
//   * otherwise 128 x 64 tiles of 4 waves (N = 1024, this | 128 x 128 of 4 waves | 128 x 64 of 2 waves: Q8_0 4096 x 4096 69 | 84 | 104 us).
template <int TYPE>
hipError_t launch_typed(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (pl.family != MMF_F16) return hipErrorInvalidValue;
    switch (pl.form) {
    case F16F_N32_H64:  return launch_cfg<TYPE, 1, 1, 2, 1, 4, 4>(w, p, N, dst, ldd, st);
    case F16F_N32_H32:  return launch_cfg<TYPE, 1, 1, 1, 1, 4, 4>(w, p, N, dst, ldd, st);
    case F16F_S4_H128:  if constexpr (!WT<TYPE>::MIN) return launch_cfg<TYPE, 1, 2, 4, 1, 4, 4>(w, p, N, dst, ldd, st); break;
    case F16F_S4_H64:   return launch_cfg<TYPE, 1, 2, 2, 1, 4, 4>(w, p, N, dst, ldd, st);
    case F16F_S4_H32:   return launch_cfg<TYPE, 1, 2, 1, 1, 4, 4>(w, p, N, dst, ldd, st);
    case F16F_S2_H128:  return launch_cfg<TYPE, 1, 2, 4, 1, 4, 2>(w, p, N, dst, ldd, st);
    case F16F_256x128:  return launch_cfg<TYPE, 2, 4, 4, 1, 4>(w, p, N, dst, ldd, st);
    case F16F_64x64:    return launch_cfg<TYPE, 1, 1, 2, 2, 4>(w, p, N, dst, ldd, st);
    case F16F_128x64:   return launch_cfg<TYPE, 1, 2, 4, 1, 4>(w, p, N, dst, ldd, st);
    default: break;
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_gemm_q16(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q5_1: return launch_typed<GGML_TYPE_Q5_1>(w, pl, p, N, dst, ldd, st);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0>(w, pl, p, N, dst, ldd, st);
    default: return hipErrorInvalidValue;
    }
}
