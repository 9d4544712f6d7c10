// gemm_qmp.hip -- K3p: quantized mat-mat for prompt-sized batches (a few hundred src1 rows) on the MX matrix path, Q4_0.
//
// COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698) with the block arithmetic of gemm_qmx.hip: one exact
// v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 tile and quant block (weights and both activation digits as bf6), then
// acc += (sumi * d1) * d0 in f32 (ggml_vec_dot_q4_0_q8_0, Ggml.cs:1136-1159).
//
// Why another form.  4096 x 4096 x 512 is 2048 output tiles for 256 CUs: eight tiles per CU.  The staged kernel of gemm_qmx.hip
// runs it on 64 x 64 workgroup tiles (two tiles per wave and k-block) and pulls 1.7 KB through the CU's vector memory path per tile
// and k-block -- each weight fragment serves two MFMAs, each activation stage two waves -- which is what bounds it (PMC,
// profiles/r03_pmc_c345_baseline.txt: 38 % of wave-cycles parked, 32 % VALU; 1400 W at 2.13 GHz, profiles/r03_power.txt): the path
// delivers about 70 GB/s per CU.  Here a WAVE owns a 128 x 64 output tile (4 x 2 MFMA tiles: 0.86 KB per tile and k-block, the
// least eight tiles allow) and the eight waves of a workgroup split K into eight contiguous ranges -- the structure of the
// batched-decode form K3s (gemm_qmx.hip) with eight tiles behind every fragment instead of one:
//   * no LDS stages, no barrier inside the K loop: every wave streams its own operands L2 -> registers one k-block ahead;
//   * weights: lanes 0..31 fetch the rows of m-tile 2p and lanes 32..63 those of m-tile 2p + 1 in ONE instruction (the MX
//     operand wants a block in both lane halves) and v_permlane32_swap hands each half the other's copy;
//   * activations: the K1b image (kind 3) straight from L2, both digit groups of a column in the two lane halves;
//   * row scales: the wave's slice of the scale plane in its own LDS slice (broadcast ds_read_b128);
//   * the eight partial tiles are added in wave order through LDS (two rounds of four tiles), every wave storing its share of
//     the rows: an eight-way tree fixed by K alone, so a row shard computes the bits of the unsplit matrix.
#include "common.h"
#include "plan.h"
#include <utility>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using i32x8 = __attribute__((ext_vector_type(8))) int;
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#ifdef K3P_TRACE            // tools/k3p_trace.hip: per-wave time stamps (s_memrealtime, 100 MHz) at the phase ends + shader cycles of the K loop
__device__ unsigned long long k3p_trace_buf[4096 * 8];
#define K3P_STAMP(k) do { if (lane == 0 && (size_t)blockIdx.x * 8 + wave < 4096) k3p_trace_buf[((size_t)blockIdx.x * 8 + wave) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define K3P_STAMP(k) do { } while (0)
#endif

constexpr int KS = 8;          // waves per workgroup = K ranges
// (K3P_TABLE_ROWS / K3P_SLICE_ROWS / K3P_MAX_SLICES: plan.h -- the scale tables whole up to K = 20480, in slices beyond: SLICED)
constexpr int WMT = 4;         // 32-row weight tiles per wave (two lane-half pairs)
constexpr int WNT = 2;         // 32-column tiles of src1 per wave

// XCD-aware tile order (speed only): workgroups b, b + 8, ... share an XCD and its L2; an XCD takes a contiguous run of the tile
// list ordered "n fastest", so the workgroups that stream the same weight rows sit behind one L2.
// (tile = blockIdx.x + r * gridDim.x for the r-th tile of a persistent workgroup; the grid is a multiple of 8 whenever r > 0, so a
// workgroup stays in its XCD's run of the list)
template <int WM = WMT>      // WM: 32-row weight tiles per wave (4; r5: 2 where a grid of 128-row tiles leaves CUs idle -- the int8 kernel)
__device__ __forceinline__ void tile_origin(int tile, int tiles_m, int tiles_n, int &m0, int &n0) {
    const int nwg = tiles_m * tiles_n;
    const int bid = tile, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    m0 = (t_lin / tiles_n) * (32 * WM);
    n0 = (t_lin % tiles_n) * (32 * WNT);
}

// This wave's slice of the row scales: nloc rows x 64 floats into its own LDS slice.  Rows at or past nbk_valid are ZERO, so a
// k-block past the end of K adds (sumi * 0) * d0 = +0 whatever the operand registers hold.  Same wave writes and reads.
// By LDS-DMA (buffer_load ... lds: no registers, every piece of the slice in flight at once): one instruction moves four table rows
// -- lane l fetches 16 bytes of row l / 16 -- to 1 KiB of the slice; rows at or past nbk_valid are beyond the descriptor and arrive as
// zeros (the range check covers the scalar offset, and the DMA form writes the zeros: tools/oob_probe.hip).  r4: it was a loop of four
// register loads per lane and trip -- three trips for K = 11008, each exposing a load latency with every workgroup of the launch
// asking at once: 4.4 us in front of the K loop, now 2.8 (4096 x 11008 x 512 Q8_0: 72.2 -> 70.4 us, tools/k3p_trace.hip).
typedef __attribute__((address_space(3))) void lds_void_p;
__device__ __forceinline__ void load_scale_table(float *tabD, const float *__restrict__ ad, int kb0, int nloc, int nbk_valid, int Npad, int n0, int lane) {
    const rsrc_t rT = make_rsrc(ad, (uint32_t)nbk_valid * (uint32_t)Npad * 4u);
    const uint32_t voff = (uint32_t)(((lane >> 4) * Npad + 4 * (lane & 15)) * 4);
    const int ngrp = nloc >> 2, rem = nloc & 3;
    for (int g = 0; g < ngrp; ++g)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, (lds_void_p *)(tabD + 256 * g), 16, (int)voff, (int)((uint32_t)((kb0 + 4 * g) * Npad + n0) * 4u), 0, 0);
    if (rem && lane < 16 * rem)                             // (the slice ends inside a group of four rows: the lanes of the rows past it stay out)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, (lds_void_p *)(tabD + 256 * ngrp), 16, (int)voff, (int)((uint32_t)((kb0 + 4 * ngrp) * Npad + n0) * 4u), 0, 0);
    // A DMA's LDS write is complete for a reader only after vmcnt(0) AND lgkmcnt(0), and the workgroup barrier is the customary third part
    // of the wait (quantize.hip K1b: with vmcnt(0) alone, now and then half of every 128-byte line still held the previous contents).
    // Every wave of the workgroup comes through here once per tile, so the barrier is uniform.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The waves' sums, added in wave order, four of the eight tiles per round (128 KB of LDS in 16-byte pieces); every wave takes its
// share of the pieces, so the additions of one element are the same, in the same order, whoever makes them.
template <int WM>
__device__ __forceinline__ void reduce_and_store(f32x16 (&acc)[WM][WNT], uint8_t *smem, float *__restrict__ dst, int M, int N, int ldd, int m0, int n0,
                                                 int wave, int lane, const mm_epilogue &ep) {
    constexpr int WMT = WM;                                  // (shadows the file's default: the index arithmetic below is the wave tile's)
    const int l31 = lane & 31, hh = lane >> 5;
    const int lda = (int)ep.ld_add, ld2 = (int)ep.ld2;
    f32x4 *xch = (f32x4 *)smem + lane;                       // [wave][tile of the round][q][lane] pieces of four accumulator registers
#pragma unroll
    for (int rnd = 0; rnd < WMT * WNT / 4; ++rnd) {          // (four tiles per round: two rounds for the 128-row wave tile, one for the 64-row one)
        __syncthreads();                                     // (round 0: every wave is past its scale table; round 1: past its reads)
        if (rnd == 0) K3P_STAMP(3); else K3P_STAMP(4);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = 4 * rnd + tt, j = t / WMT, i = t % WMT;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xch[(size_t)((wave * 4 + tt) * 4 + q) * 64] = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {                        // 4 tiles x 4 register groups = 16 pieces of 64 lanes x 16 B, 2 per wave
            const int pc = wave * 2 + k, tt = pc / 4, q = pc % 4;
            const int t = 4 * rnd + tt, j = t / WMT, i = t % WMT;
            f32x4 v = xch[(size_t)(tt * 4 + q) * 64];
#pragma unroll
            for (int g = 1; g < KS; ++g) {
                const f32x4 o = xch[(size_t)((g * 4 + tt) * 4 + q) * 64];
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            // dst[n][m]: D[row = (r & 3) + 8 (r >> 2) + 4 hh][col = lane & 31], r = 4 q + e
            const int m = m0 + 32 * i + l31;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + 32 * j + e + 8 * q + 4 * hh;
                if (n < N && m < M) {
                    if (ep.mode == 3) {                      // the row split's exchange: the same position in this rank's and every peer's [N][M] buffer
                        dst[(size_t)n * ldd + m] = v[e];
                        for (int k = 0; k < ep.npush; ++k) ep.push[k][(size_t)n * ldd + m] = v[e];
                    } else if (ep.mode == 2) {
                        dst[(size_t)n * ldd + m] = v[e] * ep.scale;
                    } else {
                        dst[(size_t)n * ldd + m] = v[e];
                        if (ep.mode == 1) ep.dst2[(size_t)n * ld2 + m] = v[e] + ep.addend[(size_t)n * lda + m];
                    }
                }
            }
        }
    }
}

template <int WM> struct WFragT { u32x4 lo[WM / 2]; u32x2 hi[WM / 2]; float d[WM / 2]; };   // lanes 0..31: m-tile 2p, lanes 32..63: m-tile 2p + 1
struct AFrag { u32x4 lo[WNT]; u32x2 hi[WNT]; };

template <bool SLICED, int WM = 4>   // (SLICED: the scale tables are refilled inside the K loop -- K > 19968; the one-table form compiles without the test.  WM: 32-row m-tiles per wave -- 4, or r5 2 where a grid of 128-row tiles leaves CUs idle)
__global__ __launch_bounds__(KS * 64, 2)
void gemm_qmx_mid_kernel(const uint8_t *__restrict__ w6a, const uint8_t *__restrict__ w6b, const float *__restrict__ wd,
                         const uint8_t *__restrict__ a6, const float *__restrict__ ad, float *__restrict__ dst, int M, int N, int Mpad,
                         int Npad, int nbkp, int nloc, int ldd, int tiles_m, int tiles_n, uint32_t w6a_bytes, uint32_t wd_bytes,
                         uint32_t a_bytes, const mm_epilogue ep, int ch_arg) {
    const int ch = SLICED ? ch_arg : nloc;
    constexpr int WMT = WM;                                  // (shadows the file's default)
    using WFrag = WFragT<WMT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    K3P_STAMP(0);

    // A workgroup is PERSISTENT: it takes tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the grid is one workgroup per CU once there
    // are more tiles than CUs).  The first operands of the next tile are requested before the reduction of the current one.
    const int nwg = tiles_m * tiles_n;
    int tile = blockIdx.x;
    int m0, n0;
    tile_origin<WMT>(tile, tiles_m, tiles_n, m0, n0);
    const int kb0 = wave * nloc;                            // this wave's k-blocks: kb0 .. kb0 + nloc - 1 (planes and image read 0 past their end)

    const rsrc_t rWa = make_rsrc(w6a, w6a_bytes), rWb = make_rsrc(w6b, w6a_bytes / 2), rWd = make_rsrc(wd, wd_bytes);
    const rsrc_t rA = make_rsrc(a6, a_bytes);
    uint32_t offA, offB, offD, voff16, voff8;               // per-lane offsets of the current tile's operands
    auto set_offsets = [&](int m0_, int n0_) {
        const int mrow = m0_ + 32 * hh + l31;               // pair p: + 64 p rows (the instruction's immediate offset)
        offA = (uint32_t)(mrow * 16); offB = (uint32_t)(mrow * 8); offD = (uint32_t)(mrow * 4);
        voff16 = (uint32_t)((hh * Npad + n0_ + l31) * 16); voff8 = (uint32_t)(32 * Npad + (hh * Npad + n0_ + l31) * 8);
    };
    set_offsets(m0, n0);
    const uint32_t wa_blk = (uint32_t)(Mpad * 16), wb_blk = (uint32_t)(Mpad * 8), d_blk = (uint32_t)(Mpad * 4);
    const uint32_t a_blk = (uint32_t)Npad * 48u;

    auto load_w = [&](WFrag &f, int kb) {
#pragma unroll
        for (int p = 0; p < WMT / 2; ++p) {
            f.lo[p] = __builtin_amdgcn_raw_buffer_load_b128(rWa, (int)(offA + 1024u * p), (int)((uint32_t)kb * wa_blk), 0);
            f.hi[p] = __builtin_amdgcn_raw_buffer_load_b64(rWb, (int)(offB + 512u * p), (int)((uint32_t)kb * wb_blk), 0);
            f.d[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rWd, (int)(offD + 256u * p), (int)((uint32_t)kb * d_blk), 0));
        }
    };
    auto load_a = [&](AFrag &f, int kb) {
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            f.lo[j] = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(voff16 + 512u * j), (int)((uint32_t)kb * a_blk), 0);
            f.hi[j] = __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(voff8 + 256u * j), (int)((uint32_t)kb * a_blk), 0);
        }
    };

    // (r4: the wave's scale table holds `ch` k-blocks at a time -- all of its range up to K = 20480, beyond that (eight whole tables would not
    // fit the 160 KB of LDS) the range goes through the slice in two or more refills, every wave of the workgroup at the same trip)
    float *const tabD = (float *)smem + (size_t)wave * ch * (32 * WNT);
    WFrag wl;                                               // the NEXT k-block's weights, as loaded (lane halves = the m-tiles of a pair)
    AFrag af;                                               // activation fragments: column tile j is refetched as soon as its last MFMA has issued
    load_w(wl, kb0);                                        // (requested first: the weights come from HBM, the table from L2)
    load_a(af, kb0);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (bool first = true;; first = false) {
    if (!first) __syncthreads();                            // (the previous tile's reduction has read its last partial sums: the tables may go over them)
    load_scale_table(tabD, ad, kb0, ch, nbkp, Npad, n0, lane);   // (K1b writes the k-blocks K is padded to as zeros)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int scale_a = hh ? 127 : 131;                     // E8M0: K group 0 (the ah digits) carries 2^4

    // v_permlane32_swap: x in both operands -> {x's lower half in both halves, x's upper half in both halves}
    auto both = [](uint32_t x, uint32_t &b0, uint32_t &b1) {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        b0 = r[0]; b1 = r[1];
    };

    // ---- K loop: one k-block per trip.  The weights of the block are spread to both lane halves (their load registers are free
    //      again and take the next block's: a whole block of look-ahead for the stream that comes from HBM); eight tiles, the MFMA of
    //      tile t + 1 issued before the scale-accumulate of tile t; column tile j's fragment is refetched right behind its last MFMA.
    //      (The look-ahead of the last trip reads the next wave's first block, or zeros past the planes: never used.) ----
    K3P_STAMP(1);
#ifdef K3P_TRACE
    const unsigned long long clk0 = __builtin_readcyclecounter();
#endif
    for (int b = 0, tb = 0; b < nloc; ++b, ++tb) {
        if (SLICED && tb == ch) {                                  // (uniform over the workgroup: the next `ch` rows of every wave's table)
            tb = 0;
            load_scale_table(tabD, ad, kb0 + b, nloc - b < ch ? nloc - b : ch, nbkp, Npad, n0, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // Two waves share a SIMD and the older one wins every arbitration: left alone, waves 0..3 finish their K range a quarter
        // early and waves 4..7 run the rest at one wave per SIMD (tools/k3p_trace.hip: 15.8 against 21.5 us).  The younger half
        // takes priority on most k-blocks, so both halves finish together.
        // r4: two blocks of three (it was every other one: the older half still finished 2.3 us early -- K loop done at 18.2 | 20.5 us after
        // the launch's first stamp, now 19.4 | 19.7; 4096 x 4096 x 512 27.6 -> 27.0 us per launch, tools/k3p_trace.hip)
        if (wave >= KS / 2) { if (b % 3 != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        i32x8 B[WMT];
        float dw[WMT];
#pragma unroll
        for (int p = 0; p < WMT / 2; ++p) {
            uint32_t b0[6], b1[6], d0u, d1u;
#ifdef K3P_NOSWAP          // timing experiment only (wrong operands): what do the 14 swaps + their copies cost?
            b0[0] = wl.lo[p][0]; b0[1] = wl.lo[p][1]; b0[2] = wl.lo[p][2]; b0[3] = wl.lo[p][3]; b0[4] = wl.hi[p][0]; b0[5] = wl.hi[p][1];
            b1[0] = wl.lo[p][1]; b1[1] = wl.lo[p][0]; b1[2] = wl.lo[p][3]; b1[3] = wl.lo[p][2]; b1[4] = wl.hi[p][1]; b1[5] = wl.hi[p][0];
            d0u = d1u = __builtin_bit_cast(uint32_t, wl.d[p]);
#else
            both(wl.lo[p][0], b0[0], b1[0]); both(wl.lo[p][1], b0[1], b1[1]); both(wl.lo[p][2], b0[2], b1[2]); both(wl.lo[p][3], b0[3], b1[3]);
            both(wl.hi[p][0], b0[4], b1[4]); both(wl.hi[p][1], b0[5], b1[5]);
            both(__builtin_bit_cast(uint32_t, wl.d[p]), d0u, d1u);
#endif
            B[2 * p] = i32x8{(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b0[4], (int)b0[5], 0, 0};
            B[2 * p + 1] = i32x8{(int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3], (int)b1[4], (int)b1[5], 0, 0};
            dw[2 * p] = __builtin_bit_cast(float, d0u); dw[2 * p + 1] = __builtin_bit_cast(float, d1u);
        }
        load_w(wl, kb0 + b + 1);
        const float *dp = tabD + (SLICED ? tb : b) * (32 * WNT) + 4 * hh;
        f32x16 x[2];
        {
            const i32x8 A0 = {(int)af.lo[0][0], (int)af.lo[0][1], (int)af.lo[0][2], (int)af.lo[0][3], (int)af.hi[0][0], (int)af.hi[0][1], 0, 0};
            x[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A0, B[0], zero, 3, 3, 0, scale_a, 0, 127);
        }
        static_for<WMT * WNT>([&](auto tc) {
            constexpr int t = decltype(tc)::value, j = t / WMT, i = t % WMT;
            if constexpr (t + 1 < WMT * WNT) {
                constexpr int j1 = (t + 1) / WMT, i1 = (t + 1) % WMT;
                const i32x8 A1 = {(int)af.lo[j1][0], (int)af.lo[j1][1], (int)af.lo[j1][2], (int)af.lo[j1][3], (int)af.hi[j1][0], (int)af.hi[j1][1], 0, 0};
                x[(t + 1) & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A1, B[i1], zero, 3, 3, 0, scale_a, 0, 127);
                if constexpr (i1 == WMT - 1) {              // the last MFMA that reads column tile j1's fragment has issued
                    af.lo[j1] = __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(voff16 + 512u * j1), (int)((uint32_t)(kb0 + b + 1) * a_blk), 0);
                    af.hi[j1] = __builtin_amdgcn_raw_buffer_load_b64(rA, (int)(voff8 + 256u * j1), (int)((uint32_t)(kb0 + b + 1) * a_blk), 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 da = *(const f32x4 *)(dp + 32 * j + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = x[t & 1][4 * q + e] * da[e];                               // (sumi * d1) ...
                    acc[i][j][4 * q + e] = __builtin_fmaf(v, dw[i], acc[i][j][4 * q + e]);      // ... * d0, Ggml.cs:1158
                }
            }
        });
    }

#ifdef K3P_TRACE
    asm volatile("" : "+v"(acc[0][0]));
    if (lane == 0 && (size_t)blockIdx.x * 8 + wave < 4096) k3p_trace_buf[((size_t)blockIdx.x * 8 + wave) * 8 + 7] = __builtin_readcyclecounter() - clk0;
#endif
    K3P_STAMP(2);
    // ---- the waves' sums, added in wave order, four of the eight tiles per round (128 KB of LDS); every wave takes its share of the
    //      result rows a lane holds, so the additions of one element are the same, in the same order, whoever makes them ----
    __builtin_amdgcn_s_setprio(0);
    const int m0c = m0, n0c = n0;
    tile += (int)gridDim.x;
    if (tile < nwg) {                                       // the next tile's first operands travel while this one is reduced and stored
        tile_origin<WMT>(tile, tiles_m, tiles_n, m0, n0);
        set_offsets(m0, n0);
        load_w(wl, kb0);
        load_a(af, kb0);
    }
    reduce_and_store<WMT>(acc, smem, dst, M, N, ldd, m0c, n0c, wave, lane, ep);
    if (tile >= nwg) break;
  }
    K3P_STAMP(5);
}

// ---- K3p on the int8 matrix cores: Q8_0 and Q5_0 (COMPUTE phase with ggml_vec_dot_q8_0_q8_0, Ggml.cs:1351-1381, and
//      ggml_vec_dot_q5_0_q8_0, Ggml.cs:1258-1301) ----------------------------------------------------------------------------------------
// Q8_0's resident planes ARE v_mfma_i32_32x32x32_i8 operands -- [k-block][half][row][16 B], the halves holding the even and the odd
// elements, the order K1's image 0 gives the activations -- so a k-block is one 16-byte load per operand, tile and lane: nothing to
// expand, nothing to spread across lane halves.  Q5_0 gets planes of the same form at upload (ggml_hip_weight::i8p, layout.hip).
// Per block, Q8_0: acc = fma((float)sumi, d1 * d0, acc) (gemm_q8s.hip, the batched-decode form of the same arithmetic); Q5_0:
// acc = fma(d0 * (float)sumi, d1, acc) -- the reference's a = d * sxy, t = a * y.d (Ggml.cs:1296-1298).  Structure as above; the weights of block b + 1 are requested at the start of block b into a second
// register set (the sets take turns: two k-blocks per trip, nloc even).
//
// r4 -- Q5_1 and Q4_1 (and through Q5_1 the Q5_K extension, whose super-blocks live as eight k-blocks of the planar Q5_1 form): the same loop on
// int8 planes of the UNSIGNED values (nib | bit << 4) in 0..31, per block acc = fma(d0 * (float)sxy, d1, acc) -- the reference's
// (d * sxy) * y.d (Ggml.cs:1344) -- and the min terms m * (s0 + s1) as a matrix product of their own IN FRONT of the K loop: sixteen
// k-blocks per v_mfma_f32_32x32x16_bf16, both operands as bf16 pieces that sum to the f32 value exactly (see the loop top).  Order of
// an element's additions in a wave: the min terms of the wave's chunks (chunk c % 8 == wave, by K alone), then its k-blocks in order;
// the waves' sums are added as for every type.  (The first form of this round -- one v_mfma_f32_32x32x2_f32 per tile and PAIR of
// k-blocks inside the loop, 128 matrix passes per tile and 16 k-blocks instead of 40 / 48 -- is docs/experiments/r4_min_term_in_loop_f32.patch:
// 85.5 us where this one takes 77 at 4096 x 11008 x 512.)
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;
template <int WM, bool TWO = false> struct WI8T { i32x4 q[WM]; float d[WM]; float m[TWO ? WM : 1]; };   // (TWO: m = the scale of the k-block's second 16-element block)

template <int TYPE, bool M3 = false, bool SLICED = false, int WM = 4>
__global__ __launch_bounds__(KS * 64, 2)
void gemm_q8_mid_kernel(const uint8_t *__restrict__ qs, const float *__restrict__ wd, const int8_t *__restrict__ a8,
                        const float *__restrict__ ad, float *__restrict__ dst, int M, int N, int Mpad, int Npad,
                        int nbk, int nloc, int ldd, int tiles_m, int tiles_n, uint32_t w_bytes, uint32_t a_bytes, const mm_epilogue ep,
                        const uint8_t *__restrict__ mp3, const uint8_t *__restrict__ sp3, int ch_arg) {
    const int ch = SLICED ? ch_arg : nloc;
    constexpr int WMT = WM;                                  // (r5: the wave tile's height is a parameter here -- 4 or 2 m-tiles; shadows the file's default)
    constexpr bool TWO = TYPE == GGML_TYPE_Q4_2;            // (r5) two 16-element blocks per k-block, a scale each: Q4_2, and Q6_K in its planar form
    using WI8 = WI8T<WMT, TWO>;
    constexpr bool MIN = TYPE == GGML_TYPE_Q5_1, MINP = MIN;   // (Q4_1 runs this instantiation: its int8 planes hold 0..15)
#ifndef K3P_DA_INPLACE
#define K3P_DA_INPLACE (MIN || SLICED || TWO)   // (the sliced forms: their extra live state took the scale look-ahead registers to scratch -- Q8_0 4096 x 28672 x 512 197 us)
#endif
    constexpr bool DA_INPLACE = K3P_DA_INPLACE;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    K3P_STAMP(0);
    const int nwg = tiles_m * tiles_n;                      // (persistent workgroups: see the MX kernel)
    int tile = blockIdx.x;
    int m0, n0;
    tile_origin<WMT>(tile, tiles_m, tiles_n, m0, n0);
    const int kb0 = wave * nloc;

    const rsrc_t rW = make_rsrc(qs, w_bytes), rD = make_rsrc(wd, w_bytes / 8), rA = make_rsrc(a8, a_bytes);
    const rsrc_t rM2 = make_rsrc(TWO ? (const void *)mp3 : (const void *)wd, w_bytes / 8);   // (TWO: the second scales' plane rides in the mp3 argument)
    uint32_t offW, offD, offA;
    auto set_offsets = [&](int m0_, int n0_) {
        offW = (uint32_t)((hh * Mpad + m0_ + l31) * 16); offD = (uint32_t)((m0_ + l31) * 4); offA = (uint32_t)((hh * Npad + n0_ + l31) * 16);
    };
    set_offsets(m0, n0);
    const uint32_t w_blk = (uint32_t)(2 * Mpad * 16), d_blk = (uint32_t)(Mpad * 4), a_blk = (uint32_t)(2 * Npad * 16);
    auto load_w = [&](WI8 &f, int kb) {
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            f.q[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(offW + 512u * i), (int)((uint32_t)kb * w_blk), 0));
            f.d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rD, (int)(offD + 128u * i), (int)((uint32_t)kb * d_blk), 0));
            if constexpr (TWO) f.m[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM2, (int)(offD + 128u * i), (int)((uint32_t)kb * d_blk), 0));
        }
    };
    float *const tabD = (float *)smem + (size_t)wave * ch * (32 * WNT);   // (`ch` k-blocks of the wave's range at a time: see the MX kernel)
    WI8 w0, w1;
    i32x4 af[WNT];
    auto load_first = [&]() {
        load_w(w0, kb0);                                    // (requested first: the weights come from HBM, the table from L2)
#pragma unroll
        for (int j = 0; j < WNT; ++j) af[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(offA + 512u * j), (int)((uint32_t)kb0 * a_blk), 0));
    };
    load_first();
    const i32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // ---- the min-term product's pieces (see the loop top) ----
    constexpr int NPIECE = M3 ? 3 : 2;                      // bf16 pieces of a min: Q5_1's is an f16 value (two hold it), Q4_1's an f32, Q5_K's an f32 product
#ifndef K3P_MIN_PRE
#define K3P_MIN_PRE 1                                       // chunks per wave requested ahead of the scale table (56 / 72 registers each, taken from the not-yet-live accumulators)
#endif
    constexpr int NPRE = MINP ? K3P_MIN_PRE : 1;            // (2 and 3 measured no better: what the product's loads cost is their bytes through L2, all workgroups at once, not their latency)
    using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
    const int nchunks = (nbk + 15) / 16;
    const rsrc_t rMP = make_rsrc(MINP ? mp3 : qs, MINP ? (uint32_t)(nchunks * 6) * (uint32_t)Mpad * 16u : 0u);
    const rsrc_t rSP = make_rsrc(MINP ? sp3 : qs, MINP ? (uint32_t)(nchunks * 6) * (uint32_t)Npad * 16u : 0u);
    // (a chunk's plane index 6c + piece is uniform: it travels in the scalar offset; the lane's row and k-group half in ONE per-lane offset per
    // operand -- with everything in per-lane offsets the compiler kept 42 address registers and spilled them.  A chunk past the end of K is
    // never requested; the second k-group of the last chunk exists in both piece planes, zero-filled: api.cpp alloc_weight, quantize.hip K1)
    auto chunk_loads = [&](int c, i32x4 (&sa_)[WNT][3], i32x4 (&mb_)[WMT][NPIECE], int m0_, int n0_) {
        const uint32_t vM = (uint32_t)((3 * hh) * Mpad + m0_ + l31) * 16u, vS = (uint32_t)((3 * hh) * Npad + n0_ + l31) * 16u;
        if (c < nchunks) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                if (pc < NPIECE)                            // (the weights' pieces first: they come from HBM, the activations' from L2)
#pragma unroll
                    for (int i = 0; i < WMT; ++i)
                        mb_[i][pc] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rMP, (int)(vM + 512u * i), (int)((uint32_t)(6 * c + pc) * (uint32_t)Mpad * 16u), 0));
#pragma unroll
                for (int j = 0; j < WNT; ++j)
                    sa_[j][pc] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rSP, (int)(vS + 512u * j), (int)((uint32_t)(6 * c + pc) * (uint32_t)Npad * 16u), 0));
            }
        } else {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                for (int i = 0; i < WMT; ++i) if (pc < NPIECE) mb_[i][pc] = i32x4{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < WNT; ++j) sa_[j][pc] = i32x4{0, 0, 0, 0};
            }
        }
    };
    // one tile's share of a chunk: the kept piece products, smallest first (activation piece PA x weight piece PB)
    auto chunk_tile = [&](const i32x4 (&sj)[3], const i32x4 (&mi)[NPIECE], f32x16 a) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        static_for<6>([&](auto cc) {
            constexpr int c6 = decltype(cc)::value;
            if constexpr (PB[c6] < NPIECE)
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sj[PA[c6]]), __builtin_bit_cast(bf16x8, mi[PB[c6]]), a, 0, 0, 0);
        });
        return a;
    };
  for (bool first = true;; first = false) {
    if (!first) __syncthreads();                            // (the previous tile's reduction has read its last partial sums)
    // (Q5_1 / Q4_1 / Q5_K) the min terms m * (s0 + s1) of SIXTEEN k-blocks of a tile as a K = 16 matrix product on the bf16 cores: both
    // operands live as three bf16 pieces that sum to the f32 value exactly (the weight's min plane at upload: ggml_hip_weight::mp3; d1 *
    // (float)sum(a) = the Q8_1 s0 + s1 of Ggml.cs:820-821 from K1: act_planes::sp3), six of the nine piece products are kept (K10d's
    // arithmetic, dense16.hip: every kept product is exact in f32, the dropped ones are below 2^-23 of the term; a Q5_1 min is an f16
    // value -- two pieces, five products) -- 40 / 48 matrix passes per tile and 16 k-blocks where the f32 instruction took 128.  Chunk c
    // (k-blocks 16c .. 16c + 15; lane half hh holds k-group 2c + hh) is wave c % 8's: by K alone, like everything else in an element's
    // summation order.  The wave's first chunk is requested HERE, in front of the scale table's loads, while the accumulators'
    // registers are still free.  (4096 x 11008 x 512: 77 us with the product, 71 without any min term, 86 with the f32 instruction inside
    // the loop; the product's MFMAs alone cost 3.5 us, its loads alone 6 -- +15 % operand bytes through L2 with every workgroup asking at
    // once, tools/k3p_trace.hip q51.)
    i32x4 sa[NPRE][WNT][3], mb[NPRE][WMT][NPIECE];
    if constexpr (MINP) {
#pragma unroll
        for (int u = 0; u < NPRE; ++u) chunk_loads(wave + KS * u, sa[u], mb[u], m0, n0);
    }
    load_scale_table(tabD, ad, kb0, ch, nbk, Npad, n0, lane);    // (image 0 does not write the k-blocks K is padded to: rows from nbk on are zero)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    f32x16 acc[WMT][WNT];
    if constexpr (MINP) {
        // tile by tile: a tile's accumulator comes to life as the pieces of its m-tile die (slot 0 always runs: zeros where the wave has no chunk)
        static_for<WMT * WNT>([&](auto tc) {
            constexpr int t = decltype(tc)::value, i = t / WNT, j = t % WNT;
            f32x16 a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            a = chunk_tile(sa[0][j], mb[0][i], a);
#pragma unroll
            for (int u = 1; u < NPRE; ++u)
                if (wave + KS * u < nchunks) a = chunk_tile(sa[u][j], mb[u][i], a);
            acc[i][j] = a;
        });
    } else {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }

    // (Q5_1 / Q4_1 / Q5_K) the rest of the wave's min-term chunks (K beyond what the slots above hold): loads, then MFMAs, chunk by chunk
    if constexpr (MINP) {
        for (int c = wave + KS * NPRE; c < nchunks; c += KS) {
            i32x4 sa_[WNT][3], mb_[WMT][NPIECE];
            chunk_loads(c, sa_, mb_, m0, n0);
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) acc[i][j] = chunk_tile(sa_[j], mb_[i], acc[i][j]);
        }
    }

    // one k-block: local index b, weights in w; the next block's go into wn; column tile j's fragment is refetched behind its last MFMA.
    // The 16 row scales of a column tile are read during the LAST tile of the column tile before it (the order pins below are
    // scheduling boundaries: a read issued where it is used costs its LDS latency twice per tile).
    f32x4 da[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) da[q] = *(const f32x4 *)(tabD + 4 * hh + 8 * q);
    i32x16 x[2];                                            // the products of the tile at hand and of the next one (x[0]: handed from block to block)
    auto block = [&](int b, int tb, WI8 &w, WI8 &wn) {     // (tb: the block's row in the table slice)
        if (wave >= KS / 2) { if (b & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }   // (see the MX kernel)
        load_w(wn, kb0 + b + 1);
        const float *dp = tabD + tb * (32 * WNT) + 4 * hh;
        // (x[0] arrives from the previous k-block: the product of a block's FIRST tile is issued in front of the previous block's last
        // scale-accumulates -- issued here it stood back to back with the second tile's, and the first conversion waited out both)
        static_for<WMT * WNT>([&](auto tc) {
            constexpr int t = decltype(tc)::value, j = t / WMT, i = t % WMT;
            if constexpr (t + 1 == WMT * WNT) {
                constexpr int jp = (t - 1) / WMT, ip = (t - 1) % WMT;
                asm volatile("" : "+v"(wn.q[0]), "+v"(acc[ip][jp]));
                x[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], wn.q[0], zero, 0, 0, 0);      // (the last block's: operands of the look-ahead, never used)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (t + 1 < WMT * WNT) {
                constexpr int j1 = (t + 1) / WMT, i1 = (t + 1) % WMT;
                // Order pin (empty statement, no instruction): the MFMA of tile t + 1 may not issue before the scale-accumulate of
                // tile t - 1 has finished -- left alone the compiler issues all eight MFMAs of a block up front (128 registers of
                // products: 860 bytes of scratch).
                if constexpr (t >= 1) {
                    constexpr int jp = (t - 1) / WMT, ip = (t - 1) % WMT;
                    asm volatile("" : "+v"(w.q[i1]), "+v"(acc[ip][jp]));
                }
                x[(t + 1) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[j1], w.q[i1], zero, 0, 0, 0);
                if constexpr (i1 == WMT - 1)
                    af[j1] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(offA + 512u * j1), (int)((uint32_t)(kb0 + b + 1) * a_blk), 0));
                // ... and not LATER than the start of tile t's scale-accumulates either (r4): left alone the compiler sinks it to the end of
                // tile t and issues the MFMAs in pairs, whose first conversion then waits out the matrix pipe, four times per k-block.
                // Nothing crosses this line; the scale reads are placed by hand (dn / in place), so they lose nothing.  Q5_0 4096 x 11008 x 512
                // 66.9 -> 65.0 us, Q5_1 71.9 -> 69.6, Q8_0 level.  (The MX loop above is better off with the compiler's pairs: the same line
                // there costs 3 to 11 %, its scale reads want to float.)
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 dn[DA_INPLACE ? 1 : 4];
            if constexpr (i == WMT - 1 && !DA_INPLACE) {    // the next column tile's scales (j + 1, or tile 0 of the next k-block: 64 floats on)
#pragma unroll
                for (int q = 0; q < 4; ++q) dn[q] = *(const f32x4 *)(dp + 32 * (j + 1) + 8 * q);
            }
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if constexpr (TYPE == GGML_TYPE_Q8_0)
                        acc[i][j][4 * q + e] = __builtin_fmaf((float)x[t & 1][4 * q + e], da[q][e] * w.d[i], acc[i][j][4 * q + e]);   // Ggml.cs:1377-1378
                    else
                        acc[i][j][4 * q + e] = __builtin_fmaf(w.d[i] * (float)x[t & 1][4 * q + e], da[q][e], acc[i][j][4 * q + e]);   // Ggml.cs:1296-1298, 1344
                // (Q5_1: no second register set for the next column tile's scales -- group q's are dead after its last m-tile and are
                // refetched into place, a whole tile step ahead of their next use)
                if constexpr (DA_INPLACE && i == WMT - 1) da[q] = *(const f32x4 *)(dp + 32 * (j + 1) + 8 * q);
            });
            if constexpr (i == WMT - 1 && !DA_INPLACE) {
#pragma unroll
                for (int q = 0; q < 4; ++q) da[q] = dn[q];
            }
        });
    };
    // (r5) the two-scale types -- Q4_2 (Ggml.cs:1217-1252), and Q6_K in its planar form: a k-block is two 16-element blocks with a scale each.  Bytes 0..7
    // of both operand planes are the first block's elements, bytes 8..15 the second's, so one v_mfma_i32_32x32x16_i8 per half gives the two integer sums
    // apart; per element sumf += (d_lo * yd) * sumi_lo, then += (d_hi * yd) * sumi_hi -- the statement of the batched-decode form (gemm_q8s.hip).  The loop
    // above with 2 x 8 steps per k-block: the product of step s + 1 is issued in front of step s's scale-accumulates.
    auto half8 = [](const i32x4 &v, int h) -> long { return (long)(uint32_t)v[2 * h] | ((long)v[2 * h + 1] << 32); };
    auto block2 = [&](int b, int tb, WI8 &w, WI8 &wn) {
        if (wave >= KS / 2) { if (b & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        load_w(wn, kb0 + b + 1);
        const float *dp = tabD + tb * (32 * WNT) + 4 * hh;
        static_for<2 * WMT * WNT>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value, t = s / 2, h = s % 2, j = t / WMT, i = t % WMT;
            if constexpr (s + 1 == 2 * WMT * WNT) {
                constexpr int tp = (s - 1) / 2, jp = tp / WMT, ip = tp % WMT;
                asm volatile("" : "+v"(wn.q[0]), "+v"(acc[ip][jp]));
                x[0] = __builtin_amdgcn_mfma_i32_32x32x16_i8(half8(af[0], 0), half8(wn.q[0], 0), zero, 0, 0, 0);   // (the last block's: operands of the look-ahead, never used)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (s + 1 < 2 * WMT * WNT) {
                constexpr int s1 = s + 1, t1 = s1 / 2, h1 = s1 % 2, j1 = t1 / WMT, i1 = t1 % WMT;
                if constexpr (s >= 1) {
                    constexpr int tp = (s - 1) / 2, jp = tp / WMT, ip = tp % WMT;
                    asm volatile("" : "+v"(w.q[i1]), "+v"(acc[ip][jp]));
                }
                x[s1 & 1] = __builtin_amdgcn_mfma_i32_32x32x16_i8(half8(af[j1], h1), half8(w.q[i1], h1), zero, 0, 0, 0);
                if constexpr (i1 == WMT - 1 && h1 == 1)
                    af[j1] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)(offA + 512u * j1), (int)((uint32_t)(kb0 + b + 1) * a_blk), 0));
                __builtin_amdgcn_sched_barrier(0);
            }
            const float sw = h ? w.m[TWO ? i : 0] : w.d[i];
            static_for<4>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[i][j][4 * q + e] = __builtin_fmaf(sw * da[q][e], (float)x[s & 1][4 * q + e], acc[i][j][4 * q + e]);   // Ggml.cs:1217-1252
                if constexpr (i == WMT - 1 && h == 1) da[q] = *(const f32x4 *)(dp + 32 * (j + 1) + 8 * q);
            });
        });
    };
    K3P_STAMP(1);
#ifdef K3P_TRACE
    const unsigned long long clk0 = __builtin_readcyclecounter();
#endif
    if constexpr (TWO) x[0] = __builtin_amdgcn_mfma_i32_32x32x16_i8(half8(af[0], 0), half8(w0.q[0], 0), zero, 0, 0, 0);
    else x[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], w0.q[0], zero, 0, 0, 0);
    for (int b = 0, tb = 0; b < nloc; b += 2, tb += 2) {    // (the look-ahead of the last trip reads past the wave's range: never used)
        if (SLICED && tb == ch) {                           // (K beyond 19968: the next `ch` rows of every wave's table; `ch` is even.  The scales the last
            tb = 0;                                         // tile prefetched from the row behind the slice are replaced here)
            load_scale_table(tabD, ad, kb0 + b, nloc - b < ch ? nloc - b : ch, nbk, Npad, n0, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int q = 0; q < 4; ++q) da[q] = *(const f32x4 *)(tabD + 4 * hh + 8 * q);
        }
        if constexpr (TWO) { block2(b, SLICED ? tb : b, w0, w1); block2(b + 1, (SLICED ? tb : b) + 1, w1, w0); }
        else { block(b, SLICED ? tb : b, w0, w1); block(b + 1, (SLICED ? tb : b) + 1, w1, w0); }
    }
#ifdef K3P_TRACE
    asm volatile("" : "+v"(acc[0][0]));
    if (lane == 0 && (size_t)blockIdx.x * 8 + wave < 4096) k3p_trace_buf[((size_t)blockIdx.x * 8 + wave) * 8 + 7] = __builtin_readcyclecounter() - clk0;
#endif
    K3P_STAMP(2);
    __builtin_amdgcn_s_setprio(0);
    const int m0c = m0, n0c = n0;
    tile += (int)gridDim.x;
    if (tile < nwg) {                                       // the next tile's first operands travel while this one is reduced and stored
        tile_origin<WMT>(tile, tiles_m, tiles_n, m0, n0);
        set_offsets(m0, n0);
        load_first();
    }
    reduce_and_store<WMT>(acc, smem, dst, M, N, ldd, m0c, n0c, wave, lane, ep);
    if (tile >= nwg) break;
  }
    K3P_STAMP(5);
}

}  // namespace

// workgroups of a launch: every tile its own up to four rounds of the chip, beyond that one persistent workgroup per CU (a multiple of 8: XCD order)
static unsigned persistent_grid(int tiles) {
    static std::atomic<int> cus[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int n = cus[dev & 63].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
        n = n / 8 * 8;
        if (n < 8) n = 8;
        cus[dev & 63].store(n, std::memory_order_relaxed);
    }
    // Persistent only from four rounds on: the hardware hands a free CU the next workgroup, a persistent grid deals the tiles out
    // in advance -- 688 tiles (11008 x 4096 x 512) are 2.7 rounds dispatched and 3 dealt (79 against 84 us), 2000 tiles 7.8 either way
    // (210 against 205 us: there the workgroup start-up and the next tile's first loads behind the reduction are what is saved).
    return (unsigned)(tiles < 4 * n ? tiles : n);
}

hipError_t launch_gemm_q8_mid(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue &ep) {
    const bool with_min = w->type == GGML_TYPE_Q5_1 || w->type == GGML_TYPE_Q4_1;
    const bool two = w->type == GGML_TYPE_Q4_2;             // (and Q6_K in its planar form: int8 planes, the k-block's second scale in the m plane)
    const uint8_t *planes = w->type == GGML_TYPE_Q8_0 ? w->qs : (w->type == GGML_TYPE_Q5_0 || with_min || two) ? w->i8p : nullptr;
    if (pl.family != MMF_K3P_I8 || !planes || !w->d || (with_min && (!w->m || !w->mp3 || !p.sp3)) || (two && !w->m)) return hipErrorInvalidValue;
    if (ldd > 0x7FFFFFFF) return hipErrorNotSupported;     // (the kernel carries the row stride as an int: callers with an unfused form fall back)
    // applicability (at least 8 k-blocks per wave, the eight scale tables within 160 KB of LDS, every offset within 32 bits) and the
    // k-blocks per wave were decided by plan.cpp (plan_k3p_i8); the checks below only guard the kernel's assumptions
    const int nbkp = (int)pad_kblocks(w->nbk);
    const int nloc = pl.nloc;
    const uint64_t wq_bytes = (uint64_t)nbkp * 2 * (uint64_t)w->Mpad * 16, aq_bytes = (uint64_t)nbkp * 2 * (uint64_t)p.Npad * 16;
    if (nloc < 8 || (nloc & 1) || KS * nloc < nbkp) return hipErrorInvalidValue;
    if ((uint64_t)(KS * nloc + 2) * 2 * (uint64_t)w->Mpad * 16 > 0xFFFFFFFFull || (uint64_t)(KS * nloc + 2) * 2 * (uint64_t)p.Npad * 16 > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const int wmt = pl.tile_m / 32;                         // (r5: 128-row wave tiles, or 64-row ones where those leave CUs idle: plan_k3p_i8 -- the same tree)
    if ((wmt != 4 && wmt != 2) || w->Mpad % (32 * wmt) != 0 || p.Npad % (32 * WNT) != 0) return hipErrorInvalidValue;
    const int tiles_m = (int)((w->M + 32 * wmt - 1) / (32 * wmt)), tiles_n = (int)((N + 32 * WNT - 1) / (32 * WNT));
    // (K > 20480: table slices of `ch` k-blocks per wave, + one row behind the last slice for the loop's look-ahead of the next row's scales)
    const bool sliced = nloc > K3P_TABLE_ROWS;
    const int nch = (nloc + K3P_SLICE_ROWS - 1) / K3P_SLICE_ROWS, ch = sliced ? ((nloc + nch - 1) / nch + 1) & ~1 : nloc;
    const size_t tab = ((size_t)KS * ch + (sliced ? 1 : 0)) * (32 * WNT) * 4, xch = (size_t)KS * 4 * 16 * 64 * 4;
    const size_t lds = tab > xch ? tab : xch;
    if (lds > 160 * 1024 || ch > nloc) return hipErrorInvalidValue;
    (void)hipGetLastError();
#define Q8MID_GO(...) do { if (wmt == 2) { if (sliced) Q8MID_GO1(__VA_ARGS__, true, 2); else Q8MID_GO1(__VA_ARGS__, false, 2); } \
                           else { if (sliced) Q8MID_GO1(__VA_ARGS__, true, 4); else Q8MID_GO1(__VA_ARGS__, false, 4); } } while (0)
#define Q8MID_GO1(...) do { \
        auto kern = gemm_q8_mid_kernel<__VA_ARGS__>; \
        static PerDeviceOnce once; \
        const hipError_t attr = once.max_dynamic_lds((const void *)kern, 160 * 1024); \
        if (attr != hipSuccess) return attr; \
        kern<<<dim3(persistent_grid(tiles_m * tiles_n)), KS * 64, lds, st>>>(planes, w->d, p.a8, p.ad, dst, (int)w->M, (int)N, (int)w->Mpad, (int)p.Npad, \
                                                                        (int)w->nbk, nloc, (int)ldd, tiles_m, tiles_n, (uint32_t)wq_bytes, (uint32_t)aq_bytes, ep, two ? (const uint8_t *)w->m : w->mp3, p.sp3, ch); } while (0)
    // (Q4_1: the kernel of Q5_1 -- unsigned values 0..15 on the int8 planes, the same min term)
    if (w->type == GGML_TYPE_Q8_0) Q8MID_GO(GGML_TYPE_Q8_0, false); else if (w->type == GGML_TYPE_Q5_0) Q8MID_GO(GGML_TYPE_Q5_0, false);
    else if (two) Q8MID_GO(GGML_TYPE_Q4_2, false);
    // (the min of a Q5_1 block is an f16 value: two bf16 pieces; Q4_1's is an f32, Q5_K's an f32 product: three)
    else if (w->ext_type != 0 || w->type == GGML_TYPE_Q4_1) Q8MID_GO(GGML_TYPE_Q5_1, true); else Q8MID_GO(GGML_TYPE_Q5_1, false);
#undef Q8MID_GO
#undef Q8MID_GO1
    return hipGetLastError();
}

// Q4_0 (plan.cpp plan_k3p_mx: at least 8 k-blocks per wave, tables within LDS, offsets within 32 bits)
hipError_t launch_gemm_qmx_mid(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue &ep) {
    if (pl.family != MMF_K3P_MX || w->type != GGML_TYPE_Q4_0 || !w->q6a || !w->q6b) return hipErrorInvalidValue;
    if (ldd > 0x7FFFFFFF) return hipErrorNotSupported;     // (see launch_gemm_q8_mid)
    const int nbkp = (int)pad_kblocks(w->nbk);
    const int nloc = pl.nloc;
    if (nloc < 8 || KS * nloc < nbkp) return hipErrorInvalidValue;
    const uint64_t nba = (uint64_t)nbkp;
    const uint64_t wq_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 16, wd_bytes = (nba + K_LOOKAHEAD) * (uint64_t)w->Mpad * 4;
    const uint64_t a_bytes = nba * 48 * (uint64_t)p.Npad;
    // (32-bit buffer offsets, the look-ahead past a wave's range included)
    if ((uint64_t)(KS * nloc + 2) * (uint64_t)w->Mpad * 16 > 0xFFFFFFFFull || (uint64_t)(KS * nloc + 2) * 48 * (uint64_t)p.Npad > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const int wmt = pl.tile_m / 32;                         // (r5: 128-row wave tiles, or 64-row ones where those leave CUs idle: plan_k3p_mx -- the same tree)
    if ((wmt != 4 && wmt != 2) || w->Mpad % (32 * wmt) != 0 || p.Npad % (32 * WNT) != 0) return hipErrorInvalidValue;
    const int tiles_m = (int)((w->M + 32 * wmt - 1) / (32 * wmt)), tiles_n = (int)((N + 32 * WNT - 1) / (32 * WNT));
    const bool sliced = nloc > K3P_TABLE_ROWS;
    const int nch = (nloc + K3P_TABLE_ROWS - 1) / K3P_TABLE_ROWS, ch = (nloc + nch - 1) / nch;
    const size_t tab = (size_t)KS * ch * (32 * WNT) * 4, xch = (size_t)KS * 4 * 16 * 64 * 4;
    const size_t lds = tab > xch ? tab : xch;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = wmt == 2 ? (sliced ? gemm_qmx_mid_kernel<true, 2> : gemm_qmx_mid_kernel<false, 2>) : (sliced ? gemm_qmx_mid_kernel<true, 4> : gemm_qmx_mid_kernel<false, 4>);
    static PerDeviceOnce once[4];
    const hipError_t attr = once[(wmt == 2 ? 2 : 0) + (sliced ? 1 : 0)].max_dynamic_lds((const void *)kern, 160 * 1024);
    if (attr != hipSuccess) return attr;
    (void)hipGetLastError();
    kern<<<dim3(persistent_grid(tiles_m * tiles_n)), KS * 64, lds, st>>>(w->q6a, w->q6b, w->d, (const uint8_t *)p.a8, p.ad, dst, (int)w->M, (int)N,
                                                                                     (int)w->Mpad, (int)p.Npad, nbkp, nloc, (int)ldd, tiles_m, tiles_n,
                                                                                     (uint32_t)wq_bytes, (uint32_t)wd_bytes, (uint32_t)a_bytes, ep, ch);
    return hipGetLastError();
}
