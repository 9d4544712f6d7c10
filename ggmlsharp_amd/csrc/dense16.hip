// dense16.hip -- K10b: dense f16 x f32 mul_mat (ggml_compute_forward_mul_mat_f16_f32, Ggml.cs:6180-6438) on the f16
// matrix cores, for every shape with more than 4 src1 rows (fewer: the mat-vec form of dense.hip).
//
// The reference's INIT phase rounds src1 to Half (Ggml.cs:6362-6379) and ggml_vec_dot_f16 (Ggml.cs:2642-2651) sums
// (float)h * (float)h in f64.  Here: convert_act_f16 does the same rounding (round-to-nearest-even, as (Half)x) into a
// panel image, v_mfma_f32_32x32x16_f16 forms the products exactly (22-bit significands) and accumulates in f32 --
// the reference's sum with f32 instead of f64 accumulation, ~sqrt(K) * 2^-24 relative, inside the 1e-3 budget.
// No scale work, no VALU in the loop: this is the one kernel of the path that the matrix pipe bounds.
//
// Layouts (both built once: the weights at upload, the activations by the INIT kernel below): k-panel-major,
// [K/8][rows][8 x f16 = 16 B], K zero-padded to whole stages (128) -- panel p of row r is one 16-byte piece, 64 rows of a
// panel are one contiguous 1-KiB DMA piece, and the MFMA fragment of lane (row l & 31, half h) for k-step ks is the
// 16-byte entry of panel 2 ks + h: one conflict-free ds_read_b128 (activations, via LDS) or one buffer_load_dwordx4
// straight into the B operand (weights, from L2, four k-steps ahead).
// Structure as gemm_qmx.hip: 4 waves, wave tile 64 (m) x 128 (n) = 2 x 4 MFMA tiles, workgroup tile 256 x 128, two
// workgroups per CU, activations by buffer_load ... lds DMA in double-buffered stages of 8 k-steps, one barrier per stage.
#include "common.h"
#include "plan.h"
#include <hip/hip_fp16.h>
#include <cstdlib>
#include <utility>

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

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

// ---- INIT: src1 f32 rows -> f16 panel image [Kpad/8][Npad][16 B]; lane = 8 * row_in_wave + t owns 4 consecutive floats ----
#define CV_CH 4   // 128-byte chunks (32 floats = 4 panels) per lane group and workgroup column
__global__ __launch_bounds__(256) void convert_act_f16_kernel(const float *__restrict__ x, int64_t N, int64_t K, int64_t Kpad, int64_t ld1,
                                                             uint8_t *__restrict__ img, int64_t Npad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, t = lane & 7;
    const int64_t n = (int64_t)blockIdx.y * 32 + wave * 8 + (lane >> 3);
    const bool live = n < N;
    const float *row = x + (live ? n : N - 1) * ld1;
#pragma unroll
    for (int j = 0; j < CV_CH; ++j) {
        const int64_t k = ((int64_t)blockIdx.x * CV_CH + j) * 32 + 4 * t;
        if (k >= Kpad) break;                                          // uniform per workgroup column
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 3 < K) v = *(const float4 *)(row + k);
        else {
            if (k + 0 < K) v.x = row[k + 0];
            if (k + 1 < K) v.y = row[k + 1];
            if (k + 2 < K) v.z = row[k + 2];
        }
        // (Half)x: round-to-nearest-even, overflow to infinity (Ggml.cs:6369)
        const uint32_t h0 = __builtin_bit_cast(uint16_t, (_Float16)v.x), h1 = __builtin_bit_cast(uint16_t, (_Float16)v.y);
        const uint32_t h2 = __builtin_bit_cast(uint16_t, (_Float16)v.z), h3 = __builtin_bit_cast(uint16_t, (_Float16)v.w);
        if (live) *(uint2 *)(img + ((k >> 3) * Npad + n) * 16 + 8 * (t & 1)) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
    }
}

// ---- weights: f16 rows -> panels [Kpad/8 + spare][Mpad][16 B] (zero past K and past M: the buffer is pre-zeroed) ----
__global__ void f16_rows_to_panels_kernel(const uint16_t *__restrict__ rows, int64_t M, int64_t K, int64_t Mpad, uint8_t *__restrict__ pan) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    if (m >= M) return;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t k = p * 8 + e;
        if (k < K) w[e >> 1] |= (uint32_t)rows[m * K + k] << (16 * (e & 1));
    }
    *(uint4 *)(pan + (p * Mpad + m) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

// Workgroup -> tile for grids of SEVERAL rounds (dense16s / dense32s): the hardware deals workgroups to the eight XCDs in turn
// (bid & 7) and an XCD keeps about 64 of them resident, so an XCD's run of the tile list is ordered in 8 x 8 blocks of tiles (m fastest
// inside a block, blocks down a block-column, block-columns across n): the 64 resident tiles share 8 weight panels and 8 activation
// panels.  With the list ordered m-fastest over ALL rows (the one-round order of the kernels above) an XCD streams every weight panel
// once per tile column: 11008 x 4096 x 4096 F16 pulled ~2.9 GB through the L2s for 210 MB of operands and ran at the HBM rate.
__device__ __forceinline__ void tile_of_blocked(int bid, int tiles_m, int tiles_n, int &tm_i, int &tn_i) {
    const int nwg = tiles_m * tiles_n, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);       // position in the blocked list
    const int colsz = tiles_m * 8, nbn = (tiles_n + 7) >> 3;
    int bj = t / colsz;
    if (bj > nbn - 1) bj = nbn - 1;
    const int u = t - bj * colsz, wd = tiles_n - bj * 8 < 8 ? tiles_n - bj * 8 : 8;               // inside block-column bj, wd tiles wide
    const int nbm = (tiles_m + 7) >> 3;
    int bi = u / (8 * wd);
    if (bi > nbm - 1) bi = nbm - 1;
    const int r = u - bi * 8 * wd, h = tiles_m - bi * 8 < 8 ? tiles_m - bi * 8 : 8;
    tm_i = bi * 8 + r % h;
    tn_i = bj * 8 + r / h;
}

constexpr int KS = 8;       // k-steps of 16 per LDS stage (128 k)
#ifndef D16_RING
#define D16_RING 4
#endif
#ifndef D16_FLATMAP       // developer A/B: 1 = the one-round tile order for every grid
#define D16_FLATMAP 0
#endif
#ifndef D16_AF_AHEAD
#define D16_AF_AHEAD 2
#endif
constexpr int RING = D16_RING;   // weight fragments in flight, in k-steps (A/B: -DD16_RING=n)

template <int WMT, int WNT, int WGM, int WGN>
struct Cfg {
    static constexpr int TM = WGM * WMT * 32, TN = WGN * WNT * 32, NT = WGM * WGN * 64;
    static constexpr int STAGE = KS * 2 * TN * 16;
    static constexpr int TOTAL = 2 * STAGE;
    static constexpr int P = NT / TN;
    static_assert(NT % TN == 0 && (KS * 2) % P == 0, "chunk decomposition");
    static constexpr int ROUNDS = KS * 2 * TN / NT;
    // DMA pieces per n-tile step: all of them are issued before the drain point (k-step KS - 2)
    static constexpr int PP = (ROUNDS + (KS - 2) * WNT - 1) / ((KS - 2) * WNT);
    static_assert(PP <= 2, "at most two DMA pieces per n-tile step");
};

// KSP = 1 | 2 | 4: K split inside the workgroup as in gemm_qmx.hip -- KSP wave groups take alternate stages (own stage buffers)
// and add their tiles through LDS in group order at the end; for batches that leave most of the chip idle otherwise.
// VS = 2: the two-way tree WITHOUT the second wave group ("virtual" split, as gemm_qmx.hip): one group runs the stage sets of both groups one
// after the other, banks the first sum and adds the second to it -- group 0 + group 1, the addition the split form makes, bit for bit -- on
// the 4-wave geometry that keeps two workgroups per CU.  For grids of many tiles, where the 8-wave form's one workgroup per CU costs rounds.
template <int WMT, int WNT, int WGM, int WGN, int KSP, int VS = 1>
__global__ __launch_bounds__(WGM * WGN * 64 * KSP, 2)
void dense16_kernel(const uint8_t *__restrict__ wpan, const uint8_t *__restrict__ apan, float *__restrict__ dst, int M, int N, int Mpad,
                    int Npad, int nstages, int ldd, int tiles_m, int tiles_n, uint32_t w_bytes, uint32_t a_bytes) {
    using C = Cfg<WMT, WNT, WGM, WGN>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave_all / (WGM * WGN), wave = wave_all % (WGM * WGN);   // K-split group, wave inside it
    const int tid = wave * 64 + lane;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wn = wave / WGM, wm_ = wave % WGM;
    uint8_t *const gsm = smem + grp * C::TOTAL;

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    int tm_i, tn_i;
    if (nwg > 512 && !D16_FLATMAP) {                          // several rounds: 8 x 8 blocks of tiles per XCD (tile_of_blocked above)
        tile_of_blocked(bid, tiles_m, tiles_n, tm_i, tn_i);
    } else if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0) {
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

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // activations: DMA, chunk c = tid + NT * i of a stage = [16 panels][TN rows] x 16 B
    const uint32_t a_pan = (uint32_t)(Npad * 16);
    const uint32_t voffA = (uint32_t)(((tid / C::TN) * Npad + n0 + tid % C::TN) * 16);
    const rsrc_t rA = make_rsrc(apan, a_bytes);
    auto dma_piece = [&](int s, int buf, auto pc) {           // stage s of K -> stage buffer buf of this group
        constexpr int i = decltype(pc)::value;
        uint8_t *sp = gsm + (buf & 1) * C::STAGE;
        blds16(rA, sp + (size_t)(wave * 64 + C::NT * i) * 16, voffA, ((uint32_t)s * KS * 2 + C::P * i) * a_pan);
    };

    // weights: registers, RING k-steps ahead
    const uint32_t w_pan = (uint32_t)(Mpad * 16);
    const uint32_t voffW = (uint32_t)((hh * Mpad + m0 + wm_ * WMT * 32 + l31) * 16);
    const rsrc_t rW = make_rsrc(wpan, w_bytes);
    u32x4 bq[RING][WMT];
    auto load_b = [&](int gstep, auto rc) {                 // global k-step -> ring slot
        constexpr int slot = decltype(rc)::value;
#pragma unroll
        for (int i = 0; i < WMT; ++i)
            bq[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(voffW + 512u * i), (int)((uint32_t)gstep * 2 * w_pan), 0);
    };

    // The stage's A fragments (activations, LDS) run AF_AHEAD n-tile steps ahead of the MFMAs that use them: a ds_read_b128 issued
    // right in front of its MFMAs exposes the whole LDS latency 32 times per stage (what hipcc makes of the plain loop).
    constexpr int AF_AHEAD = D16_AF_AHEAD, NSTEP = KS * WNT;
    static_assert(VS == 1 || KSP == 1, "virtual split: one wave group");
    auto compute = [&](int s, int sn, int buf) {              // stage s; sn = the stage this group runs next (prefetch target)
        const uint8_t *sA = gsm + (buf & 1) * C::STAGE + ((size_t)(hh * C::TN + wn * WNT * 32 + l31)) * 16;
        f16x8 afr[AF_AHEAD + 1];
        auto fetch_af = [&](auto gc) {
            constexpr int g = decltype(gc)::value, ks = g / WNT, j = g % WNT;
            afr[g % (AF_AHEAD + 1)] = *(const f16x8 *)(sA + (ks * 2 * C::TN + 32 * j) * 16);
        };
        static_for<AF_AHEAD>([&](auto gc) { fetch_af(gc); });
        static_for<KS>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            static_for<WNT>([&](auto jc) {
                constexpr int j = decltype(jc)::value, g = ks * WNT + j;
                static_for<C::PP>([&](auto uc) {
                    constexpr int pc = C::PP * (ks * WNT + j) + decltype(uc)::value;
                    if constexpr (pc < C::ROUNDS) dma_piece(sn, buf + 1, std::integral_constant<int, pc>{});
                });
                if constexpr (g + AF_AHEAD < NSTEP) fetch_af(std::integral_constant<int, g + AF_AHEAD>{});
                const f16x8 af = afr[g % (AF_AHEAD + 1)];
#pragma unroll
                for (int i = 0; i < WMT; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(f16x8, bq[ks % RING][i]), acc[i][j], 0, 0, 0);
                // hipcc's scheduler otherwise sinks the look-ahead reads (LDS fragments, weight ring) down to their uses
                __builtin_amdgcn_sched_barrier(0);
            });
            // The DMA pieces of the next stage must have landed before the barrier.  They are all issued by k-step KS_DMA; vector-memory
            // operations complete in issue order, so waiting until only the weight loads issued SINCE then are outstanding is enough --
            // vmcnt(0) here (and again at the barrier) made every wave sit out the L2 latency of its youngest weight loads once per stage.
#ifndef D16_NODRAIN   /* timing ablation only: without the drain the kernel races */
            if constexpr (ks == KS - 2) {
                constexpr int KS_DMA = ((C::ROUNDS - 1) / C::PP) / WNT;               // k-step of the last DMA piece
                constexpr int YOUNGER = (KS - 2 - KS_DMA) * WMT;                        // weight loads issued behind it (end of k-steps KS_DMA .. KS - 3)
                static_assert(KS_DMA <= KS - 3 && YOUNGER >= 0 && YOUNGER < 64, "drain point");
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            }
#endif
            // this group's next k-steps: the rest of the stage, then its next stage
            load_b(ks + RING < KS ? s * KS + ks + RING : sn * KS + (ks + RING - KS), std::integral_constant<int, ks % RING>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    static_assert(RING <= KS, "weight look-ahead stays within two stages of a group");
    static_for<C::ROUNDS>([&](auto pc) { dma_piece(grp, 0, pc); });
    static_for<RING>([&](auto rc) { load_b(grp * KS + decltype(rc)::value, rc); });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (VS == 1) {
        const int niter = (nstages + KSP - 1) / KSP;          // every wave passes the same number of barriers
        for (int it = 0; it < niter; ++it) {
            const int s = it * KSP + grp;
            if (KSP == 1 || s < nstages) compute(s, s + KSP, it);
            // (KSP > 1: a group may skip its last stage, and with it the drain inside compute; KSP == 1: the weight loads in flight stay in flight)
            if constexpr (KSP == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    } else {
        // passes 0 .. VS-1 over stages p, p + VS, ...; the launcher guarantees nstages >= VS (no pass is empty)
        f32x16 sav[WMT][WNT];
        int s = 0, pass = 0;
        for (int it = 0; it < nstages; ++it) {
            int sn = s + VS, pn = pass;
            if (sn >= nstages) { pn = pass + 1; sn = pn < VS ? pn : nstages; }
            compute(s, sn, it);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (pn != pass && pn < VS) {                      // (uniform) pass boundary: bank this pass's sum in group order, start the next from zero
#pragma unroll
                for (int i = 0; i < WMT; ++i)
#pragma unroll
                    for (int j = 0; j < WNT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            sav[i][j][r] = pass == 0 ? acc[i][j][r] : sav[i][j][r] + acc[i][j][r];
                            acc[i][j][r] = 0.0f;
                        }
            }
            s = sn; pass = pn;
        }
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = sav[i][j][r] + acc[i][j][r];
    }

    // ---- K split: groups 1 .. KSP-1 hand their tiles to group 0 through LDS; group 0 adds them in group order ----
    if constexpr (KSP > 1) {
        constexpr int NTILE = WMT * WNT, GRP_FLOATS = WGM * WGN * NTILE * 16 * 64;
        static_assert((KSP - 1) * GRP_FLOATS * 4 <= KSP * C::TOTAL, "K-split exchange fits the stage buffers");
        float *xch = (float *)smem + (size_t)wave * (NTILE * 16 * 64) + lane;
        if (grp != 0) {
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];
                        xch[(size_t)(grp - 1) * GRP_FLOATS + ((i * WNT + j) * 16 + r) * 64] = v;
                    }
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

    // dst[n][m] (Ggml.cs:6692-6697): D[row = (r&3) + 8*(r>>2) + 4*hh][col = lane & 31]
    const rsrc_t rD = make_rsrc(dst + (size_t)n0 * ldd + m0, 0xFFFFFFFFu);
    const bool full = n0 + C::TN <= N && m0 + C::TM <= M;
    const uint32_t lane_off = (uint32_t)((4 * hh * ldd + l31) * 4);
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int mb = (wm_ * WMT + i) * 32, nb = (wn * WNT + j) * 32;
            if (!full && (m0 + mb >= M || n0 + nb >= N)) continue;
            const bool mok = full || m0 + mb + l31 < M;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nr = nb + (r & 3) + 8 * (r >> 2);
                const float v = acc[i][j][r];     // (a bit_cast applied directly to the vector element stores element 0 every time)
                if (full || (mok && n0 + nr + 4 * hh < N))
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), rD, (int)lane_off,
                                                          (int)((uint32_t)(nr * ldd + mb) * 4u), 0);
            }
        }
}

// ---- the same product on v_mfma_f32_16x16x32_f16 (K10c): the forms that do not split K (more than 512 src1 rows) ----
// Same panels, same stage image, same DMA, same wave tile (64 x 128 or 64 x 64 outputs); the tile is cut into 16 x 16 MFMA tiles and a
// k-step is 32: lane (row l & 15, k-group g = l >> 4) takes the 16-byte entry of panel 4 ks + g -- a 16-lane group reads 256
// contiguous bytes of LDS.  Flops per cycle are those of the 32 x 32 x 16 shape; what differs is the clock the chip holds under this
// load (MI355X_MICROARCH.md, DVFS give-back item 7: bare loops on random data ran 1.12-1.15 x the FLOP/s on this shape at equal
// cycles).  The sum of an output element: k in groups of 32 inside the MFMA, groups in order -- fixed by N and K like every form here.
template <int WMT, int WNT, int WGM, int WGN>
struct CfgS {
    static constexpr int TM = WGM * WMT * 16, TN = WGN * WNT * 16, NT = WGM * WGN * 64;
    static constexpr int KS32 = KS / 2;                       // k-steps of 32 per stage
    static constexpr int STAGE = KS * 2 * TN * 16, TOTAL = 2 * STAGE;
    static constexpr int P = NT / TN;
    static_assert(NT % TN == 0 && (KS * 2) % P == 0, "chunk decomposition");
    static constexpr int ROUNDS = KS * 2 * TN / NT;          // DMA pieces per stage, one per n-tile step from the top of the stage
    static constexpr int NSTEP = KS32 * WNT;
    static_assert(ROUNDS <= (KS32 - 1) * WNT, "all DMA pieces are issued before the stage's last k-step");
};

template <int WMT, int WNT, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64, 2)
void dense16s_kernel(const uint8_t *__restrict__ wpan, const uint8_t *__restrict__ apan, float *__restrict__ dst, int M, int N, int Mpad,
                     int Npad, int nstages, int ldd, int tiles_m, int tiles_n, uint32_t w_bytes, uint32_t a_bytes) {
    using C = CfgS<WMT, WNT, WGM, WGN>;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    constexpr int KS32 = C::KS32, RING32 = 2;                 // weight fragments in flight, in k-steps of 32
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wn = wave / WGM, wm_ = wave % WGM;

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    int tm_i, tn_i;
    if (nwg > 512 && !D16_FLATMAP) {                          // more than one round of the chip: 8 x 8 blocks of tiles per XCD
        tile_of_blocked(bid, tiles_m, tiles_n, tm_i, tn_i);
    } else if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0) {    // 2 x 4 blocks of the tile grid per XCD, as dense16_kernel
        const int hm = tiles_m >> 1, l = bid >> 3;
        tm_i = (xcd & 1) * hm + l % hm;
        tn_i = (xcd >> 1) * (tiles_n >> 2) + l / hm;
    } else {
        const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        tm_i = t_lin % tiles_m;
        tn_i = t_lin / tiles_m;
    }
    const int m0 = tm_i * C::TM, n0 = tn_i * C::TN;

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    const uint32_t a_pan = (uint32_t)(Npad * 16);
    const uint32_t voffA = (uint32_t)(((tid / C::TN) * Npad + n0 + tid % C::TN) * 16);
    const rsrc_t rA = make_rsrc(apan, a_bytes);
    auto dma_piece = [&](int s, int buf, auto pc) {
        constexpr int i = decltype(pc)::value;
        uint8_t *sp = smem + (buf & 1) * C::STAGE;
        blds16(rA, sp + (size_t)(wave * 64 + C::NT * i) * 16, voffA, ((uint32_t)s * KS * 2 + C::P * i) * a_pan);
    };

    const uint32_t w_pan = (uint32_t)(Mpad * 16);
    const uint32_t voffW = (uint32_t)((g4 * Mpad + m0 + wm_ * WMT * 16 + l15) * 16);
    const rsrc_t rW = make_rsrc(wpan, w_bytes);
    u32x4 bq[RING32][WMT];
    auto load_b = [&](int gstep, auto rc) {                 // global k-step of 32 -> ring slot
        constexpr int slot = decltype(rc)::value;
#pragma unroll
        for (int i = 0; i < WMT; ++i)
            bq[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(voffW + 256u * i), (int)((uint32_t)gstep * 4 * w_pan), 0);
    };

    constexpr int AF_AHEAD = D16_AF_AHEAD;
    auto compute = [&](int s, int buf) {
        const uint8_t *sA = smem + (buf & 1) * C::STAGE + ((size_t)(g4 * C::TN + wn * WNT * 16 + l15)) * 16;
        f16x8 afr[AF_AHEAD + 1];
        auto fetch_af = [&](auto gc) {
            constexpr int g = decltype(gc)::value, ks = g / WNT, j = g % WNT;
            afr[g % (AF_AHEAD + 1)] = *(const f16x8 *)(sA + (ks * 4 * C::TN + 16 * j) * 16);
        };
        static_for<AF_AHEAD>([&](auto gc) { fetch_af(gc); });
        static_for<KS32>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            static_for<WNT>([&](auto jc) {
                constexpr int j = decltype(jc)::value, g = ks * WNT + j;
                if constexpr (g < C::ROUNDS) dma_piece(s + 1, buf + 1, std::integral_constant<int, g>{});
                if constexpr (g + AF_AHEAD < C::NSTEP) fetch_af(std::integral_constant<int, g + AF_AHEAD>{});
                const f16x8 af = afr[g % (AF_AHEAD + 1)];
#pragma unroll
                for (int i = 0; i < WMT; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, bq[ks % RING32][i]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            // next stage's DMA pieces landed before the barrier: all but the weight loads issued since the last piece (see dense16_kernel)
            if constexpr (ks == KS32 - 2) {
                constexpr int KS_DMA = (C::ROUNDS - 1) / WNT;
                constexpr int YOUNGER = (KS32 - 2 - KS_DMA) * WMT;
                static_assert(KS_DMA <= KS32 - 2 && YOUNGER >= 0, "drain point");
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            }
            load_b(ks + RING32 < KS32 ? s * KS32 + ks + RING32 : (s + 1) * KS32 + (ks + RING32 - KS32), std::integral_constant<int, ks % RING32>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    static_for<C::ROUNDS>([&](auto pc) { dma_piece(0, 0, pc); });
    static_for<RING32>([&](auto rc) { load_b(decltype(rc)::value, rc); });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // r4: two workgroups share a CU, one wave of each per SIMD, and the arbiter serves the older wave first (gemm_qmx.hip has the
    // measurement): on one-round grids the younger wave -- odd wave slot -- takes priority on two stages of three, so both workgroups finish
    // together.  Speed only: 4096^3 129.2 -> 126.4 us, 4096 x 11008 x 2048 193.4 -> 185.4 us (INIT + COMPUTE, A/B in one gpurun call); the
    // split-bf16 F32 kernel measured level (632 us) and goes without.
#ifndef D16_PRIO
#define D16_PRIO 3
#endif
    bool younger = false;
    if constexpr (D16_PRIO != 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        younger = (hwid & 1u) != 0 && tiles_m * tiles_n <= 512;
    }
    for (int it = 0; it < nstages; ++it) {
        if constexpr (D16_PRIO != 0) { if (younger) { if (it % D16_PRIO != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } }
        compute(it, it);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // dst[n][m]: the MFMA leaves D[row = 4 * (lane >> 4) + r][col = lane & 15] of a 16 x 16 tile -- a store of one accumulator register
    // would write four 64-byte pieces of four rows.  The wave's four m-tiles of one n-tile are 64 consecutive m: transpose (m-tile
    // index) x (lane group) with two rounds of lane swaps, and a store writes ONE row of dst, 256 contiguous bytes.
    static_assert(WMT == 4, "the store transpose pairs four m-tiles with the four 16-lane groups");
    const rsrc_t rD = make_rsrc(dst + (size_t)n0 * ldd + m0, 0xFFFFFFFFu);
    const bool full = n0 + C::TN <= N && m0 + C::TM <= M;
    const int mw = wm_ * WMT * 16;                            // the wave's first m inside the workgroup tile
    const bool mok = full || m0 + mw + lane < M;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int nb = (wn * WNT + j) * 16;
        if (!full && n0 + nb >= N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float f = acc[i][j][r]; v[i] = __builtin_bit_cast(uint32_t, f); }
            // (a, b) -> ({a.lo, b.lo}, {a.hi, b.hi}) on halves, then ({a.r0, b.r0, a.r2, b.r2}, {a.r1, b.r1, a.r3, b.r3}) on rows of 16
            const u32x2 p02 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false), p13 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
            const u32x2 q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false), q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
            const uint32_t t[4] = {q01[0], q01[1], q23[0], q23[1]};       // t[g]: row nb + 4 g + r, m = mw + lane
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nr = nb + 4 * g + r;
                if (full || (mok && n0 + nr < N))
                    __builtin_amdgcn_raw_buffer_store_b32(t[g], rD, (int)(lane * 4), (int)((uint32_t)(nr * ldd + mw) * 4u), 0);
            }
        }
    }
}

template <int WMT, int WNT, int WGM, int WGN>
hipError_t launch_cfg_s(const ggml_hip_weight *w, const uint8_t *apan, int64_t N, int64_t Npad, float *dst, int64_t ldd, hipStream_t st) {
    using C = CfgS<WMT, WNT, WGM, WGN>;
    auto kern = dense16s_kernel<WMT, WNT, WGM, WGN>;
    static PerDeviceOnce once;
    const hipError_t attr = once.max_dynamic_lds((const void *)kern, C::TOTAL);
    if (attr != hipSuccess) return attr;
    if (w->Mpad % C::TM != 0 || Npad % C::TN != 0) return hipErrorInvalidValue;
    const int64_t Kpad = dense16_kpad(w->K);
    const int tiles_m = (int)((w->M + C::TM - 1) / C::TM), tiles_n = (int)((N + C::TN - 1) / C::TN);
    const uint64_t w_bytes = (uint64_t)(Kpad / 8 + DENSE16_SPARE_PANELS) * w->Mpad * 16, a_bytes = (uint64_t)(Kpad / 8) * Npad * 16;
    if (w_bytes > 0xFFFFFFFFull || a_bytes > 0xFFFFFFFFull || (uint64_t)C::TN * ldd * 4 > 0xFFFFFFFFull) return hipErrorNotSupported;
    kern<<<dim3((unsigned)(tiles_m * tiles_n)), C::NT, C::TOTAL, st>>>(w->p16, apan, dst, (int)w->M, (int)N, (int)w->Mpad, (int)Npad,
                                                                       (int)(Kpad / (16 * KS)), (int)ldd, tiles_m, tiles_n, (uint32_t)w_bytes,
                                                                       (uint32_t)a_bytes);
    return hipGetLastError();
}

template <int WMT, int WNT, int WGM, int WGN, int KSP = 1, int VS = 1>
hipError_t launch_cfg(const ggml_hip_weight *w, const uint8_t *apan, int64_t N, int64_t Npad, float *dst, int64_t ldd, hipStream_t st) {
    using C = Cfg<WMT, WNT, WGM, WGN>;
    auto kern = dense16_kernel<WMT, WNT, WGM, WGN, KSP, VS>;
        static PerDeviceOnce once;   // per kernel instantiation; the attribute is set once per device
    const hipError_t attr = once.max_dynamic_lds((const void *)kern, C::TOTAL * KSP);
    if (attr != hipSuccess) return attr;
    if (w->Mpad % C::TM != 0 || Npad % C::TN != 0) return hipErrorInvalidValue;
    const int64_t Kpad = dense16_kpad(w->K);
    const int tiles_m = (int)((w->M + C::TM - 1) / C::TM), tiles_n = (int)((N + C::TN - 1) / C::TN);
    const uint64_t w_bytes = (uint64_t)(Kpad / 8 + DENSE16_SPARE_PANELS) * w->Mpad * 16, a_bytes = (uint64_t)(Kpad / 8) * Npad * 16;
    if (w_bytes > 0xFFFFFFFFull || a_bytes > 0xFFFFFFFFull || (uint64_t)C::TN * ldd * 4 > 0xFFFFFFFFull) return hipErrorNotSupported;
    kern<<<dim3((unsigned)(tiles_m * tiles_n)), C::NT * KSP, C::TOTAL * KSP, st>>>(w->p16, apan, dst, (int)w->M, (int)N, (int)w->Mpad, (int)Npad,
                                                                     (int)(Kpad / (16 * KS)), (int)ldd, tiles_m, tiles_n, (uint32_t)w_bytes,
                                                                     (uint32_t)a_bytes);
    return hipGetLastError();
}


// ---- K10d: dense F32 x F32 on the bf16 matrix cores, every f32 operand split EXACTLY into three bf16 pieces -------------------------
// ggml_compute_forward_mul_mat_f32 (Ggml.cs:5969-6178; ggml_vec_dot_f32 2631-2640: f32 products, f64 running sum).  The f32 matrix
// instruction (v_mfma_f32_32x32x2_f32, dense.hip) runs at 1/16 of the bf16 rate.  a = a0 + a1 + a2 with bf16 pieces (truncation: the
// three pieces hold the 24 significand bits exactly), likewise b, and
//     a * b = a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0) + [a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|, dropped]
// Every kept product of two bf16 values is exact in f32; the six partial products go through six v_mfma_f32_16x16x32_bf16 per tile and
// k-step into ONE f32 accumulator.  Error against the exact product: 2^-23 relative -- that of rounding the product to f32, which the
// reference does -- and the sum is f32 as in dense.hip (the reference: f64): same class as the kernel it replaces, 2.2 x its speed.
// Images: panel-plane q = 3 * (k / 8) + piece, [Kpad/8 * 3][rows][16 B] for both operands (weights at upload, src1 by the INIT kernel).
// Workgroup 128 x 128, four waves of 64 x 64 (4 x 4 MFMA tiles); a stage is ONE k-step of 32 (4 panels x 3 pieces x 128 rows = 24 KB of
// LDS, double-buffered), 96 MFMAs per wave and stage; the weights of the next stage travel into a second register set meanwhile.
// (split3: common.h -- K3p's min-term product uses the same pieces)

// INIT: src1 f32 rows -> split image; a thread owns 8 consecutive k of one row = one 16-byte entry of each of the three planes
__global__ __launch_bounds__(256) void convert_act_split_kernel(const float *__restrict__ x, int64_t N, int64_t K, int64_t Kpad, int64_t ld1,
                                                               uint8_t *__restrict__ img, int64_t Npad) {
    const int64_t n = (int64_t)blockIdx.y * 32 + (threadIdx.x >> 3);
    const int64_t p = (int64_t)blockIdx.x * 8 + (threadIdx.x & 7);             // panel (8 k)
    if (n >= N || p * 8 >= Kpad) return;
    const float *row = x + n * ld1 + p * 8;
    float v[8];
    if (p * 8 + 7 < K) {
        const float4 a = *(const float4 *)row, b = *(const float4 *)(row + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p * 8 + e < K ? row[e] : 0.0f;
    }
    uint32_t w[3][4] = {};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        uint32_t q0, q1, q2;
        split3(v[e], q0, q1, q2);
        w[0][e >> 1] |= q0 << (16 * (e & 1)); w[1][e >> 1] |= q1 << (16 * (e & 1)); w[2][e >> 1] |= q2 << (16 * (e & 1));
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(uint4 *)(img + ((p * 3 + pl) * Npad + n) * 16) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

// weights: f32 rows -> split panels [Kpad/8 * 3 + spare][Mpad][16 B] (zero past K and past M: the buffer is pre-zeroed)
__global__ void f32_rows_to_split_panels_kernel(const float *__restrict__ rows, int64_t M, int64_t K, int64_t Mpad, uint8_t *__restrict__ pan) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = blockIdx.y;
    if (m >= M) return;
    uint32_t w[3][4] = {};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t k = p * 8 + e;
        uint32_t q0 = 0, q1 = 0, q2 = 0;
        if (k < K) split3(rows[m * K + k], q0, q1, q2);
        w[0][e >> 1] |= q0 << (16 * (e & 1)); w[1][e >> 1] |= q1 << (16 * (e & 1)); w[2][e >> 1] |= q2 << (16 * (e & 1));
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *(uint4 *)(pan + ((p * 3 + pl) * Mpad + m) * 16) = make_uint4(w[pl][0], w[pl][1], w[pl][2], w[pl][3]);
}

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__global__ __launch_bounds__(256, 2)
void dense32s_kernel(const uint8_t *__restrict__ wpan, const uint8_t *__restrict__ apan, float *__restrict__ dst, int M, int N, int Mpad,
                     int Npad, int nstages, int ldd, int tiles_m, int tiles_n, uint32_t w_bytes, uint32_t a_bytes) {
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    constexpr int WMT = 4, WNT = 4, TM = 128, TN = 128, NT = 256;
    constexpr int STAGE = 12 * TN * 16;                       // 4 panels x 3 pieces
    constexpr int ROUNDS = 12 * TN / NT;                      // DMA pieces per stage (6)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wn = wave >> 1, wm_ = wave & 1;

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    int tm_i, tn_i;
    if (nwg > 512 && !D16_FLATMAP) {                          // more than one round of the chip: 8 x 8 blocks of tiles per XCD
        tile_of_blocked(bid, tiles_m, tiles_n, tm_i, tn_i);
    } else if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0) {    // 2 x 4 blocks of the tile grid per XCD, as dense16_kernel
        const int hm = tiles_m >> 1, l = bid >> 3;
        tm_i = (xcd & 1) * hm + l % hm;
        tn_i = (xcd >> 1) * (tiles_n >> 2) + l / hm;
    } else {
        const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        tm_i = t_lin % tiles_m;
        tn_i = t_lin / tiles_m;
    }
    const int m0 = tm_i * TM, n0 = tn_i * TN;

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // activations: DMA, chunk c = tid + NT * i of a stage = [12 panel-planes][TN rows] x 16 B (two panel-planes per round)
    const uint32_t a_pan = (uint32_t)(Npad * 16);
    const uint32_t voffA = (uint32_t)(((tid / TN) * Npad + n0 + tid % TN) * 16);
    const rsrc_t rA = make_rsrc(apan, a_bytes);
    auto dma_piece = [&](int s, int buf, auto pc) {
        constexpr int i = decltype(pc)::value;
        uint8_t *sp = smem + (buf & 1) * STAGE;
        blds16(rA, sp + (size_t)(wave * 64 + NT * i) * 16, voffA, ((uint32_t)s * 12 + 2 * i) * a_pan);
    };
    // weights: registers, one stage ahead; lane (row l15, k-group g4) takes panel-plane 3 * (4 s + g4) + piece
    const uint32_t w_pan = (uint32_t)(Mpad * 16);
    const uint32_t voffW = (uint32_t)((3 * g4 * Mpad + m0 + wm_ * 64 + l15) * 16);
    const rsrc_t rW = make_rsrc(wpan, w_bytes);
    u32x4 bq[2][3][WMT];
    auto load_b = [&](int s, auto sc) {
        constexpr int slot = decltype(sc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < WMT; ++i)
                bq[slot][pl][i] = __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(voffW + (uint32_t)pl * w_pan + 256u * i), (int)((uint32_t)s * 12 * w_pan), 0);
    };

    // piece pairs (src1, weight), smallest products first
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    auto compute = [&](int s, auto sc) {
        constexpr int slot = decltype(sc)::value;
        const uint8_t *sA = smem + (s & 1) * STAGE + ((size_t)(3 * g4 * TN + wn * 64 + l15)) * 16;
        u32x4 af[2][3];
        auto fetch_af = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[j & 1][pl] = *(const u32x4 *)(sA + (pl * TN + 16 * j) * 16);
        };
        fetch_af(std::integral_constant<int, 0>{});
        // the next stage: its DMA pieces first (they must have landed at this stage's barrier), its weights behind them
        static_for<ROUNDS>([&](auto pc) { dma_piece(s + 1, s + 1, pc); });
        load_b(s + 1, std::integral_constant<int, 1 - slot>{});
        __builtin_amdgcn_sched_barrier(0);
        // The six products of an n-tile and k-step go into a FRESH accumulator t (the matrix pipe truncates when it aligns its partial sums
        // with the accumulator: harmless against a block's own small sum, a bias of up to 768 half-ulps against the running sum -- measured
        // 1.6e-5 of the rms at K = 4096 with everything in one accumulator); t joins the running sum by a rounded VALU add, issued behind the
        // MFMAs of the NEXT n-tile so that the pipe never waits for it.
        f32x4 t[2][WMT];
        static_for<WNT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 1 < WNT) fetch_af(std::integral_constant<int, j + 1>{});
            // (everything requested at the top of the stage is three n-tile steps = 72 MFMAs old here)
            if constexpr (j == WNT - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int i = 0; i < WMT; ++i)
                    t[j & 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[j & 1][PA[c]]), __builtin_bit_cast(bf16x8, bq[slot][PB[c]][i]),
                                                                          c == 0 ? (f32x4){0.0f, 0.0f, 0.0f, 0.0f} : t[j & 1][i], 0, 0, 0);
            if constexpr (j > 0) {
#pragma unroll
                for (int i = 0; i < WMT; ++i) acc[i][j - 1] += t[(j - 1) & 1][i];
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int i = 0; i < WMT; ++i) acc[i][WNT - 1] += t[(WNT - 1) & 1][i];
    };

    static_for<ROUNDS>([&](auto pc) { dma_piece(0, 0, pc); });
    load_b(0, std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int s = 0; s < nstages; s += 2) {                    // (nstages is even: K is padded to whole groups of 128)
        compute(s, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        compute(s + 1, std::integral_constant<int, 1>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // dst[n][m]: the store transpose of dense16s_kernel (one 256-byte row of dst per store)
    const rsrc_t rD = make_rsrc(dst + (size_t)n0 * ldd + m0, 0xFFFFFFFFu);
    const bool full = n0 + TN <= N && m0 + TM <= M;
    const int mw = wm_ * 64;
    const bool mok = full || m0 + mw + lane < M;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int nb = (wn * WNT + j) * 16;
        if (!full && n0 + nb >= N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint32_t v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float f = acc[i][j][r]; v[i] = __builtin_bit_cast(uint32_t, f); }
            const u32x2 p02 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false), p13 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
            const u32x2 q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false), q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
            const uint32_t t[4] = {q01[0], q01[1], q23[0], q23[1]};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nr = nb + 4 * g + r;
                if (full || (mok && n0 + nr < N))
                    __builtin_amdgcn_raw_buffer_store_b32(t[g], rD, (int)(lane * 4), (int)((uint32_t)(nr * ldd + mw) * 4u), 0);
            }
        }
    }
}

}  // namespace

hipError_t launch_f16_rows_to_panels(ggml_hip_weight *w, hipStream_t st) {
    if (!w->p16 || w->M <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 255) / 256), (unsigned)((w->K + 7) / 8));
    f16_rows_to_panels_kernel<<<grid, 256, 0, st>>>((const uint16_t *)w->dense, w->M, w->K, w->Mpad, w->p16);
    return hipGetLastError();
}

hipError_t launch_dense16_init(const float *x, int64_t N, int64_t K, int64_t ld1, void *work, hipStream_t st) {
    const int64_t Kpad = dense16_kpad(K), Npad = pad_act(N);
    dim3 grid((unsigned)((Kpad / 32 + CV_CH - 1) / CV_CH), (unsigned)((N + 31) / 32));
    convert_act_f16_kernel<<<grid, 256, 0, st>>>(x, N, K, Kpad, ld1, (uint8_t *)work, Npad);
    return hipGetLastError();
}

// The form was chosen by plan.cpp (plan_dense: the shape of the matrix instruction and the K split by N and K, the tile by the tile count):
//   * above 512 src1 rows v_mfma_f32_16x16x32_f16 (the chip holds a higher clock on it), 256 x 128 tiles from 384 tiles on, else 128 x 128;
//   * up to 128 rows K split four ways inside the workgroup, on 32-row tiles, or on 128-row tiles of 16 waves where those cover the chip;
//   * prompt-sized batches: two wave groups splitting K -- on 128 x 128 tiles from 160 such tiles on (4096 x 4096 x 512 44.1 -> 37.5 us), as a
//     VIRTUAL split on 4-wave workgroups for vocabulary-sized matrices (32000 x 4096 x 512: 229 us on the 8-wave form -> 203), else 128 x 64.
hipError_t launch_dense16(const ggml_hip_weight *w, const mm_plan &pl, const void *work, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    const int64_t Npad = pad_act(N);
    const uint8_t *a = (const uint8_t *)work;
    if (pl.family != MMF_DENSE16 || !w->p16) return hipErrorInvalidValue;
    switch (pl.form) {
    case D16F_S_256x128:  return launch_cfg_s<4, 8, 4, 1>(w, a, N, Npad, dst, ldd, st);
    case D16F_S_128x128:  return launch_cfg_s<4, 4, 2, 2>(w, a, N, Npad, dst, ldd, st);
    case D16F_256x128:    return launch_cfg<2, 4, 4, 1>(w, a, N, Npad, dst, ldd, st);
    case D16F_S4_H128:    return launch_cfg<1, 2, 4, 1, 4>(w, a, N, Npad, dst, ldd, st);
    case D16F_S4_H32:     return launch_cfg<1, 2, 1, 1, 4>(w, a, N, Npad, dst, ldd, st);
    case D16F_V2_128x128: return launch_cfg<2, 2, 2, 2, 1, 2>(w, a, N, Npad, dst, ldd, st);
    case D16F_S2_128x128: return launch_cfg<2, 2, 2, 2, 2>(w, a, N, Npad, dst, ldd, st);
    case D16F_S2_128x64:  return launch_cfg<1, 2, 4, 1, 2>(w, a, N, Npad, dst, ldd, st);
    case D16F_128x128:    return launch_cfg<2, 2, 2, 2>(w, a, N, Npad, dst, ldd, st);
    default: break;
    }
    return hipErrorInvalidValue;
}

// ---- K10d host side ----
hipError_t launch_f32_rows_to_split_panels(ggml_hip_weight *w, hipStream_t st) {
    if (!w->p32 || w->M <= 0) return hipSuccess;
    dim3 grid((unsigned)((w->M + 255) / 256), (unsigned)(dense16_kpad(w->K) / 8));
    f32_rows_to_split_panels_kernel<<<grid, 256, 0, st>>>((const float *)w->dense, w->M, w->K, w->Mpad, w->p32);
    return hipGetLastError();
}

// (whether this form serves a shape -- more than 256 src1 rows, by N alone -- is plan.cpp's decision: plan_dense.  A workgroup's K loop is one
// latency chain (K = 4096: ~165 us whatever the grid), so the form pays once the 128 x 128 tiles cover the chip: 4096 x 4096 x N, this kernel |
// dense.hip in us -- N = 128 167 | 120, 256 168 | 122, 512 171 | 173, 1024 186 | 287, 4096 623 | 1200.)
hipError_t launch_dense32_init(const float *x, int64_t N, int64_t K, int64_t ld1, void *work, hipStream_t st) {
    const int64_t Kpad = dense16_kpad(K), Npad = pad_act(N);
    dim3 grid((unsigned)((Kpad / 8 + 7) / 8), (unsigned)((N + 31) / 32));
    convert_act_split_kernel<<<grid, 256, 0, st>>>(x, N, K, Kpad, ld1, (uint8_t *)work, Npad);
    return hipGetLastError();
}

hipError_t launch_dense32(const ggml_hip_weight *w, const void *work, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    const int64_t Npad = pad_act(N), Kpad = dense16_kpad(w->K);
    constexpr int LDS = 2 * 12 * 128 * 16;
    static PerDeviceOnce once;
    const hipError_t attr = once.max_dynamic_lds((const void *)dense32s_kernel, LDS);
    if (attr != hipSuccess) return attr;
    if (w->Mpad % 128 != 0 || Npad % 128 != 0) return hipErrorInvalidValue;
    const int tiles_m = (int)((w->M + 127) / 128), tiles_n = (int)((N + 127) / 128);
    const uint64_t w_bytes = (uint64_t)(Kpad / 8 * 3 + DENSE32_SPARE_PANELS) * w->Mpad * 16, a_bytes = (uint64_t)(Kpad / 8 * 3) * Npad * 16;
    if (w_bytes > 0xFFFFFFFFull || a_bytes > 0xFFFFFFFFull || (uint64_t)128 * ldd * 4 > 0xFFFFFFFFull) return hipErrorNotSupported;
    dense32s_kernel<<<dim3((unsigned)(tiles_m * tiles_n)), 256, LDS, st>>>(w->p32, (const uint8_t *)work, dst, (int)w->M, (int)N, (int)w->Mpad, (int)Npad,
                                                                           (int)(Kpad / 32), (int)ldd, tiles_m, tiles_n, (uint32_t)w_bytes, (uint32_t)a_bytes);
    return hipGetLastError();
}
