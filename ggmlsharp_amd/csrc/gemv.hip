// gemv.hip -- K2: quantized mat-vec / skinny mat-mat (N <= GEMV_MAX_N), HBM-bandwidth bound.
//
// Computes the COMPUTE phase of ggml_compute_forward_mul_mat_q_f32 (Ggml.cs:6676-6698) for few src1 rows:
//   dst[n*ldd + m] = sum_b dw[m,b] * da[n,b] * sum_{k in b} w[m,k] * a[n,k]
// with the integer block sums exact (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1136-1159, _q5_0_q8_0 1270-1298,
// _q8_0_q8_0 1362-1378, _q4_1_q8_1 1176-1198, _q5_1_q8_1 1318-1344, _q4_2_q8_0 1216-1252 -- the last two with IEEE-half
// scales, SURVEY D7 intent) and the f32 scale-accumulate per block as in :1158.
// Only the order of the f32 additions over blocks differs from the scalar loop.
//
// Mapping, chosen for the planar weight layout ([k-block][row][16 B]):
//   workgroup = 16 weight rows x all K; lane = (row r = lane & 15, k-lane kq = lane >> 4); the 8 waves x 4 k-lanes
//   are 32 "k-workers", worker u takes blocks u, u+32, u+64, ...  A wave-wide 16-byte load therefore touches four
//   256-byte contiguous segments, every lane keeps 4 independent loads in flight, and a row's dot product needs
//   only two xor-shuffles plus one LDS pass to combine -- no atomics, fixed summation tree (deterministic).
//   The Q8 activation vector (written by K1) is staged into LDS once per 128 k-blocks and read as broadcasts.
#include "common.h"
#include <cstring>

namespace {

// GV_ROWS (template parameter) = weight rows per workgroup; 16 everywhere (see launch_typed)
#ifndef GV_MAX_WGS
#define GV_MAX_WGS 512              // persistent grid: 2 workgroups per CU (A/B at M = 32000: 512 -> 17.3 us, 1024 -> 18.5, 2048 -> 19.2)
#endif
#ifndef GV_WAVES
#define GV_WAVES 8                     // waves per workgroup (A/B: 4 -> 8 waves: 4096^2 batch-1 5.81 -> 5.40 us, 4096 x 11008 12.1 -> 10.6,
                                       // N = 8 14.8 -> 12.0, M = 32000 17.25 -> 17.1; 16 waves: 5.23 us at 4096^2 but 19.9 us at M = 32000 with
                                       // one resident workgroup per CU, and 10 % slower on Q8_0 -- and the choice must not depend on M)
#endif
#define GV_THREADS (64 * GV_WAVES)
#define GV_NKQ (64 / GV_ROWS)          // k-lanes per wave
#define GV_WORKERS (GV_WAVES * GV_NKQ) // k-workers per workgroup
#define GV_CHUNK 128  // k-blocks of activations staged in LDS at a time (N <= 8)
// wider batches (9..16 src1 rows, two-step form only) stage fewer k-blocks at a time so that the image still fits 64 KB of LDS
// r4: a pseudo-type for weights living in the planar Q4_2 form ON INT8 PLANES (the Q6_K extension, kquants.hip): Q8_0's operand loads (two planes of
// int8 per k-block, no unpacking) with Q4_2's arithmetic (two 16-element sub-blocks per k-block, the second's scale in the m plane)
constexpr int GV_TYPE_I8X2 = 100;
template <int TYPE> constexpr bool GV_W_I8 = TYPE == GGML_TYPE_Q8_0 || TYPE == GV_TYPE_I8X2;
template <int TYPE> constexpr bool GV_TWO_SC = TYPE == GGML_TYPE_Q4_2 || TYPE == GV_TYPE_I8X2;

template <int NC> struct GvChunk { static constexpr int value = NC <= 8 ? GV_CHUNK : 64; };
#ifndef GV_NT
#define GV_NT 1       // non-temporal weight loads (A/B: -DGV_NT=0)
#endif
#ifndef GV_PF_WGS
#define GV_PF_WGS 256                  // the look-ahead form (N = 1, K <= 4096) keeps two row tiles per wave in flight: ONE workgroup per CU
                                       // (A/B 512 -> 256: M = 11008 8.0 -> 7.6 us, M = 32000 16.1 -> 15.7, M = 65536 28.9 -> 28.5; the forms
                                       // without look-ahead need the second workgroup: 32000 x 4096 x 8 37.1 -> 41.8 us with 256)
#endif
#ifndef GV_OLD_FUSED
#define GV_OLD_FUSED 0                 // A/B: the former fused kernel (block-wide staging behind a barrier)
#endif

using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;
__device__ __forceinline__ uint4 ld_w(const uint8_t *p) {
#if GV_NT
    const u32x4v v = __builtin_nontemporal_load((const u32x4v *)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *(const uint4 *)p;
#endif
}

// A result store the compiler's wait-count bookkeeping does not see.  On gfx9 loads and stores share `vmcnt` but only
// operations of one kind return in order, so with a store possibly in flight hipcc turns every later "wait for this load"
// into "wait for everything" -- which drained the look-ahead weight stream once per row tile.  Hidden in asm, the store
// still occupies a counter slot, which only makes the counted waits of the loads conservative (never short).
__device__ __forceinline__ void st_result(float *p, float v) {
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
}

// bit e of qh for e = 8i + {0,2,4,6} (sel = 0) or 8i + {1,3,5,7} (sel = 1), moved to bit 4 of bytes 0..3
__device__ __forceinline__ uint32_t q5_high_bits(uint32_t qh, int i, int sel) {
    const uint32_t t = ((qh >> (8 * i + sel)) & 0x55u);
    return (t * 0x00410410u) & 0x10101010u;
}

// FUSED = true : src1 is f32; every workgroup quantizes the activation chunk itself (the INIT phase of
//                 Ggml.cs:6641-6654 / quantize_row_q8_0 733-762, same arithmetic as K1, bit-exact) straight into LDS,
//                 so the whole mul_mat at small N is ONE launch and one dependent memory round trip.
// FUSED = false: src1 was quantized earlier (K1 planes, or reference Q8 blocks through ggml_hip_vec_dot).
// The weight loads of a chunk are issued before the activations are staged, so both latencies overlap.
template <int TYPE, int NC, bool FUSED, int GV_ROWS>
__global__ __launch_bounds__(GV_THREADS) void gemv_q_kernel(const uint8_t *__restrict__ qs, const uint32_t *__restrict__ gs,
                                                    const float *__restrict__ x, int64_t ld1,
                                                    const int8_t *__restrict__ a8, const float *__restrict__ ad,
                                                    const int32_t *__restrict__ as, float *__restrict__ dst, int64_t M,
                                                    int64_t Mpad, int64_t Npad, int64_t nbk, int64_t ldd, int N, int ntiles) {
    constexpr int CH = GvChunk<NC>::value;
    static_assert(GV_ROWS == 16, "the side image (ggml_hip_weight::gs) is laid out in 16-row tiles");
    static_assert(!FUSED || NC <= 8, "the fused form quantizes N columns per workgroup: N <= 8 only");
    __shared__ uint4 sA[CH * 2 * NC];
    __shared__ float sD[CH * NC];
    __shared__ int sS[CH * NC];
    __shared__ float sRed[GV_WAVES][NC][GV_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane % GV_ROWS, kq = lane / GV_ROWS, u = wave * GV_NKQ + kq;
    constexpr int BPL = GvChunk<NC>::value / GV_WORKERS;     // k-blocks per lane per chunk
    // Persistent over row tiles: virtual block vb = blockIdx.x, + gridDim.x, ... takes tile f(vb).  When all of K fits one
    // LDS chunk the activations are quantized / staged ONCE per workgroup instead of once per 16 rows (at M = 32000 the
    // 2000 redundant quantizations were ~8 % of the kernel).  Per-row arithmetic and summation tree are unchanged.
    // (Also loading the next tile's weights a tile ahead measured 10 % SLOWER: 19.3 vs 17.3 us at M = 32000.)
    const bool single_chunk = nbk <= GvChunk<NC>::value;
    bool staged = false;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    const int tile = vb;   // launch order (see the fused kernel below)
    const int64_t row = (int64_t)tile * GV_ROWS + r;  // < Mpad by construction

    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0f;

    for (int64_t cb = 0; cb < nbk; cb += CH) {
        const int nbc = (int)((nbk - cb) < CH ? (nbk - cb) : CH);

        // 1. all of this lane's weight loads for the chunk go out first (4 x 16 B + scales in flight per lane, 512 lanes)
        uint4 q[BPL], q2[GV_W_I8<TYPE> ? BPL : 1];
        constexpr bool HAS_M = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1 || GV_TWO_SC<TYPE>;   // Q4_2: second scale
        constexpr bool HAS_H = TYPE == GGML_TYPE_Q5_0 || TYPE == GGML_TYPE_Q5_1;
        constexpr int NP = 1 + (HAS_M ? 1 : 0) + (HAS_H ? 1 : 0);
        float dw[BPL], mw[HAS_M ? BPL : 1];
        uint32_t hb[HAS_H ? BPL : 1];
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int bl = u + GV_WORKERS * j;
            const bool ok = bl < nbc;
            const int64_t b = cb + (ok ? bl : 0);
            // once-read weight stream: non-temporal loads (MI355X_MICROARCH.md nt-weights: issued -> landed -18 %)
            if (GV_W_I8<TYPE>) {
                q[j] = ld_w(qs + ((b * 2 + 0) * Mpad + row) * 16);
                q2[j] = ld_w(qs + ((b * 2 + 1) * Mpad + row) * 16);
            } else {
                q[j] = ld_w(qs + (b * Mpad + row) * 16);
            }
            // scales / mins / fifth bits from the tile-major side image (common.h ggml_hip_weight::gs)
            const uint32_t *g = gs + (((int64_t)tile * nbk + b) * NP) * 16 + r;
            dw[j] = ok ? __uint_as_float(g[0]) : 0.0f;   // dw = 0 kills the contribution of a block past the end
            if (HAS_M) mw[j] = ok ? __uint_as_float(g[16]) : 0.0f;
            if (HAS_H) hb[j] = g[16 * (NP - 1)];
        }

        // 2. activations of the chunk -> LDS (int8 even/odd planes + scale + block sum); once per workgroup if K fits
        if (!(single_chunk && staged)) {
        __syncthreads();  // previous chunk fully consumed
        if constexpr (FUSED) {
            const int t = tid & 7, grp = tid >> 3;             // 8 lanes per 32-element block, GV_THREADS / 8 groups
            constexpr int NGRP = GV_THREADS / 8;
            constexpr int ITEMS = GV_CHUNK * NC / NGRP;       // (column, k-block) items per group
            constexpr int UNR = ITEMS < 4 ? ITEMS : 4;          // loads in flight per lane
#pragma unroll 1
            for (int it0 = 0; it0 < ITEMS; it0 += UNR) {
                float4 v[UNR];
#pragma unroll
                for (int k = 0; k < UNR; ++k) {
                    const int w0 = grp + NGRP * (it0 + k);
                    const int c = w0 / GV_CHUNK, bl = w0 % GV_CHUNK;
                    const int cc = c < N ? c : N - 1;
                    const int blc = bl < nbc ? bl : nbc - 1;
                    v[k] = *(const float4 *)(x + (int64_t)cc * ld1 + (cb + blc) * QK + 4 * t);
                }
#pragma unroll
                for (int k = 0; k < UNR; ++k) {
                    const int w0 = grp + NGRP * (it0 + k);
                    const int c = w0 / GV_CHUNK, bl = w0 % GV_CHUNK;
                    const bool live = c < N && bl < nbc;       // uniform over the 8 lanes of the group
                    float amax = fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w)));
                    amax = group8_max(amax);
                    const float d = amax / 127.0f;                  // Ggml.cs:751
                    const float id = d != 0.0f ? 1.0f / d : 0.0f;   // Ggml.cs:752
                    const int q0 = (int)rintf(v[k].x * id), q1 = (int)rintf(v[k].y * id);   // Ggml.cs:758-759 (D1, D2)
                    const int q2_ = (int)rintf(v[k].z * id), q3 = (int)rintf(v[k].w * id);
                    const int sum = group8_sum(q0 + q1 + q2_ + q3);
                    const uint32_t e16 = ((uint32_t)q0 & 0xFFu) | (((uint32_t)q2_ & 0xFFu) << 8);
                    const uint32_t o16 = ((uint32_t)q1 & 0xFFu) | (((uint32_t)q3 & 0xFFu) << 8);
                    const bool even_lane = (t & 1) == 0;
                    const uint32_t recv = (uint32_t)dpp_i<DPP_XOR1>((int)(even_lane ? o16 : e16));
                    const uint32_t word = even_lane ? (e16 | (recv << 16)) : (recv | (o16 << 16));
                    if (live) {
                        uint8_t *base = (uint8_t *)sA;
                        const int h = even_lane ? 0 : 1, off = even_lane ? 2 * t : 2 * t - 2;
                        *(uint32_t *)(base + ((bl * 2 + h) * NC + c) * 16 + off) = word;
                        if (t == 0) { sD[bl * NC + c] = d; sS[bl * NC + c] = sum; }
                    }
                }
            }
        } else {
            for (int i = tid; i < nbc * 2 * NC; i += GV_THREADS) {
                const int c = i % NC, bh = i / NC;  // bh = b_local*2 + h
                const int cc = c < N ? c : N - 1;
                sA[i] = *(const uint4 *)(a8 + (((cb * 2 + bh) * Npad) + cc) * 16);
            }
            for (int i = tid; i < nbc * NC; i += GV_THREADS) {
                const int c = i % NC, bl = i / NC;
                const int cc = c < N ? c : N - 1;
                sD[i] = ad[(cb + bl) * Npad + cc];
                sS[i] = as[(cb + bl) * Npad + cc];
            }
        }
        __syncthreads();
        staged = true;
        }

        // 3. integer block dots + f32 scale-accumulate (Ggml.cs:1136-1159)
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int bl = (u + GV_WORKERS * j) < nbc ? (u + GV_WORKERS * j) : 0;
            const uint32_t qq[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
            uint32_t lo[4], hi[4];
            if (GV_W_I8<TYPE>) {
                lo[0] = q[j].x; lo[1] = q[j].y; lo[2] = q[j].z; lo[3] = q[j].w;
                hi[0] = q2[j].x; hi[1] = q2[j].y; hi[2] = q2[j].z; hi[3] = q2[j].w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = qq[i] & 0x0F0F0F0Fu;          // elements 8i+0,2,4,6  (Ggml.cs:1149)
                    hi[i] = (qq[i] >> 4) & 0x0F0F0F0Fu;   // elements 8i+1,3,5,7  (Ggml.cs:1150)
                    if (HAS_H) {                          // Ggml.cs:1285-1289 / 1330-1334
                        lo[i] |= q5_high_bits(hb[j], i, 0);
                        hi[i] |= q5_high_bits(hb[j], i, 1);
                    }
                    if (TYPE == GGML_TYPE_Q4_2) {         // (nib - 8) bytewise: the two half-block sums need their own offsets
                        lo[i] = ((lo[i] | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
                        hi[i] = ((hi[i] | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const uint4 a0 = sA[(bl * 2 + 0) * NC + c];
                const uint4 a1 = sA[(bl * 2 + 1) * NC + c];
                const float da = sD[bl * NC + c];
                const int sa = sS[bl * NC + c];
                if (GV_TWO_SC<TYPE>) {
                    // elements 0..15 (the first 16-element block, scale dw) are words 0, 1 of both planes, 16..31 (scale
                    // mw) words 2, 3: sumf += (d0 * yd) * sumi_0; sumf += (d1 * yd) * sumi_1 (Ggml.cs:1249-1250)
                    int s0 = 0, s1 = 0;
                    s0 = dot4(lo[0], a0.x, s0); s0 = dot4(lo[1], a0.y, s0); s0 = dot4(hi[0], a1.x, s0); s0 = dot4(hi[1], a1.y, s0);
                    s1 = dot4(lo[2], a0.z, s1); s1 = dot4(lo[3], a0.w, s1); s1 = dot4(hi[2], a1.z, s1); s1 = dot4(hi[3], a1.w, s1);
                    acc[c] = fmaf(dw[j] * da, (float)s0, acc[c]);
                    acc[c] = fmaf(mw[j] * da, (float)s1, acc[c]);
                    continue;
                }
                int s = 0;
                s = dot4(lo[0], a0.x, s); s = dot4(lo[1], a0.y, s); s = dot4(lo[2], a0.z, s); s = dot4(lo[3], a0.w, s);
                s = dot4(hi[0], a1.x, s); s = dot4(hi[1], a1.y, s); s = dot4(hi[2], a1.z, s); s = dot4(hi[3], a1.w, s);
                if (TYPE == GGML_TYPE_Q4_0) s -= 8 * sa;   // (nib - 8) * a summed = nib*a summed - 8 * sum(a)
                if (TYPE == GGML_TYPE_Q5_0) s -= 16 * sa;
                acc[c] = fmaf(dw[j] * da, (float)s, acc[c]);
                if (TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1) acc[c] = fmaf(mw[j], da * (float)sa, acc[c]);   // + m * (s0 + s1)
            }
        }
    }

#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float v = acc[c];
#pragma unroll
        for (int sft = GV_ROWS; sft < 64; sft <<= 1) v += __shfl_xor(v, sft);
        if (kq == 0) sRed[wave][c][r] = v;
    }
    __syncthreads();
    for (int o = tid; o < GV_ROWS * NC; o += GV_THREADS) {
        const int c = o / GV_ROWS, rr = o % GV_ROWS;
        const int64_t m = (int64_t)tile * GV_ROWS + rr;
        float quad[GV_WAVES / 4];                                                                   // fixed tree over the waves
#pragma unroll
        for (int q = 0; q < GV_WAVES / 4; ++q)
            quad[q] = (sRed[4 * q][c][rr] + sRed[4 * q + 1][c][rr]) + (sRed[4 * q + 2][c][rr] + sRed[4 * q + 3][c][rr]);
        const float tot = GV_WAVES == 4 ? quad[0] : GV_WAVES == 8 ? quad[0] + quad[1 % (GV_WAVES / 4)]
                                        : (quad[0] + quad[1 % (GV_WAVES / 4)]) + (quad[2 % (GV_WAVES / 4)] + quad[3 % (GV_WAVES / 4)]);
        if (m < M && c < N) dst[(int64_t)c * ldd + m] = tot;
    }
    __syncthreads();   // sRed is rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K2f: the FUSED form (src1 is f32, N <= 8), restated around the latency chain instead of the barrier.
//
// What bounded the kernel above at small M was not bandwidth but its dependent chain: weight loads, THEN the activation
// loads (vector-memory results return in issue order, so the x rows queued behind 40 KB of weights per workgroup), quantize,
// __syncthreads, dots, reduce (PMC round 1: 64-77 % of wave-cycles in s_waitcnt / s_barrier).  Here
//   * every WAVE quantizes exactly the k-blocks it consumes (its 4 k-lanes x 4 blocks of a 128-block chunk) into an LDS slice
//     of its own: no workgroup barrier between the INIT arithmetic and the dots, waves drift apart freely;
//   * the activation loads are issued FIRST, the weight stream right behind them: quantizing runs while the weights are in
//     flight (the compiler's vmcnt for the x registers leaves the younger weight loads outstanding);
//   * one barrier per row tile remains (the cross-wave reduction), on a double-buffered sRed.
// Arithmetic, block -> lane assignment and summation tree are those of the kernel above, so the bits are too.
// SC: all of K fits one chunk of 128 k-blocks (K <= 4096), a launch-time fact -- two instantiations instead of one kernel
// that carries both schedules in its registers (the combined form spilled: 24 B of scratch at 128 registers)
// PRO: the rms_norm -> mul prologue (common.h mm_prologue): the quantized row is (x * rms_scale) * g, computed here.
// MULTI: several weight matrices behind one activation matrix (common.h mv_set) -- its own kernels below: with the per-item
// selects in every kernel the single-matrix call lost 5-12 % (4096 x 4096: 4.6 -> 5.2 us)
// K8 (r4): the activations are quantized by the Q8_K rule of the k-quant extension (kquants.hip; quantize.hip K1 with K8 = true: one scale per 256
// elements, iscale = -128 / the first element of largest magnitude, q = min(127, nearest(iscale x))): a wave consumes HALF a super-block at four
// places of a chunk, so the scales of all (column, super-block) pairs are computed once per workgroup into LDS ahead of the items -- behind the
// first item's loads, like the rms_norm prologue's row scale -- and the wave-private quantization reads them.  K <= 32768 (128 super-blocks).
constexpr int GV_K8_SB = 128;
template <int TYPE, int NC, int GV_ROWS, bool SC, bool PRO, bool MULTI, bool K8 = false>
__device__ __forceinline__ void gemv_fused_body(const mv_set &ws, const float *__restrict__ x, int64_t ld1, int64_t nbk, int N, int ntiles,
                                                const mm_epilogue &ep, const mm_prologue &pro) {
    static_assert(GV_ROWS == 16 && GV_NKQ == 4, "lane = (row, k-lane) with 4 k-lanes per wave");
    constexpr int CH = GV_CHUNK;                       // k-blocks per chunk, all waves together
    constexpr int BPL = CH / GV_WORKERS;               // k-blocks per lane per chunk (4)
    constexpr int WBLK = GV_NKQ * BPL;                 // k-blocks per WAVE per chunk (16): local id i = kq + 4 * j
    // wave-private slice: per local block a slot of NC x {16 B even plane, 16 B odd plane} (+16 B so that the four k-lanes of
    // a wave, NC * 32 B apart, never start on the same bank), then NC scales and NC block sums
    constexpr int QSLOT = NC * 32 + 16;
    constexpr int WSLICE = WBLK * QSLOT + WBLK * NC * 8;
    __shared__ __attribute__((aligned(16))) uint8_t sAct[GV_WAVES * WSLICE];
    __shared__ float sRed[2][GV_WAVES][NC][GV_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane % GV_ROWS, kq = lane / GV_ROWS, u = wave * GV_NKQ + kq;
    uint8_t *const myq = sAct + wave * WSLICE;
    float *const myd = (float *)(myq + WBLK * QSLOT);
    int *const mys = (int *)(myd + WBLK * NC);
    constexpr bool single_chunk = SC;
    bool staged = false;
    int parity = 0;
    __shared__ float sScale[NC];
    __shared__ float sK8[K8 ? NC * GV_K8_SB * 2 : 1];      // (column, super-block) -> iscale, d
    static_assert(!(K8 && (PRO || MULTI)), "the Q8_K rule: the plain single-matrix call only");
    constexpr bool HAS_M = TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1 || GV_TWO_SC<TYPE>;   // Q4_2: second scale
    constexpr bool HAS_H = TYPE == GGML_TYPE_Q5_0 || TYPE == GGML_TYPE_Q5_1;
    constexpr int NP = 1 + (HAS_M ? 1 : 0) + (HAS_H ? 1 : 0);

    // The work of a workgroup is a sequence of ITEMS (row tile, chunk of 128 k-blocks), tiles taken round-robin over the
    // persistent grid.  The weight registers are double-buffered across items: item w+1's weights are requested BEFORE item
    // w is consumed, so the memory pipe never drains between tiles (a tile's dots, reduction and barrier used to sit between
    // two bursts of loads: ~10 % of a round at M = 32000).
    const int nchunks = (int)((nbk + CH - 1) / CH);
    const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nitems = my_tiles * nchunks;
    auto tile_of = [&](int w) {
        // row tiles in launch order (neighbouring workgroups stream neighbouring 256-byte pieces of every k-block row; the
        // per-XCD ranges this used to hand out existed for the [k-block][row] scale plane, whose 128-byte lines two tiles
        // shared -- with the tile-major side image they cost 3-6 % at M >= 11008)
        return (int)blockIdx.x + (w / nchunks) * (int)gridDim.x;
    };
    // the matrix a global row tile belongs to (common.h mv_set; one matrix: the selects fold to its fields) and the tile in it
    struct Sel { const uint8_t *qs; const uint32_t *gs; float *dst; int64_t M, Mpad, ldd; int tile; };
    auto select = [&](int gt) {
        Sel o;
        if constexpr (!MULTI) {
            o.qs = ws.qs[0]; o.gs = ws.gs[0]; o.dst = ws.dst[0]; o.M = ws.M[0]; o.Mpad = ws.Mpad[0]; o.ldd = ws.ldd[0]; o.tile = gt;
            return o;
        }
        const int k = (gt >= ws.tile_end[0]) + (gt >= ws.tile_end[1]) + (gt >= ws.tile_end[2]);
        o.qs = k == 0 ? ws.qs[0] : k == 1 ? ws.qs[1] : k == 2 ? ws.qs[2] : ws.qs[3];
        o.gs = k == 0 ? ws.gs[0] : k == 1 ? ws.gs[1] : k == 2 ? ws.gs[2] : ws.gs[3];
        o.dst = k == 0 ? ws.dst[0] : k == 1 ? ws.dst[1] : k == 2 ? ws.dst[2] : ws.dst[3];
        o.M = k == 0 ? ws.M[0] : k == 1 ? ws.M[1] : k == 2 ? ws.M[2] : ws.M[3];
        o.Mpad = k == 0 ? ws.Mpad[0] : k == 1 ? ws.Mpad[1] : k == 2 ? ws.Mpad[2] : ws.Mpad[3];
        o.ldd = k == 0 ? ws.ldd[0] : k == 1 ? ws.ldd[1] : k == 2 ? ws.ldd[2] : ws.ldd[3];
        o.tile = gt - (k == 0 ? 0 : k == 1 ? ws.tile_end[0] : k == 2 ? ws.tile_end[1] : ws.tile_end[2]);
        return o;
    };
    // (wider batches keep one register set: their activation registers already fill the budget, and the second set cost
    // them a resident workgroup -- 32000 x 4096 x 8: 50.5 us with it against 41.9 without)
    // (the look-ahead form: one column, and three / four -- 11008 x 4096 x 4 15.6 -> 12.6 us, 32000 x 4096 x 4 29.6 -> 23.3 with
    // one workgroup per CU; two columns measured level to worse: 11008 x 4096 x 2 8.3 -> 9.1 us)
    constexpr bool PF = SC && (NC == 1 || NC == 4);
    uint4 q[BPL], q2[GV_W_I8<TYPE> ? BPL : 1], qn[PF ? BPL : 1], q2n[PF && GV_W_I8<TYPE> ? BPL : 1];
    float dw[BPL], mw[HAS_M ? BPL : 1], dwn[PF ? BPL : 1], mwn[PF && HAS_M ? BPL : 1];
    uint32_t hb[HAS_H ? BPL : 1], hbn[PF && HAS_H ? BPL : 1];
    // the weight stream of one item (4 x 16 B + scales in flight per lane)
    auto load_item = [&](int w, uint4 *Q, uint4 *Q2, float *DW, float *MW, uint32_t *HB) {
        const Sel sel = select(tile_of(w));
        const uint8_t *const qs = sel.qs;
        const uint32_t *const gs = sel.gs;
        const int64_t Mpad = sel.Mpad;
        const int tl = sel.tile;
        const int64_t row = (int64_t)tl * GV_ROWS + r;  // < Mpad by construction
        const int64_t cb = (int64_t)(w % nchunks) * CH;
        const int nbc = (int)((nbk - cb) < CH ? (nbk - cb) : CH);
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int bl = u + GV_WORKERS * j;
            const bool ok = bl < nbc;
            const int64_t b = cb + (ok ? bl : 0);
            if (GV_W_I8<TYPE>) {
                Q[j] = ld_w(qs + ((b * 2 + 0) * Mpad + row) * 16);
                Q2[j] = ld_w(qs + ((b * 2 + 1) * Mpad + row) * 16);
            } else {
                Q[j] = ld_w(qs + (b * Mpad + row) * 16);
            }
            // (unconditional loads from a clamped block: no branch and no use of a result inside the load sequence -- the item's
            // consumer zeroes the scales of a block past the end of K)
            // scales / mins / fifth bits from the tile-major side image (common.h ggml_hip_weight::gs): the four k-lanes of a
            // wave read four adjacent 64-byte pieces
            const uint32_t *g = gs + (((int64_t)tl * nbk + b) * NP) * 16 + r;
            DW[j] = __uint_as_float(g[0]);
            if (HAS_M) MW[j] = __uint_as_float(g[16]);
            if (HAS_H) HB[j] = g[16 * (NP - 1)];
        }
    };

    float acc[NC];
    constexpr int ITEMS = WBLK * NC / 8;           // activation passes of the wave (8 groups of 8 lanes per pass)
    const int t = lane & 7, grp = lane >> 3;
    // prologue: the first item's activations go out FIRST, its weights right behind them
    // (at most XB passes are held in registers at a time: all of them for N <= 2; wider batches fetch the later passes
    // while they quantize the earlier ones)
    constexpr int XB = NC >= 8 ? 2 : (ITEMS < 4 ? ITEMS : 4);
    float4 v[XB], vg[PRO ? XB : 1];
    auto load_x = [&](int w, int p0) {
        const int64_t cb = (int64_t)(w % nchunks) * CH;
        const int nbc = (int)((nbk - cb) < CH ? (nbk - cb) : CH);
#pragma unroll
        for (int pp = 0; pp < XB; ++pp) {
            const int p = p0 + pp;
            const int it = grp + 8 * p, c = it / WBLK, i = it % WBLK;
            const int bl = wave * GV_NKQ + (i & 3) + GV_WORKERS * (i >> 2);        // block of the chunk
            const int cc = c < N ? c : N - 1, blc = bl < nbc ? bl : nbc - 1;
            v[pp] = *(const float4 *)(x + (int64_t)cc * ld1 + (cb + blc) * QK + 4 * t);
            if constexpr (PRO) vg[pp] = *(const float4 *)(pro.g + (int64_t)cc * pro.ld_g + (cb + blc) * QK + 4 * t);
        }
    };
    if constexpr (PRO) {
        // rms_norm's row scale (Ggml.cs:5889-5915): wave c computes row c's, in the element order and f64 tree every kernel
        // shares (common.h rms_row_scale).  The other waves put their first item's loads in flight BEFORE they wait for it,
        // the computing waves right after: the weight stream starts at launch, not behind the norm.
        if (wave < N) {
            const float sc = rms_row_scale(x + (int64_t)wave * ld1, nbk * QK, lane);
            if (lane == 0) sScale[wave] = sc;
        }
        if (nitems > 0) { load_x(0, 0); load_item(0, q, q2, dw, mw, hb); }
        __syncthreads();
    } else {
        if constexpr (!K8) {
            if (nitems > 0) { load_x(0, 0); load_item(0, q, q2, dw, mw, hb); }
        } else {
            const int nsb = (int)(nbk >> 3);                // (K % 256 == 0 for the k-quants)
            // K1's own arrangement (quantize.hip, K8 = true): EIGHT LANES per super-block, lane t holding elements 32 j + 4 t .. + 3 of its eight
            // k-blocks, joined by three DPP steps -- 64 super-blocks per trip of the workgroup (a wave per super-block and six rounds of
            // ds_bpermute shuffles measured 4096 x 4096 x 1 / x 4 6.5 / 12.9 us against the reference types' 4.0 / 6.8).  The first trip's loads
            // go out BEHIND the first item's activations and IN FRONT of its weights (vector-memory results return in issue order).
            const int g8 = tid >> 3, t8 = tid & 7;
            float4 e8[8];
            auto k8_load = [&](int base) {
                const int item = base + g8;
                const int it2 = item < N * nsb ? item : 0;
                const float *src = x + (int64_t)(it2 / nsb) * ld1 + (int64_t)(it2 % nsb) * 256 + 4 * t8;
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) e8[jb] = *(const float4 *)(src + 32 * jb);
            };
            auto k8_reduce = [&](int base) {
                const int item = base + g8;
                float am = 0.0f, mx = 0.0f;
                int ix = 0;                                 // this lane's first element of largest magnitude (element order)
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) {
                    const float e[4] = {e8[jb].x, e8[jb].y, e8[jb].z, e8[jb].w};
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        if (fabsf(e[q4]) > am) { am = fabsf(e[q4]); mx = e[q4]; ix = 32 * jb + 4 * t8 + q4; }
                }
                auto join = [&](float oa, float om, int oi) {   // the larger magnitude wins, equal magnitudes the earlier element
                    const bool take = oa > am || (oa == am && oi < ix);
                    am = take ? oa : am; mx = take ? om : mx; ix = take ? oi : ix;
                };
                join(dpp_f<DPP_XOR1>(am), dpp_f<DPP_XOR1>(mx), dpp_i<DPP_XOR1>(ix));
                join(dpp_f<DPP_XOR2>(am), dpp_f<DPP_XOR2>(mx), dpp_i<DPP_XOR2>(ix));
                join(dpp_f<DPP_HALF_MIRROR>(am), dpp_f<DPP_HALF_MIRROR>(mx), dpp_i<DPP_HALF_MIRROR>(ix));
                const float isc = am != 0.0f ? -128.0f / mx : 0.0f;
                if (t8 == 0 && item < N * nsb) {
                    const int c = item / nsb, sb = item % nsb;
                    sK8[(c * GV_K8_SB + sb) * 2] = isc; sK8[(c * GV_K8_SB + sb) * 2 + 1] = am != 0.0f ? 1.0f / isc : 0.0f;
                }
            };
            if (nitems > 0) load_x(0, 0);
            k8_load(0);
            if (nitems > 0) load_item(0, q, q2, dw, mw, hb);
            k8_reduce(0);
            for (int base = GV_THREADS / 8; base < N * nsb; base += GV_THREADS / 8) { k8_load(base); k8_reduce(base); }
            __syncthreads();
        }
    }

    // one item: INIT arithmetic for this wave's blocks (first item of a single-chunk K, every item otherwise), block dots on
    // the weight registers handed in, and -- on a row tile's last chunk -- the reduction and the store
    auto do_item = [&](int w, const uint4 *Q, const uint4 *Q2, const float *DW, const float *MW, const uint32_t *HB) {
        const int cidx = w % nchunks;
        const int64_t cb = (int64_t)cidx * CH;
        const int nbc = (int)((nbk - cb) < CH ? (nbk - cb) : CH);
        const Sel sel = select(tile_of(w));
        const int tile = sel.tile;
        float *const dst = sel.dst;
        const int64_t M = sel.M, ldd = sel.ldd;
        if (cidx == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.0f;
        }
        const bool stage_now = !(single_chunk && staged);
        // 3. INIT phase for this wave's blocks (Ggml.cs:6641-6654 / quantize_row_q8_0 733-762, the arithmetic of K1)
        if (stage_now) {
            if (!single_chunk) __builtin_amdgcn_wave_barrier();   // (the previous chunk's reads of the slice are done: same wave, in order)
#pragma unroll
            for (int p = 0; p < ITEMS; ++p) {
                if (p > 0 && p % XB == 0) load_x(w, p);
                float4 vp = v[p % XB];
                const int it = grp + 8 * p, c = it / WBLK, i = it % WBLK;
                const int bl = wave * GV_NKQ + (i & 3) + GV_WORKERS * (i >> 2);
                const bool live = bl < nbc;                        // uniform over the 8 lanes of the group
                if constexpr (PRO) {
                    const int cc = c < N ? c : N - 1;
                    const float sc = sScale[cc];
                    const float4 gg = vg[p % XB];
                    const float4 nn = make_float4(vp.x * sc, vp.y * sc, vp.z * sc, vp.w * sc);          // the rms_norm node
                    vp = make_float4(nn.x * gg.x, nn.y * gg.y, nn.z * gg.z, nn.w * gg.w);              // the mul node
                    if (blockIdx.x == 0 && live && c < N) {       // both nodes' data, written once (every workgroup computes the same)
                        const int64_t e = (int64_t)c * (nbk * QK) + (cb + bl) * QK + 4 * t;
                        *(float4 *)(pro.n_out + e) = nn;
                        *(float4 *)(pro.y_out + e) = vp;
                    }
                }
                float d, id;
                if constexpr (K8) {                              // the super-block's scale, computed ahead of the items
                    const int cc = c < N ? c : N - 1;
                    const int sbi = (int)((cb + (live ? bl : 0)) >> 3);
                    id = sK8[(cc * GV_K8_SB + sbi) * 2]; d = sK8[(cc * GV_K8_SB + sbi) * 2 + 1];
                    vp = make_float4(fminf(127.0f, rintf(vp.x * id)), fminf(127.0f, rintf(vp.y * id)), fminf(127.0f, rintf(vp.z * id)), fminf(127.0f, rintf(vp.w * id)));
                    id = 1.0f;                                   // (vp holds the quants: x * 1 and the nearest integer of an integer are exact)
                } else {
                    float amax = fmaxf(fmaxf(fabsf(vp.x), fabsf(vp.y)), fmaxf(fabsf(vp.z), fabsf(vp.w)));
                    amax = group8_max(amax);
                    d = amax / 127.0f;                           // Ggml.cs:751
                    id = d != 0.0f ? 1.0f / d : 0.0f;            // Ggml.cs:752
                }
                const int q0 = (int)rintf(vp.x * id), q1 = (int)rintf(vp.y * id);   // Ggml.cs:758-759 (D1, D2)
                const int q2_ = (int)rintf(vp.z * id), q3 = (int)rintf(vp.w * id);
                const int sum = group8_sum(q0 + q1 + q2_ + q3);
                const uint32_t e16 = ((uint32_t)q0 & 0xFFu) | (((uint32_t)q2_ & 0xFFu) << 8);
                const uint32_t o16 = ((uint32_t)q1 & 0xFFu) | (((uint32_t)q3 & 0xFFu) << 8);
                const bool even_lane = (t & 1) == 0;
                const uint32_t recv = (uint32_t)dpp_i<DPP_XOR1>((int)(even_lane ? o16 : e16));
                const uint32_t word = even_lane ? (e16 | (recv << 16)) : (recv | (o16 << 16));
                const int h = even_lane ? 0 : 1, off = even_lane ? 2 * t : 2 * t - 2;
                // a block past the end of K is written as zeros (its weights carry dw = 0; 0 * garbage could be NaN)
                *(uint32_t *)(myq + i * QSLOT + (c * 2 + h) * 16 + off) = live ? word : 0u;
                if (t == 0) { myd[i * NC + c] = live ? d : 0.0f; mys[i * NC + c] = live ? sum : 0; }
            }
            // same wave wrote and reads: LDS operations of one wave complete in order; only the compiler needs the fence
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            staged = true;
        }

        // 4. integer block dots + f32 scale-accumulate (Ggml.cs:1136-1159)
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int i = kq + GV_NKQ * j;
            const bool ok = u + GV_WORKERS * j < nbc;
            const float dwj = ok ? DW[j] : 0.0f;          // dw = 0 kills the contribution of a block past the end
            const float mwj = HAS_M && ok ? MW[HAS_M ? j : 0] : 0.0f;
            const uint32_t qq[4] = {Q[j].x, Q[j].y, Q[j].z, Q[j].w};
            uint32_t lo[4], hi[4];
            if (GV_W_I8<TYPE>) {
                lo[0] = Q[j].x; lo[1] = Q[j].y; lo[2] = Q[j].z; lo[3] = Q[j].w;
                hi[0] = Q2[j].x; hi[1] = Q2[j].y; hi[2] = Q2[j].z; hi[3] = Q2[j].w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lo[k] = qq[k] & 0x0F0F0F0Fu;          // elements 8k+0,2,4,6  (Ggml.cs:1149)
                    hi[k] = (qq[k] >> 4) & 0x0F0F0F0Fu;   // elements 8k+1,3,5,7  (Ggml.cs:1150)
                    if (HAS_H) {                          // Ggml.cs:1285-1289 / 1330-1334
                        lo[k] |= q5_high_bits(HB[j], k, 0);
                        hi[k] |= q5_high_bits(HB[j], k, 1);
                    }
                    if (TYPE == GGML_TYPE_Q4_2) {         // (nib - 8) bytewise: the two half-block sums need their own offsets
                        lo[k] = ((lo[k] | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
                        hi[k] = ((hi[k] | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const uint4 a0 = *(const uint4 *)(myq + i * QSLOT + (c * 2 + 0) * 16);
                const uint4 a1 = *(const uint4 *)(myq + i * QSLOT + (c * 2 + 1) * 16);
                const float da = myd[i * NC + c];
                const int sa = mys[i * NC + c];
                if (GV_TWO_SC<TYPE>) {
                    int s0 = 0, s1 = 0;
                    s0 = dot4(lo[0], a0.x, s0); s0 = dot4(lo[1], a0.y, s0); s0 = dot4(hi[0], a1.x, s0); s0 = dot4(hi[1], a1.y, s0);
                    s1 = dot4(lo[2], a0.z, s1); s1 = dot4(lo[3], a0.w, s1); s1 = dot4(hi[2], a1.z, s1); s1 = dot4(hi[3], a1.w, s1);
                    acc[c] = fmaf(dwj * da, (float)s0, acc[c]);
                    acc[c] = fmaf(mwj * da, (float)s1, acc[c]);
                    continue;
                }
                int sdot = 0;
                sdot = dot4(lo[0], a0.x, sdot); sdot = dot4(lo[1], a0.y, sdot); sdot = dot4(lo[2], a0.z, sdot); sdot = dot4(lo[3], a0.w, sdot);
                sdot = dot4(hi[0], a1.x, sdot); sdot = dot4(hi[1], a1.y, sdot); sdot = dot4(hi[2], a1.z, sdot); sdot = dot4(hi[3], a1.w, sdot);
                if (TYPE == GGML_TYPE_Q4_0) sdot -= 8 * sa;   // (nib - 8) * a summed = nib*a summed - 8 * sum(a)
                if (TYPE == GGML_TYPE_Q5_0) sdot -= 16 * sa;
                acc[c] = fmaf(dwj * da, (float)sdot, acc[c]);
                if (TYPE == GGML_TYPE_Q4_1 || TYPE == GGML_TYPE_Q5_1) acc[c] = fmaf(mwj, da * (float)sa, acc[c]);   // + m * (s0 + s1)
            }
        }
        if (cidx != nchunks - 1) return;                // the row tile's last chunk: reduce and store below

        // 5. k-lanes by two xor-shuffles, waves through LDS: the fixed tree of the kernel above.  sRed alternates between two
        //    buffers, so ONE barrier per tile orders everything (tile t+2's writes come after tile t+1's barrier, which every
        //    wave reaches only after its tile-t reads)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float vsum = acc[c];
#pragma unroll
            for (int sft = GV_ROWS; sft < 64; sft <<= 1) vsum += __shfl_xor(vsum, sft);
            if (kq == 0) sRed[parity][wave][c][r] = vsum;
        }
        __syncthreads();
        for (int o = tid; o < GV_ROWS * NC; o += GV_THREADS) {
            const int c = o / GV_ROWS, rr = o % GV_ROWS;
            const int64_t m = (int64_t)tile * GV_ROWS + rr;
            float quad[GV_WAVES / 4];                                                                   // fixed tree over the waves
#pragma unroll
            for (int qd = 0; qd < GV_WAVES / 4; ++qd)
                quad[qd] = (sRed[parity][4 * qd][c][rr] + sRed[parity][4 * qd + 1][c][rr]) + (sRed[parity][4 * qd + 2][c][rr] + sRed[parity][4 * qd + 3][c][rr]);
            const float tot = GV_WAVES == 4 ? quad[0] : GV_WAVES == 8 ? quad[0] + quad[1 % (GV_WAVES / 4)]
                                            : (quad[0] + quad[1 % (GV_WAVES / 4)]) + (quad[2 % (GV_WAVES / 4)] + quad[3 % (GV_WAVES / 4)]);
            if (m < M && c < N) {
                // the node that follows the mul_mat, applied as the product is stored (common.h mm_epilogue)
                st_result(dst + (int64_t)c * ldd + m, ep.mode == 2 ? tot * ep.scale : tot);
                if (ep.mode == 1) st_result(ep.dst2 + (int64_t)c * ep.ld2 + m, tot + ep.addend[(int64_t)c * ep.ld_add + m]);
            }
        }
        parity ^= 1;
    };

    if constexpr (PF) {
        // The look-ahead pays where a workgroup walks several row tiles of one chunk (4096 x 4096: 5.06 -> 4.66 us, M = 32000:
        // 19.3 -> 18.7).  Two register sets take turns (the loop is unrolled by two items): a copy "current = next" at the end
        // of an item would make the wave wait for the look-ahead data THERE, i.e. drain the memory pipe once per row tile.
        // The steady-state loop issues its look-ahead loads UNCONDITIONALLY (a load sequence behind a branch makes hipcc's
        // wait counts assume the worst of both paths: every wait for the current set then also waited for half of the set
        // just requested); the last one or two items run below it.
        int w = 0;
        for (; w + 2 < nitems; w += 2) {
            load_item(w + 1, qn, q2n, dwn, mwn, hbn);                          // the NEXT item's weight stream goes out first
            do_item(w, q, q2, dw, mw, hb);
            load_item(w + 2, q, q2, dw, mw, hb);
            do_item(w + 1, qn, q2n, dwn, mwn, hbn);
        }
        if (w + 1 < nitems) {
            load_item(w + 1, qn, q2n, dwn, mwn, hbn);
            do_item(w, q, q2, dw, mw, hb);
            do_item(w + 1, qn, q2n, dwn, mwn, hbn);
        } else if (w < nitems) {
            do_item(w, q, q2, dw, mw, hb);
        }
    } else {
        // with K in several chunks the activations of every item have to be fetched and quantized as well, and asking for them
        // first, weights behind, in the item's own iteration measured better -- again with the two-set look-ahead of the
        // single-chunk form extended to the activations (4096 x 11008: 8.1 against 8.7 us, 11008 x 11008: 16.5 against 18.1),
        // and against ONE chunk of 384 k-blocks (all 11 block loads of a lane at once, same bits: any chunk size that is a
        // multiple of 32 leaves a lane's block sequence alone): 4096 x 11008 9.4 us, 4096 x 8192 8.8 against 6.6
        for (int w = 0; w < nitems; ++w) {
            if (w > 0) {
                if (!(single_chunk && staged)) load_x(w, 0);
                load_item(w, q, q2, dw, mw, hb);
            }
            do_item(w, q, q2, dw, mw, hb);
        }
    }
}

// one weight matrix: the arguments as they always were (the set of one is built in registers; a kernel that takes the whole
// mv_set in its argument segment ran the single-matrix call 5 % slower: 4096 x 4096 4.60 -> 4.85 us)
template <int TYPE, int NC, int GV_ROWS, bool SC, bool PRO = false, bool K8 = false>
__global__ __launch_bounds__(GV_THREADS, (NC <= 2 && !PRO ? 4 : 2)) void gemv_fused_kernel(   // two resident workgroups per CU up to 2 columns (4 columns: 133 registers, spills under 128)
    const uint8_t *__restrict__ qs, const uint32_t *__restrict__ gs, const float *__restrict__ x, int64_t ld1, float *__restrict__ dst,
    int64_t M, int64_t Mpad, int64_t nbk, int64_t ldd, int N, int ntiles, const mm_epilogue ep, const mm_prologue pro) {
    mv_set ws;
    ws.n = 1; ws.qs[0] = qs; ws.gs[0] = gs; ws.dst[0] = dst; ws.M[0] = M; ws.Mpad[0] = Mpad; ws.ldd[0] = ldd;
    gemv_fused_body<TYPE, NC, GV_ROWS, SC, PRO, false, K8>(ws, x, ld1, nbk, N, ntiles, ep, pro);
}
// several weight matrices behind one activation matrix (common.h mv_set)
template <int TYPE, int NC, int GV_ROWS, bool SC, bool PRO>
__global__ __launch_bounds__(GV_THREADS, (NC <= 2 && !PRO ? 4 : 2)) void gemv_fused_multi_kernel(
    const mv_set ws, const float *__restrict__ x, int64_t ld1, int64_t nbk, int N, int ntiles, const mm_prologue pro) {
    const mm_epilogue ep{0, nullptr, 0, nullptr, 0, 1.0f};
    gemv_fused_body<TYPE, NC, GV_ROWS, SC, PRO, true>(ws, x, ld1, nbk, N, ntiles, ep, pro);
}

// the prologue of the call in flight on this host thread (launch_gemv_q_fused_pro sets it around the type dispatch)
thread_local const mm_prologue *t_prologue = nullptr;

template <int TYPE, bool FUSED, int ROWS>
hipError_t launch_rows(const ggml_hip_weight *w, const float *x, int64_t ld1, act_planes p, int64_t N, float *dst,
                       int64_t ldd, hipStream_t st, const mm_epilogue *epp) {
    const mm_epilogue ep = epp ? *epp : mm_epilogue{0, nullptr, 0, nullptr, 0, 1.0f};
    const int ntiles = (int)((w->M + ROWS - 1) / ROWS);
    dim3 grid((unsigned)(ntiles < GV_MAX_WGS ? ntiles : GV_MAX_WGS));
    if constexpr (FUSED && !GV_OLD_FUSED) {
        if ((N <= 1 || N >= 3) && w->nbk <= GV_CHUNK && ntiles > GV_PF_WGS) grid = dim3((unsigned)GV_PF_WGS);
        // up to 4 columns; 5..8 stay on the block-staged kernel below (118 registers = two resident workgroups per CU; the
        // wave-private form needs 146 there: 32000 x 4096 x 8 41.9 against 48.5 us)
        if (N <= 4) {
#define GVF_ARGS (TYPE == GV_TYPE_I8X2 ? w->i8p : w->qs), w->gs, x, ld1, dst, w->M, w->Mpad, w->nbk, ldd, (int)N, ntiles, ep
#define GVF_LAUNCH(NC) do { \
        if (w->ext_type != 0) {                 /* a k-quant weight (planar Q5_1 form): activations by the Q8_K rule */ \
            if constexpr (TYPE == GGML_TYPE_Q5_1 || TYPE == GV_TYPE_I8X2) { \
                if (t_prologue || ep.mode != 0 || w->nbk % 8 != 0 || w->nbk / 8 > GV_K8_SB || (TYPE == GV_TYPE_I8X2 && !w->i8p)) return hipErrorNotSupported; \
                if (w->nbk <= GV_CHUNK) gemv_fused_kernel<TYPE, NC, ROWS, true, false, true><<<grid, GV_THREADS, 0, st>>>(GVF_ARGS, mm_prologue{nullptr, 0, nullptr, nullptr}); \
                else gemv_fused_kernel<TYPE, NC, ROWS, false, false, true><<<grid, GV_THREADS, 0, st>>>(GVF_ARGS, mm_prologue{nullptr, 0, nullptr, nullptr}); \
            } else return hipErrorNotSupported; \
        } else if constexpr (TYPE == GV_TYPE_I8X2) return hipErrorNotSupported;   /* (the form exists for the k-quant extension only) */ \
        else if (t_prologue) { \
            if (w->nbk <= GV_CHUNK) gemv_fused_kernel<TYPE, NC, ROWS, true, true><<<grid, GV_THREADS, 0, st>>>(GVF_ARGS, *t_prologue); \
            else gemv_fused_kernel<TYPE, NC, ROWS, false, true><<<grid, GV_THREADS, 0, st>>>(GVF_ARGS, *t_prologue); \
        } else if (w->nbk <= GV_CHUNK) gemv_fused_kernel<TYPE, NC, ROWS, true><<<grid, GV_THREADS, 0, st>>>(GVF_ARGS, mm_prologue{nullptr, 0, nullptr, nullptr}); \
        else gemv_fused_kernel<TYPE, NC, ROWS, false><<<grid, GV_THREADS, 0, st>>>(GVF_ARGS, mm_prologue{nullptr, 0, nullptr, nullptr}); } while (0)
            if (N <= 1) GVF_LAUNCH(1);
            else if (N <= 2) GVF_LAUNCH(2);
            else GVF_LAUNCH(4);
#undef GVF_LAUNCH
#undef GVF_ARGS
            return hipGetLastError();
        }
    }
    if (ep.mode != 0 || t_prologue) return hipErrorNotSupported;          // (callers ask gemv_fused_has_epilogue first)
    if (FUSED && w->ext_type != 0) return hipErrorNotSupported;           // (the block-staged fused form quantizes by the Q8_0 rule: plan.cpp keeps k-quants off it)
#define GV_LAUNCH(NC) gemv_q_kernel<TYPE, NC, FUSED, ROWS><<<grid, GV_THREADS, 0, st>>>((TYPE == GV_TYPE_I8X2 ? w->i8p : w->qs), w->gs, x, ld1, p.a8, p.ad, p.as, dst, w->M, w->Mpad, p.Npad, w->nbk, ldd, (int)N, ntiles)
    if (N <= 1) GV_LAUNCH(1);
    else if (N <= 2) GV_LAUNCH(2);
    else if (N <= 4) GV_LAUNCH(4);
    else if (N <= 8) GV_LAUNCH(8);
    else if constexpr (!FUSED) GV_LAUNCH(16);   // 9..16 rows: the weights are still streamed once, 16 dot products per block
#undef GV_LAUNCH
    return hipGetLastError();
}

template <int TYPE, bool FUSED>
hipError_t launch_typed(const ggml_hip_weight *w, const float *x, int64_t ld1, act_planes p, int64_t N, float *dst,
                        int64_t ldd, hipStream_t st, const mm_epilogue *ep) {
    // 32 rows per workgroup measured 5 % faster at M = 32000 (4.46 vs 4.24 TB/s) and 13 % slower at M = 4096; a choice by
    // M would change the summation tree between a row shard and the unsplit matrix, and the multi-GPU path promises
    // bit-identical results for any split -- so one shape for every M.
    return launch_rows<TYPE, FUSED, 16>(w, x, ld1, p, N, dst, ldd, st, ep);
}

template <bool FUSED>
hipError_t launch_any(const ggml_hip_weight *w, const float *x, int64_t ld1, act_planes p, int64_t N, float *dst,
                      int64_t ldd, hipStream_t st, const mm_epilogue *ep = nullptr) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    if (N > (FUSED ? GEMV_MAX_N : GEMV_WIDE_MAX_N)) return hipErrorInvalidValue;
    switch (w->type) {
    case GGML_TYPE_Q4_0: return launch_typed<GGML_TYPE_Q4_0, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep);
    case GGML_TYPE_Q4_1: return launch_typed<GGML_TYPE_Q4_1, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep);
    case GGML_TYPE_Q5_0: return launch_typed<GGML_TYPE_Q5_0, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep);
    case GGML_TYPE_Q4_2:
        if (w->ext_type != 0) return w->i8p ? launch_typed<GV_TYPE_I8X2, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep) : hipErrorInvalidValue;   // (Q6_K: int8 planes)
        return launch_typed<GGML_TYPE_Q4_2, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep);
    case GGML_TYPE_Q5_1: return launch_typed<GGML_TYPE_Q5_1, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep);
    case GGML_TYPE_Q8_0: return launch_typed<GGML_TYPE_Q8_0, FUSED>(w, x, ld1, p, N, dst, ldd, st, ep);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_gemv_q(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st) {
    return launch_any<false>(w, nullptr, 0, p, N, dst, ldd, st);
}

hipError_t launch_gemv_q_fused(const ggml_hip_weight *w, const float *x, int64_t ld1, int64_t N, float *dst, int64_t ldd,
                               hipStream_t st, const mm_epilogue *ep) {
    act_planes none = {nullptr, nullptr, nullptr, 0};
    return launch_any<true>(w, x, ld1, none, N, dst, ldd, st, ep);
}

namespace {
template <int TYPE>
hipError_t launch_multi_typed(const mv_set &ws, const float *x, int64_t ld1, const mm_prologue *pro, int64_t nbk, int64_t N, int ntiles, hipStream_t st) {
    const mm_prologue none{nullptr, 0, nullptr, nullptr};
    const bool sc = nbk <= GV_CHUNK;
    const int cap = ((N <= 1 || N >= 3) && sc) ? GV_PF_WGS : GV_MAX_WGS;
    dim3 grid((unsigned)(ntiles < cap ? ntiles : cap));
#define GVM(NC) do { \
        if (pro) { if (sc) gemv_fused_multi_kernel<TYPE, NC, 16, true, true><<<grid, GV_THREADS, 0, st>>>(ws, x, ld1, nbk, (int)N, ntiles, *pro); \
                   else gemv_fused_multi_kernel<TYPE, NC, 16, false, true><<<grid, GV_THREADS, 0, st>>>(ws, x, ld1, nbk, (int)N, ntiles, *pro); } \
        else if (sc) gemv_fused_multi_kernel<TYPE, NC, 16, true, false><<<grid, GV_THREADS, 0, st>>>(ws, x, ld1, nbk, (int)N, ntiles, none); \
        else gemv_fused_multi_kernel<TYPE, NC, 16, false, false><<<grid, GV_THREADS, 0, st>>>(ws, x, ld1, nbk, (int)N, ntiles, none); } while (0)
    if (N <= 1) GVM(1);
    else if (N <= 2) GVM(2);
    else GVM(4);
#undef GVM
    return hipGetLastError();
}
}  // namespace

hipError_t launch_gemv_q_fused_multi(const ggml_hip_weight *const *w, int n_w, const float *x, int64_t ld1, const mm_prologue *pro, int64_t N,
                                     float *const *dst, const int64_t *ldd, hipStream_t st) {
    if (n_w < 1 || n_w > 4 || N < 1 || N > 4 || GV_OLD_FUSED) return hipErrorNotSupported;
    mv_set ws;
    memset(&ws, 0, sizeof ws);
    ws.n = n_w;
    int ntiles = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < n_w) {
            if (w[i]->type != w[0]->type || w[i]->nbk != w[0]->nbk || w[i]->M <= 0) return hipErrorInvalidValue;
            ntiles += (int)((w[i]->M + 15) / 16);
            ws.qs[i] = w[i]->qs; ws.gs[i] = w[i]->gs; ws.dst[i] = dst[i]; ws.M[i] = w[i]->M; ws.Mpad[i] = w[i]->Mpad; ws.ldd[i] = ldd[i];
        }
        ws.tile_end[i] = ntiles;
    }
    switch (w[0]->type) {
    case GGML_TYPE_Q4_0: return launch_multi_typed<GGML_TYPE_Q4_0>(ws, x, ld1, pro, w[0]->nbk, N, ntiles, st);
    case GGML_TYPE_Q4_1: return launch_multi_typed<GGML_TYPE_Q4_1>(ws, x, ld1, pro, w[0]->nbk, N, ntiles, st);
    case GGML_TYPE_Q5_0: return launch_multi_typed<GGML_TYPE_Q5_0>(ws, x, ld1, pro, w[0]->nbk, N, ntiles, st);
    case GGML_TYPE_Q4_2: return launch_multi_typed<GGML_TYPE_Q4_2>(ws, x, ld1, pro, w[0]->nbk, N, ntiles, st);
    case GGML_TYPE_Q5_1: return launch_multi_typed<GGML_TYPE_Q5_1>(ws, x, ld1, pro, w[0]->nbk, N, ntiles, st);
    case GGML_TYPE_Q8_0: return launch_multi_typed<GGML_TYPE_Q8_0>(ws, x, ld1, pro, w[0]->nbk, N, ntiles, st);
    default: return hipErrorInvalidValue;
    }
}

bool gemv_fused_has_epilogue(int64_t N) { return N >= 1 && N <= 4 && !GV_OLD_FUSED; }

hipError_t launch_gemv_q_fused_pro(const ggml_hip_weight *w, const float *x, int64_t ld1, const mm_prologue &pro, int64_t N, float *dst,
                                   int64_t ldd, hipStream_t st, const mm_epilogue *ep) {
    if (!gemv_fused_has_epilogue(N)) return hipErrorNotSupported;
    struct Scope {
        explicit Scope(const mm_prologue *p) { t_prologue = p; }
        ~Scope() { t_prologue = nullptr; }
    } scope(&pro);
    act_planes none = {nullptr, nullptr, nullptr, 0};
    return launch_any<true>(w, x, ld1, none, N, dst, ldd, st, ep);
}
