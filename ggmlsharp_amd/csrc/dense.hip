// dense.hip -- K10: dense f32 x f32 and f16 x f32 mul_mat on the f32-input matrix cores.
//
// ggml_compute_forward_mul_mat_f32 (Ggml.cs:5969-6178) with ggml_vec_dot_f32 (Ggml.cs:2631-2640: f32 products,
// f64 running sum) and ggml_compute_forward_mul_mat_f16_f32 (Ggml.cs:6180-6438: INIT rounds src1 to Half
// :6362-6379, then ggml_vec_dot_f16 :2642-2651 multiplies (float)h * (float)h).
// v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered f32 fmaf chain (no reduced-precision shortcut exists on
// gfx950), i.e. the reference's sum with f32 instead of f64 accumulation: ~1e-6 relative at these K, inside the
// 1e-3 budget.  Same orientation as the quantized kernel: MFMA rows = src1 rows n, MFMA cols = weight rows m.
#include "common.h"
#include "plan.h"
#include <cstdlib>
#include <hip/hip_fp16.h>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

#define DT 64    // tile edge (n and m)
#define DK 32    // k per stage
#define DLD 33   // padded LDS row (floats): column reads hit 32 distinct banks

// VEC: K % 8 == 0, ld1 % 4 == 0 and src1 16-byte aligned -- a thread's 8 consecutive k are two 16-byte loads per operand
// (one for f16 weights) instead of 8 scalar ones (the scalar form spends its time in the texture addresser: 28 -> 70 TFLOP/s)
template <bool W_F16, bool VEC>
__global__ __launch_bounds__(256) void dense_kernel(const void *__restrict__ wv, const float *__restrict__ x,
                                                   float *__restrict__ dst, int64_t M, int64_t N, int64_t K, int64_t ld1,
                                                   int64_t ldd) {
    __shared__ float sX[DT * DLD];
    __shared__ float sW[DT * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wm_ = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * DT, n0 = (int64_t)blockIdx.y * DT;

    const int srow = tid >> 2, sk = (tid & 3) * 8;
    const int64_t xr = (n0 + srow) < N ? (n0 + srow) : (N - 1);
    const float *xp = x + xr * ld1;
    const int64_t wr = m0 + srow;  // < Mpad, padded rows are zero

    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (int64_t k0 = 0; k0 < K; k0 += DK) {
        float xv[8], wv8[8];
        if constexpr (VEC) {
            const float4 x0 = *(const float4 *)(xp + k0 + sk), x1 = *(const float4 *)(xp + k0 + sk + 4);
            xv[0] = x0.x; xv[1] = x0.y; xv[2] = x0.z; xv[3] = x0.w; xv[4] = x1.x; xv[5] = x1.y; xv[6] = x1.z; xv[7] = x1.w;
            if (W_F16) {
                const uint4 q = *(const uint4 *)((const __half *)wv + wr * K + k0 + sk);
                const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wv8[2 * e + 0] = __half2float(__ushort_as_half((unsigned short)(u[e] & 0xFFFFu)));
                    wv8[2 * e + 1] = __half2float(__ushort_as_half((unsigned short)(u[e] >> 16)));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = __half2float(__float2half_rn(xv[e]));   // (Half)src1, Ggml.cs:6369
            } else {
                const float4 w0 = *(const float4 *)((const float *)wv + wr * K + k0 + sk), w1 = *(const float4 *)((const float *)wv + wr * K + k0 + sk + 4);
                wv8[0] = w0.x; wv8[1] = w0.y; wv8[2] = w0.z; wv8[3] = w0.w; wv8[4] = w1.x; wv8[5] = w1.y; wv8[6] = w1.z; wv8[7] = w1.w;
            }
        } else
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t k = k0 + sk + e;
            float xe = k < K ? xp[k] : 0.0f;
            float we;
            if (W_F16) {
                const __half *wp = (const __half *)wv + wr * K;
                we = k < K ? __half2float(wp[k]) : 0.0f;
                xe = __half2float(__float2half_rn(xe));  // (Half)src1, Ggml.cs:6369
            } else {
                const float *wp = (const float *)wv + wr * K;
                we = k < K ? wp[k] : 0.0f;
            }
            xv[e] = xe;
            wv8[e] = we;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sX[srow * DLD + sk + e] = xv[e];
            sW[srow * DLD + sk + e] = wv8[e];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < DK / 2; ++s) {
            const float a = sX[(wn * 32 + l31) * DLD + 2 * s + hh];
            const float b = sW[(wm_ * 32 + l31) * DLD + 2 * s + hh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int64_t m = m0 + wm_ * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < N && m < M) dst[n * ldd + m] = acc[r];
    }
}

// ---- F32 weights, large shapes: 128 x 128 tiles, 2 x 2 MFMA tiles per wave.  The 64 x 64 kernel above moves 16 KB from L2 per
//      workgroup and 32-k stage for 16 MFMAs per wave -- ~48 B per clock and CU, the L1's limit; a 128 x 128 tile halves the
//      bytes per flop and the LDS reads per MFMA.  Same arithmetic, same order (k ascending, two per v_mfma_f32_32x32x2_f32):
//      bit for bit the result of the kernel above.  XCD-aware 1-D grid (2 x 4 blocks of the tile grid per XCD, gemm_qmx.hip).
//      Needs K % 32 == 0, ld1 % 4 == 0, 16-byte aligned src1.
__global__ __launch_bounds__(256) void dense_f32_big_kernel(const float *__restrict__ w, const float *__restrict__ x, float *__restrict__ dst,
                                                           int64_t M, int64_t N, int64_t K, int64_t ld1, int64_t ldd, int tiles_m,
                                                           int tiles_n) {
    constexpr int BT = 128;
    __shared__ float sX[BT * DLD];
    __shared__ float sW[BT * DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wm_ = wave & 1;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    int tm_i, tn_i;
    if ((tiles_m & 1) == 0 && (tiles_n & 3) == 0) {
        const int hm = tiles_m >> 1, l = bid >> 3;
        tm_i = (xcd & 1) * hm + l % hm;
        tn_i = (xcd >> 1) * (tiles_n >> 2) + l / hm;
    } else {
        const int t_lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        tm_i = t_lin % tiles_m;
        tn_i = t_lin / tiles_m;
    }
    const int64_t m0 = (int64_t)tm_i * BT, n0 = (int64_t)tn_i * BT;

    const int srow = tid >> 1, sk = (tid & 1) * 16;          // a thread stages 16 consecutive k of one row of each operand
    const int64_t xr = (n0 + srow) < N ? (n0 + srow) : (N - 1);
    const float *xp = x + xr * ld1 + sk;
    const float *wp = w + (m0 + srow) * K + sk;              // < Mpad rows (padded rows are zero)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 xv[4], wv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { xv[e] = *(const float4 *)(xp + 4 * e); wv[e] = *(const float4 *)(wp + 4 * e); }
    for (int64_t k0 = 0; k0 < K; k0 += DK) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float *px = &sX[srow * DLD + sk + 4 * e], *pw = &sW[srow * DLD + sk + 4 * e];
            px[0] = xv[e].x; px[1] = xv[e].y; px[2] = xv[e].z; px[3] = xv[e].w;
            pw[0] = wv[e].x; pw[1] = wv[e].y; pw[2] = wv[e].z; pw[3] = wv[e].w;
        }
        __syncthreads();
        if (k0 + DK < K) {                                    // the next stage's loads fly during this stage's MFMAs
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[e] = *(const float4 *)(xp + k0 + DK + 4 * e); wv[e] = *(const float4 *)(wp + k0 + DK + 4 * e); }
        }
#pragma unroll
        for (int s = 0; s < DK / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) a[j] = sX[(wn * 64 + 32 * j + l31) * DLD + 2 * s + hh];
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = sW[(wm_ * 64 + 32 * i + l31) * DLD + 2 * s + hh];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = m0 + wm_ * 64 + 32 * i + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t n = n0 + wn * 64 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const float v = acc[i][j][r];
                if (n < N && m < M) dst[n * ldd + m] = v;
            }
        }
}

// ---- F32 weights, 5 .. 256 src1 rows (r4; the bounds are plan.cpp plan_dense's): K split over the eight waves of a workgroup.  The tile kernels above give a workgroup the whole
//      K: at these sizes a 4096 x 4096 matrix is 64 .. 256 tiles of 64 x 64, every wave runs 2048 sixteen-pass MFMAs one after the other and the
//      launch takes a flat 120 us whatever N is (the weights stream in 12).  Here a workgroup owns ONE 32 x 32 tile and each wave an eighth of
//      K (a contiguous range of whole 32-k stages), staged through the wave's own LDS slice (no workgroup barrier in the K loop: a thread
//      brings 16 consecutive k of one row of each operand, 16-byte pieces where src1 allows it, the next stage's loads fly during this
//      stage's MFMAs); the eight partial tiles are added in wave order through LDS, every wave taking two of the sixteen accumulator
//      registers.  Arithmetic inside a range as above (k ascending, two per v_mfma_f32_32x32x2_f32); tree: 8 contiguous ranges -- by K alone.
template <bool VEC>
__global__ __launch_bounds__(512) void dense_f32_ksplit_kernel(const float *__restrict__ w, const float *__restrict__ x, float *__restrict__ dst,
                                                              int64_t M, int64_t N, int64_t K, int64_t ld1, int64_t ldd, int kr) {
    extern __shared__ __attribute__((aligned(16))) float sS[];   // 8 * 2 * 32 * DLD floats (66 KB: dynamic, opted in by the launcher) -- per wave 32 rows of src1 and 32 of weights, 32 k each (+ pad)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * 32, n0 = (int64_t)blockIdx.y * 32;
    float *sX = sS + wave * (2 * 32 * DLD), *sW = sX + 32 * DLD;
    const int srow = lane >> 1, sk = (lane & 1) * 16;        // a lane stages 16 consecutive k of one row of each operand
    const int64_t xr = (n0 + srow) < N ? (n0 + srow) : (N - 1);
    const int64_t kb = (int64_t)wave * kr, ke = kb + kr < K ? kb + kr : K;      // this wave's range (whole stages; may be empty)
    const float *xp = x + xr * ld1 + sk;
    const float *wp = w + (m0 + srow) * K + sk;              // < Mpad rows (padded rows are zero)
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float4 xv[4], wv[4];
    auto load = [&](int64_t k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (VEC) xv[e] = *(const float4 *)(xp + k0 + 4 * e);
            else { const float *q = xp + k0 + 4 * e; xv[e] = make_float4(q[0], q[1], q[2], q[3]); }
            wv[e] = *(const float4 *)(wp + k0 + 4 * e);      // (rows of the resident copy: K % 4 == 0 keeps them 16-byte aligned)
        }
    };
    if (kb < ke) load(kb);
    for (int64_t k0 = kb; k0 < ke; k0 += DK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float *px = &sX[srow * DLD + sk + 4 * e], *pw = &sW[srow * DLD + sk + 4 * e];
            px[0] = xv[e].x; px[1] = xv[e].y; px[2] = xv[e].z; px[3] = xv[e].w;
            pw[0] = wv[e].x; pw[1] = wv[e].y; pw[2] = wv[e].z; pw[3] = wv[e].w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (k0 + DK < ke) load(k0 + DK);                     // the next stage's loads fly during this stage's MFMAs
#pragma unroll
        for (int s = 0; s < DK / 2; ++s) {
            const float a = sX[l31 * DLD + 2 * s + hh];
            const float b = sW[l31 * DLD + 2 * s + hh];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                     // (the slice is rewritten by the next trip)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- the waves' partial tiles, added in wave order; wave v takes accumulator registers 2v and 2v + 1 ----
    __syncthreads();
    float *xch = sS + lane;                                   // [wave][register][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64] = acc[r];
    __syncthreads();
    const int64_t m = m0 + l31;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = 2 * wave + q;
        float v = xch[r * 64];
#pragma unroll
        for (int g = 1; g < 8; ++g) v += xch[(g * 16 + r) * 64];
        const int64_t n = n0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < N && m < M) dst[n * ldd + m] = v;
    }
}

// ---- N <= 8 (more rows in passes of 8): mat-vec, bandwidth-bound.  One wave per DGR weight rows: lanes stride K in 16-byte
//      pieces, f32 products and f32 partial sums per lane (the reference sums the same products in f64, Ggml.cs:2633 / 2644:
//      ~1e-6 relative at these K), a fixed xor-shuffle tree across the wave.  F16 weights: src1 is rounded to Half first
//      (Ggml.cs:6369).  A piece of src1 is loaded once for the wave's DGR rows (with one row per wave the L2 reads of src1
//      were 2 .. 16 x the weight stream: 32000 x 4096 f16, N = 8: 146 us). ----
template <bool W_F16, int NC, int DGR>                       // DGR rows per wave: 1 for one or two columns (more waves in flight), 4 above
__global__ __launch_bounds__(256) void dense_gemv_kernel(const void *__restrict__ wv, const float *__restrict__ x, float *__restrict__ dst,
                                                        int64_t M, int N, int64_t K, int64_t ld1, int64_t ldd) {
    const int lane = threadIdx.x & 63;
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * DGR;
    if (m0 >= M) return;                                     // uniform per wave
    float acc[DGR][NC];
#pragma unroll
    for (int r = 0; r < DGR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = 0.0f;
    constexpr int E = W_F16 ? 8 : 4;                         // weight elements per 16-byte piece
    const int64_t Kv = K / E * E;
    for (int64_t k = (int64_t)lane * E; k < Kv; k += 64 * E) {
        float wf[DGR][E];
#pragma unroll
        for (int r = 0; r < DGR; ++r) {
            const int64_t m = m0 + r < M ? m0 + r : M - 1;   // rows past the end repeat the last one (not stored)
            if (W_F16) {
                const uint4 q = *(const uint4 *)((const __half *)wv + m * K + k);
                const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wf[r][2 * e + 0] = __half2float(__ushort_as_half((unsigned short)(u[e] & 0xFFFFu)));
                    wf[r][2 * e + 1] = __half2float(__ushort_as_half((unsigned short)(u[e] >> 16)));
                }
            } else {
                const float4 q = *(const float4 *)((const float *)wv + m * K + k);
                wf[r][0] = q.x; wf[r][1] = q.y; wf[r][2] = q.z; wf[r][3] = q.w;
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float *xp = x + (int64_t)(c < N ? c : N - 1) * ld1 + k;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float xe = xp[e];
                if (W_F16) xe = __half2float(__float2half_rn(xe));
#pragma unroll
                for (int r = 0; r < DGR; ++r) acc[r][c] = fmaf(wf[r][e], xe, acc[r][c]);
            }
        }
    }
    if (lane == 0)                                           // K tail (K not a multiple of the piece)
        for (int64_t k = Kv; k < K; ++k)
#pragma unroll
            for (int r = 0; r < DGR; ++r) {
                const int64_t m = m0 + r < M ? m0 + r : M - 1;
                const float we = W_F16 ? __half2float(((const __half *)wv)[m * K + k]) : ((const float *)wv)[m * K + k];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float xe = x[(int64_t)(c < N ? c : N - 1) * ld1 + k];
                    if (W_F16) xe = __half2float(__float2half_rn(xe));
                    acc[r][c] = fmaf(we, xe, acc[r][c]);
                }
            }
#pragma unroll
    for (int r = 0; r < DGR; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float v = acc[r][c];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
            if (lane == 0 && c < N && m0 + r < M) dst[(int64_t)c * ldd + m0 + r] = v;
        }
}

}  // namespace

hipError_t launch_dense(const ggml_hip_weight *w, const mm_plan &pl, const float *x, int64_t N, int64_t ld1, float *dst, int64_t ldd,
                        hipStream_t st) {
    if (N <= 0 || w->M <= 0) return hipSuccess;
    if (pl.family != MMF_DENSE && pl.family != MMF_DENSE_GEMV) return hipErrorInvalidValue;
    // mat-vec form: rows of the resident copy are K elements apart, so 16-byte pieces need K % 8 (f16) / K % 4 (f32) == 0
    const bool f16 = w->type == GGML_TYPE_F16;
    // up to 16 rows (plan.cpp plan_dense): passes of 8 columns over the weights (4096 x 4096 x 16: two passes 53 us, the tile kernel
    // below 121 us; at 32 rows and M = 11008 the tile kernel is ahead, 121 against 243 us)
    if (pl.family == MMF_DENSE_GEMV) {
        for (int64_t c0 = 0; c0 < N; c0 += 8) {
            const int n = (int)(N - c0 < 8 ? N - c0 : 8);
            const float *xc = x + c0 * ld1;
            float *dc = dst + c0 * ldd;
#define DG(F, NC) dense_gemv_kernel<F, NC, (NC <= 2 ? 1 : 4)><<<dim3((unsigned)((w->M + 4 * (NC <= 2 ? 1 : 4) - 1) / (4 * (NC <= 2 ? 1 : 4)))), 256, 0, st>>>(w->dense, xc, dc, w->M, n, w->K, ld1, ldd)
            if (f16) { if (n <= 1) DG(true, 1); else if (n <= 2) DG(true, 2); else if (n <= 4) DG(true, 4); else DG(true, 8); }
            else { if (n <= 1) DG(false, 1); else if (n <= 2) DG(false, 2); else if (n <= 4) DG(false, 4); else DG(false, 8); }
#undef DG
        }
        return hipGetLastError();
    }
    dim3 grid((unsigned)((w->M + DT - 1) / DT), (unsigned)((N + DT - 1) / DT));
    // K % 32 == 0 keeps every 8-element piece of a stage inside the row
    const bool vec = w->K % DK == 0 && ld1 % 4 == 0 && ((uintptr_t)x & 15) == 0;
    if (pl.form == DNF_KSPLIT) {                            // F32, 5 .. 256 rows, K % 256 == 0 (plan.cpp): a wave's range is K / 8, whole stages
        if (f16 || w->K % 256 != 0) return hipErrorInvalidValue;
        const dim3 g2((unsigned)((w->M + 31) / 32), (unsigned)((N + 31) / 32));
        const int kr = (int)(w->K / 8);
        const size_t lds = (size_t)8 * 2 * 32 * DLD * 4;
        (void)hipGetLastError();
#define DKS_GO(V) do { \
            auto kern = dense_f32_ksplit_kernel<V>; \
            static PerDeviceOnce once; \
            const hipError_t attr = once.max_dynamic_lds((const void *)kern, (int)lds); \
            if (attr != hipSuccess) return attr; \
            kern<<<g2, 512, lds, st>>>((const float *)w->dense, x, dst, w->M, N, w->K, ld1, ldd, kr); } while (0)
        if (vec) DKS_GO(true); else DKS_GO(false);           // (a strided or misaligned src1: the same sums from scalar loads)
#undef DKS_GO
        return hipGetLastError();
    }
    {   // F32 weights, enough 128 x 128 tiles to fill the chip (plan.cpp): the big-tile kernel (bitwise the same result; it reads src1 in
        // 16-byte pieces -- a strided or misaligned src1 takes the 64 x 64 kernel, whose fma chain is the same)
        const int64_t tm = (w->M + 127) / 128, tn = (N + 127) / 128;
        if (pl.form == DNF_BIG && vec) {
            dense_f32_big_kernel<<<dim3((unsigned)(tm * tn)), 256, 0, st>>>((const float *)w->dense, x, dst, w->M, N, w->K, ld1, ldd, (int)tm, (int)tn);
            return hipGetLastError();
        }
    }
    if (w->type == GGML_TYPE_F16) {
        if (vec) dense_kernel<true, true><<<grid, 256, 0, st>>>(w->dense, x, dst, w->M, N, w->K, ld1, ldd);
        else dense_kernel<true, false><<<grid, 256, 0, st>>>(w->dense, x, dst, w->M, N, w->K, ld1, ldd);
    } else {
        if (vec) dense_kernel<false, true><<<grid, 256, 0, st>>>(w->dense, x, dst, w->M, N, w->K, ld1, ldd);
        else dense_kernel<false, false><<<grid, 256, 0, st>>>(w->dense, x, dst, w->M, N, w->K, ld1, ldd);
    }
    return hipGetLastError();
}
