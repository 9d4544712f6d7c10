// valu_ubench.hip -- developer microbenchmark (not part of the product): VALU issue cost on gfx950 for the
// instruction mix of the block-scale epilogue, at 1/2/4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o gpurun_out/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x2 = __attribute__((ext_vector_type(2))) float;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using i32x16 = __attribute__((ext_vector_type(16))) int;

#define ITERS 2000

// MODE 0: 32 x v_fma_f32   1: 16 x v_pk_fma_f32   2: 32 x v_cvt_f32_i32   3: epilogue mix (16 cvt + 8 pk_mul + 8 pk_fma)
// 4: mix with single ops (16 cvt + 16 mul + 16 fma)  5: mix + one i8 MFMA per iteration  6: mfma only
template <int MODE>
__global__ void k(float *out, long long *cyc, int n) {
    float a[16], s[16];
    int t[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.5f + i; s[i] = 1.0f + 1e-6f * (threadIdx.x + i); t[i] = threadIdx.x + i; }
    float da = out[threadIdx.x & 7], dw = out[8 + (threadIdx.x & 7)];
    i32x4 fa = {t[0], t[1], t[2], t[3]}, fb = {t[4], t[5], t[6], t[7]};
    i32x16 T = {0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[r] = fmaf(a[r], s[r], da); s[r] = fmaf(s[r], dw, a[r]); }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 x = {a[r], a[r + 1]}, y = {s[r], s[r + 1]}, z = {da, dw};
                x = __builtin_elementwise_fma(x, y, z);
                y = __builtin_elementwise_fma(y, z, x);
                a[r] = x[0]; a[r + 1] = x[1]; s[r] = y[0]; s[r + 1] = y[1];
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[r] = (float)t[r]; t[r] = __float_as_int(a[r]) + it; }
        } else if (MODE == 3 || MODE == 5) {
            if (MODE == 5) T = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, T, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 tt = {(float)(MODE == 5 ? T[r] : t[r]), (float)(MODE == 5 ? T[r + 1] : t[r + 1])};
                f32x2 sc = {s[r] * dw, s[r + 1] * dw};
                f32x2 ac = {a[r], a[r + 1]};
                ac = __builtin_elementwise_fma(tt, sc, ac);
                a[r] = ac[0]; a[r + 1] = ac[1];
            }
            if (MODE == 3) { t[it & 15] += it; }
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[r] = fmaf((float)t[r], s[r] * dw, a[r]); }
            t[it & 15] += it;
        } else if (MODE == 6) {
            T = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, T, 0, 0, 0);
        } else if (MODE == 7) {   // 32 v_mul_f32_dpp row_newbcast
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[r]) : "v"(s[r]), "v"(dw));
                asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(s[r]) : "v"(a[r]), "v"(da));
            }
        } else if (MODE == 8) {   // f16 epilogue: 16 dpp mul + 16 fma + 2 f16 MFMAs
            using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
            using f32x16 = __attribute__((ext_vector_type(16))) float;
            f16x8 ha = __builtin_bit_cast(f16x8, fa), hb = __builtin_bit_cast(f16x8, fb);
            f32x16 X = {0};
            X = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, X, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, X, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sc;
                asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(sc) : "v"(s[r]), "v"(dw));
                a[r] = fmaf(X[r], sc, a[r]);
            }
        } else if (MODE == 9) {   // 32 v_mul_lo_u32
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[r] = t[r] * (it + 3); t[r] = t[r] * (t[(r + 1) & 15] | 1); }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(a[r]), "+v"(s[r]), "+v"(t[r]));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i] + s[i] + t[i] + T[i];
    out[blockIdx.x * blockDim.x + threadIdx.x + 16] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd, float *d_out, long long *d_cyc) {
    dim3 grid(256), block(64 * 4 * waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(d_out, d_cyc, 10);
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(d_out, d_cyc, ITERS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d: %8.1f us, %7.1f shader-cycles/iter/wave (memtime), %6.1f cycles/iter per SIMD, clock %.2f GHz\n", name,
           waves_per_simd, ms * 1e3, (double)cyc / ITERS, (double)cyc / ITERS / waves_per_simd, cyc / (ms * 1e6));
}

int main() {
    float *d_out; long long *d_cyc;
    hipMalloc(&d_out, 256 * 1024 * 4 + 64); hipMalloc(&d_cyc, 8);
    std::vector<float> h(16, 1.0f);
    hipMemcpy(d_out, h.data(), 64, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0>("32 v_fma_f32", w, d_out, d_cyc);
        run<1>("16 v_pk_fma_f32", w, d_out, d_cyc);
        run<2>("16 v_cvt_f32_i32 + 16 v_add", w, d_out, d_cyc);
        run<3>("mix 16cvt+8pk_mul+8pk_fma", w, d_out, d_cyc);
        run<4>("mix 16cvt+16mul+16fma", w, d_out, d_cyc);
        run<5>("mix(pk) + 1 mfma_i8", w, d_out, d_cyc);
        run<6>("1 mfma_i32_32x32x32_i8", w, d_out, d_cyc);
        run<7>("32 v_mul_f32_dpp", w, d_out, d_cyc);
        run<8>("f16 tile: 2 mfma + 16 dpp + 16 fma", w, d_out, d_cyc);
        run<9>("32 v_mul_lo_u32", w, d_out, d_cyc);
    }
    return 0;
}
