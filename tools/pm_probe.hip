// Developer probe (not product): an outer product of f32 scale vectors on the bf16 matrix pipe.
// P[n][m] = d1[n] * d0[m] from three bf16 pieces per scale (exact split by truncation) and the eight largest cross terms
// in the 8 k-slots of lanes 0..31 of one v_mfma_f32_32x32x16_bf16 (lanes 32..63 = k 8..15 hold zeros).  Reports the
// worst error against the correctly rounded f32 product in ulps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ void split3(float d, unsigned &p0, unsigned &p1, unsigned &p2) {   // top halves of the f32 patterns = bf16 bits
    const unsigned b0 = __float_as_uint(d) & 0xFFFF0000u;
    const float r1 = d - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b1);
    p0 = b0 >> 16; p1 = b1 >> 16; p2 = __float_as_uint(r2) >> 16;          // r2 has <= 8 significant bits: exact
}

__global__ void k(const float *d1, const float *d0, float *P) {
    const int lane = threadIdx.x, l31 = lane & 31, hh = lane >> 5;
    unsigned a0, a1, a2, w0, w1, w2;
    split3(d1[l31], a0, a1, a2);
    split3(d0[l31], w0, w1, w2);
    // slots: (a0,w0) (a0,w1) (a1,w0) (a1,w1) (a0,w2) (a2,w0) (a1,w2) (a2,w1)
    u32x4 A = {a0 | (a0 << 16), a1 | (a1 << 16), a0 | (a2 << 16), a1 | (a2 << 16)};
    u32x4 B = {w0 | (w1 << 16), w0 | (w1 << 16), w2 | (w0 << 16), w2 | (w1 << 16)};
    if (hh) { A = (u32x4){0, 0, 0, 0}; B = A; }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
        P[row * 32 + l31] = acc[r];
    }
}

int main() {
    float h1[32], h0[32], hp[1024], *d1, *d0, *dp;
    hipMalloc(&d1, 128); hipMalloc(&d0, 128); hipMalloc(&dp, 4096);
    double worst = 0; long nbad = 0, ntot = 0, hist[9] = {0};
    srand(1);
    for (int trial = 0; trial < 200; ++trial) {
        for (int i = 0; i < 32; ++i) {
            h1[i] = ldexpf((float)rand() / RAND_MAX + 0.5f, rand() % 40 - 30);
            h0[i] = ldexpf((float)rand() / RAND_MAX + 0.5f, rand() % 30 - 20) * ((rand() & 1) ? -1.f : 1.f);
            if (trial == 0 && i < 4) { h1[i] = i == 0 ? 0.f : h1[i]; h0[i] = i == 1 ? 0.f : h0[i]; }
        }
        hipMemcpy(d1, h1, 128, hipMemcpyHostToDevice); hipMemcpy(d0, h0, 128, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d1, d0, dp); hipMemcpy(hp, dp, 4096, hipMemcpyDeviceToHost);
        for (int n = 0; n < 32; ++n) for (int m = 0; m < 32; ++m) {
            const float want = h1[n] * h0[m], got = hp[n * 32 + m];
            unsigned uw, ug; memcpy(&uw, &want, 4); memcpy(&ug, &got, 4);
            if (want == 0.0f) { if (got != 0.0f) printf("zero product came out %g\n", got); continue; }
            const double ulps = fabs((double)(int)(uw - ug));
            ++ntot; if (ulps > 0) ++nbad; if (ulps > worst) worst = ulps;
            if (ulps < 8) ++hist[(int)ulps]; else ++hist[8];
        }
    }
    printf("outer product by bf16 pieces: %ld of %ld differ from the rounded f32 product, worst %.0f ulp\n", nbad, ntot, worst);
    for (int i = 0; i < 9; ++i) printf("  %s%d ulp: %ld\n", i == 8 ? ">=" : "", i, hist[i]);
    return 0;
}
