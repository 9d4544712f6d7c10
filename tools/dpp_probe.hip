// dpp_probe.hip -- developer tool: semantics of DPP row_newbcast on gfx950 (v_mul_f32_dpp vdst, src0, src1 row_newbcast:r reads
// src0 from lane r of the reading lane's own row of 16 lanes).  The mat-mat kernels use it to take the 16 activation-row scales of a
// 32x32 accumulator tile from ONE register (lane 16 g + j holds the scale of accumulator register j for lane half g >> 1).
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/dpp_probe tools/dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int R>
__device__ __forceinline__ float bmul(float s, float x) {
    float r;
    asm volatile("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(s), "v"(x), "n"(R));
    return r;
}

__global__ void probe(float *out) {
    const float s = (float)threadIdx.x, x = 2.0f;
    float r[16];
    r[0] = bmul<0>(s, x); r[1] = bmul<1>(s, x); r[2] = bmul<2>(s, x); r[3] = bmul<3>(s, x);
    r[4] = bmul<4>(s, x); r[5] = bmul<5>(s, x); r[6] = bmul<6>(s, x); r[7] = bmul<7>(s, x);
    r[8] = bmul<8>(s, x); r[9] = bmul<9>(s, x); r[10] = bmul<10>(s, x); r[11] = bmul<11>(s, x);
    r[12] = bmul<12>(s, x); r[13] = bmul<13>(s, x); r[14] = bmul<14>(s, x); r[15] = bmul<15>(s, x);
    for (int i = 0; i < 16; ++i) out[i * 64 + threadIdx.x] = r[i];
}

int main() {
    float *d, h[16 * 64];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int l = 0; l < 64; ++l)
            if (h[i * 64 + l] != 2.0f * (float)((l & ~15) + i)) { if (bad < 8) printf("r %d lane %d: got %g want %g\n", i, l, h[i * 64 + l], 2.0f * ((l & ~15) + i)); ++bad; }
    printf("row_newbcast probe: %d mismatches of 1024\n", bad);
    return bad != 0;
}
