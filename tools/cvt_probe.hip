// Developer probe (not product): semantics of v_cvt_scalef32_2xpk16_bf6_f32 on gfx950 -- element order inside the 192-bit
// result, the role of the scale operand, and that every integer in [-8, 8] gets the e3m2 code K1 / layout.hip emit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const float *x, unsigned *o, float sc) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) { a[i] = x[i]; b[i] = x[16 + i]; }
    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, sc);
    for (int i = 0; i < 6; ++i) o[i] = r[i];
}
static unsigned code(int v) { const unsigned t[9] = {0, 12, 16, 18, 20, 21, 22, 23, 24}; return t[v < 0 ? -v : v] | (v < 0 ? 32u : 0u); }
int main() {
    float *dx; unsigned *d_o; hipMalloc(&dx, 128); hipMalloc(&d_o, 24);
    float hx[32]; unsigned ho[6];
    const float scales[3] = {1.0f, 16.0f, 0.0625f};
    for (int pat = 0; pat < 3; ++pat)
        for (int s = 0; s < 3; ++s) {
            for (int e = 0; e < 32; ++e) hx[e] = pat == 0 ? (float)((e % 17) - 8) : pat == 1 ? (float)(e == 5 ? 3 : 0) : (float)(((e * 7) % 17) - 8);
            if (s == 1) for (int e = 0; e < 32; ++e) hx[e] *= 16.0f;
            if (s == 2) for (int e = 0; e < 32; ++e) hx[e] *= 0.0625f;
            hipMemcpy(dx, hx, 128, hipMemcpyHostToDevice);
            k<<<1, 1>>>(dx, d_o, scales[s]); hipMemcpy(ho, d_o, 24, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int e = 0; e < 32; ++e) {
                const int bit = 6 * e; unsigned c = (ho[bit >> 5] >> (bit & 31)); if ((bit & 31) > 26) c |= ho[(bit >> 5) + 1] << (32 - (bit & 31));
                c &= 63u;
                const int v = (int)(hx[e] / scales[s]);
                if (c != code(v)) { if (bad < 4) printf("  pat %d scale %g e %d value %d: got code %u want %u\n", pat, scales[s], e, v, c, code(v)); ++bad; }
            }
            printf("pattern %d scale %g: %d mismatches  words %08x %08x %08x %08x %08x %08x\n", pat, scales[s], bad, ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
        }
    return 0;
}
