// mx_probe.hip -- developer probe (not part of the product): operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with
// bf6 (e3m2) operands and E8M0 block scales, checked with exact integer data against a CPU product.
// build: hipcc --offload-arch=gfx950 -O2 tools/mx_probe.hip -o tools/bin/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__host__ __device__ inline uint32_t bf6_of_int(int v) {          // |v| <= 8
    const uint32_t tab[9] = {0, 12, 16, 18, 20, 21, 22, 23, 24};
    return (v < 0 ? 32u : 0u) | tab[v < 0 ? -v : v];
}

// A: [32 rows][64 k] ints, B: [64 k][32 cols] ints (stored as Bt[col][k]); lane l holds row/col l&31, k = 32*(l>>5) .. +31,
// element e at bits [6e, 6e+5] of the lane's 192-bit fragment.  scale byte: sa[l], sb[l] per lane.
template <int CBSZ, int BLGP>
__global__ void probe(const int *A, const int *Bt, const int *sa, const int *sb, float *D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    uint32_t fa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 32; ++e) {
        const uint32_t ca = bf6_of_int(A[r * 64 + 32 * h + e]), cb = bf6_of_int(Bt[r * 64 + 32 * h + e]);
        const int bit = 6 * e, w = bit >> 5, s = bit & 31;
        fa[w] |= ca << s; fb[w] |= cb << s;
        if (s > 26) { fa[w + 1] |= ca >> (32 - s); fb[w + 1] |= cb >> (32 - s); }
    }
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)fa[i]; b[i] = (int)fb[i]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];   // D[row][col], col = lane & 31
}

int main() {
    std::vector<int> A(32 * 64), Bt(32 * 64), sa(64), sb(64);
    srand(5);
    for (auto &v : A) v = rand() % 17 - 8;
    for (auto &v : Bt) v = rand() % 16 - 8;
    for (int l = 0; l < 64; ++l) { sa[l] = l < 32 ? 131 : 127; sb[l] = 127; }   // A's K-block 0 scaled by 2^4
    int *dA, *dB, *dsa, *dsb; float *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    std::vector<float> D(32 * 32);
    auto check = [&](const char *name) {
        hipDeviceSynchronize();
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int bad_scaled = 0, bad_plain = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                long s0 = 0, s1 = 0;
                for (int k = 0; k < 32; ++k) { s0 += (long)A[i * 64 + k] * Bt[j * 64 + k]; s1 += (long)A[i * 64 + 32 + k] * Bt[j * 64 + 32 + k]; }
                if (D[i * 32 + j] != (float)(16 * s0 + s1)) ++bad_scaled;
                if (D[i * 32 + j] != (float)(s0 + s1)) ++bad_plain;
            }
        printf("%s: mismatches vs 16*blk0+blk1: %d   vs blk0+blk1: %d   D[0][0..3] = %g %g %g %g\n", name, bad_scaled, bad_plain, D[0], D[1], D[2], D[3]);
    };
    probe<3, 3><<<1, 64>>>(dA, dB, dsa, dsb, dD); check("bf6 x bf6 (cbsz 3, blgp 3)");
    probe<2, 2><<<1, 64>>>(dA, dB, dsa, dsb, dD); check("fp6 x fp6 (cbsz 2, blgp 2) [expected wrong: e2m3 codes]");
    return 0;
}
