// mx16_probe.hip -- developer probe (not part of the product): operand and scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 with bf6 operands,
// found with exact integer data.  build: hipcc --offload-arch=gfx950 -O2 tools/mx16_probe.hip -o tools/bin/mx16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__host__ __device__ inline uint32_t bf6_of_int(int v) {          // |v| <= 8
    const uint32_t tab[9] = {0, 12, 16, 18, 20, 21, 22, 23, 24};
    return (v < 0 ? 32u : 0u) | tab[v < 0 ? -v : v];
}
// raw fragments: FA[lane][32 elements], FB[lane][32 elements] as small ints; scales per lane; D[16][16] out (row = 4*(l>>4)+r, col = l&15)
__global__ void probe(const int *FA, const int *FB, const int *sa, const int *sb, float *D) {
    const int l = threadIdx.x;
    uint32_t fa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 32; ++e) {
        const uint32_t ca = bf6_of_int(FA[l * 32 + e]), cb = bf6_of_int(FB[l * 32 + e]);
        const int bit = 6 * e, w = bit >> 5, s = bit & 31;
        fa[w] |= ca << s; fb[w] |= cb << s;
        if (s > 26) { fa[w + 1] |= ca >> (32 - s); fb[w + 1] |= cb >> (32 - s); }
    }
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)fa[i]; b[i] = (int)fb[i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 3, 3, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

int main() {
    std::vector<int> FA(64 * 32), FB(64 * 32), sa(64), sb(64);
    int *dA, *dB, *dsa, *dsb; float *dD;
    hipMalloc(&dA, FA.size() * 4); hipMalloc(&dB, FB.size() * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 256 * 4);
    std::vector<float> D(256);
    auto run = [&]() {
        hipMemcpy(dA, FA.data(), FA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, FB.data(), FB.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dsa, dsb, dD);
        hipDeviceSynchronize();
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    };
    // test 1: hypothesis H1 (lane l: row / col l & 15, k = 32 (l >> 4) + e), per-group scales 2^0, 2^4, 2^8, 2^12 on A
    {
        std::vector<int> A(16 * 128), Bt(16 * 128);
        srand(5);
        for (auto &v : A) v = rand() % 17 - 8;
        for (auto &v : Bt) v = rand() % 16 - 8;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 32; ++e) { FA[l * 32 + e] = A[(l & 15) * 128 + 32 * (l >> 4) + e]; FB[l * 32 + e] = Bt[(l & 15) * 128 + 32 * (l >> 4) + e]; }
        for (int l = 0; l < 64; ++l) { sa[l] = 127 + 4 * (l >> 4); sb[l] = 127; }
        run();
        int bad = 0, bad_plain = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0, p = 0;
                for (int g = 0; g < 4; ++g) {
                    long sg = 0;
                    for (int k = 0; k < 32; ++k) sg += (long)A[i * 128 + 32 * g + k] * Bt[j * 128 + 32 * g + k];
                    s += (double)sg * (double)(1 << (4 * g)); p += (double)sg;
                }
                if (D[i * 16 + j] != (float)s) ++bad;
                if (D[i * 16 + j] != (float)p) ++bad_plain;
            }
        printf("test 1 (H1, A scales 2^(4g) by lane group): mismatches %d  (vs unscaled sum: %d)  D[0][0..3] = %g %g %g %g\n", bad, bad_plain, D[0], D[1], D[2], D[3]);
    }
    // test 2: which (lane group, element half) of A pairs with which of B: ones in one (group, half) of each, D[0][0] = number of common k
    printf("test 2: rows = A (lane group, element half 0: e < 16 / 1: e >= 16), columns = B (same): common k count\n");
    for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
    for (int ga = 0; ga < 4; ++ga)
        for (int ha = 0; ha < 2; ++ha) {
            printf("  A(g%d,h%d):", ga, ha);
            for (int gb = 0; gb < 4; ++gb)
                for (int hb = 0; hb < 2; ++hb) {
                    for (auto &v : FA) v = 0;
                    for (auto &v : FB) v = 0;
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 32; ++e) {
                            if ((l >> 4) == ga && (e >> 4) == ha) FA[l * 32 + e] = 1;
                            if ((l >> 4) == gb && (e >> 4) == hb) FB[l * 32 + e] = 1;
                        }
                    run();
                    printf(" %3g", D[0]);
                }
            printf("\n");
        }
    // test 3: whose scale applies to which (lane group, element half) of A: all B ones; A ones in one (group, half); scale 2^4 in ONE lane group of A
    printf("test 3: rows = A ones in (lane group, element half), columns = lane group whose A scale is 2^4: D[0][0] (16 = unscaled, 256 = scaled)\n");
    for (int ga = 0; ga < 4; ++ga)
        for (int ha = 0; ha < 2; ++ha) {
            printf("  A(g%d,h%d):", ga, ha);
            for (int gs = 0; gs < 4; ++gs) {
                for (auto &v : FA) v = 0;
                for (auto &v : FB) v = 1;
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 32; ++e)
                        if ((l >> 4) == ga && (e >> 4) == ha) FA[l * 32 + e] = 1;
                for (int l = 0; l < 64; ++l) { sa[l] = (l >> 4) == gs ? 131 : 127; sb[l] = 127; }
                run();
                printf(" %4g", D[0]);
            }
            printf("\n");
        }
    // test 4: the same for B's scale
    printf("test 4: rows = B ones in (lane group, element half), columns = lane group whose B scale is 2^4: D[0][0]\n");
    for (int gb = 0; gb < 4; ++gb)
        for (int hb = 0; hb < 2; ++hb) {
            printf("  B(g%d,h%d):", gb, hb);
            for (int gs = 0; gs < 4; ++gs) {
                for (auto &v : FA) v = 1;
                for (auto &v : FB) v = 0;
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 32; ++e)
                        if ((l >> 4) == gb && (e >> 4) == hb) FB[l * 32 + e] = 1;
                for (int l = 0; l < 64; ++l) { sb[l] = (l >> 4) == gs ? 131 : 127; sa[l] = 127; }
                run();
                printf(" %4g", D[0]);
            }
            printf("\n");
        }
    return 0;
}
