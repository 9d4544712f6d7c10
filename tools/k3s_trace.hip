// k3s_trace.hip -- developer tool: where does the time of the small-batch MX form (gemm_qmx.hip K3s) go?  The kernel is compiled
// in with time stamps (s_memrealtime, 100 MHz) at its phase ends -- wave start, loads issued + scale table in LDS, last pair done,
// barrier passed, partial sums exchanged, summed, stored -- and run on zero-filled planes of the given shape, rotating over copies
// so that neither L2 nor the Infinity Cache holds the weights.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DK3S_TRACE -I ggmlsharp_amd/csrc -o tools/bin/k3s_trace tools/k3s_trace.hip
//   run:   tools/bin/k3s_trace [M K]
#include "../ggmlsharp_amd/csrc/gemm_qmx.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int M = argc > 2 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096, N = 32, Npad = 32, KS = 8;
    const int nbk = K / 32, nbkp = (nbk + 3) / 4 * 4, Mpad = (M + 255) / 256 * 256;
    int nloc = (nbkp + KS - 1) / KS; nloc += nloc & 1;
    const size_t wa = (size_t)(nbkp + 2) * Mpad * 16, wb = wa / 2, wdb = (size_t)(nbkp + 2) * Mpad * 4;
    const size_t ab = (size_t)nbkp * 48 * Npad, adb = (size_t)nbkp * Npad * 4;
    const int copies = 24;
    std::vector<uint8_t *> A(copies), B(copies); std::vector<float *> D(copies);
    for (int c = 0; c < copies; ++c) {
        CK(hipMalloc(&A[c], wa)); CK(hipMalloc(&B[c], wb)); CK(hipMalloc(&D[c], wdb));
        CK(hipMemset(A[c], 0, wa)); CK(hipMemset(B[c], 0, wb)); CK(hipMemset(D[c], 0, wdb));
    }
    uint8_t *a6; float *ad, *dst;
    CK(hipMalloc(&a6, ab)); CK(hipMalloc(&ad, adb)); CK(hipMalloc(&dst, (size_t)N * M * 4));
    CK(hipMemset(a6, 0, ab)); CK(hipMemset(ad, 0, adb));
    const mm_epilogue ep{0, nullptr, 0, nullptr, 0, 1.0f};
    const int rows = nloc <= 8 ? 8 : nloc <= 16 ? 16 : nloc;
    const int tab = KS * rows * 32 * 4, xch = KS * 4 * 16 * 64 * 4, lds = std::max(tab, xch);
    const int t32 = (M + 31) / 32;
    const int wmt = getenv("K3S_TILES") ? atoi(getenv("K3S_TILES")) : (t32 <= 256 ? 1 : t32 <= 512 ? 2 : 4);   // tiles per workgroup (1 | 2 | 4)
    const bool two = wmt == 2;
    const int grid = (M + 32 * wmt - 1) / (32 * wmt);
    auto go = [&](int c) {
#define GO(NP, ROT, WMT) do { auto kern = gemm_qmx_small_kernel<GGML_TYPE_Q4_0, 8, NP, ROT, WMT>; \
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        kern<<<grid, KS * 64, lds>>>(A[c], B[c], D[c], D[c], a6, ad, ad, dst, M, N, Mpad, Npad, nbkp, nloc, M, (uint32_t)wa, (uint32_t)wdb, (uint32_t)ab, (uint32_t)adb, ep, grid); } while (0)
        if (wmt == 4) GO(2, true, 4); else if (two) { if (nloc <= 8) GO(4, false, 2); else GO(4, true, 2); } else if (nloc <= 8) GO(4, false, 1); else if (nloc <= 16) GO(8, false, 1); else GO(8, true, 1);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int c = 0; c < copies; ++c) go(c);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 0; it < 4; ++it) for (int c = 0; c < copies; ++c) go(c);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("M %d K %d N %d: %d workgroups of %d waves, %d k-blocks per wave; back to back over %d weight copies: %.2f us per launch\n", M, K, N, grid, KS, nloc, copies,
           ms * 1e3 / (4 * copies));
    std::vector<unsigned long long> t((size_t)2048 * 2 * 8);
    CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(k3s_trace_buf), t.size() * 8));
    const int ng = std::min(grid, 2048);
    unsigned long long first = ~0ull;
    for (int g = 0; g < ng; ++g) for (int w = 0; w < 2; ++w) first = std::min(first, t[((size_t)g * 2 + w) * 8]);
    const char *names[7] = {"wave start", "loads issued, table in LDS", "last pair done", "barrier passed", "sums exchanged", "", "summed and stored"};
    for (int w = 0; w < 2; ++w) {
        printf("  wave %d of the workgroups, us after the first wave of the launch started (median | max over workgroups):\n", w ? KS - 1 : 0);
        for (int k = 0; k < 7; ++k) {
            if (k == 5) continue;
            std::vector<double> v;
            for (int g = 0; g < ng; ++g) v.push_back((double)(t[((size_t)g * 2 + w) * 8 + k] - first) * 0.01);
            std::sort(v.begin(), v.end());
            printf("    %-28s %7.2f | %7.2f\n", names[k], v[v.size() / 2], v.back());
        }
    }
    return 0;
}
