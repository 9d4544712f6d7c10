// k3p_trace.hip -- developer tool: where does the time of the prompt-sized MX form (gemm_qmp.hip K3p) go?  The kernel is compiled in
// with -DK3P_TRACE: every wave stamps s_memrealtime (100 MHz) at kernel entry, K loop start, K loop end, after the first barrier, after
// the barrier in front of the second reduction round and at its end, plus the shader cycles (s_memtime) of its K loop; operands are
// random bf6 codes / scales.  Reported per wave index 0..7: median over workgroups, in us after the launch's first stamp.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DK3P_TRACE -I ggmlsharp_amd/csrc -o tools/bin/k3p_trace tools/k3p_trace.hip
//   run:   tools/bin/k3p_trace [M K N]
#include "../ggmlsharp_amd/csrc/gemm_qmp.hip"
#include "../ggmlsharp_amd/csrc/plan.cpp"      // (r4: the launchers consume the plan of the product)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 4096, K = argc > 3 ? atoi(argv[2]) : 4096, N = argc > 3 ? atoi(argv[3]) : 512;
    const bool q51 = argc > 4 && !strcmp(argv[4], "q51");    // Q5_1 on the int8 form: the min-term product on the bf16 cores in front of the K loop
    const bool i8 = q51 || (argc > 4 && !strcmp(argv[4], "i8"));      // Q8_0 on the int8 form instead of Q4_0 on the MX form
    const int nbk = K / 32, nbkp = (int)pad_kblocks(nbk), Mpad = (int)pad_rows(M), Npad = (int)pad_act(N);
    const size_t wa = (size_t)(nbkp + K_LOOKAHEAD) * Mpad * 16, wb = wa / 2, wdb = (size_t)(nbkp + K_LOOKAHEAD) * Mpad * 4;
    const size_t ab = (size_t)nbkp * 48 * Npad, adb = (size_t)nbkp * Npad * 4;
    const int copies = argc > 5 ? atoi(argv[5]) : 16;         // (1: the weights stay in the caches from launch to launch)
    std::vector<ggml_hip_weight> W(copies);
    std::vector<uint8_t> h(std::max(2 * wa, ab)); uint32_t s = 12345;
    auto fill = [&](void *d, size_t n) { for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(s >> 24); } CK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice)); };
    std::vector<float> f(std::max(wdb, adb) / 4);
    auto fillf = [&](void *d, size_t n) { for (size_t i = 0; i < n / 4; ++i) { s = s * 1664525u + 1013904223u; f[i] = 0.5f + (float)(s >> 8) / 33554432.0f; } CK(hipMemcpy(d, f.data(), n, hipMemcpyHostToDevice)); };
    for (auto &w : W) {
        memset(&w, 0, sizeof(w));
        w.type = q51 ? GGML_TYPE_Q5_1 : i8 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q4_0; w.M = M; w.K = K; w.Mpad = Mpad; w.nbk = nbk;
        CK(hipMalloc(&w.q6a, 2 * wa)); CK(hipMalloc(&w.q6b, wb)); CK(hipMalloc(&w.d, wdb));
        fill(w.q6a, 2 * wa); fill(w.q6b, wb); fillf(w.d, wdb);
        w.qs = w.q6a;                                         // (int8 planes: 32 B per row and k-block)
        if (q51) {                                            // (piece planes: random finite bf16 values)
            const size_t mpb = (size_t)((nbk + 15) / 16 * 2 * 3) * Mpad * 16;
            w.i8p = w.q6a; w.m = w.d;
            CK(hipMalloc(&w.mp3, mpb));
            std::vector<uint16_t> hp(mpb / 2);
            for (auto &v : hp) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3C00u | ((s >> 20) & 0x83FFu)); }
            CK(hipMemcpy(w.mp3, hp.data(), mpb, hipMemcpyHostToDevice));
        }
    }
    uint8_t *a6; float *ad, *dst;
    CK(hipMalloc(&a6, ab)); CK(hipMalloc(&ad, adb)); CK(hipMalloc(&dst, (size_t)N * M * 4));
    fill(a6, ab); fillf(ad, adb);
    act_planes p; p.a8 = (int8_t *)a6; p.ad = ad; p.as = (int32_t *)ad; p.Npad = Npad; p.sp3 = nullptr;
    if (q51) {
        const size_t spb = (size_t)((nbk + 15) / 16 * 2 * 3) * Npad * 16;
        CK(hipMalloc(&p.sp3, spb));
        std::vector<uint16_t> hp(spb / 2);
        for (auto &v : hp) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3C00u | ((s >> 20) & 0x83FFu)); }
        CK(hipMemcpy(p.sp3, hp.data(), spb, hipMemcpyHostToDevice));
    }
    const mm_epilogue ep{0, nullptr, 0, nullptr, 0, 1.0f};
    const mm_plan pl = plan_mul_mat(q51 ? GGML_TYPE_Q5_1 : i8 ? GGML_TYPE_Q8_0 : GGML_TYPE_Q4_0, 0, M, K, N, false);
    if (pl.family != (i8 ? MMF_K3P_I8 : MMF_K3P_MX)) { printf("this shape is not served by K3p (plan family %d)\n", pl.family); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 200; ++it) CK((i8 ? launch_gemm_q8_mid : launch_gemm_qmx_mid)(&W[it % copies], pl, p, N, dst, M, 0, ep));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int iters = 2000;
    for (int it = 0; it < iters; ++it) CK((i8 ? launch_gemm_q8_mid : launch_gemm_qmx_mid)(&W[it % copies], pl, p, N, dst, M, 0, ep));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nwg = ((M + 127) / 128) * ((N + 63) / 64);
    printf("M %d K %d N %d: %d workgroups of 8 waves; back to back over %d weight copies: %.2f us per launch\n", M, K, N, nwg, copies, ms * 1e3 / iters);
    std::vector<unsigned long long> t((size_t)4096 * 8);
    CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(k3p_trace_buf), t.size() * 8));
    const int nw = std::min(nwg * 8, 4096);
    unsigned long long first = ~0ull;
    for (int i = 0; i < nw; ++i) first = std::min(first, t[(size_t)i * 8]);
    const char *names[6] = {"wave start", "table in LDS, first loads issued", "K loop done", "first barrier passed", "round 1 done, barrier passed", "stored"};
    printf("%-36s", "us after the launch's first stamp");
    for (int wv = 0; wv < 8; ++wv) printf("  wave %d", wv);
    printf("\n");
    for (int k = 0; k < 6; ++k) {
        printf("%-36s", names[k]);
        for (int wv = 0; wv < 8; ++wv) {
            std::vector<double> v;
            for (int g = 0; g * 8 + wv < nw; ++g) v.push_back((double)(t[((size_t)g * 8 + wv) * 8 + k] - first) * 0.01);
            std::sort(v.begin(), v.end());
            printf(" %7.2f", v[v.size() / 2]);
        }
        printf("\n");
    }
    {
        std::vector<double> ghz;
        for (int i = 0; i < nw; ++i) { const double us = (double)(t[(size_t)i * 8 + 2] - t[(size_t)i * 8 + 1]) * 0.01; if (us > 0) ghz.push_back((double)t[(size_t)i * 8 + 7] / us * 1e-3); }
        std::sort(ghz.begin(), ghz.end());
        printf("in-kernel clock over the K loop (median over waves): %.3f GHz\n", ghz[ghz.size() / 2]);
        std::vector<double> last;
        for (int g = 0; g * 8 < nw; ++g) { double mx = 0; for (int wv = 0; wv < 8; ++wv) mx = std::max(mx, (double)(t[((size_t)g * 8 + wv) * 8 + 5] - first) * 0.01); last.push_back(mx); }
        std::sort(last.begin(), last.end());
        printf("workgroup end (median | max over workgroups): %.2f | %.2f us\n", last[last.size() / 2], last.back());
    }
    return 0;
}
