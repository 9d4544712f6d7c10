// k3m_clock.hip -- developer tool: the clock and the board power the shipped headline kernel (gemm_qmx.hip K3m, Q4_0 4096^3 form) runs at.
// The kernel is compiled in with -DK3M_CLOCK: every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its K loop
// into a buffer nothing else reads; in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over all waves of the last launch
// after >= SECONDS of back-to-back launches (MI355X_MICROARCH.md, DVFS give-back item 6).  A host thread samples the hwmon power and
// sclk files of the card every 10 ms meanwhile.  Operands: random bf6 codes / random scales in [0.5, 1) ("random"), or all zero ("zero").
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DK3M_CLOCK -I ggmlsharp_amd/csrc -o tools/bin/k3m_clock tools/k3m_clock.hip -lpthread
//   run:   tools/bin/k3m_clock [seconds] [M K N]
#include "../ggmlsharp_amd/csrc/gemm_qmx.hip"
#include "../ggmlsharp_amd/csrc/plan.cpp"      // (r4: the launchers consume the plan of the product)
// (K3p lives in gemm_qmp.hip, which this tool does not build: the shapes it clocks are served by the staged forms)
hipError_t launch_gemm_qmx_mid(const ggml_hip_weight *, const mm_plan &, act_planes, int64_t, float *, int64_t, hipStream_t, const mm_epilogue &) { return hipErrorNotSupported; }
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cctype>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static std::vector<std::string> find(const char *pat) {
    std::vector<std::string> r; glob_t g;
    if (glob(pat, 0, nullptr, &g) == 0) for (size_t i = 0; i < g.gl_pathc; ++i) r.push_back(g.gl_pathv[i]);
    globfree(&g);
    return r;
}
static double read_num(const std::string &f) {
    FILE *fp = fopen(f.c_str(), "r"); if (!fp) return -1; double v = -1; if (fscanf(fp, "%lf", &v) != 1) v = -1; fclose(fp); return v;
}
static std::string read_txt(const std::string &f) {
    FILE *fp = fopen(f.c_str(), "r"); if (!fp) return ""; char b[512]; size_t n = fread(b, 1, sizeof(b) - 1, fp); b[n] = 0; fclose(fp); return b;
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int M = argc > 4 ? atoi(argv[2]) : 4096, K = argc > 4 ? atoi(argv[3]) : 4096, N = argc > 4 ? atoi(argv[4]) : 4096;
    const int nbk = K / 32, nbkp = (int)pad_kblocks(nbk), Mpad = (int)pad_rows(M), Npad = (int)pad_act(N);
    const size_t wa = (size_t)(nbkp + K_LOOKAHEAD) * Mpad * 16, wb = wa / 2, wdb = (size_t)(nbkp + K_LOOKAHEAD) * Mpad * 4;
    const size_t ab = (size_t)nbkp * 48 * Npad, adb = (size_t)nbkp * Npad * 4;
    ggml_hip_weight w; memset(&w, 0, sizeof(w));
    w.type = GGML_TYPE_Q4_0; w.M = M; w.K = K; w.Mpad = Mpad; w.nbk = nbk;
    uint8_t *a6; float *ad, *dst;
    CK(hipMalloc(&w.q6a, wa)); CK(hipMalloc(&w.q6b, wb)); CK(hipMalloc(&w.d, wdb));
    CK(hipMalloc(&a6, ab)); CK(hipMalloc(&ad, adb)); CK(hipMalloc(&dst, (size_t)N * M * 4));
    act_planes p; p.a8 = (int8_t *)a6; p.ad = ad; p.as = (int32_t *)ad; p.Npad = Npad;
    // hwmon files of THIS device's card: the host shows all its GPUs in sysfs (other tenants' load included), so match the PCI address
    char bus[64] = {0};
    CK(hipDeviceGetPCIBusId(bus, sizeof(bus), 0));
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    std::string card;
    for (auto &d : find("/sys/class/drm/card*/device")) {
        char real[512]; if (!realpath(d.c_str(), real)) continue;
        if (strstr(real, bus)) { card = d; break; }
    }
    printf("device 0 = PCI %s = %s\n", bus, card.empty() ? "(no matching card: summing every card)" : card.c_str());
    const std::string base = card.empty() ? std::string("/sys/class/drm/card*/device") : card;
    std::vector<std::string> pw = find((base + "/hwmon/hwmon*/power1_average").c_str());
    if (pw.empty()) pw = find((base + "/hwmon/hwmon*/power1_input").c_str());
    std::vector<std::string> fq = find((base + "/hwmon/hwmon*/freq1_input").c_str());
    std::vector<std::string> cap = find((base + "/hwmon/hwmon*/power1_cap").c_str());
    printf("hwmon: %zu power file(s), %zu sclk file(s); power1_cap %s\n", pw.size(), fq.size(), cap.empty() ? "n/a" : std::to_string(read_num(cap[0]) / 1e6).c_str());
    for (const char *mode : {"zero", "random", "zero", "random"}) {
        const bool rnd = mode[0] == 'r';
        if (rnd) {
            std::vector<uint8_t> h(std::max(wa, ab)); uint32_t s = 12345;
            auto fill = [&](void *d, size_t n) { for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)(s >> 24); } CK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice)); };
            fill(w.q6a, wa); fill(w.q6b, wb); fill(a6, ab);
            std::vector<float> f(std::max(wdb, adb) / 4);
            auto fillf = [&](void *d, size_t n) { for (size_t i = 0; i < n / 4; ++i) { s = s * 1664525u + 1013904223u; f[i] = 0.5f + (float)(s >> 8) / 33554432.0f; } CK(hipMemcpy(d, f.data(), n, hipMemcpyHostToDevice)); };
            fillf(w.d, wdb); fillf(ad, adb);
        } else {
            CK(hipMemset(w.q6a, 0, wa)); CK(hipMemset(w.q6b, 0, wb)); CK(hipMemset(w.d, 0, wdb)); CK(hipMemset(a6, 0, ab)); CK(hipMemset(ad, 0, adb));
        }
        std::atomic<bool> stop{false};
        std::vector<double> Pw, Fq;
        std::thread sampler([&] {
            while (!stop.load()) {
                double pt = 0; for (auto &f : pw) { const double v = read_num(f); if (v > 0) pt += v / 1e6; }
                double fm = 0; for (auto &f : fq) fm = std::max(fm, read_num(f) / 1e6);
                Pw.push_back(pt); Fq.push_back(fm);
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        });
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const auto t0 = std::chrono::steady_clock::now();
        int launches = 0; float last_ms = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) CK(launch_gemm_qmx(&w, plan_mul_mat(w.type, 0, w.M, w.K, N, false), p, N, dst, M, 0, nullptr));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&last_ms, e0, e1)); launches += 50;
        }
        stop.store(true); sampler.join();
        std::vector<unsigned long long> t(8192 * 2);
        CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(k3m_clock_buf), t.size() * 8));
        std::vector<double> ghz, loop_us;
        for (int i = 0; i < 8192; ++i) if (t[2 * i + 1] > 0) { ghz.push_back((double)t[2 * i] / (double)t[2 * i + 1] * 0.1); loop_us.push_back(t[2 * i + 1] * 0.01); }
        std::sort(ghz.begin(), ghz.end()); std::sort(loop_us.begin(), loop_us.end());
        auto med = [](std::vector<double> &v) { return v.empty() ? -1.0 : v[v.size() / 2]; };
        // power / sclk over the second half of the run (steady state)
        std::vector<double> P2(Pw.begin() + Pw.size() / 2, Pw.end()), F2(Fq.begin() + Fq.size() / 2, Fq.end());
        std::sort(P2.begin(), P2.end()); std::sort(F2.begin(), F2.end());
        printf("%-6s M %d K %d N %d: %d launches over %.1f s, last 50 launches %.1f us each; in-kernel clock (median | p10 | p90 over %zu waves) %.3f | %.3f | %.3f GHz, "
               "K loop %.1f us; board power median %.0f W (min %.0f, max %.0f; %zu samples), hwmon sclk median %.0f MHz\n",
               mode, M, K, N, launches, seconds, last_ms * 1e3 / 50, ghz.size(), med(ghz), ghz.empty() ? -1 : ghz[ghz.size() / 10], ghz.empty() ? -1 : ghz[ghz.size() * 9 / 10],
               med(loop_us), med(P2), P2.empty() ? -1 : P2.front(), P2.empty() ? -1 : P2.back(), P2.size(), med(F2));
        if (rnd) {                     // (r4) ... and WHERE the slow ones ran: K loop by XCD, by wave slot of the SIMD (0 = the older workgroup of the CU), by SIMD
            std::vector<unsigned> wh(8192 * 2);
            CK(hipMemcpyFromSymbol(wh.data(), HIP_SYMBOL(k3m_where_buf), wh.size() * 4));
            auto report = [&](const char *what, int nkeys, auto key) {
                printf("       K loop median by %s:", what);
                for (int k = 0; k < nkeys; ++k) {
                    std::vector<double> v;
                    for (int i = 0; i < 8192; ++i) if (t[2 * i + 1] > 0 && key(wh[2 * i], wh[2 * i + 1]) == k) v.push_back(t[2 * i + 1] * 0.01);
                    std::sort(v.begin(), v.end());
                    if (!v.empty()) printf("  %d: %.1f (%zu)", k, v[v.size() / 2], v.size());
                }
                printf("\n");
            };
            report("XCD", 8, [](unsigned, unsigned xcc) { return (int)(xcc & 15); });
            printf("       in-kernel clock median by XCD (GHz):");
            for (int k = 0; k < 8; ++k) {
                std::vector<double> v;
                for (int i = 0; i < 8192; ++i) if (t[2 * i + 1] > 0 && (int)(wh[2 * i + 1] & 15) == k) v.push_back((double)t[2 * i] / (double)t[2 * i + 1] * 0.1);
                std::sort(v.begin(), v.end());
                if (!v.empty()) printf("  %d: %.3f", k, v[v.size() / 2]);
            }
            printf("\n");
            report("wave slot", 4, [](unsigned hw, unsigned) { return (int)(hw & 15); });
            report("SIMD", 4, [](unsigned hw, unsigned) { return (int)((hw >> 4) & 3); });
            report("shader engine", 8, [](unsigned hw, unsigned) { return (int)((hw >> 13) & 7); });
            report("CU in its array", 16, [](unsigned hw, unsigned) { return (int)((hw >> 8) & 15); });
        }
        if (!loop_us.empty())          // (r4) how far apart the waves finish their K loops: two workgroups share a CU, one wave of each per SIMD
            printf("       K loop per wave: min %.1f  p10 %.1f  p25 %.1f  median %.1f  p75 %.1f  p90 %.1f  max %.1f us\n", loop_us.front(), loop_us[loop_us.size() / 10],
                   loop_us[loop_us.size() / 4], med(loop_us), loop_us[loop_us.size() * 3 / 4], loop_us[loop_us.size() * 9 / 10], loop_us.back());
    }
    for (auto &f : find((base + "/pp_dpm_sclk").c_str())) printf("pp_dpm_sclk (%s) after the run:\n%s", f.c_str(), read_txt(f).c_str());
    return 0;
}
