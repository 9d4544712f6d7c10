// atomic_probe.hip -- developer tool: cost of a shared ticket counter (one returning atomicAdd per ticket, thread 0 of each
// workgroup) for dynamic tile scheduling: same-address device-scope atomics from every CU of the 8 XCDs.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/atomic_probe tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: one counter, agent scope; 1: one counter per XCD (by HW_REG_XCC_ID), agent scope; 2: per XCD, workgroup scope (L2-local)
template <int MODE>
__global__ void tickets(unsigned *ctr, int per_wg, unsigned long long *lat, unsigned *sink) {
    if (threadIdx.x != 0) return;
    unsigned xcc = 0;
    if (MODE >= 1) xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15;      // HW_REG_XCC_ID, bits [3:0]
    unsigned *p = ctr + 64 * xcc;
    unsigned acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < per_wg; ++i) {
        unsigned v = MODE == 2 ? __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                               : __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += v;                                              // dependent: the next atomic is issued after this one returned
        asm volatile("" : "+v"(acc));
    }
    lat[blockIdx.x] = wall_clock64() - t0;
    sink[blockIdx.x] = acc + xcc;
}

int main() {
    unsigned *ctr, *sink; unsigned long long *lat;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&sink, 4096 * 4)); CK(hipMalloc(&lat, 4096 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kern, int wgs, int per) {
        CK(hipMemset(ctr, 0, 4096));
        kern<<<wgs, 64>>>(ctr, per, lat, sink); CK(hipDeviceSynchronize());
        CK(hipMemset(ctr, 0, 4096));
        CK(hipEventRecord(e0)); kern<<<wgs, 64>>>(ctr, per, lat, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[4096]; unsigned c[1024];
        CK(hipMemcpy(h, lat, wgs * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, ctr, 4096, hipMemcpyDeviceToHost));
        double s = 0, mx = 0; for (int i = 0; i < wgs; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
        unsigned tot = 0; for (int x = 0; x < 16; ++x) tot += c[64 * x];
        printf("%-44s %4d workgroups x %2d tickets: kernel %7.2f us, per ticket (mean over workgroups) %6.0f ns, slowest workgroup %6.2f us, counted %u (want %d), per-XCD %u %u %u %u %u %u %u %u\n",
               name, wgs, per, ms * 1e3, s / wgs / per * 10.0, mx / 100.0, tot, wgs * per, c[0], c[64], c[128], c[192], c[256], c[320], c[384], c[448]);
    };
    for (int per : {1, 4, 8}) {
        run("one counter, agent scope", tickets<0>, 512, per);
        run("one counter per XCD, agent scope", tickets<1>, 512, per);
        run("one counter per XCD, workgroup scope (L2)", tickets<2>, 512, per);
    }
    run("one counter, agent scope", tickets<0>, 1, 64);
    return 0;
}
