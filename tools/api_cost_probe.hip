// api_cost_probe.hip -- developer probe: host-side cost of the runtime calls the Seam-1 pipeline makes per chunk (seams.cpp issue_chunks):
// hipMemcpyAsync (pinned, 2 MB, each way), hipEventRecord, hipStreamWaitEvent, a small kernel launch.  Enqueue time on the calling thread, us.
//   build: hipcc --offload-arch=gfx950 -O2 tools/api_cost_probe.hip -o tools/bin/api_cost_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 2u << 20;
    void *h; float *d; CK(hipHostMalloc(&h, 4 * n, hipHostMallocDefault)); CK(hipMalloc((void **)&d, 4 * n));
    hipStream_t s1, s2, s3; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(64); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int R = 200;
    auto report = [&](const char *what, double t) { printf("%-44s %7.2f us per call\n", what, t / R); };
    CK(hipDeviceSynchronize());
    double t0 = now(); for (int i = 0; i < R; ++i) CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1)); double t1 = now(); CK(hipDeviceSynchronize()); report("hipMemcpyAsync H2D 2 MB pinned (enqueue)", t1 - t0);
    t0 = now(); for (int i = 0; i < R; ++i) CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s2)); t1 = now(); CK(hipDeviceSynchronize()); report("hipMemcpyAsync D2H 2 MB pinned (enqueue)", t1 - t0);
    t0 = now(); for (int i = 0; i < R; ++i) CK(hipEventRecord(ev[i % 64], s1)); t1 = now(); CK(hipDeviceSynchronize()); report("hipEventRecord", t1 - t0);
    t0 = now(); for (int i = 0; i < R; ++i) CK(hipStreamWaitEvent(s3, ev[i % 64], 0)); t1 = now(); CK(hipDeviceSynchronize()); report("hipStreamWaitEvent", t1 - t0);
    t0 = now(); for (int i = 0; i < R; ++i) tiny<<<1, 64, 0, s3>>>(d); t1 = now(); CK(hipDeviceSynchronize()); report("kernel launch (1 block)", t1 - t0);
    // a whole chunk as the pipeline issues it today (8 calls, 3 streams) and as ONE stream would (4 calls)
    t0 = now();
    for (int i = 0; i < R; ++i) {
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1)); CK(hipEventRecord(ev[0], s1)); CK(hipStreamWaitEvent(s3, ev[0], 0));
        tiny<<<1, 64, 0, s3>>>(d); tiny<<<1, 64, 0, s3>>>(d);
        CK(hipEventRecord(ev[1], s3)); CK(hipStreamWaitEvent(s2, ev[1], 0)); CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s2));
    }
    t1 = now(); CK(hipDeviceSynchronize()); double t2 = now(); report("chunk as today: 8 calls over 3 streams (enqueue)", t1 - t0); printf("   ... wall per chunk incl. execution %7.2f us\n", (t2 - t0) / R);
    t0 = now();
    for (int i = 0; i < R; ++i) {
        hipStream_t s = (i & 1) ? s1 : s2;
        CK(hipMemcpyAsync(d + (i & 1) * (n / 8), (char *)h + (i & 1) * (n / 2), n / 2, hipMemcpyHostToDevice, s)); tiny<<<1, 64, 0, s>>>(d); tiny<<<1, 64, 0, s>>>(d);
        CK(hipMemcpyAsync((char *)h + 2 * n + (i & 1) * (n / 2), d + (i & 1) * (n / 8), n / 2, hipMemcpyDeviceToHost, s));
    }
    t1 = now(); CK(hipDeviceSynchronize()); t2 = now(); report("chunk on one of two streams: 4 calls (enqueue)", t1 - t0); printf("   ... wall per chunk incl. execution %7.2f us (1 MB each way per chunk)\n", (t2 - t0) / R);
    return 0;
}
