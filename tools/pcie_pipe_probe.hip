// pcie_pipe_probe.hip -- developer probe: what is the floor of the Seam-1 pattern (src1 down, two kernels, dst up) at 4096 x 4096 x 512 on this box,
// with raw runtime calls and no library?  8 MB each way through a hipHostRegister'ed pool (the C# host's context pool is registered that way),
// a stand-in kernel of about the product's duration per chunk.  Wall time from the first call to the end of the last synchronize, median of 21.
//   build: hipcc --offload-arch=gfx950 -O2 tools/pcie_pipe_probe.hip -o tools/bin/pcie_pipe_probe     run: pcie_pipe_probe [spin]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(const float *__restrict__ x, float *__restrict__ y, size_t n, int spin) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.0f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a += x[i];
    for (int k = 0; k < spin; ++k) a = a * 1.0000001f + 1e-9f;
    y[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// INIT stand-in: src1 rows from the host mapping -> a device buffer; COMPUTE stand-in: device buffer -> dst rows in the host mapping (coalesced float4 stores)
__global__ void pull(const float4 *__restrict__ hx, float4 *__restrict__ dx, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dx[i] = hx[i];
}
__global__ void push(const float4 *__restrict__ dd, float4 *__restrict__ hd, size_t n4, int spin) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = dd[i];
        for (int k = 0; k < spin; ++k) v.x = v.x * 1.0000001f + 1e-9f;
        hd[i] = v;
    }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "spin")) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    const size_t N = 512, K = 4096, M = 4096, xb = N * K * 4, db = N * M * 4;
    void *pool = aligned_alloc(4096, xb + db);
    memset(pool, 1, xb + db);
    CK(hipHostRegister(pool, xb + db, hipHostRegisterPortable | hipHostRegisterMapped));
    char *hx = (char *)pool, *hd = (char *)pool + xb;
    float *dx, *dd; CK(hipMalloc((void **)&dx, xb)); CK(hipMalloc((void **)&dd, db));
    hipStream_t s[3]; for (auto &q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(64); for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto kern = [&](hipStream_t st, size_t r0, size_t n) {   // two launches per chunk, like INIT + COMPUTE (~4 + ~7 us per 128 rows)
        work<<<256, 256, 0, st>>>(dx + r0 * K, dd + r0 * M, n * K, 200);
        work<<<256, 256, 0, st>>>(dx + r0 * K, dd + r0 * M, n * K, 1500);
    };
    auto median = [&](auto &&f) { std::vector<double> t; for (int i = 0; i < 21; ++i) { (void)hipDeviceSynchronize(); double t0 = now(); f(); t.push_back(now() - t0); } std::sort(t.begin(), t.end()); return t[10]; };
    printf("%s wait; 8 MB each way, registered pool\n", argc > 1 ? "spin" : "default");
    printf("H2D alone                       %7.1f us\n", median([&] { (void)hipMemcpyAsync(dx, hx, xb, hipMemcpyHostToDevice, s[0]); (void)hipStreamSynchronize(s[0]); }));
    printf("D2H alone                       %7.1f us\n", median([&] { (void)hipMemcpyAsync(hd, dd, db, hipMemcpyDeviceToHost, s[2]); (void)hipStreamSynchronize(s[2]); }));
    printf("both at once (two streams)      %7.1f us\n", median([&] { (void)hipMemcpyAsync(dx, hx, xb, hipMemcpyHostToDevice, s[0]); (void)hipMemcpyAsync(hd, dd, db, hipMemcpyDeviceToHost, s[2]);
                                                                       (void)hipStreamSynchronize(s[0]); (void)hipStreamSynchronize(s[2]); }));
    printf("kernels of all 512 rows alone   %7.1f us\n", median([&] { kern(s[1], 0, N); (void)hipStreamSynchronize(s[1]); }));
    printf("empty synchronize               %7.1f us\n", median([&] { (void)hipStreamSynchronize(s[1]); }));
    // uneven chunks over three streams: a small first chunk starts the download early, a small last one shortens the drain
    {
        const std::vector<std::vector<int>> splits = {{128, 128, 128, 128}, {64, 128, 160, 160}, {64, 192, 192, 64}, {96, 160, 160, 96}, {32, 160, 160, 160}, {170, 171, 171}, {128, 192, 192},
                                                      {64, 64, 128, 128, 128}, {102, 102, 102, 103, 103}, {256, 128, 128}, {192, 192, 128}, {224, 160, 128}};
        for (const auto &sp : splits) {
            double t = median([&] {
                size_t r0 = 0;
                for (size_t k = 0; k < sp.size(); ++k) {
                    const size_t c = (size_t)sp[k];
                    (void)hipMemcpyAsync((char *)dx + r0 * K * 4, hx + r0 * K * 4, c * K * 4, hipMemcpyHostToDevice, s[0]);
                    (void)hipEventRecord(ev[2 * k], s[0]); (void)hipStreamWaitEvent(s[1], ev[2 * k], 0);
                    kern(s[1], r0, c);
                    (void)hipEventRecord(ev[2 * k + 1], s[1]); (void)hipStreamWaitEvent(s[2], ev[2 * k + 1], 0);
                    (void)hipMemcpyAsync(hd + r0 * M * 4, (char *)dd + r0 * M * 4, c * M * 4, hipMemcpyDeviceToHost, s[2]);
                    r0 += c;
                }
                (void)hipStreamSynchronize(s[2]);
            });
            printf("three streams, chunks of");
            for (int c : sp) printf(" %d", c);
            printf(" rows: %7.1f us\n", t);
        }
    }
    for (int nch : {1, 2, 4, 8, 16}) {
        const size_t c = N / nch;
        double t3 = median([&] {                        // today's form: three streams, two event hops per chunk
            for (int k = 0; k < nch; ++k) {
                (void)hipMemcpyAsync((char *)dx + k * c * K * 4, hx + k * c * K * 4, c * K * 4, hipMemcpyHostToDevice, s[0]);
                (void)hipEventRecord(ev[2 * k], s[0]); (void)hipStreamWaitEvent(s[1], ev[2 * k], 0);
                kern(s[1], k * c, c);
                (void)hipEventRecord(ev[2 * k + 1], s[1]); (void)hipStreamWaitEvent(s[2], ev[2 * k + 1], 0);
                (void)hipMemcpyAsync(hd + k * c * M * 4, (char *)dd + k * c * M * 4, c * M * 4, hipMemcpyDeviceToHost, s[2]);
            }
            (void)hipStreamSynchronize(s[2]);
        });
        double t2 = median([&] {                        // upload and kernels on one stream, download on another: one hop per chunk
            for (int k = 0; k < nch; ++k) {
                (void)hipMemcpyAsync((char *)dx + k * c * K * 4, hx + k * c * K * 4, c * K * 4, hipMemcpyHostToDevice, s[0]);
                kern(s[0], k * c, c);
                (void)hipEventRecord(ev[k], s[0]); (void)hipStreamWaitEvent(s[2], ev[k], 0);
                (void)hipMemcpyAsync(hd + k * c * M * 4, (char *)dd + k * c * M * 4, c * M * 4, hipMemcpyDeviceToHost, s[2]);
            }
            (void)hipStreamSynchronize(s[2]);
        });
        double t2b = median([&] {                       // chunk k whole on stream k % 2: no events at all
            for (int k = 0; k < nch; ++k) {
                hipStream_t q = s[k & 1 ? 2 : 0];
                (void)hipMemcpyAsync((char *)dx + k * c * K * 4, hx + k * c * K * 4, c * K * 4, hipMemcpyHostToDevice, q);
                kern(q, k * c, c);
                (void)hipMemcpyAsync(hd + k * c * M * 4, (char *)dd + k * c * M * 4, c * M * 4, hipMemcpyDeviceToHost, q);
            }
            (void)hipStreamSynchronize(s[0]); (void)hipStreamSynchronize(s[2]);
        });
        double tz = median([&] {                        // the kernels read src1 through the pool's device mapping; download by DMA on a second stream
            float *mx; (void)hipHostGetDevicePointer((void **)&mx, hx, 0);
            for (int k = 0; k < nch; ++k) {
                work<<<256, 256, 0, s[1]>>>(mx + k * c * K, dd + k * c * M, c * K, 200);
                work<<<256, 256, 0, s[1]>>>(dx + k * c * K, dd + k * c * M, c * K, 1500);
                (void)hipEventRecord(ev[k], s[1]); (void)hipStreamWaitEvent(s[2], ev[k], 0);
                (void)hipMemcpyAsync(hd + k * c * M * 4, (char *)dd + k * c * M * 4, c * M * 4, hipMemcpyDeviceToHost, s[2]);
            }
            (void)hipStreamSynchronize(s[2]);
        });
        float *mx, *md; (void)hipHostGetDevicePointer((void **)&mx, hx, 0); (void)hipHostGetDevicePointer((void **)&md, hd, 0);
        for (int wg : {256, 1024}) {
            double tk1 = median([&] {                   // no DMA at all: both kernels touch the pool through its mapping, one stream
                for (int k = 0; k < nch; ++k) {
                    pull<<<wg, 256, 0, s[1]>>>((const float4 *)(mx + k * c * K), (float4 *)(dx + k * c * K), c * K / 4);
                    push<<<wg, 256, 0, s[1]>>>((const float4 *)(dd + k * c * M), (float4 *)(md + k * c * M), c * M / 4, 300);
                }
                (void)hipStreamSynchronize(s[1]);
            });
            double tk2 = median([&] {                   // ... chunk k on stream k % 2
                for (int k = 0; k < nch; ++k) {
                    hipStream_t q = s[k & 1 ? 2 : 0];
                    pull<<<wg, 256, 0, q>>>((const float4 *)(mx + k * c * K), (float4 *)(dx + k * c * K), c * K / 4);
                    push<<<wg, 256, 0, q>>>((const float4 *)(dd + k * c * M), (float4 *)(md + k * c * M), c * M / 4, 300);
                }
                (void)hipStreamSynchronize(s[0]); (void)hipStreamSynchronize(s[2]);
            });
            double tk3 = median([&] {                   // ... pulls on one stream, pushes on another, an event per chunk
                for (int k = 0; k < nch; ++k) {
                    pull<<<wg, 256, 0, s[0]>>>((const float4 *)(mx + k * c * K), (float4 *)(dx + k * c * K), c * K / 4);
                    (void)hipEventRecord(ev[k], s[0]); (void)hipStreamWaitEvent(s[2], ev[k], 0);
                    push<<<wg, 256, 0, s[2]>>>((const float4 *)(dd + k * c * M), (float4 *)(md + k * c * M), c * M / 4, 300);
                }
                (void)hipStreamSynchronize(s[2]);
            });
            printf("%2d chunks, kernels only (%4d workgroups): one stream %7.1f | alternate streams %7.1f | pull stream + push stream %7.1f us\n", nch, wg, tk1, tk2, tk3);
        }
        printf("%2d chunks: three streams %7.1f | upload + kernels on one, download on another %7.1f | alternate streams, no events %7.1f | zero-copy read + DMA download %7.1f us\n", nch, t3, t2, t2b, tz);
    }
    return 0;
}
