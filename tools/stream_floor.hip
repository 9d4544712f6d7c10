// stream_floor.hip -- developer tool: what does the chip give a kernel that does NOTHING but stream a 10.5 MB (or larger) buffer
// from HBM, launched back to back over rotating copies (so neither L2 nor the 256 MB Infinity Cache serves it)?  This is the
// floor the batch-1 mat-vec (gemv.hip K2f) is measured against in DESIGN.md 5.0.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/stream_floor tools/stream_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int UN, bool NT>
__global__ void __launch_bounds__(512) stream_kernel(const v4i *__restrict__ p, size_t n16, int *__restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int acc = 0;
    for (; i + (UN - 1) * stride < n16; i += UN * stride) {
        v4i v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { v4i v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678) out[blockIdx.x] = acc;           // (never: keeps the loads alive)
}

// the mat-vec's weight addressing without its arithmetic: persistent workgroups over 16-row tiles, lane = (row, k-lane), 8 waves x 4
// k-lanes = 32 k-workers taking k-blocks u, u+32, u+64, u+96 of a 128-block row (K = 4096).  LAYOUT 0: the resident planar
// form [k-block][row][16 B] (a wave instruction = four 256-byte runs, 16 * Mpad bytes apart); 1: tile-major [tile][k-block][16 rows]
// [16 B] (a tile = one contiguous 32 KB run, a wave instruction = four 256-byte runs 8 KB apart); 2: tile-major with the wave's 16
// blocks adjacent ([tile][wave][j][kq][row]: a wave instruction = 1 KB contiguous, a wave's tile share = 4 KB contiguous)
template <int LAYOUT, bool PF>
__global__ void __launch_bounds__(512) tiled_kernel(const uint8_t *__restrict__ p, int64_t Mpad, int ntiles, int *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4, u = wave * 4 + kq;
    int acc = 0;
    v4i q[4], qn[4];
    auto load = [&](int tile, v4i *Q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t b = u + 32 * j;
            const uint8_t *a = LAYOUT == 0 ? p + (b * Mpad + (int64_t)tile * 16 + r) * 16
                             : LAYOUT == 1 ? p + (int64_t)tile * 32768 + b * 256 + r * 16
                                           : p + (int64_t)tile * 32768 + wave * 4096 + j * 1024 + lane * 16;
            Q[j] = __builtin_nontemporal_load((const v4i *)a);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) load(tile, q);
    for (; tile < ntiles; tile += gridDim.x) {
        if (PF && tile + (int)gridDim.x < ntiles) load(tile + gridDim.x, qn);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= q[j].x ^ q[j].y ^ q[j].z ^ q[j].w;
        if (PF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = qn[j];
        } else if (tile + (int)gridDim.x < ntiles) load(tile + gridDim.x, q);
    }
    if (acc == 0x12345678) out[blockIdx.x] = acc;
}

// the same addressing with the mat-vec's other parts added one at a time (FLAGS: 1 = the scale plane, one 4-byte load per lane
// and k-block; 2 = the eight dot4 + scale-accumulate per block against LDS-resident activations; 4 = the cross-lane / cross-wave
// reduction with its barrier and the 16 result stores per tile).  K = 4096, grid = 2 persistent workgroups per CU.
template <int FLAGS>
__global__ void __launch_bounds__(512, 4) matvec_like(const uint8_t *__restrict__ p, const float *__restrict__ d, int64_t Mpad, int ntiles,
                                                      float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t sAct[8 * 16 * 48];
    __shared__ float sRed[2][8][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4, u = wave * 4 + kq;
    for (int i = threadIdx.x; i < 8 * 16 * 48 / 4; i += 512) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
        ((uint32_t *)sAct)[i] = (FLAGS & 32) ? h : 0x01020304u * (i & 3);          // 32: activations with every bit toggling
    }
    __syncthreads();
    unsigned long long c0 = 0, r0 = 0;
    if (FLAGS & 64) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }      // 64: shader clock over the kernel -> out[]
    const uint8_t *myq = sAct + wave * 16 * 48;
    v4i q[4], qn[4];
    float dw[4], dwn[4];
    auto load = [&](int tile, v4i *Q, float *D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t b = u + 32 * j;
            Q[j] = __builtin_nontemporal_load((const v4i *)(p + (b * Mpad + (int64_t)tile * 16 + r) * 16));
            if ((FLAGS & 1) && !(FLAGS & 24)) D[j] = d[b * Mpad + (int64_t)tile * 16 + r];
            else if (FLAGS & 8) D[j] = d[((int64_t)tile * 128 + b) * 16 + r];       // 8: scale plane tile-major [tile][k-block][16 rows]
            else if (!(FLAGS & 16)) D[j] = 1.0f;
        }
        if (FLAGS & 16) {                                                            // 16: the lane's four scales as ONE 16-byte load
            const float4 t = *(const float4 *)(d + (((int64_t)tile * 8 + wave) * 64 + lane) * 4);
            D[0] = t.x; D[1] = t.y; D[2] = t.z; D[3] = t.w;
        }
    };
    int parity = 0;
    auto item = [&](int tile, const v4i *Q, const float *D) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (FLAGS & 2) {
                const int i = kq + 4 * j;
                const uint4 a0 = *(const uint4 *)(myq + i * 48), a1 = *(const uint4 *)(myq + i * 48 + 16);
                const uint32_t qq[4] = {(uint32_t)Q[j].x, (uint32_t)Q[j].y, (uint32_t)Q[j].z, (uint32_t)Q[j].w};
                const uint32_t aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                int sd = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sd = __builtin_amdgcn_sdot4((int)(qq[k] & 0x0F0F0F0Fu), (int)aa[k], sd, false);
                    sd = __builtin_amdgcn_sdot4((int)((qq[k] >> 4) & 0x0F0F0F0Fu), (int)aa[4 + k], sd, false);
                }
                sd -= 8 * (int)(aa[0] & 0xFF);
                acc = fmaf(D[j] * __uint_as_float(0x3f800000u | (aa[1] & 0xFFFF)), (float)sd, acc);
            } else {
                acc += __uint_as_float(((uint32_t)(Q[j].x ^ Q[j].y ^ Q[j].z ^ Q[j].w)) & 0x3fffffffu) * D[j];
            }
        }
        if (FLAGS & 4) {
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            if (kq == 0) sRed[parity][wave][r] = acc;
            __syncthreads();
            if (threadIdx.x < 16) {
                float t = 0.0f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) t += sRed[parity][w8][threadIdx.x];
                asm volatile("global_store_dword %0, %1, off" : : "v"(out + (int64_t)tile * 16 + threadIdx.x), "v"(t) : "memory");
            }
            parity ^= 1;
        } else if (acc == 1.2345f) out[threadIdx.x] = acc;
    };
    const int g = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    load(tile, q, dw);
    for (; tile + 2 * g < ntiles; tile += 2 * g) {
        load(tile + g, qn, dwn);
        item(tile, q, dw);
        load(tile + 2 * g, q, dw);
        item(tile + g, qn, dwn);
    }
    if (tile + g < ntiles) {
        load(tile + g, qn, dwn);
        item(tile, q, dw);
        item(tile + g, qn, dwn);
    } else {
        item(tile, q, dw);
    }
    if ((FLAGS & 64) && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
        out[Mpad + 16] = (float)(c1 - c0);
        out[Mpad + 17] = (float)(r1 - r0);
    }
}

// the skeleton with DYNAMIC tile scheduling: two static rounds (tiles b, b + G), then tickets from one counter per group of G / NG
// workgroups (blockIdx % NG), fetched two items ahead by thread 0 and handed to the workgroup through LDS at the per-tile barrier.
// Exactly 2 * (G / NG) + (pool size) atomics reach a counter per launch, so the workgroup that receives the last value resets it:
// the counters are zero again at kernel end without a second counter or a memset.
template <int FLAGS, int NG>
__global__ void __launch_bounds__(512, 4) matvec_dyn(const uint8_t *__restrict__ p, const float *__restrict__ d, int64_t Mpad, int ntiles,
                                                     float *__restrict__ out, unsigned *__restrict__ ctr) {
    __shared__ __attribute__((aligned(16))) uint8_t sAct[8 * 16 * 48];
    __shared__ float sRed[2][8][16];
    __shared__ int sTick[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4, u = wave * 4 + kq;
    for (int i = threadIdx.x; i < 8 * 16 * 48 / 4; i += 512) ((uint32_t *)sAct)[i] = 0x01020304u * (i & 3);
    const int G = gridDim.x, grp = blockIdx.x % NG, S = G / NG;
    unsigned *my = ctr + 32 * grp;                                   // one counter per 128-byte line
    const int pool = ntiles > 2 * G + grp ? (ntiles - 2 * G - grp + NG - 1) / NG : 0;     // tickets 0 .. pool-1 are tiles 2G + grp + NG * v
    const unsigned last = (unsigned)(2 * S + pool) - 1u;            // the value the very last atomic of this launch returns
    unsigned pa = 0, pb = 0;                                         // thread 0: the two tickets in flight (pa older)
    bool fetching = true, am_last = false;
    if (threadIdx.x == 0) {
        pa = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pb = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const uint8_t *myq = sAct + wave * 16 * 48;
    v4i q[4], qn[4];
    float dw[4], dwn[4];
    auto load = [&](int tile, v4i *Q, float *D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t b = u + 32 * j;
            Q[j] = __builtin_nontemporal_load((const v4i *)(p + (b * Mpad + (int64_t)tile * 16 + r) * 16));
            if (FLAGS & 8) D[j] = d[((int64_t)tile * 128 + b) * 16 + r];
            else D[j] = d[b * Mpad + (int64_t)tile * 16 + r];
        }
    };
    int parity = 0;
    // returns the tile of the item two ahead (>= ntiles: none)
    auto item = [&](int tile, const v4i *Q, const float *D) -> int {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = kq + 4 * j;
            const uint4 a0 = *(const uint4 *)(myq + i * 48), a1 = *(const uint4 *)(myq + i * 48 + 16);
            const uint32_t qq[4] = {(uint32_t)Q[j].x, (uint32_t)Q[j].y, (uint32_t)Q[j].z, (uint32_t)Q[j].w};
            const uint32_t aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            int sd = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sd = __builtin_amdgcn_sdot4((int)(qq[k] & 0x0F0F0F0Fu), (int)aa[k], sd, false);
                sd = __builtin_amdgcn_sdot4((int)((qq[k] >> 4) & 0x0F0F0F0Fu), (int)aa[4 + k], sd, false);
            }
            sd -= 8 * (int)(aa[0] & 0xFF);
            acc = fmaf(D[j] * __uint_as_float(0x3f800000u | (aa[1] & 0xFFFF)), (float)sd, acc);
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (kq == 0) sRed[parity][wave][r] = acc;
        if (threadIdx.x == 0) {
            // hand the older ticket to the workgroup; replace it while tickets are still valid
            am_last |= pa == last;
            sTick[parity] = pa < (unsigned)pool ? 2 * G + grp + NG * (int)pa : 0x7fffffff;
            if (fetching && pa >= (unsigned)pool) fetching = false;
            pa = pb;
            if (fetching) pb = __hip_atomic_fetch_add(my, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else pb = 0xffffffffu;
        }
        __syncthreads();
        const int nn = sTick[parity];
        if (threadIdx.x < 16) {
            float t = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) t += sRed[parity][w8][threadIdx.x];
            asm volatile("global_store_dword %0, %1, off" : : "v"(out + (int64_t)tile * 16 + threadIdx.x), "v"(t) : "memory");
        }
        parity ^= 1;
        return nn;
    };
    int cur = blockIdx.x, nxt = blockIdx.x + G, nn;
    if (cur < ntiles) {
        load(cur, q, dw);
        while (true) {
            if (nxt >= ntiles) { item(cur, q, dw); break; }
            load(nxt, qn, dwn);
            nn = item(cur, q, dw);
            cur = nxt; nxt = nn;
            if (nxt >= ntiles) { item(cur, qn, dwn); break; }
            load(nxt, q, dw);
            nn = item(cur, qn, dwn);
            cur = nxt; nxt = nn;
        }
    }
    // (a workgroup reads every ticket it asked for, so one of them sees the launch's last value)
    if (threadIdx.x == 0) {
        am_last |= pa == last || pb == last;
        if (am_last) __hip_atomic_store(my, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void fill_random(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;
        x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
        p[i] = x;
    }
}

__global__ void empty_kernel(int *out) { if (out == nullptr) out[0] = 1; }

template <typename F> static double period_us(F launch, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 40; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

// the same launches as ONE captured graph of `n` kernel nodes, replayed: the launch path bench.py's side configs use
template <typename F> static double graph_period_us(F launch, int n, int reps) {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i) launch(i, st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
    return ms * 1e3 / (reps * (double)n);
}

int main(int argc, char **argv) {
    const size_t sizes[] = {10485760 + 2097152 / 4, 4 * 10485760ull, 82000000ull, 168000000ull};   // ~ Q4_0 4096^2, 4x, 32000x4096, 65536x4096
    int *out; CK(hipMalloc(&out, 1 << 16));
    printf("empty kernel back to back: %.2f us per launch\n", period_us([&](int) { empty_kernel<<<1, 64>>>(out); }, 2000));
    printf("empty kernel, graph of 64 nodes replayed: %.2f us per node\n", graph_period_us([&](int, hipStream_t st) { empty_kernel<<<1, 64, 0, st>>>(out); }, 64, 20));
    for (size_t bytes : sizes) {
        bytes &= ~(size_t)15;
        const int copies = (int)((400ull << 20) / bytes) + 2;
        std::vector<v4i *> buf(copies);
        // random bits by default (the weights of a real matrix); "const" as the first argument fills with one byte value
        const bool constant = argc > 1 && argv[1][0] == 'c';
        int bi = 0;
        for (auto &b : buf) {
            CK(hipMalloc(&b, bytes));
            if (constant) CK(hipMemset(b, 1, bytes));
            else fill_random<<<1024, 256>>>((uint32_t *)b, bytes / 4, 77u * ++bi);
        }
        CK(hipDeviceSynchronize());
        printf("%.1f MB x %d copies:\n", bytes / 1e6, copies);
        const int grids[] = {256, 512, 1024, 2048};
        for (int g : grids) {
            double a = period_us([&](int i) { stream_kernel<4, true><<<g, 512>>>(buf[i % copies], bytes / 16, out); }, 400);
            double b = period_us([&](int i) { stream_kernel<4, false><<<g, 512>>>(buf[i % copies], bytes / 16, out); }, 400);
            double c = period_us([&](int i) { stream_kernel<8, true><<<g, 512>>>(buf[i % copies], bytes / 16, out); }, 400);
            double d = period_us([&](int i) { stream_kernel<2, true><<<g, 512>>>(buf[i % copies], bytes / 16, out); }, 400);
            printf("  grid %4d x 512: nt x4 %.2f us = %.2f TB/s | plain x4 %.2f | nt x8 %.2f | nt x2 %.2f\n", g, a, bytes / a * 1e-6, b, c, d);
        }
        for (int g : grids) {
            const int n = copies * 2;
            double a = graph_period_us([&](int i, hipStream_t st) { stream_kernel<4, true><<<g, 512, 0, st>>>(buf[i % copies], bytes / 16, out); }, n, 10);
            double d = graph_period_us([&](int i, hipStream_t st) { stream_kernel<2, true><<<g, 512, 0, st>>>(buf[i % copies], bytes / 16, out); }, n, 10);
            printf("  graph replay, grid %4d x 512: nt x4 %.2f us = %.2f TB/s | nt x2 %.2f\n", g, a, bytes / a * 1e-6, d);
        }
        {
            const int64_t M = (int64_t)(bytes / 2048 / 16) * 16;            // rows of a K = 4096 nibble plane of this many bytes
            const int ntiles = (int)(M / 16);
            const int n = copies * 2;
            for (int g : {256, 512, 1024}) {
                const int gg = g < ntiles ? g : ntiles;
                double t0 = graph_period_us([&](int i, hipStream_t st) { tiled_kernel<0, true><<<gg, 512, 0, st>>>((const uint8_t *)buf[i % copies], M, ntiles, out); }, n, 10);
                double t0n = graph_period_us([&](int i, hipStream_t st) { tiled_kernel<0, false><<<gg, 512, 0, st>>>((const uint8_t *)buf[i % copies], M, ntiles, out); }, n, 10);
                double t1 = graph_period_us([&](int i, hipStream_t st) { tiled_kernel<1, true><<<gg, 512, 0, st>>>((const uint8_t *)buf[i % copies], M, ntiles, out); }, n, 10);
                double t2 = graph_period_us([&](int i, hipStream_t st) { tiled_kernel<2, true><<<gg, 512, 0, st>>>((const uint8_t *)buf[i % copies], M, ntiles, out); }, n, 10);
                double t2n = graph_period_us([&](int i, hipStream_t st) { tiled_kernel<2, false><<<gg, 512, 0, st>>>((const uint8_t *)buf[i % copies], M, ntiles, out); }, n, 10);
                printf("  mat-vec addressing, %lld rows, grid %4d: planar %.2f us (no look-ahead %.2f) | tile-major %.2f | tile-major, wave-contiguous %.2f (no look-ahead %.2f)  [%.2f TB/s best]\n",
                       (long long)M, gg, t0, t0n, t1, t2, t2n, M * 2048.0 / (t2 < t1 ? (t2 < t0 ? t2 : t0) : (t1 < t0 ? t1 : t0)) * 1e-6);
            }
        }
        {
            // the whole weight of a K = 4096 Q4_0 matrix in `bytes`: 16/20 of it nibbles, 4/20 scales
            const int64_t M = (int64_t)(bytes / 2560 / 16) * 16;
            const int ntiles = (int)(M / 16), n = copies * 2;
            float *res; CK(hipMalloc(&res, (M + 64) * 4));
            auto run = [&](auto kern) {
                return graph_period_us([&](int i, hipStream_t st) {
                    const uint8_t *base = (const uint8_t *)buf[i % copies];
                    kern<<<ntiles < 512 ? ntiles : 512, 512, 0, st>>>(base, (const float *)(base + M * 2048), M, ntiles, res); }, n, 10);
            };
            const double t9 = run(matvec_like<9>), t15 = run(matvec_like<15>), t17 = run(matvec_like<17>), t23 = run(matvec_like<23>);
            printf("  mat-vec skeleton, scale plane tile-major: + scales %.2f us, all %.2f | one 16-byte scale load per lane: + scales %.2f, all %.2f\n", t9, t15, t17, t23);
            {
                int wrate = 0; CK(hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0));     // kHz
                auto clk = [&](const char *nm, auto kern) {
                    const double t = run(kern);
                    float cr[2]; CK(hipMemcpy(cr, res + M + 16, 8, hipMemcpyDeviceToHost));
                    printf("  %s: %.2f us, workgroup 0 ran %.0f shader cycles in %.2f us = %.2f GHz\n", nm, t, cr[0], cr[1] / (wrate * 1e-3), cr[0] / (cr[1] / (wrate * 1e-3)) * 1e-3);
                };
                clk("skeleton, tile-major scales, constant activations", matvec_like<15 + 64>);
                clk("skeleton, tile-major scales, random activations  ", matvec_like<15 + 32 + 64>);
                clk("skeleton, nibbles only                           ", matvec_like<0 + 64>);
            }
            {
                unsigned *ctr; CK(hipMalloc(&ctr, 128 * 128)); CK(hipMemset(ctr, 0, 128 * 128));
                auto rund = [&](auto kern) {
                    return graph_period_us([&](int i, hipStream_t st) {
                        const uint8_t *base = (const uint8_t *)buf[i % copies];
                        kern<<<512, 512, 0, st>>>(base, (const float *)(base + M * 2048), M, ntiles, res, ctr); }, n, 10);
                };
                if (ntiles >= 3 * 512) {
                    const double d8 = rund(matvec_dyn<15, 8>), d32 = rund(matvec_dyn<15, 32>), d64 = rund(matvec_dyn<15, 64>), d128 = rund(matvec_dyn<15, 128>);
                    unsigned hc[32 * 128]; CK(hipMemcpy(hc, ctr, sizeof hc, hipMemcpyDeviceToHost));
                    unsigned bad = 0; for (unsigned v : hc) bad |= v;
                    printf("  skeleton (tile-major scales, all parts), DYNAMIC tiles: 8 counters %.2f us | 32 counters %.2f | 64 counters %.2f | 128 counters %.2f  (counters back at zero: %s)\n",
                           d8, d32, d64, d128, bad ? "NO" : "yes");
                }
                CK(hipFree(ctr));
            }
            const double t0 = run(matvec_like<0>), t1 = run(matvec_like<1>), t3 = run(matvec_like<3>), t5 = run(matvec_like<5>), t7 = run(matvec_like<7>), t6 = run(matvec_like<6>);
            printf("  mat-vec skeleton, %lld rows: nibbles only %.2f us | + scales %.2f (%.2f TB/s) | + dots %.2f | scales + reduce %.2f | all %.2f (%.2f TB/s) | all but scales %.2f\n",
                   (long long)M, t0, t1, M * 2560.0 / t1 * 1e-6, t3, t5, t7, M * 2560.0 / t7 * 1e-6, t6);
            CK(hipFree(res));
        }
        for (auto &b : buf) CK(hipFree(b));
    }
    return 0;
}
