// Developer probe: does the register-bank pattern of a VALU instruction's operands change its issue rate on gfx950?
// A loop of 16 v_fmac_f32 on 16 accumulators whose three operands sit (a) in ONE VGPR bank (index mod 4 equal -- what an accumulator tile,
// a product tile and a scale vector give when all are 4-aligned tuples indexed alike), (b) in three different banks; at one and at two waves
// per SIMD.  Cycles per instruction per wave from the shader clock.  Fixed registers v16..v51, named in the asm.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/vgpr_bank_probe tools/vgpr_bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters) {
    asm volatile("v_mov_b32 v16, 1.0\n\tv_mov_b32 v17, 1.0\n\tv_mov_b32 v18, 1.0\n\tv_mov_b32 v19, 1.0\n\tv_mov_b32 v20, 1.0\n\tv_mov_b32 v21, 1.0\n\tv_mov_b32 v22, 1.0\n\tv_mov_b32 v23, 1.0\n\tv_mov_b32 v24, 1.0\n\tv_mov_b32 v25, 1.0\n\tv_mov_b32 v26, 1.0\n\tv_mov_b32 v27, 1.0\n\tv_mov_b32 v28, 1.0\n\tv_mov_b32 v29, 1.0\n\tv_mov_b32 v30, 1.0\n\tv_mov_b32 v31, 1.0\n\tv_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v34, 1.0\n\tv_mov_b32 v35, 1.0\n\tv_mov_b32 v36, 1.0\n\tv_mov_b32 v37, 1.0\n\tv_mov_b32 v38, 1.0\n\tv_mov_b32 v39, 1.0\n\tv_mov_b32 v40, 1.0\n\tv_mov_b32 v41, 1.0\n\tv_mov_b32 v42, 1.0\n\tv_mov_b32 v43, 1.0\n\tv_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v46, 1.0\n\tv_mov_b32 v47, 1.0\n\tv_mov_b32 v48, 0.5\n\tv_mov_b32 v49, 0.5\n\tv_mov_b32 v50, 0.5\n\tv_mov_b32 v51, 0.5" ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");     // (normal values: uninitialised registers may hold denormals or NaNs)
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)
            asm volatile(
                "v_fmac_f32 v16, v32, v48\n\t"
                "v_fmac_f32 v17, v33, v49\n\t"
                "v_fmac_f32 v18, v34, v50\n\t"
                "v_fmac_f32 v19, v35, v51\n\t"
                "v_fmac_f32 v20, v36, v48\n\t"
                "v_fmac_f32 v21, v37, v49\n\t"
                "v_fmac_f32 v22, v38, v50\n\t"
                "v_fmac_f32 v23, v39, v51\n\t"
                "v_fmac_f32 v24, v40, v48\n\t"
                "v_fmac_f32 v25, v41, v49\n\t"
                "v_fmac_f32 v26, v42, v50\n\t"
                "v_fmac_f32 v27, v43, v51\n\t"
                "v_fmac_f32 v28, v44, v48\n\t"
                "v_fmac_f32 v29, v45, v49\n\t"
                "v_fmac_f32 v30, v46, v50\n\t"
                "v_fmac_f32 v31, v47, v51\n\t"
                ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        else
            asm volatile(
                "v_fmac_f32 v16, v33, v50\n\t"
                "v_fmac_f32 v17, v34, v51\n\t"
                "v_fmac_f32 v18, v35, v48\n\t"
                "v_fmac_f32 v19, v36, v49\n\t"
                "v_fmac_f32 v20, v37, v50\n\t"
                "v_fmac_f32 v21, v38, v51\n\t"
                "v_fmac_f32 v22, v39, v48\n\t"
                "v_fmac_f32 v23, v40, v49\n\t"
                "v_fmac_f32 v24, v41, v50\n\t"
                "v_fmac_f32 v25, v42, v51\n\t"
                "v_fmac_f32 v26, v43, v48\n\t"
                "v_fmac_f32 v27, v44, v49\n\t"
                "v_fmac_f32 v28, v45, v50\n\t"
                "v_fmac_f32 v29, v46, v51\n\t"
                "v_fmac_f32 v30, v47, v48\n\t"
                "v_fmac_f32 v31, v32, v49\n\t"
                ::: "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    unsigned long long *out;
    CK(hipMalloc(&out, 256 * 8 * 8));
    const int iters = 200000;
    for (int waves = 4; waves <= 8; waves += 4)
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) k<0><<<256, waves * 64>>>(out, iters); else k<1><<<256, waves * 64>>>(out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("   (kernel %.3f ms = %.2f ns per instruction per wave)\n", ms, ms * 1e6 / ((double)iters * 16));
            std::vector<unsigned long long> h(256 * waves);
            CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            printf("%d wave(s) per SIMD, %s: %.2f cycles per v_fmac per wave (median over %zu waves)\n", waves / 4, mode == 0 ? "three operands in ONE bank   " : "three operands in THREE banks",
                   (double)h[h.size() / 2] / ((double)iters * 16), h.size());
        }
    return 0;
}
