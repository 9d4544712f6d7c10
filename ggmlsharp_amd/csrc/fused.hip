// fused.hip -- the f32 neighbours of mul_mat FUSED: pairs of graph nodes that one launch serves,
// every node's data still materialised (the reference's contract: after ggml_graph_compute each node's data is in host
// memory), each value produced by exactly the reference's operation sequence -- bit for bit what the separate kernels of
// eltwise.hip give.
//   rms_norm -> mul   (the pre-mul_mat norm of a transformer block: Ggml.cs:5858-5920 then 5007-5035): n = x * scale and
//                     y = n * g are written in the same pass over the row: one launch, 3 reads + 2 writes of a tensor from
//                     HBM become 2 + 2 (the row is re-read from L1 / L2 between the sum of squares and the scaling);
//   silu -> mul       (the SwiGLU gate, Ggml.cs:5705-5748 then 5007-5035): s = silu(a), y = s * b, one pass.
// The epilogue side (mul_mat -> add, mul_mat -> scale) lives in the mat-mul kernels themselves: gemv.hip (N <= 4) and
// gemm_qmx.hip apply it to the accumulators as they are stored.
#include "common.h"

namespace {

// one wave per row, the loop structure of rms_norm_f32_kernel (eltwise.hip) -- same element order, same f64 tree, hence the
// same bits as the unfused pair; the second pass re-reads the row from L1 / L2 (a row is a few KB), not from HBM
__global__ __launch_bounds__(256) void rms_norm_mul_kernel(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ n_out,
                                                           float *__restrict__ y_out, int64_t nr, int64_t nc) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nr) return;
    const float *xr = x + row * nc, *gr = g + row * nc;
    float *nrw = n_out + row * nc, *yr = y_out + row * nc;
    const float scale = rms_row_scale(xr, nc, lane);
#pragma unroll 8
    for (int64_t i = lane; i < nc; i += 64) {
        const float nn = xr[i] * scale;                     // ggml_vec_scale_f32 (Ggml.cs:5917)
        nrw[i] = nn;
        yr[i] = nn * gr[i];                                 // ggml_vec_mul_f32 (Ggml.cs:5029)
    }
}

// s = silu(a) in the reference's GGML_SILU_FP16 form (eltwise.hip silu_f32_kernel), y = s * b
__global__ __launch_bounds__(256) void silu_mul_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ s_out,
                                                       float *__restrict__ y_out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = (float)(_Float16)a[i];
    const float e = (float)exp(-(double)f);
    const float s = (float)(_Float16)(f / (1.0f + e));
    s_out[i] = s;
    y_out[i] = s * b[i];
}

}  // namespace

hipError_t launch_rms_norm_mul_f32(const float *x, const float *g, float *n_out, float *y_out, int64_t nr, int64_t nc, hipStream_t st) {
    if (nr <= 0 || nc <= 0) return hipSuccess;
    rms_norm_mul_kernel<<<dim3((unsigned)((nr + 3) / 4)), 256, 0, st>>>(x, g, n_out, y_out, nr, nc);
    return hipGetLastError();
}

hipError_t launch_silu_mul_f32(const float *a, const float *b, float *s_out, float *y_out, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    silu_mul_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(a, b, s_out, y_out, n);
    return hipGetLastError();
}
