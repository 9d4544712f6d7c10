// eltwise.hip -- the f32 element-wise neighbours of mul_mat in a transformer block, HBM-bound:
//   add      ggml_compute_forward_add_f32      Ggml.cs:4622-4682 (ggml_vec_add_f32)
//   mul      ggml_compute_forward_mul_f32      Ggml.cs:5007-5035 (ggml_vec_mul_f32)
//   scale    ggml_compute_forward_scale_f32    Ggml.cs:6746-6778 (in place: dst is a view of src0, ggml_scale_impl :8265)
//   rms_norm ggml_compute_forward_rms_norm_f32 Ggml.cs:5858-5920 (f32 squares summed in f64, eps = 1e-6)
//   silu     ggml_compute_forward_silu_f32     Ggml.cs:5705-5748 (half-table form, see silu_f32_kernel)
// One IEEE operation per reference operator (built with -ffp-contract=off, correctly rounded divide / sqrt), so add, mul and
// scale are bit-exact; rms_norm differs from the reference only in the order of its f64 additions (a wave-wide tree
// instead of a sequential loop), which reaches the f32 result only when the f64 sum sits on a float rounding boundary.
#include "common.h"

namespace {

template <int OP>   // 0 add, 1 mul
__global__ __launch_bounds__(256) void binary_f32_kernel(const float4 *__restrict__ x, const float4 *__restrict__ y,
                                                        float4 *__restrict__ z, int64_t n4, const float *xs, const float *ys,
                                                        float *zs, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const float4 a = x[i], b = y[i];
        z[i] = OP == 0 ? make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w) : make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
    }
    const int64_t t = n4 * 4 + i;                      // tail (n % 4 elements), handled by the first threads
    if (i < 4 && t < n) zs[t] = OP == 0 ? xs[t] + ys[t] : xs[t] * ys[t];
}

__global__ __launch_bounds__(256) void scale_f32_kernel(float *__restrict__ z, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] *= v;                               // ggml_vec_scale_f32
}

// silu in the reference's GGML_SILU_FP16 form (Ggml.cs:2737-2746): the argument is rounded to half, the result is the half
// table entry silu(f) = f / (1 + exp(-f)) rounded to half (table built at Ggml.cs:1455-1471; indexed by bit pattern, SURVEY
// A2 intent).  65536 possible results: computed here instead of looked up -- exp in f64 and rounded to f32 (= the
// correctly rounded expf), then the reference's own f32 add and divide, then the round to half.  tests/ checks all 65536.
__global__ __launch_bounds__(256) void silu_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = (float)(_Float16)x[i];
    const float e = (float)exp(-(double)f);
    const float s = f / (1.0f + e);
    y[i] = (float)(_Float16)s;
}

// one wave per row
__global__ __launch_bounds__(256) void rms_norm_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t nr, int64_t nc) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nr) return;
    const float *xr = x + row * nc;
    const float scale = rms_row_scale(xr, nc, lane);
    float *yr = y + row * nc;
#pragma unroll 8
    for (int64_t i = lane; i < nc; i += 64) yr[i] = xr[i] * scale;
}

}  // namespace

hipError_t launch_binary_f32(int op, const float *x, const float *y, float *z, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t n4 = n / 4;
    dim3 grid((unsigned)((n4 + 255) / 256 + (n4 % 256 == 0 ? 1 : 0)));
    if (op == 0) binary_f32_kernel<0><<<grid, 256, 0, st>>>((const float4 *)x, (const float4 *)y, (float4 *)z, n4, x, y, z, n);
    else binary_f32_kernel<1><<<grid, 256, 0, st>>>((const float4 *)x, (const float4 *)y, (float4 *)z, n4, x, y, z, n);
    return hipGetLastError();
}

hipError_t launch_scale_f32(float *z, int64_t n, float v, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    scale_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(z, n, v);
    return hipGetLastError();
}

hipError_t launch_silu_f32(const float *x, float *y, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    silu_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>(x, y, n);
    return hipGetLastError();
}

hipError_t launch_rms_norm_f32(const float *x, float *y, int64_t nr, int64_t nc, hipStream_t st) {
    if (nr <= 0 || nc <= 0) return hipSuccess;
    rms_norm_f32_kernel<<<dim3((unsigned)((nr + 3) / 4)), 256, 0, st>>>(x, y, nr, nc);
    return hipGetLastError();
}
