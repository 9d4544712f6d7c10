/* The program of INTEGRATION.md section 3, in C99 against include/ggml.h: ggml_init -> new tensors -> ggml_mul_mat ->
 * ggml_build_forward -> ggml_graph_compute -> read dst.  Proves that the headers are valid C and that the C-ABI links
 * from plain C; on a machine without a GPU ggml_graph_compute reports GGML_HIP_ERR_NO_DEVICE (there is no CPU fallback)
 * and the program prints "no-device" and exits 0.  Reads like Test1/Program.cs of the reference (f32 mul_mat 64x128x256). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ggml.h"

int main(void) {
    const int M = 64, K = 128, N = 256;
    struct ggml_init_params ip;
    memset(&ip, 0, sizeof ip);
    ip.mem_size = 16u << 20;
    struct ggml_context *ctx = ggml_init(&ip);
    if (!ctx) return 2;
    struct ggml_tensor *W = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, M);
    struct ggml_tensor *X = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, N);
    if (!W || !X) return 3;
    if (W->ne[0] != K || W->nb[1] != (uint64_t)K * 4 || sizeof(struct ggml_tensor) != 176) return 4;   /* Test0-style layout checks */
    float *w = (float *)W->data, *x = (float *)X->data;
    for (int i = 0; i < M * K; ++i) w[i] = (float)((i * 7) % 13 - 6) * 0.125f;
    for (int i = 0; i < N * K; ++i) x[i] = (float)((i * 5) % 11 - 5) * 0.25f;
    struct ggml_tensor *Y = ggml_mul_mat(ctx, W, X);
    if (!Y || Y->ne[0] != M || Y->ne[1] != N) return 5;
    struct ggml_cgraph *gf = (struct ggml_cgraph *)calloc(1, sizeof *gf);
    ggml_build_forward(gf, Y);
    if (gf->n_nodes != 1 || gf->n_leafs != 2) return 6;
    int rc = ggml_graph_compute(ctx, gf);
    if (rc == GGML_HIP_ERR_NO_DEVICE) { printf("no-device\n"); free(gf); ggml_free(ctx); return 0; }
    if (rc != GGML_HIP_OK) { printf("error %d: %s\n", rc, ggml_hip_last_error()); return 7; }
    const float *y = (const float *)Y->data;
    double worst = 0.0;
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < M; ++m) {
            double s = 0.0;                                  /* ggml_vec_dot_f32: f32 products, f64 sum (Ggml.cs:2631-2640) */
            for (int k = 0; k < K; ++k) s += (double)(w[m * K + k] * x[n * K + k]);
            const double e = s - y[n * M + m];
            if ((e < 0 ? -e : e) > worst) worst = e < 0 ? -e : e;
        }
    printf("ok max abs err %.3g\n", worst);
    free(gf);
    ggml_free(ctx);
    return worst < 1e-3 ? 0 : 8;
}
