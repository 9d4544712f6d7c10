/*
 * ggml.h -- host-side mirror of the slice of GGMLSharp's public API that the quantized mul_mat path needs
 * (SURVEY.md 8(b) "Public C# surface that must remain source-compatible for this path").
 *
 * In the reference this surface is C# (`public static` members of GGMLSharp.Ggml).  No .NET toolchain exists in the
 * build image, so the host side above the C-ABI (ggml_hip.h) is written in C++ with the same names, argument
 * meaning, struct layouts and pool arithmetic, exported with C linkage so the parity tests read like the
 * reference's own Test0/Test3 programs.  GGML_OP_MUL_MAT nodes are computed by the HIP path through
 * ggml_hip_compute_forward_mul_mat, GGML_OP_CPY (f32/f16 -> quantized) and GGML_OP_ADD (quantized + f32) through
 * ggml_hip_compute_forward_cpy / _add -- the product has no CPU compute path; any other op in a graph is reported
 * as unsupported (out of scope).
 *
 * Deviations from the reference signatures, all forced by C linkage or by the missing error channel:
 *   - ggml_init takes the params struct by pointer (C# passes it by value);
 *   - ggml_build_forward writes into a caller-provided ggml_cgraph (C# returns the 98 KB struct by value,
 *     Ggml.cs:7653-7673);
 *   - ggml_graph_compute returns an int status (the reference returns void and only Debug.Asserts).
 */
#ifndef GGML_HOST_MIRROR_H
#define GGML_HOST_MIRROR_H

#include "ggml_hip_ext.h"

#ifdef __cplusplus
extern "C" {
#endif

/* TypeDefs:24-30 */
struct ggml_init_params {
    uint64_t mem_size;    /* bytes */
    void    *mem_buffer;  /* if NULL, memory is allocated internally */
    uint8_t  no_alloc;    /* C# bool */
};

/* TypeDefs:48-56 (32 bytes) */
struct ggml_object {
    uint64_t offs;
    uint64_t size;
    struct ggml_object *next;
    uint8_t padding[8];
};

/* TypeDefs:58-63 */
struct ggml_scratch {
    uint64_t offs;
    uint64_t size;
    void    *data;
};

/* TypeDefs:32-46 (88 bytes) */
struct ggml_context {
    uint64_t mem_size;
    void    *mem_buffer;
    uint8_t  mem_buffer_owned;
    uint8_t  no_alloc;
    uint8_t  _pad[2];
    int32_t  n_objects;
    struct ggml_object *objects_begin;
    struct ggml_object *objects_end;
    struct ggml_scratch scratch;
    struct ggml_scratch scratch_save;
};

/* TypeDefs:102-121 (98 360 bytes) */
struct ggml_cgraph {
    int32_t n_nodes;
    int32_t n_leafs;
    int32_t n_threads;
    int32_t _pad0;
    size_t  work_size;
    struct ggml_tensor *work;
    struct ggml_tensor *nodes[GGML_MAX_NODES];
    struct ggml_tensor *grads[GGML_MAX_NODES];
    struct ggml_tensor *leafs[GGML_MAX_NODES];
    int32_t perf_runs;
    int32_t _pad1;
    int64_t perf_cycles;
    int64_t perf_time_us;
};

/* Ggml.cs:1447-1564: claims one of 64 context slots, NULL when none is free (:1529-1536) */
struct ggml_context *ggml_init(const struct ggml_init_params *params);
/* Ggml.cs:1566-1601; additionally drops cached device copies of tensors that lived in this pool */
void ggml_free(struct ggml_context *ctx);
size_t ggml_used_mem(const struct ggml_context *ctx);

/* Ggml.cs:2347-2395 -> ggml_new_tensor_impl 7722-7866.  NULL when the pool is exhausted (:7757-7763). */
struct ggml_tensor *ggml_new_tensor(struct ggml_context *ctx, int type, int n_dims, const int64_t *ne);
struct ggml_tensor *ggml_new_tensor_1d(struct ggml_context *ctx, int type, int64_t ne0);
struct ggml_tensor *ggml_new_tensor_2d(struct ggml_context *ctx, int type, int64_t ne0, int64_t ne1);
struct ggml_tensor *ggml_new_tensor_3d(struct ggml_context *ctx, int type, int64_t ne0, int64_t ne1, int64_t ne2);
struct ggml_tensor *ggml_new_tensor_4d(struct ggml_context *ctx, int type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3);

/* Ggml.cs:3766-3785 */
int64_t ggml_nelements(const struct ggml_tensor *t);
int64_t ggml_nrows(const struct ggml_tensor *t);
size_t  ggml_nbytes(const struct ggml_tensor *t);
int     ggml_blck_size(int type);
size_t  ggml_type_size(int type);
int     ggml_is_quantized(int type);                           /* Ggml.cs:8355 */
int     ggml_is_contiguous(const struct ggml_tensor *t);       /* Ggml.cs:8365 */
int     ggml_can_mul_mat(const struct ggml_tensor *t0, const struct ggml_tensor *t1);  /* Ggml.cs:8345 */

/* Ggml.cs:2501-2565, 2802-2850 (F32 tensors only here) */
struct ggml_tensor *ggml_set_f32(struct ggml_tensor *t, float value);
float ggml_get_f32_1d(const struct ggml_tensor *t, int i);
void  ggml_set_f32_1d(struct ggml_tensor *t, int i, float value);

/* Ggml.cs:7137-7151 -> ggml_mul_mat_impl 8222-8246: result F32 {a.ne1, b.ne1, a.ne2, b.ne3}; NULL if !ggml_can_mul_mat
 * or a is transposed (the reference Debug.Asserts, :8228-8229) */
struct ggml_tensor *ggml_mul_mat(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);

/* Ggml.cs:3751-3763, 2425-2428 */
struct ggml_tensor *ggml_view_tensor(struct ggml_context *ctx, struct ggml_tensor *src);
struct ggml_tensor *ggml_dup_tensor(struct ggml_context *ctx, const struct ggml_tensor *src);
/* Ggml.cs:7169-7175 -> ggml_cpy_impl 8275-8299: a view of b with op = CPY, src0 = a, src1 = b.
 * The only public way to make a quantized tensor; NULL if element counts differ (:8281). */
struct ggml_tensor *ggml_cpy(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
/* Ggml.cs:6846-6852 -> ggml_add_impl 7868-7891: result has a's type and shape; NULL unless same shape (:7874). */
struct ggml_tensor *ggml_add(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
/* Ggml.cs:6878-6884 -> ggml_mul_impl 7918-7946: result = dup(a); NULL unless same shape (:7924). */
struct ggml_tensor *ggml_mul(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
/* Ggml.cs:7153-7159 -> ggml_scale_impl 8248-8273: b a scalar (:8254); the result is a view of a -- the op works in place. */
struct ggml_tensor *ggml_scale(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
/* Ggml.cs:7123-7128 -> ggml_rms_norm_impl 8199-8220: result = dup(a). */
struct ggml_tensor *ggml_rms_norm(struct ggml_context *ctx, struct ggml_tensor *a);
/* Ggml.cs:7095-7107 -> ggml_silu_impl 8154-8174: result = dup(a), or a view of a for the in-place form. */
struct ggml_tensor *ggml_silu(struct ggml_context *ctx, struct ggml_tensor *a);
struct ggml_tensor *ggml_silu_inplace(struct ggml_context *ctx, struct ggml_tensor *a);

/* Ggml.cs:7648-7673 */
void ggml_build_forward(struct ggml_cgraph *out, struct ggml_tensor *tensor);
void ggml_build_forward_expand(struct ggml_cgraph *cgraph, struct ggml_tensor *tensor);

/* Ggml.cs:3209-3736.  MUL_MAT nodes go through the INIT / COMPUTE / FINALIZE protocol (:3553-3670) into
 * ggml_hip_compute_forward_mul_mat with ith = 0, n_tasks = 1 (the reference's own offload slot, :3368-3370).
 * Returns a ggml_hip_status. */
int ggml_graph_compute(struct ggml_context *ctx, struct ggml_cgraph *cgraph);

#ifdef __cplusplus
}
#endif
#endif
