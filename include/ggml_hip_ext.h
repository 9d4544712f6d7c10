/*
 * ggml_hip_ext.h -- the rest of libggml_hip.so's C-ABI: what goes beyond the drop-in core of ggml_hip.h (SURVEY.md 8(f) "next"
 * rows and the multi-device forms).  A host that only wants GGML_OP_MUL_MAT on the GPU needs ggml_hip.h alone.
 *   - named / output-only graph scopes, counters                       (ggml_hip_graph_begin_keyed, _graph_outputs, _debug_*_counters)
 *   - introspection and the explicit INIT step                          (ggml_hip_act_image_kind, ggml_hip_quantize_act_dev)
 *   - neighbours of the path and their fused forms                      (ggml_hip_compute_forward_{cpy,add,mul,scale,rms_norm,silu,...})
 *   - device-level fused / grouped products                             (ggml_hip_norm_mul_mat_dev, _mul_mat_multi_dev, _mul_mat_epilogue_dev, ...)
 *   - several devices in one process, one process per device           (ggml_hip_split_weight_*, _mul_mat_split_dev, _ipc_*, _push_columns_dev, ...)
 *   - the k-quant extension type                                        (GGML_HIP_TYPE_Q5_K)
 *   - TEST HOOKS (ggml_hip_debug_*): inert unless called; ggml_hip_debug_force_gemm acts on the CALLING THREAD only.
 */
#ifndef GGML_HIP_EXT_H
#define GGML_HIP_EXT_H

#include "ggml_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: the two headers ARE its export list */
#endif

/* EXTENSION, not a reference type: the reference's enum stops at Q8_1 / I32 (TypeDefs:153-169) and holds no k-quants
 * (SURVEY 8(a) row K), while BASELINE.json's north_star and config 4 name Q5_K.  Built to the PUBLISHED upstream format
 * (ggml k_quants, 2023-06: 176-byte super-blocks of 256 weights; activations by the Q8_K rule) as an unpinned extra: no oracle
 * exists in the reference, tests/np_kquants.py restates the published algorithm.  Accepted by ggml_hip_weight_upload /
 * _from_device / _download, ggml_hip_mul_mat{,_init,_compute}_dev, ggml_hip_mul_mat_work_size,
 * ggml_hip_dequantize_rows_dev and (r4) ggml_hip_quantize_rows_dev (quantize_row_q5_K_reference / _q4_K_reference of the published
 * format restated: make_qkx1_quants per sub-block, 6-bit scales / mins against the super-block's d / dmin) only -- never inside a ggml_tensor (the reference cannot express the type). */
#define GGML_HIP_TYPE_Q5_K 113
/* r4: Q4_K of the same published format -- { half d; half dmin; u8 scales[12]; u8 qs[128] }, 144 bytes per 256 weights, the super-block of
 * Q5_K without its fifth-bit bytes and with the same scale / min packing and the same dot rule against Q8_K.  It lives in the same resident
 * form (eight k-blocks of the planar Q5_1 form, fifth-bit plane zero) and runs the same kernels; the same "unpinned extra" status. */
#define GGML_HIP_TYPE_Q4_K 112
/* r4: Q6_K of the same published format -- { u8 ql[128]; u8 qh[64]; i8 scales[16]; half d }, 210 bytes per 256 weights: sixteen sub-blocks
 * of 16 six-bit weights (q - 32) with a signed 8-bit scale each, no min.  Resident as eight k-blocks of the planar Q4_2 form (two scales per
 * k-block) on int8 operand planes; served by the int8 kernels that take two scales per k-block (its own mat-vec, the batched-decode form and
 * the staged int8 form: WHICH one serves a shape is ggml_hip_mm_plan's answer, the only authority -- no range is restated here).  ggml_hip_quantize_rows_dev: quantize_row_q6_K_reference WITHOUT the least-squares refinement of the
 * sub-block scales (make_qx_quants in its plain form) -- a valid encoder of the published structure.  Unpinned like the other two. */
#define GGML_HIP_TYPE_Q6_K 114


/* OPT-IN, and a deviation from the reference's contract (which leaves EVERY node's data in host memory, Ggml.cs:3539-3704):
 * called inside an open scope, before its nodes, it names the tensors (by data pointer) whose data the caller will read after
 * ggml_hip_graph_end; no other result of the scope is copied to the host -- its host memory keeps whatever it held.  The
 * library still copies what IT needs on the host (a buffer that is recycled, a source it has to upload again).  At prompt-sized
 * batches the copies are the whole cost of a graph: 7B decoder layer at batch 512, 4.26 ms with every node's data, see DESIGN 8
 * with the last node's only.  The key of a named scope must cover the list (the host mirror never calls this). */
int  ggml_hip_graph_outputs(const void *const *host_ptrs, int n);
/* The same scope, NAMED: `key` (non-zero) identifies the graph the caller is about to run -- a hash over what decides the
 * calls it will make: every node's op, the data pointers, types, ne / nb of the node and of its sources, and the scalar
 * operands it reads on the host (the factor of a SCALE node).  Inside a scope the seams cost the host one kernel launch per
 * node and the device -> host copies of all results go out together at the end; a named scope that needed nothing else is
 * captured into a hipGraph the second time it is seen and replayed with ONE launch from the third time on (the seams
 * return at once; leaf tensors are re-read from host memory by the captured copies, so their CONTENTS may change between
 * runs -- a decoder's token loop).  Everything the key covers must be unchanged when a key is reused; a weight that is
 * rewritten must be invalidated as always (ggml_hip_invalidate*), which also drops the captured scopes.  One device slot
 * only; with several the call is ggml_hip_graph_begin.  (7B decoder layer at batch 1 through ggml_graph_compute: 439 us per
 * graph with per-node copies, 97 us in a plain scope, see DESIGN 8 for the replayed figure.) */
int  ggml_hip_graph_begin_keyed(uint64_t key);
/* Named scopes so far by what became of them: observed clean, captured, replayed, refused (tests, tuning). */
void ggml_hip_debug_scope_counters(uint64_t *observed, uint64_t *captured, uint64_t *replayed, uint64_t *refused);
/* Bytes moved over PCIe by seam 1 so far and the number of src1 operands served from a resident dst (tests, tuning). */
void ggml_hip_debug_transfer_counters(uint64_t *h2d_bytes, uint64_t *d2h_bytes, uint64_t *resident_hits);
/* The Seam-1 weight cache (device forms of leaf src0 tensors, keyed by host pointer + shape; Ggml.cs:1545: tensor data is stable for the
 * life of its context).  It is bounded by the device's memory: when an upload cannot be allocated, least-recently-used entries are evicted
 * and the upload is tried once more.  _budget sets a smaller bound per slot in bytes (0 = none) so that a test can watch the eviction happen
 * on a small problem -- the entry the running call uses always stays; _stats reports entries, resident bytes and evictions so far. */
void ggml_hip_debug_weight_cache_budget(size_t bytes_per_slot);
void ggml_hip_debug_weight_cache_stats(uint64_t *entries, uint64_t *bytes, uint64_t *evictions);


/* Which layout step 1 writes into d_work for this weight type, K and N (introspection for tests and profiling tools):
 * 0 = int8 planes (mat-vec and int8-MFMA kernels), 1 / 2 = f16 images (gemm_q16.hip), 3 = bf6 digit image (gemm_qmx.hip).
 * A function of the type, K and N ONLY -- never of the number of weight rows -- so a row shard runs the kernel form of the
 * unsplit matrix (what makes a shard's result bit for bit the matching columns).  The one exception is stated, not hidden: a
 * weight whose planes do not fit 32-bit buffer offsets (more than 4 GiB per plane) takes kind 0 whatever this says. */
int    ggml_hip_act_image_kind(int type, int64_t K, int64_t N);
/* THE PLAN of mul_mat(type, M, K, N) as ggml_hip_mul_mat_dev will run it (csrc/plan.cpp: the one place where a product's kernel is decided;
 * every launcher consumes the same structure).  No device is needed to ask.  What the row split over several GPUs stands on
 * (Ggml.cs:6665-6672: contiguous row ranges, each computed independently) is visible here and tested without a GPU:
 *     tree_id -- the ORDER OF AN ELEMENT'S ADDITIONS (arithmetic form of a block term and of the min term, number of partial sums, how K is
 *     divided among them) -- is a function of (type, K, N) and never of M,
 * so a row shard computes, bit for bit, the matching columns of the unsplit product; family / form / tiles may follow M.  The one
 * exception carries GGML_HIP_PLAN_WIDE: planes beyond 32-bit buffer offsets (> 4 GiB per plane) are served by the int8 family. */
enum {  /* ggml_hip_mm_plan_t.family */
    GGML_HIP_MMF_GEMV_FUSED = 1, GGML_HIP_MMF_GEMV_ROWS = 2, GGML_HIP_MMF_K3S_MX = 3, GGML_HIP_MMF_K3S_I8 = 4, GGML_HIP_MMF_K3P_MX = 5,
    GGML_HIP_MMF_K3P_I8 = 6, GGML_HIP_MMF_MX = 7, GGML_HIP_MMF_F16 = 8, GGML_HIP_MMF_I8 = 9, GGML_HIP_MMF_DENSE = 10,
    GGML_HIP_MMF_DENSE_GEMV = 11, GGML_HIP_MMF_DENSE16 = 12, GGML_HIP_MMF_DENSE32 = 13
};
enum {  /* ggml_hip_mm_plan_t.flags */
    GGML_HIP_PLAN_WIDE = 1,            /* the 32-bit-offset exception applied */
    GGML_HIP_PLAN_EPILOGUE_FUSED = 2,  /* an add / scale node behind the product runs in the kernel's store phase */
    GGML_HIP_PLAN_PERSISTENT = 4, GGML_HIP_PLAN_Q8K = 8,
    GGML_HIP_PLAN_NEEDS_WORK = 16,     /* the product needs a work buffer of ggml_hip_mul_mat_work_size bytes */
    GGML_HIP_PLAN_MIN_PIECES = 32      /* INIT writes image 0 AND the bf16 piece planes of d * sum (K3p-int8 behind Q5_1 / Q4_1 / Q5_K): image_kind 0 + 64 */
};
typedef struct ggml_hip_mm_plan_t {
    int32_t  family, image_kind, form;       /* which kernel, what INIT writes (-1 nothing, 0..3 K1's images, 0 + 64 image 0 with the min-term piece planes, 32 / 33 dense panels), which instantiation */
    uint32_t tree_id;                        /* hash of (arith, ksplit, kstyle, kunit): what fixes an element's bits */
    int32_t  arith, ksplit, kstyle, kunit;   /* kstyle: 0 one chain over K, 1 stage sets taken in turn, 2 contiguous ranges, 3 interleaved workers */
    int32_t  tile_m, tile_n, waves, tiles_per_wave;
    int64_t  workgroups;
    int32_t  flags;
} ggml_hip_mm_plan_t;
int    ggml_hip_mm_plan(int type, int64_t M, int64_t K, int64_t N, ggml_hip_mm_plan_t *out);
/* TEST HOOK, per calling thread (a host thread that never calls it is never affected): which matrix-core kernel serves
 * N > 8 on THIS thread's calls -- 0 automatic (by type, N and K), 1 int8 MFMA
 * (gemm_q.hip), 2 f16 MFMA (gemm_q16.hip), 3 MX (gemm_qmx.hip; for Q5_0 / Q8_0 its two-digit form, which needs the
 * weight to have been uploaded while 3 was in force -- the digit planes are not built otherwise).  Same results within
 * the documented tolerance whichever runs; -DGGML_HIP_DEV builds read GGML_HIP_GEMM=i8|f16|mx as the initial value. */
void   ggml_hip_debug_force_gemm(int which);
/* Step 1 alone with an explicit layout: every src1 row -> Q8_0 (quantize_row_q8_0, Ggml.cs:733-762, the loop of
 * Ggml.cs:6641-6654) written as image `image_kind` (see above) into d_work.  image_kind + 16 (kinds 0..2, K % 256 == 0):
 * the Q8_K rule of the k-quant extension instead (one scale per 256 elements; see GGML_HIP_TYPE_Q5_K).  image_kind + 64 (kind 0 only,
 * K >= 256): beside image 0, d * (float)sum(q) of every block -- the Q8_1 s0 + s1 of Ggml.cs:820-821 -- as three bf16 pieces that sum to it
 * exactly, in the half of the image region image 0 leaves free (same work size): what ggml_hip_act_image_kind returns for Q5_1 / Q4_1
 * weights wherever the min terms run as a matrix product of their own (gemm_qmp.hip, gemm_q8s.hip; the row ranges are the plan's --
 * ask ggml_hip_act_image_kind / ggml_hip_mm_plan, they are not restated here). */
int    ggml_hip_quantize_act_dev(const float *d_src1, int64_t N, int64_t K, int64_t ld1, void *d_work, size_t work_bytes,
                                 int image_kind, void *stream);

/* ---------------- neighbours of the path (SURVEY.md 8(f) "next") ----------------
 * ggml_compute_forward_cpy -> ggml_compute_forward_dup_f32 / _dup_f16, quantizing branch (Ggml.cs:8659-8663,
 * 4339-4363, 3935-3966): src0 F32 or F16 with contiguous rows, dst a contiguous Q4_0 / Q4_1 / Q4_2 / Q5_0 / Q5_1 / Q8_0 tensor with
 * the same element count.  This is the only public way to produce a quantized tensor in the reference; on the device
 * it uses the intended quantize_row_q4_0 (== _reference), not the broken AVX packNibbles path.
 * Same offload convention as Seam 1 (acts for ith == 0, COMPUTE phase). */
int ggml_hip_compute_forward_cpy(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 struct ggml_tensor *dst);
/* ggml_compute_forward_add for a quantized src0 = ggml_compute_forward_add_q_f32 (Ggml.cs:4797-4906):
 * dst row = quantize_row_q(dequantize_row_q(src0 row) + src1 row); src1 F32, dst the type and shape of src0;
 * for an F32 src0 = ggml_compute_forward_add_f32 (Ggml.cs:4622-4682), same-shape contiguous operands, bit-exact. */
int ggml_hip_compute_forward_add(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 const struct ggml_tensor *src1, struct ggml_tensor *dst);
/* The f32 element-wise neighbours of mul_mat in a transformer block; contiguous F32 tensors, same
 * offload convention as Seam 1; inside a graph scope operands and results stay in HBM (see ggml_hip_graph_begin).
 *   mul      ggml_compute_forward_mul_f32      Ggml.cs:5007-5035   dst = src0 * src1, same shape           (bit-exact)
 *   scale    ggml_compute_forward_scale_f32    Ggml.cs:6746-6778   dst *= *(float *)src1->data, IN PLACE: dst is a view
 *                                                                  of src0 (ggml_scale_impl Ggml.cs:8265)     (bit-exact)
 *   rms_norm ggml_compute_forward_rms_norm_f32 Ggml.cs:5858-5920   y = x / sqrt(mean(x^2) + 1e-6), squares summed in f64
 *                                                                  (only the order of the f64 additions differs) */
int ggml_hip_compute_forward_mul(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 const struct ggml_tensor *src1, struct ggml_tensor *dst);
int ggml_hip_compute_forward_scale(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                   const struct ggml_tensor *src1, struct ggml_tensor *dst);
int ggml_hip_compute_forward_rms_norm(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                      struct ggml_tensor *dst);
/*   silu     ggml_compute_forward_silu_f32     Ggml.cs:5705-5748   the GGML_SILU_FP16 build (GGMLSharp.csproj:9): argument
 *                                                                  rounded to half, y = half(silu(x)) widened; the table
 *                                                                  of Ggml.cs:1455-1471 indexed by bit pattern (SURVEY A2,
 *                                                                  intent).  In-place form (ggml_silu_inplace): dst is a
 *                                                                  view of src0.                              (bit-exact) */
int ggml_hip_compute_forward_silu(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                  struct ggml_tensor *dst);
/* ---------------- fused neighbours (adjacent ops as prologues / epilogues of mul_mat) ----------------
 * Two graph nodes served by one call; BOTH nodes' data are produced (the reference's contract: every node's data is in host
 * memory after ggml_graph_compute), each value by the reference's own operation sequence -- bit for bit what the separate
 * seams give.  The host's node loop (Ggml.cs:3539-3704) calls one of these for node i and skips node i + 1 when it sees the
 * pair; every pair has the unfused seams as its fallback.
 *   rms_norm_mul   norm_dst = rms_norm(x), mul_dst = norm_dst * g                 one launch (fused.hip)
 *   silu_mul       silu_dst = silu(a),     mul_dst = silu_dst * b  (SwiGLU gate)  one launch (fused.hip)
 *   norm_mul_mat   the three (or, with an add node, four) nodes rms_norm, mul, mul_mat [, add] as ONE launch for N <= 4
 *   mul_mat_add    mm_dst = mul_mat(src0, src1), add_dst = mm_dst + addend        the add is applied to the accumulators in
 *   mul_mat_scale  mm_dst = scale_dst = mul_mat(src0, src1) * scalar (in place)   the store phase of the mat-mul kernels
 * (epilogue forms exist in the fused mat-vec, the MX mat-mat forms, the batched-decode forms and K3p; whether the form that serves a
 * given (weight, N) has one is ggml_hip_mul_mat_epilogue_fused's answer -- the plan's MM_FLAG_EPILOGUE_FUSED -- and elsewhere the node's
 * own kernel runs behind the mat-mul inside the same call). */
int ggml_hip_compute_forward_rms_norm_mul(const struct ggml_compute_params *params, const struct ggml_tensor *x,
                                          const struct ggml_tensor *g, struct ggml_tensor *norm_dst, struct ggml_tensor *mul_dst);
int ggml_hip_compute_forward_silu_mul(const struct ggml_compute_params *params, const struct ggml_tensor *a,
                                      const struct ggml_tensor *b, struct ggml_tensor *silu_dst, struct ggml_tensor *mul_dst);
int ggml_hip_compute_forward_mul_mat_add(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                         const struct ggml_tensor *src1, struct ggml_tensor *mm_dst,
                                         const struct ggml_tensor *addend, struct ggml_tensor *add_dst);
int ggml_hip_compute_forward_mul_mat_scale(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                           const struct ggml_tensor *src1, struct ggml_tensor *mm_dst,
                                           const struct ggml_tensor *scalar, struct ggml_tensor *scale_dst);
/* rms_norm -> mul -> mul_mat [-> add]: the norm in front of a projection and the residual behind it, for decode-sized
 * batches ONE launch (x: the norm's operand, g: the mul's other operand; addend / add_dst NULL when no add node follows) */
int ggml_hip_compute_forward_norm_mul_mat(const struct ggml_compute_params *params, const struct ggml_tensor *x,
                                          const struct ggml_tensor *g, struct ggml_tensor *norm_dst, struct ggml_tensor *mul_dst,
                                          const struct ggml_tensor *src0, struct ggml_tensor *mm_dst,
                                          const struct ggml_tensor *addend, struct ggml_tensor *add_dst);
/* Several MUL_MAT nodes with the SAME src1 (the q / k / v or gate / up projections), optionally with the rms_norm -> mul pair
 * that produces that src1 in front (pro_x / pro_g / pro_norm as x / g / norm_dst above, src1 = the mul node; NULL: no pair):
 * ONE launch inside a graph scope for batches of up to 4 rows when every src0 is a cached quantized leaf of one type and K;
 * otherwise the nodes run through their own seams one after the other.  Every node's data is produced either way. */
int ggml_hip_compute_forward_mul_mat_multi(const struct ggml_compute_params *params, int n, const struct ggml_tensor *const *src0,
                                           const struct ggml_tensor *src1, struct ggml_tensor *const *dst,
                                           const struct ggml_tensor *pro_x, const struct ggml_tensor *pro_g,
                                           struct ggml_tensor *pro_norm);
/* Device form of the prologue + epilogue: d_norm = rms_norm(d_x), d_y = d_norm * d_g (both [N][K] contiguous), then the
 * product of w and d_y with the epilogue `mode` (0 none).  ggml_hip_norm_mul_mat_fused: 1 when it is one launch. */
int ggml_hip_norm_mul_mat_dev(const ggml_hip_weight *w, const float *d_x, int64_t ld_x, const float *d_g, int64_t ld_g, int64_t N,
                              float *d_norm, float *d_y, float *d_dst, int64_t ldd, void *d_work, size_t work_bytes, int mode,
                              const float *d_addend, int64_t ld_add, float *d_dst2, int64_t ldd2, float scale, void *stream);
int ggml_hip_norm_mul_mat_fused(const ggml_hip_weight *w, int64_t N);
/* Several weight matrices behind ONE activation matrix -- the q / k / v or the gate / up projections of a transformer block
 * (each its own MUL_MAT node with the same src1, Ggml.cs:6714) -- as one launch for N <= 4: 2..4 resident matrices of one
 * quantized type and K; dst[i] receives matrix i's product ([N][M_i], row stride ldd[i]); every row is bit for bit what
 * ggml_hip_mul_mat_dev gives.  With d_g the launch also computes the rms_norm -> mul pair in front (d_src1 is then the
 * norm's input x; d_norm / d_y receive both nodes' data, as in ggml_hip_norm_mul_mat_dev).  ggml_hip_mul_mat_multi_fused
 * says whether the form exists for these matrices and N (else: one call per matrix). */
int ggml_hip_mul_mat_multi_fused(const ggml_hip_weight *const *w, int n_w, int64_t N);
int ggml_hip_mul_mat_multi_dev(const ggml_hip_weight *const *w, int n_w, const float *d_src1, int64_t ld1, int64_t N,
                               float *const *d_dst, const int64_t *ldd, const float *d_g, int64_t ld_g, float *d_norm,
                               float *d_y, void *stream);
/* The same for a batch of any size, with the scratch a batch needs (ggml_hip_mul_mat_work_size(type, K, N) bytes): src1 is
 * quantized once -- the INIT phase (Ggml.cs:6641-6654) is the same for every matrix of one type and K -- and the 1..4 matrices
 * follow, in ONE launch where the library has the form (5 <= N <= 64, Q4_0 / Q4_1, K >= 2048: three 4096-row projections fill
 * the chip that one of them half uses), else one COMPUTE after the other behind the shared image.  Every row is bit for bit
 * what ggml_hip_mul_mat_dev gives for that matrix. */
int ggml_hip_mul_mat_multi_work_dev(const ggml_hip_weight *const *w, int n_w, const float *d_src1, int64_t ld1, int64_t N,
                                    float *const *d_dst, const int64_t *ldd, void *d_work, size_t work_bytes, void *stream);
/* The pair kernel alone on contiguous device rows: d_norm = rms_norm(d_x) (Ggml.cs:5858-5920), d_y = d_norm * d_g. */
int ggml_hip_rms_norm_mul_rows_dev(const float *d_x, const float *d_g, float *d_norm, float *d_y, int64_t nrows, int64_t k, void *stream);
/* Device form of the epilogue: mode 1 add (d_dst keeps the product, d_dst2 = product + d_addend), mode 2 scale (d_dst =
 * product * scale), mode 0 = ggml_hip_mul_mat_dev. */
int ggml_hip_mul_mat_epilogue_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, float *d_dst, int64_t ldd,
                                  void *d_work, size_t work_bytes, int mode, const float *d_addend, int64_t ld_add, float *d_dst2,
                                  int64_t ldd2, float scale, void *stream);
int ggml_hip_mul_mat_epilogue_fused(const ggml_hip_weight *w, int64_t N);   /* 1: the kernel form serving N applies it itself */

/* Device forms.  src_type F32 or F16; source rows ld elements apart; blocks of all rows contiguous. */
int ggml_hip_quantize_rows_src_dev(int type, int src_type, const void *d_x, int64_t ld, int64_t nrows, int64_t k,
                                   void *d_blocks, void *stream);
int ggml_hip_add_q_f32_rows_dev(int type, const void *d_blocks_in, const float *d_x, int64_t nrows, int64_t k,
                                void *d_blocks_out, void *stream);

/* ---------------- several devices, one process (SURVEY 8(b) "n_devices", 8(e)) ----------------
 * Row split of one weight matrix over the device slots with the reference's thread partition (Ggml.cs:6665-6672: dr =
 * ceil(M / G), slot g owns rows [dr*g, min(dr*(g+1), M))).  ggml_hip_mul_mat_split_dev: d_src1[g] / d_dst[g] are slot g's
 * device buffers (src1 [N][ld1] replicated, dst [N][ldd >= M]); every slot computes its rows on its own stream and writes
 * them as columns [r0, r1) of ITS dst (the kernels take a row stride: no [G][N][Ms] intermediate, no re-layout pass); the
 * exchange then completes every slot's dst.  Stream-ordered on the slots' streams: ggml_hip_sync_slots() waits.
 * Exchange forms (ggml_hip_set_exchange): 0 = peer DMA over xGMI, one strided 2-D copy per (slot, peer) -- the default;
 * 1 = RCCL ncclAllGather of contiguous shards + the re-layout kernel below (librccl is loaded at run time; needs distinct
 * devices); 2 (r4) = no exchange pass: every slot's GEMM stores its rows into EVERY slot's dst from its own store phase
 * (ggml_hip_mul_mat_push_dev below; needs every device to reach every other's memory, else the call runs form 0).  All three only
 * move data: identical bits.  Every element equals the single-device result bit for bit (the kernel form is a function of N, K and
 * the type, never of M). */
typedef struct ggml_hip_split_weight ggml_hip_split_weight;
int  ggml_hip_split_weight_upload(int type, const void *host_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                                  ggml_hip_split_weight **out);
void ggml_hip_split_weight_free(ggml_hip_split_weight *w);
int  ggml_hip_split_weight_rows(const ggml_hip_split_weight *w, int slot, int64_t *row_begin, int64_t *row_end);
int  ggml_hip_mul_mat_split_dev(const ggml_hip_split_weight *w, const float *const *d_src1, int64_t N, int64_t ld1,
                                float *const *d_dst, int64_t ldd);
int  ggml_hip_set_exchange(int mode);
int  ggml_hip_sync_slots(void);
int  ggml_hip_debug_rccl_selftest(void);             /* the RCCL exchange form on slot 0 alone (one rank), bytes checked */
/* Device memory of a slot for hosts without their own HIP binding (the C# host): plain hipMalloc / hipMemcpy. */
void *ggml_hip_slot_malloc(int slot, size_t bytes);
void  ggml_hip_slot_free(int slot, void *p);
int   ggml_hip_slot_upload(int slot, void *d_dst, const void *host_src, size_t bytes);
int   ggml_hip_slot_download(int slot, void *host_dst, const void *d_src, size_t bytes);
/* One process PER device, direct exchange (ggmlsharp_amd/dist.py, exchange "push"): every rank allocates its reference-
 * layout dst [N][M] with ggml_hip_ipc_alloc, ships the 64-byte handle to its peers (any transport), opens theirs, and after
 * computing its rows stores them as columns [col0, col0 + Ms) of EVERY rank's dst with one kernel (d_peers: HOST array of
 * n_peers <= 16 device pointers, NULL entries skipped; compute units store over xGMI, one hop, all links at once, final
 * layout -- SURVEY 8(e) "epilogue peer-writes").  The caller orders consumers behind a barrier of its own. */
int ggml_hip_ipc_alloc(size_t bytes, void **d_ptr, uint8_t *handle64);
int ggml_hip_ipc_open(const uint8_t *handle64, void **d_ptr);
int ggml_hip_ipc_close(void *d_ptr);
int ggml_hip_ipc_free(void *d_ptr);
int ggml_hip_push_columns_dev(const float *d_shard, int64_t lds, int64_t N, int64_t Ms, float *const *d_peers, int n_peers,
                              int64_t ldd, int64_t col0, void *stream);
/* The product AND the exchange in one call (r4; SURVEY 8(e): "epilogue peer-writes straight into each peer's final [N][M] buffer"):
 * this rank's rows of W (Ggml.cs:6665-6672) against all of src1, every element stored as column col0 + m of EVERY rank's reference-layout
 * dst [N][ld_total] (Ggml.cs:6692-6697).  d_peers: HOST array of n_peers <= 16 device pointers to the buffers' BASES, this rank's own at
 * index `own`, NULL entries skipped.  Where the kernel form that serves this weight at N rows has the store-phase exchange -- the staged MX
 * forms, K3p and (r5) the batched-decode forms, up to 8 destinations: ggml_hip_mul_mat_push_fused says so -- the GEMM's store phase writes to all of them (no shard pass, no
 * second launch); otherwise the product lands in this rank's buffer and ggml_hip_push_columns_dev's kernel follows.  Same bytes either
 * way.  The caller orders consumers behind a barrier of its own. */
int ggml_hip_mul_mat_push_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, float *const *d_peers, int n_peers,
                              int own, int64_t ld_total, int64_t col0, void *d_work, size_t work_bytes, void *stream);
int ggml_hip_mul_mat_push_fused(const ggml_hip_weight *w, int64_t N, int n_peers);
/* One process PER device (torch.distributed / RCCL ranks, ggmlsharp_amd/dist.py): after an all-gather of per-rank dst
 * shards ([G][N][Ms], rank-major) produce the reference layout [N][G*Ms -> M] (SURVEY.md 8(e) "layout catch"); rows of
 * the last rank beyond M are dropped. */
int ggml_hip_relayout_gathered_dev(const float *d_gathered, int G, int64_t N, int64_t Ms, float *d_dst,
                                   int64_t M, int64_t ldd, void *stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GGML_HIP_EXT_H */
