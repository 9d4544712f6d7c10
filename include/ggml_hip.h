/*
 * ggml_hip.h -- C-ABI boundary of the MI355X-native quantized mul_mat path for GGMLSharp.
 *
 * One shared library (libggml_hip.so) with plain pointers and sizes in every signature; this is
 * what the reference's C# would bind with [DllImport] (INTEGRATION.md shows the stub).
 * "Reference" below = kant2002/GGMLSharp; Ggml.cs = GGMLSharp/Ggml.cs, TypeDefs = GGMLSharp/TypeDefinitions.cs.
 *
 * Two seams of the reference are occupied:
 *   Seam 1  ggml_compute_forward_mul_mat (Ggml.cs:6714-6744), reached from the GGML_OP_MUL_MAT case of
 *           ggml_compute_forward (Ggml.cs:8649-8653)          -> ggml_hip_compute_forward_mul_mat
 *   Seam 2  the quantize_fns_t slots (TypeDefs:334-342, table Ggml.cs:219-290)
 *                                                             -> ggml_hip_{quantize,dequantize}_row, ggml_hip_vec_dot
 * plus the lifecycle hook the reference left dead (ggml_init_cublas, Ggml.cs:1499-1504) -> ggml_hip_init.
 *
 * Error behaviour: the reference has no error channel on this path (Debug.Assert only, which vanishes in
 * Release).  Every entry point here returns an int status (0 = ok) and ggml_hip_last_error() gives text.
 * There is NO CPU fallback: without a usable GPU every compute entry point returns GGML_HIP_ERR_NO_DEVICE.
 */
#ifndef GGML_HIP_H
#define GGML_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: this header IS its export list */
#endif

/* ---------------- status codes ---------------- */
enum ggml_hip_status {
    GGML_HIP_OK = 0,
    GGML_HIP_ERR_NO_DEVICE = -1,   /* no HIP device / runtime failure at init */
    GGML_HIP_ERR_TYPE = -2,        /* src0 type not supported as a weight (Q4_3 / Q8_1: Ggml.cs:248, 278-282) */
    GGML_HIP_ERR_SHAPE = -3,       /* violates the asserts of Ggml.cs:6026-6046 / 6222-6241 / 6477-6504 / 8228-8229 */
    GGML_HIP_ERR_ARG = -4,         /* null pointer, workspace too small, ... */
    GGML_HIP_ERR_RUNTIME = -5      /* a HIP call failed */
};

/* ---------------- types mirrored from the reference ---------------- */

/* TypeDefs:153-169 */
enum ggml_type {
    GGML_TYPE_F32 = 0,
    GGML_TYPE_F16 = 1,
    GGML_TYPE_Q4_0 = 2,
    GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q4_2 = 4,
    GGML_TYPE_Q4_3 = 5,
    GGML_TYPE_Q5_0 = 6,
    GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,
    GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_I8 = 10,
    GGML_TYPE_I16 = 11,
    GGML_TYPE_I32 = 12,
    GGML_TYPE_COUNT = 13
};

/* TypeDefs:172-220 (only the ids this path touches are named) */
enum ggml_op {
    GGML_OP_NONE = 0,
    GGML_OP_DUP = 1,
    GGML_OP_ADD = 2,
    GGML_OP_MUL = 4,
    GGML_OP_SILU = 17,
    GGML_OP_RMS_NORM = 19,
    GGML_OP_MUL_MAT = 20,
    GGML_OP_SCALE = 21,
    GGML_OP_CPY = 22,
    GGML_OP_COUNT = 39
};

/* TypeDefs:292-297 */
enum ggml_task_type { GGML_TASK_INIT = 0, GGML_TASK_COMPUTE = 1, GGML_TASK_FINALIZE = 2 };

/* EXTENSION, not a reference type: the reference's enum stops at Q8_1 / I32 (TypeDefs:153-169) and holds no k-quants
 * (SURVEY 8(a) row K), while BASELINE.json's north_star and config 4 name Q5_K.  Built to the PUBLISHED upstream format
 * (ggml k_quants, 2023-06: 176-byte super-blocks of 256 weights; activations by the Q8_K rule) as an unpinned extra: no oracle
 * exists in the reference, tests/np_kquants.py restates the published algorithm.  Accepted by ggml_hip_weight_upload /
 * _from_device / _download, ggml_hip_mul_mat{,_init,_compute}_dev, ggml_hip_mul_mat_work_size and
 * ggml_hip_dequantize_rows_dev only -- never inside a ggml_tensor (the reference cannot express the type). */
#define GGML_HIP_TYPE_Q5_K 113

#define GGML_MAX_DIMS 4
#define GGML_MAX_OPT 4
#define GGML_MAX_NODES 4096
#define GGML_MEM_ALIGN 16          /* Ggml.cs:16 */
#define GGML_DEFAULT_N_THREADS 4   /* Ggml.cs:22 */

/* TypeDefs:65-99, C# sequential layout, 176 bytes (SURVEY.md 8(b) offset table) */
struct ggml_tensor {
    int32_t  type;                  /*   0 ggml_type */
    int32_t  n_dims;                /*   4 */
    int64_t  ne[GGML_MAX_DIMS];     /*   8 elements per dim */
    uint64_t nb[GGML_MAX_DIMS];     /*  40 stride in bytes; nb[0] = block bytes for Q types */
    int32_t  op;                    /*  72 ggml_op */
    uint8_t  is_param;              /*  76 C# bool */
    uint8_t  _pad0[3];
    struct ggml_tensor *grad;       /*  80 */
    struct ggml_tensor *src0;       /*  88 */
    struct ggml_tensor *src1;       /*  96 */
    int64_t  opt[GGML_MAX_OPT];     /* 104 pointers stored as integers */
    int32_t  n_tasks;               /* 136 */
    int32_t  perf_runs;             /* 140 */
    int64_t  perf_cycles;           /* 144 */
    int64_t  perf_time_us;          /* 152 */
    void    *data;                  /* 160 */
    uint8_t  padding[8];            /* 168 */
};

/* TypeDefs:299-308 */
struct ggml_compute_params {
    int32_t type;                   /* ggml_task_type */
    int32_t ith, nth;
    size_t  wsize;
    void   *wdata;
};

/* Ggml.cs:55-87 */
int    ggml_hip_blck_size(int type);
size_t ggml_hip_type_size(int type);

/* ---------------- lifecycle (replaces the dead ggml_init_cublas hook, Ggml.cs:1499-1504) ----------------
 * The library keeps one context per DEVICE SLOT (streams, scratch, weight cache, graph-scope residency): no process-wide
 * device state.  ggml_hip_init(d) = one slot on device d.  ggml_hip_init_devices(n, ids) = n slots (ids NULL: devices
 * 0..n-1): Seam 1 then row-splits every mul_mat over the slots with the reference's own partition (Ggml.cs:6665-6672) --
 * the `n_devices` of SURVEY 8(b).  Two slots may name the same device (how the split is rehearsed on a one-GPU box).
 * Re-initialising on a different device set is refused (GGML_HIP_ERR_ARG) until ggml_hip_shutdown.
 * Threading: calls may arrive on any host thread.  A thread that called ggml_hip_bind_thread(s) runs every
 * seam on slot s alone, concurrently with threads bound to other slots (one graph per device); an unbound thread
 * (s = -1, the default) uses all slots. */
int         ggml_hip_device_count(void);             /* 0 when no GPU is visible; never fails */
int         ggml_hip_init(int device);
int         ggml_hip_init_devices(int n_devices, const int *device_ids);
int         ggml_hip_n_slots(void);
int         ggml_hip_slot_device(int slot);          /* HIP device ordinal of a slot, -1 if out of range */
int         ggml_hip_bind_thread(int slot);
void        ggml_hip_shutdown(void);                 /* frees the weight caches and internal buffers of every slot */
const char *ggml_hip_last_error(void);               /* thread-local, never NULL */
const char *ggml_hip_arch(void);                     /* gcnArchName of slot 0's device, "" before init */
/* The reference's context pool is ONE host allocation (NativeMemory.AlignedAlloc, Ggml.cs:1545) that every tensor's data
 * lives in.  Registering it (hipHostRegister) lets Seam 1 move src1 / dst by asynchronous DMA and overlap host -> device,
 * kernels and device -> host in chunks of src1 rows on three streams; unregistered (pageable) memory still works, without
 * the overlap.  Call unregister before freeing the pool (ggml_free, Ggml.cs:1584-1588): it waits for in-flight copies and
 * drops every cached / resident device copy made from that memory. */
int         ggml_hip_register_host_pool(void *ptr, size_t bytes);
int         ggml_hip_unregister_host_pool(void *ptr);

/* ---------------- Seam 1: ggml_compute_forward_mul_mat (Ggml.cs:6714-6744) ----------------
 * Host pointers in, host pointers out.  Follows the offload convention of the reference's own dead GPU
 * blocks (Ggml.cs:6510-6521): acts only for params->ith == 0 && params->type == GGML_TASK_COMPUTE and
 * returns GGML_HIP_OK immediately otherwise; loops the (i03, i02) slices (Ggml.cs:6566-6570).
 * src0 (the weights) is uploaded once, re-laid-out on the device and cached keyed by (src0->data, type, ne, nb, row
 * shard) -- only when src0 is a LEAF (op == GGML_OP_NONE): a src0 that a node computes is rebuilt every time, from the
 * device copy its producer left in the graph scope when there is one.  Every seam of this library that writes host
 * tensor memory drops the cache entries and resident copies overlapping what it wrote; a HOST that rewrites a leaf
 * (ggml_set_f32, a direct store through tensor->data) or frees it must call ggml_hip_invalidate{,_range}
 * (ggml_free gives no callback, Ggml.cs:1566-1601).  params->wdata is not used. */
int  ggml_hip_compute_forward_mul_mat(const struct ggml_compute_params *params,
                                      const struct ggml_tensor *src0,
                                      const struct ggml_tensor *src1,
                                      struct ggml_tensor *dst);
void ggml_hip_invalidate(const void *host_ptr);      /* drop every cached / resident device copy whose host range holds this byte */
void ggml_hip_invalidate_range(const void *host_ptr, size_t bytes);   /* ... that overlaps [host_ptr, host_ptr + bytes) */
void ggml_hip_invalidate_all(void);
/* Graph scope around the node loop of ggml_graph_compute (Ggml.cs:3539-3704), SURVEY 8(f) row 3.  Between begin and end
 * the device copy of every offloaded node's dst is kept (keyed by tensor->data): a later MUL_MAT whose src1 is that
 * tensor reads it from HBM instead of copying it host -> device again, and the device -> host copies of the node
 * results are synchronised once, in ggml_hip_graph_end -- after which every node's data is in host memory, as the
 * reference guarantees on return from ggml_graph_compute.  Calls nest; buffers are recycled across graphs. */
int  ggml_hip_graph_begin(void);
int  ggml_hip_graph_end(void);
/* Inside a scope the host copy of a node's result is OWED, not made at once (the copies of a scope go out together: one
 * launch per node is all a node costs the host).  A node the host computes itself between offloaded ones calls this for each
 * source it is about to read -- the range is paid and waited for when it is owed or still in flight, else nothing happens --
 * and ggml_hip_invalidate_range for what it wrote (a resident copy of that range is stale). */
int  ggml_hip_host_read(const void *host_ptr, size_t bytes);
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

/* ---------------- resident weights (device level) ---------------- */
typedef struct ggml_hip_weight ggml_hip_weight;      /* opaque: one 2-D weight matrix, re-laid-out on the device */

/* Upload rows [row_begin, row_end) of a [K = ne00, M = ne01] weight matrix whose rows are nb01 bytes apart
 * in reference block format (TypeDefs:236-290) from HOST memory.  A sub-range is a row shard
 * (the reference's own thread split is a contiguous row partition, Ggml.cs:6665-6672).
 * type: Q4_0, Q4_1, Q4_2, Q5_0, Q5_1, Q8_0, F32 or F16.  (Q4_2 / Q5_1: half scales as IEEE bit patterns -- the intent
 * policy of SURVEY D7; the C# as written stores them through numeric casts.  ne00 is a multiple of 32 for every
 * quantized type: the dot products run against 32-element Q8 blocks, Ggml.cs:1209-1211.) */
int  ggml_hip_weight_upload(int type, const void *host_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                            int64_t row_begin, int64_t row_end, void *stream, ggml_hip_weight **out);
/* Same, from DEVICE memory holding reference-format rows (no host round trip). */
int  ggml_hip_weight_from_device(int type, const void *dev_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                                 int64_t row_begin, int64_t row_end, void *stream, ggml_hip_weight **out);
/* Write the weight back as reference-format rows (nb01 = type_size * K / 32, contiguous) to HOST memory;
 * byte-identical to what was uploaded. */
int  ggml_hip_weight_download(const ggml_hip_weight *w, void *host_rows, void *stream);
void ggml_hip_weight_free(ggml_hip_weight *w);
int64_t ggml_hip_weight_rows(const ggml_hip_weight *w);
int64_t ggml_hip_weight_cols(const ggml_hip_weight *w);
int     ggml_hip_weight_type(const ggml_hip_weight *w);

/* ---------------- the hot path on resident data ----------------
 * dst[n * ldd + m] = sum_k deq(W[m,k]) * q8(X[n,k])   (ggml_compute_forward_mul_mat_q_f32, Ggml.cs:6440-6712)
 *   step 1 = INIT phase (Ggml.cs:6641-6654): every src1 row -> Q8_0 (quantize_row_q8_0, Ggml.cs:733-762),
 *            or Q8_1 for Q4_1 weights (Ggml.cs:781-823);
 *   step 2 = COMPUTE phase (Ggml.cs:6676-6698): integer block dot products with f32 block scales
 *            (ggml_vec_dot_q4_0_q8_0 Ggml.cs:1125-1162, _q5_0_q8_0 1258-1301, _q8_0_q8_0 1351-1381, _q4_1_q8_1 1165-1201).
 * For F32 / F16 weights it is ggml_compute_forward_mul_mat_f32 / _f16_f32 (Ggml.cs:5969-6178, 6180-6438).
 * d_src1: device f32 [N rows][K], row stride ld1 ELEMENTS; d_dst: device f32 [N][M], row stride ldd ELEMENTS
 * (the reference's dst layout: element (i01, ic) at ic*ne0 + i01, Ggml.cs:6692-6697).
 * d_work: device scratch of ggml_hip_mul_mat_work_size() bytes (the reference's wdata, Ggml.cs:3365-3378).
 * Alignment: for quantized weights d_src1 must be 16-byte aligned and ld1 a multiple of 4 (the INIT kernels read rows in
 * 16-byte pieces); GGML_HIP_ERR_SHAPE otherwise.  The device the weight lives on is made current for the launch. */
size_t ggml_hip_mul_mat_work_size(int type, int64_t K, int64_t N);
/* Which layout step 1 writes into d_work for this weight type, M rows, K and N (introspection for tests and profiling tools):
 * 0 = int8 planes (mat-vec and int8-MFMA kernels), 1 / 2 = f16 images (gemm_q16.hip), 3 = bf6 digit image (gemm_qmx.hip). */
int    ggml_hip_act_image_kind(int type, int64_t M, int64_t K, int64_t N);
/* Developer / test switch: which matrix-core kernel serves N > 8 -- 0 automatic (by type and grid size), 1 int8 MFMA
 * (gemm_q.hip), 2 f16 MFMA (gemm_q16.hip), 3 MX (gemm_qmx.hip; for Q5_0 / Q8_0 its two-digit form, which needs the
 * weight to have been uploaded while 3 was in force -- the digit planes are not built otherwise).  Same results within
 * the documented tolerance whichever runs; the environment variable GGML_HIP_GEMM=i8|f16|mx sets the initial value. */
void   ggml_hip_debug_force_gemm(int which);
/* Step 1 alone with an explicit layout: every src1 row -> Q8_0 (quantize_row_q8_0, Ggml.cs:733-762, the loop of
 * Ggml.cs:6641-6654) written as image `image_kind` (see above) into d_work.  image_kind + 16 (kinds 0..2, K % 256 == 0):
 * the Q8_K rule of the k-quant extension instead (one scale per 256 elements; see GGML_HIP_TYPE_Q5_K). */
int    ggml_hip_quantize_act_dev(const float *d_src1, int64_t N, int64_t K, int64_t ld1, void *d_work, size_t work_bytes,
                                 int image_kind, void *stream);
int    ggml_hip_mul_mat_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1,
                            float *d_dst, int64_t ldd, void *d_work, size_t work_bytes, void *stream);
/* The two steps separately (same arguments), so a harness can time the dominant kernel alone. */
int    ggml_hip_mul_mat_init_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1,
                                 void *d_work, size_t work_bytes, void *stream);
int    ggml_hip_mul_mat_compute_dev(const ggml_hip_weight *w, int64_t N, float *d_dst, int64_t ldd,
                                    const void *d_work, size_t work_bytes, void *stream);

/* ---------------- Seam 2: row functions (quantize_fns_t, TypeDefs:334-342) ----------------
 * Batched device variants: rows are contiguous, k elements each, blocks in reference format.
 * quantize: Q4_0 (Ggml.cs:334-377), Q4_1 (487-528), Q5_0 (609-653), Q8_0 (733-762), Q8_1 (781-823).
 * dequantize: Q4_0 (886-910), Q4_1 (962-987), Q5_0 (1025-1061), Q8_0 (1104-1122).  Bit-exact. */
int ggml_hip_quantize_rows_dev(int type, const float *d_x, int64_t nrows, int64_t k, void *d_blocks, void *stream);
int ggml_hip_dequantize_rows_dev(int type, const void *d_blocks, int64_t nrows, int64_t k, float *d_y, void *stream);
/* Host-pointer forms with exactly the slot signatures' argument meaning (x, y, k) / (n, s, vx, vy);
 * they copy to the device, run the kernel, and copy back. */
int ggml_hip_quantize_row(int type, const float *x, void *y, int k);
int ggml_hip_dequantize_row(int type, const void *x, float *y, int k);
int ggml_hip_vec_dot(int type, int n, float *s, const void *vx, const void *vy);  /* vy in vec_dot_type blocks */

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
 * (epilogue forms exist in the fused mat-vec, N <= 4, in the MX mat-mat of Q4_0 / Q4_1 -- N > 8, and 5..8 rows where K >= 2048 --
 * and in the batched-decode form of Q8_0, 5..64 rows with 2048 <= K <= 16384:
 * ggml_hip_mul_mat_epilogue_fused; elsewhere the node's own kernel runs behind the mat-mul inside the same call). */
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
 * devices).  Both only move data: identical bits.  Every element equals the single-device result bit for bit (the kernel
 * form is a function of N, K and the type, never of M). */
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
#endif /* GGML_HIP_H */
