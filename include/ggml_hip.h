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
 * This header is the DROP-IN CORE -- what SURVEY.md 8(b) lists and the C# patch of INTEGRATION.md binds: lifecycle, the host pool,
 * Seam 1 with its invalidation hooks and the graph scope, Seam 2, resident weights and the two-phase product on resident data.
 * Everything else the library exports (the neighbours of the path and their fused forms, named / output-only scopes, several
 * devices, IPC exchange, the k-quant extension type, test hooks) is declared in ggml_hip_ext.h, which includes this file.
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
 * d_work: device scratch of ggml_hip_mul_mat_work_size() bytes (the reference's wdata, Ggml.cs:3365-3378).  REQUIRED whenever
 *         ggml_hip_mul_mat_work_size(type, K, N) is not 0 -- also for F32 weights above 256 src1 rows, where the reference needs no
 *         wdata (Ggml.cs:3371-3373) and this library takes 6 B per src1 element: src1 as three bf16 pieces for the matrix cores.  A
 *         missing or too small buffer is GGML_HIP_ERR_ARG; no other kernel is chosen in its place (the same product gives the same
 *         bits whoever calls it).  The Seam-1 entry allocates its own scratch: a C# caller never sees this.
 * Alignment: d_src1 must be 16-byte aligned and ld1 a multiple of 4 for quantized weights and for the matrix-core forms of dense
 *         weights (F16 above 4 src1 rows, F32 above 256: the INIT kernels read rows in 16-byte pieces); GGML_HIP_ERR_SHAPE otherwise.
 *         The device the weight lives on is made current for the launch. */
size_t ggml_hip_mul_mat_work_size(int type, int64_t K, int64_t N);
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
 * dequantize: Q4_0 (886-910), Q4_1 (962-987), Q5_0 (1025-1061), Q8_0 (1104-1122).  Bit-exact.
 * (Both also take the k-quant extension types of ggml_hip_ext.h -- GGML_HIP_TYPE_Q4_K / Q5_K / Q6_K, k % 256 == 0, rows 16-byte aligned --
 * which the reference does not have: see there for what "exact" means for them.) */
int ggml_hip_quantize_rows_dev(int type, const float *d_x, int64_t nrows, int64_t k, void *d_blocks, void *stream);
int ggml_hip_dequantize_rows_dev(int type, const void *d_blocks, int64_t nrows, int64_t k, float *d_y, void *stream);
/* Host-pointer forms with exactly the slot signatures' argument meaning (x, y, k) / (n, s, vx, vy);
 * they copy to the device, run the kernel, and copy back. */
int ggml_hip_quantize_row(int type, const float *x, void *y, int k);
int ggml_hip_dequantize_row(int type, const void *x, float *y, int k);
int ggml_hip_vec_dot(int type, int n, float *s, const void *vx, const void *vy);  /* vy in vec_dot_type blocks */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GGML_HIP_H */
