// plan.h -- ONE decision per product: which kernel family and form serves mul_mat(type, M, K, N), and what that fixes.
//
// Every launcher consumes an mm_plan (no second decision inside the .hip files); api.cpp's image / epilogue questions are answered
// from the same plan; ggml_hip_mm_plan (include/ggml_hip_ext.h) hands it to callers and to the CPU test that sweeps
// type x K x N x M and asserts the invariant the multi-GPU path stands on:
//
//     the ORDER OF AN ELEMENT'S ADDITIONS (tree_id) is a function of (type, K, N) -- never of M.
//
// A row shard therefore computes, bit for bit, the matching columns of the unsplit product.  Only the geometry (tile height, tiles
// per workgroup, persistent grids) follows M.  The one stated exception carries a flag: planes beyond 32-bit buffer offsets
// (> 4 GiB per plane) are served by the int8 family whatever the type (MM_FLAG_WIDE).
#pragma once
#include <stdint.h>

enum mm_family {
    MMF_NONE = 0,
    MMF_GEMV_FUSED = 1,   // gemv.hip K2f / K2: INIT + COMPUTE in one launch, N <= 8
    MMF_GEMV_ROWS = 2,    // gemv.hip two-step form on K1's planes (Q4_2 at 9..16 rows where K < 2048; the COMPUTE-only entry for N <= 8)
    MMF_K3S_MX = 3,       // gemm_qmx.hip K3s: stage-free batched decode, MX (Q4_0, Q4_1)
    MMF_K3S_I8 = 4,       // gemm_q8s.hip: the same on the int8 cores (Q8_0, Q5_0, Q5_1 / Q5_K / Q4_K, Q4_2 / Q6_K; r5: Q4_1 from 65 rows)
    // (r5: 3 | 5 for Q4_0 and 4 | 6 are ONE summation tree each -- the same tree_id -- and plan_mul_mat picks between them by M where both serve)
    MMF_K3P_MX = 5,       // gemm_qmp.hip K3p: prompt-sized batches, MX (Q4_0)
    MMF_K3P_I8 = 6,       // gemm_qmp.hip K3p on the int8 cores (Q8_0, Q5_0, Q5_1, Q4_1; r5: Q4_2 and Q6_K in its form)
    MMF_MX = 7,           // gemm_qmx.hip staged forms
    MMF_F16 = 8,          // gemm_q16.hip staged forms
    MMF_I8 = 9,           // gemm_q.hip staged forms
    MMF_DENSE = 10,       // dense.hip tile kernels (f32 fma chain in k order)
    MMF_DENSE_GEMV = 11,  // dense.hip mat-vec form
    MMF_DENSE16 = 12,     // dense16.hip F16 x F32 on the f16 cores
    MMF_DENSE32 = 13,     // dense16.hip K10d: F32 x F32 as split bf16
};

// how K is divided among the partial sums of an element
enum mm_kstyle { MMK_CHAIN = 0, MMK_STAGE_SETS = 1, MMK_RANGES = 2, MMK_WORKERS = 3 };

enum { MM_FLAG_WIDE = 1, MM_FLAG_EPILOGUE_FUSED = 2, MM_FLAG_PERSISTENT = 4, MM_FLAG_Q8K = 8, MM_FLAG_NEEDS_WORK = 16,
       MM_FLAG_MIN_PIECES = 32 /* INIT writes image 0 AND the three bf16 piece planes of d * sum (K3p-int8, min-term types) */ };

// forms of the staged MX family (gemm_qmx.hip launch_typed): <WMT, WNT, WGM, WGN, KB, FB, KSP, VS>
enum mx_form {
    MXF_256x128 = 0,      // <2,4,4,1,4,2>       unsplit, 8 tiles per wave (the headline form)
    MXF_256x128_ALT,      // <4,2,2,2,4,1>       developer A/B only
    MXF_N32_H64,          // <1,1,2,1,4,2,4>     up to 32 rows, four-way, 64-row tiles
    MXF_N32_H32,          // <1,1,1,1,4,2,4>
    MXF_S4_H128,          // <1,2,4,1,4,FB,4>    four-way, 128-row tiles (Q4_0)
    MXF_S4_H64,           // <1,2,2,1,4,FB,4>
    MXF_S4_H32,           // <1,2,1,1,4,FB,4>
    MXF_S2V2_H64,         // <1,2,2,1,4,2,2,2>   four-way tree as two wave groups x two banked passes
    MXF_S2_H128,          // <1,2,4,1,4,2,2>     two-way
    MXF_S2_H64,           // <1,2,2,1,4,2,2>
    MXF_128x128,          // <2,2,2,2,4,2>       unsplit
    MXF_64x64,            // <1,1,2,2,4,2>       unsplit, one tile per wave (Q4_0)
    MXF_128x64,           // <1,2,4,1,4,2>       unsplit
};
// forms of the staged f16 family (gemm_q16.hip): <WMT, WNT, WGM, WGN, KB, KSP>
enum f16_form { F16F_N32_H64 = 0, F16F_N32_H32, F16F_S4_H128, F16F_S4_H64, F16F_S4_H32, F16F_S2_H128, F16F_256x128, F16F_64x64, F16F_128x64 };
// staged int8 family (gemm_q.hip): <IT, JT>
enum i8_form { I8F_64x64 = 0, I8F_128x128 };
// dense16.hip F16 forms
enum d16_form { D16F_S_256x128 = 0, D16F_S_128x128, D16F_256x128, D16F_S4_H128, D16F_S4_H32, D16F_V2_128x128, D16F_S2_128x128, D16F_S2_128x64, D16F_128x128 };
// dense.hip
enum dense_form { DNF_TILE = 0, DNF_BIG = 1, DNF_KSPLIT = 2 /* F32, 5..256 rows (plan.cpp plan_dense): 32 x 32 tiles, K over the workgroup's eight waves */ };

// K3p (gemm_qmp.hip): k-blocks of a wave's scale table -- 80 fit whole (8 waves x 80 x 256 B = the 160 KB of LDS: K <= 20480); beyond that the
// K loop refills slices of at most 78 rows (the int8 loop keeps one spare row behind the last wave's slice), four slices at most (K <= 79872)
constexpr int K3P_TABLE_ROWS = 80, K3P_SLICE_ROWS = 78, K3P_MAX_SLICES = 4;

struct mm_plan {
    int family;           // mm_family
    int image;            // what INIT writes: 0 int8 planes, 1 / 2 f16 images, 3 bf6 image; -1 nothing (fused mat-vec, dense f32 direct);
                          //   32 f16 panels (dense16), 33 split-bf16 panels (dense32)
    int form;             // the family's form number (one template instantiation)
    // ---- what fixes an element's bits: functions of (type, K, N) only ----
    int arith;            // block-term / min-term arithmetic of the family and type
    int ksplit;           // partial sums per element, added in a fixed order at the end
    int kstyle;           // mm_kstyle
    int kunit;            // k-blocks (dense: k-steps) per stage / per range
    // ---- geometry: may follow M ----
    int tile_m, tile_n;   // output tile of a workgroup
    int waves;            // waves per workgroup
    int tiles_per_wave;
    int64_t wgs;          // workgroups launched
    int flags;            // MM_FLAG_*
    // family-specific launch parameters (K3s / K3p / q8s): k-blocks per wave, weight tiles per workgroup
    int nloc, wmt;
};

// the plan of mul_mat(type, M, K, N); ext_type = GGML_HIP_TYPE_Q5_K / _Q4_K for a k-quant weight living in the planar Q5_1 form (type = Q5_1),
// _Q6_K for one living in the planar Q4_2 form on int8 planes alone (type = Q4_2).
// one_call = the product is computed by one entry (ggml_hip_mul_mat_dev: the fused mat-vec exists); false = the COMPUTE-only entry.
mm_plan plan_mul_mat(int type, int ext_type, int64_t M, int64_t K, int64_t N, bool one_call = true);
uint32_t plan_tree_id(const mm_plan &p);
// the K1 image for (type, K, N) with no weight at hand (no M: the exception cannot apply)
int plan_image_kind(int type, int64_t K, int64_t N);
// thread-local developer / test switch (ggml_hip_debug_force_gemm): 0 auto, 1 int8, 2 f16, 3 MX
int plan_force_gemm();
void plan_set_force_gemm(int which);
