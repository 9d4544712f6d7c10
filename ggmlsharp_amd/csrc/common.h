// common.h -- shared declarations for the gfx950 kernels and the C-ABI layer (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/ggml_hip_ext.h"

#define QK 32

// ---- reference block layouts (TypeDefinitions.cs:236-290); AoS, as they sit in a ggml_tensor ----
#pragma pack(push, 1)
struct block_q4_0 { float d; uint8_t qs[16]; };
struct block_q4_1 { float d; float m; uint8_t qs[16]; };
struct block_q4_2 { uint16_t d; uint8_t qs[8]; };               // 16 elements; d is an IEEE half (intent)
struct block_q5_0 { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; };
struct block_q5_1 { uint16_t d; uint16_t m; uint8_t qh[4]; uint8_t qs[16]; };   // d, m IEEE halves (D7, intent)
struct block_q8_0 { float d; int8_t qs[32]; };
struct block_q8_1 { float d; float s0; float s1; int8_t qs[32]; };
#pragma pack(pop)
static_assert(sizeof(block_q4_0) == 20, "Ggml.cs:76");
static_assert(sizeof(block_q4_1) == 24, "Ggml.cs:77");
static_assert(sizeof(block_q4_2) == 10, "Ggml.cs:78");
static_assert(sizeof(block_q5_0) == 22, "Ggml.cs:80");
static_assert(sizeof(block_q5_1) == 24, "Ggml.cs:81");
static_assert(sizeof(block_q8_0) == 36, "Ggml.cs:82");
static_assert(sizeof(block_q8_1) == 44, "Ggml.cs:83");
static_assert(sizeof(ggml_tensor) == 176, "TypeDefinitions.cs:65-99");
static_assert(offsetof(ggml_tensor, op) == 72 && offsetof(ggml_tensor, grad) == 80 &&
              offsetof(ggml_tensor, n_tasks) == 136 && offsetof(ggml_tensor, data) == 160, "layout");

// Rows of the weight planes are padded to a multiple of ROW_PAD (the widest weight tile of any kernel), rows of the
// activation planes to ACT_PAD (the widest activation tile).
#define ROW_PAD 256
#define ACT_PAD 256
// k-blocks: every plane (weights and activation scratch) is allocated for K_STAGE_PAD-aligned k-blocks, zero-filled
// past the real end (zero scale => the pad blocks add +0), plus K_LOOKAHEAD spare blocks on the weight side for the
// register prefetch of gemm_q16.hip.  Kernels that loop over the real nbk are unaffected.
#define K_STAGE_PAD 4
#define K_LOOKAHEAD 2
__host__ __device__ static inline int64_t pad_kblocks(int64_t nbk) { return (nbk + K_STAGE_PAD - 1) / K_STAGE_PAD * K_STAGE_PAD; }
static inline int64_t pad_rows(int64_t n) { return (n + ROW_PAD - 1) / ROW_PAD * ROW_PAD; }
static inline int64_t pad_act(int64_t n) { return (n + ACT_PAD - 1) / ACT_PAD * ACT_PAD; }

// ---- resident weight: planar (block-major) layout, see DESIGN.md "Data layout in HBM" ----
//   Q4_0/Q4_1/Q5_0/Q5_1: qs [nbk][Mpad][16] bytes of nibbles exactly as in the reference block
//   Q4_2          : the same plane; k-block b = the reference's 16-element blocks 2b (bytes 0..7) and 2b+1 (bytes 8..15),
//                   which is byte for byte the nibble order of a Q4_0 block (Ggml.cs:1217-1238 vs 1136-1150)
//   Q8_0          : qs [nbk][2][Mpad][16] int8, plane 0 = even elements, plane 1 = odd elements of the block
//   d   [nbk][Mpad] f32 (Q5_0 / Q5_1 / Q4_2: the half scale widened, exact; Q4_2: of its first 16-element block)
//   m   [nbk][Mpad] f32 (Q4_1, Q5_1: the min; Q4_2: the scale of its second 16-element block)
//   qh  [nbk][Mpad] u32 (Q5_0, Q5_1)
//   F32/F16: dense [Mpad][K] row-major copy
struct ggml_hip_weight {
    int      type;
    int64_t  M, K, Mpad, nbk;
    uint8_t *qs;
    uint32_t *qh;
    float   *d;
    float   *m;
    uint8_t *q6a;     // bf6 (e3m2) digit codes, [nbk][NF][Mpad][16 B] = the first 16 bytes of each 24-byte MFMA fragment
    uint8_t *q6b;     //   (element e at bits [6e, 6e+5]) and [nbk][NF][Mpad][8 B] = the last 8; NF = 1 (Q4_0, Q4_1: nib - 8)
                      //   or 2 (Q5_0, Q8_0: w = 16 * wh + wl, fragment 0 = wl, fragment 1 = wh) -- gemm_qmx.hip
    uint8_t *i8p;     // Q5_0 / Q5_1 / Q4_1 (the latter two unsigned, no - 16; r4: Q4_2 too, nib - 8): int8 operand planes of the int8 matrix cores, [nbk][2][Mpad][16 B] like Q8_0's qs -- plane h byte j =
                      //   element 2j + h, value (nib | bit << 4) - 16 (Ggml.cs:1285-1289); built once at upload (layout.hip), 1 B / weight
                      //   beside the 0.69 B / weight of the format, for prompt-sized batches (gemm_qmp.hip)
    void    *dense;
    uint8_t *p16;     // F16 only: k-panel-major copy [Kpad/8 + spare][Mpad][16 B] for dense16.hip
    uint8_t *p32;     // F32 only: the rows split exactly into three bf16 pieces, [Kpad/8 * 3 + spare][Mpad][16 B] (dense16.hip K10d): 6 B / weight
    uint32_t *gs;     // mat-vec side image (gemv.hip): the per-block words a 16-row tile needs beside its nibbles, tile-major
                      //   [Mpad/16][nbk][NP][16 rows] 4-byte words, planes in the order d, m, qh (NP = 1..3 by type).  A copy of
                      //   d / m / qh: in their [k-block][row] planes a tile's share of a k-block is one 64-byte piece per plane
                      //   (a fifth of the bytes cost a third of the mat-vec's time, tools/stream_floor.hip)
    uint8_t *mp3;     // Q5_1 / Q4_1 (and the Q5_K extension living in the Q5_1 form): the min plane split EXACTLY into three bf16 pieces,
                      //   [ceil(nbk / 8) * 3 (+ pad to whole pairs of k-groups)][Mpad][16 B]: plane 3 * (b / 8) + piece holds 8 consecutive k-blocks of a row --
                      //   the B operand of v_mfma_f32_32x32x16_bf16 for K3p's min-term product (gemm_qmp.hip); 0.19 B / weight
    uint8_t *khdr;    // k-quants only (Q6_K: 32 B per super-block, scales[16] + d): the 16 header bytes (d, dmin, scales[12]) of every super-block, [K/256][Mpad][16 B]
    int      ext_type; // 0, or GGML_HIP_TYPE_Q5_K / _Q4_K (type == Q5_1) / _Q6_K (type == Q4_2): the weight was uploaded as k-quant super-blocks and lives in the planar Q5_1 form (type == Q5_1)
    size_t   bytes;
    int      device;
    uint64_t uid;     // never reused: identifies the weight in cached launch graphs
};

// ---- activation scratch ("wdata"), planar like the weights; K1 writes one of two images into the `a8` region ----
//   int8 image (mat-vec kernel):  [nbk][2][Npad][16] int8, plane 0 = even elements of the block, plane 1 = odd
//   f16 images (gemm_q16.hip)  :  [nbk][4][Npad][16 B] f16, panel p = 2*kk + h, k-slot orders in gemm_q16.hip
//   bf6 image  (gemm_qmx.hip)  :  per k-block [2][Npad][16 B] then [2][Npad][8 B]: half 0 = digits ah, half 1 = digits al
//                                 of a = 16*ah + al, 32 bf6 codes per 24-byte fragment
//   ad [nbk][Npad] f32 block scales, as [nbk][Npad] i32 block sums of the quants (MFMA images: the float d * sum)
//   sp3 (int8 image + ACT_IMAGE_MIN_PIECES only): d * sum split exactly into three bf16 pieces, [ceil(nbk / 8) * 3][Npad][16 B] like
//       ggml_hip_weight::mp3 -- in the upper half of the a8 region, which the int8 image leaves unused
struct act_planes {
    int8_t  *a8;
    float   *ad;
    int32_t *as;
    int64_t  Npad;
    uint8_t *sp3;
};
#define ACT_IMAGE_MIN_PIECES 64      /* flag on image kind 0: K1 also writes sp3 (K3p-int8 with a min-term weight type; needs K / 32 >= 8) */
static inline size_t act_bytes(int64_t K, int64_t Npad) {
    const int64_t nbk = pad_kblocks(K / QK);
    return (size_t)nbk * 4 * Npad * 16 + (size_t)nbk * Npad * 4 * 2;
}
static inline act_planes act_carve(void *base, int64_t K, int64_t Npad) {
    const int64_t nbk = pad_kblocks(K / QK);
    act_planes p;
    p.a8 = (int8_t *)base;
    p.ad = (float *)((uint8_t *)base + (size_t)nbk * 4 * Npad * 16);
    p.as = (int32_t *)((uint8_t *)p.ad + (size_t)nbk * Npad * 4);
    p.Npad = Npad;
    p.sp3 = (uint8_t *)base + (size_t)nbk * 2 * Npad * 16;   // ((nbk / 8 rounded up) * 3 planes <= 2 nbk planes from two k-blocks on)
    return p;
}

// ---- cross-lane moves inside a group of 8 lanes as DPP modifiers (no LDS round trip, unlike __shfl_xor = ds_bpermute) ----
#ifdef __HIPCC__
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// K3s (gemm_qmx.hip / gemm_q8s.hip): workgroup -> (row tile, column tile), XCD-aware (speed only; a bijection whatever ntw and ncol are).
// Workgroups b and b + 8 share an XCD and its L2: the ncol column tiles of one row tile sit 8 apart in launch order, so the weight tile they
// all stream reaches that L2 once (with "row tile fastest" they were ntw launches apart and each fetched it again: what made the forms pay
// by M above 32 src1 rows).  Row tiles in whole groups of eight; the last ntw % 8 keep the plain order.
__device__ __forceinline__ void k3s_tile_of(int wg, int ntw, int ncol, int &rt, int &ct) {
    const int full = ntw & ~7;
    const int g = wg / (8 * ncol);
    if (g * 8 < full) { const int idx = wg - g * 8 * ncol; rt = g * 8 + (idx & 7); ct = idx >> 3; }
    else { const int w2 = wg - full * ncol, nt = ntw - full; rt = full + w2 % nt; ct = w2 / nt; }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }
#define DPP_XOR1 0xB1        /* quad_perm:[1,0,3,2] */
#define DPP_XOR2 0x4E        /* quad_perm:[2,3,0,1] */
#define DPP_HALF_MIRROR 0x141 /* lane i <-> 7 - i inside each group of 8: joins the two quads once they are uniform */
#define DPP_NEXT 0x101       /* row_shl:1: lane i reads lane i + 1 (inside a row of 16) */
// max / sum over the 8 lanes t = lane & 7 of a group, result in every lane
__device__ __forceinline__ float group8_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    return fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
}
__device__ __forceinline__ int group8_sum(int v) {
    v += dpp_i<DPP_XOR1>(v);
    v += dpp_i<DPP_XOR2>(v);
    return v + dpp_i<DPP_HALF_MIRROR>(v);
}
// ---- an f32 value as three bf16 pieces that sum to it exactly (truncation: 8 + 8 + 8 significand bits); dense16.hip K10d, gemm_qmp.hip ----
__device__ __forceinline__ void split3(float a, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
    const uint32_t u = __float_as_uint(a);
    const uint32_t b0 = u & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(b0);                       // exact
    const uint32_t b1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b1);                      // exact, at most 8 significant bits
    p0 = b0 >> 16; p1 = b1 >> 16; p2 = __float_as_uint(r2) >> 16;
    // r4: an infinity or a NaN.  As written above, inf - inf put a NaN into the remainder pieces and every output that touched an
    // infinite operand came out NaN where the reference's f32 product gives +-inf.  The value goes into the THIRD piece alone: of the six
    // kept piece products only a2 * b0 (b0 * a2) then sees it -- against the other operand's LEADING piece, which is zero only if that operand
    // is (inf * 0 is NaN in the reference too) -- while in the first piece it would meet remainder pieces that are zero for one operand in
    // 256 (inf * 0).  A NaN whose payload sits in the low 16 bits keeps a set quiet bit, so truncation cannot turn it into an infinity.
    // (Two infinities at the same k meet in no kept product: that one case gives NaN where the reference gives inf.)
    if ((u & 0x7F800000u) == 0x7F800000u) { p0 = 0; p1 = 0; p2 = (u >> 16) | ((u & 0x007FFFFFu) ? 0x0040u : 0u); }
}
#endif

// ---- developer A/B switches: compiled in only with -DGGML_HIP_DEV (tools/build_variant.sh); the product library reads
// no environment variable, so kernel selection in a host process never depends on its environment ----
#include <stdlib.h>
#ifdef GGML_HIP_DEV
static inline int dev_env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool dev_env_set(const char *name) { return getenv(name) != nullptr; }
static inline const char *dev_env_str(const char *name) { return getenv(name); }
#else
static inline int dev_env_int(const char *, int dflt) { return dflt; }
static inline bool dev_env_set(const char *) { return false; }
static inline const char *dev_env_str(const char *) { return nullptr; }
#endif

// ---- once per kernel AND device: hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel, and one
// process may drive several devices (ggml_hip_init_devices).  One object per kernel instantiation (function-local static).
#ifdef __cplusplus
#include <atomic>
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    hipError_t max_dynamic_lds(const void *kern, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
        return e;
    }
};
#endif

// ---- epilogue applied to the accumulators of a mat-mul kernel as they are stored (the add / scale node
// that follows a mul_mat node, fused into its store phase).  mode 0: none.  mode 1 (add, Ggml.cs:4622-4682): dst keeps the
// product, dst2[n][m] = product + addend[n][m] -- both nodes' data are materialised, one f32 add, bit for bit the separate
// kernel.  mode 2 (scale, Ggml.cs:6746-6778, in place: the scale node is a view of the product): dst[n][m] = product * scale.
// mode 3 (r4, the exchange of a row split fused into the store phase -- SURVEY 8(e): "epilogue peer-writes straight into each peer's final
// [N][M] buffer"): dst is this rank's own [N][M] buffer at its column offset, push[0 .. npush) are the SAME position in every peer's buffer
// (IPC-mapped, or other slots' buffers of one process); every element is stored to all of them with dst's row stride (Ggml.cs:6692-6697
// is the layout every rank must end with).  Kernels that apply it: the staged MX forms, K3p (MX and int8).
#define MM_PUSH_MAX 7
struct mm_epilogue {
    int mode;
    const float *addend; int64_t ld_add;
    float *dst2; int64_t ld2;
    float scale;
    int npush;
    float *push[MM_PUSH_MAX];
};

// ---- prologue of the fused mat-vec (prologue side): the activation row the kernel quantizes is not read
// but COMPUTED, y = (x * rms_scale(x)) * g -- the rms_norm -> mul pair in front of a mul_mat (Ggml.cs:5858-5920, 5007-5035) --
// with the reference's own operation sequence (f32 squares summed in f64 in the unfused kernel's order, one f32 multiply per
// node); workgroup 0 also writes both nodes' results (n_out = the norm, y_out = the product), [N][K] contiguous.
struct mm_prologue {
    const float *g; int64_t ld_g;
    float *n_out; float *y_out;
};

#ifdef __HIPCC__
// rms_norm's row scale 1 / sqrt(mean(x^2) + 1e-6) (Ggml.cs:5889-5915: f32 squares summed in f64), computed by ONE wave.  The
// reference adds the squares one after the other; any order of the f64 additions is within ~1e-16 of that, and ONE order is
// shared by every kernel that needs the scale (eltwise.hip, fused.hip, the mat-vec's prologue), so they agree bit for bit:
//   lane L holds four partial sums, partial q taking elements L + 64 (4 j + q), j = 0, 1, ... in order; lane total =
//   (p0 + p1) + (p2 + p3); lanes are joined inside rows of 16 by DPP (xor 1, xor 2, half mirror, row mirror -- after two
//   xor steps a quad is uniform, so a mirror is a swap of uniform groups), then rows by xor 16 and xor 32.
// Written for latency: a decode-sized norm is one wave working alone -- the loads of 32 iterations are in flight together (a
// plain loop cost one L2 round trip per iteration: 64 x ~100 ns at K = 4096), the f64 chain is 16 deep instead of 64, and
// only the last two lane steps go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)dpp_i<CTRL>((int)(uint32_t)u), hi = (uint32_t)dpp_i<CTRL>((int)(uint32_t)(u >> 32));
    return __builtin_bit_cast(double, (uint64_t)lo | ((uint64_t)hi << 32));
}
#define DPP_ROW_MIRROR 0x140  /* lane i <-> 15 - i inside each row of 16 */
__device__ __forceinline__ float rms_row_scale(const float *__restrict__ xr, int64_t nc, int lane) {
    constexpr int UN = 32;
    double p[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i0 = 0; i0 < nc; i0 += 64 * UN) {
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t i = i0 + lane + 64 * u;
            v[u] = i < nc ? xr[i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t i = i0 + lane + 64 * u;
            if (i < nc) {
                const float sq = v[u] * v[u];           // float product, then widened (Ggml.cs:5903)
                p[u & 3] += (double)sq;
            }
        }
    }
    double sum = (p[0] + p[1]) + (p[2] + p[3]);
    sum += dpp_d<DPP_XOR1>(sum);
    sum += dpp_d<DPP_XOR2>(sum);
    sum += dpp_d<DPP_HALF_MIRROR>(sum);
    sum += dpp_d<DPP_ROW_MIRROR>(sum);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = (float)(sum / (double)nc);       // Ggml.cs:5906
    return 1.0f / sqrtf(mean + 1e-6f);                  // Ggml.cs:5889, 5915
}
#endif

// ---- kernel launchers (implemented in the .hip files) ----
// layout.hip
hipError_t launch_repack_to_planar(int type, const uint8_t *aos, uint64_t nb01, int64_t row_begin, int64_t rows,
                                   ggml_hip_weight *w, hipStream_t st);
hipError_t launch_planar_to_aos(const ggml_hip_weight *w, uint8_t *aos, hipStream_t st);
hipError_t launch_nibbles_to_bf6(ggml_hip_weight *w, hipStream_t st);
hipError_t launch_q5_to_i8(ggml_hip_weight *w, hipStream_t st);     // Q5_0 / Q5_1 / Q4_1 nibble (+ fifth-bit) planes -> w->i8p
hipError_t launch_min_pieces(ggml_hip_weight *w, hipStream_t st);   // Q5_1 / Q4_1 min plane -> w->mp3 (three bf16 pieces, K3p's min-term product)
// up to 32 device buffers -> device-visible (mapped host) destinations in one launch; sizes and addresses multiples of 4
hipError_t launch_scatter_copy(const void *const *src, void *const *dst, const size_t *bytes, int n, hipStream_t st);
hipError_t launch_gemv_side_image(ggml_hip_weight *w, hipStream_t st);   // d / m / qh planes -> w->gs (after every write of the planes)
static inline int gemv_side_planes(int type) {
    return 1 + ((type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_2) ? 1 : 0) + ((type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1) ? 1 : 0);
}
hipError_t launch_relayout_gathered(const float *g, int G, int64_t N, int64_t Ms, float *dst, int64_t M, int64_t ldd,
                                    hipStream_t st);
hipError_t launch_push_columns(const float *src, int64_t lds, int64_t N, int64_t Ms, float *const *peers, int G, int64_t ldd,
                               int64_t col0, hipStream_t st);
// quantize.hip
hipError_t launch_quantize_act(const float *x, int64_t N, int64_t K, int64_t ld1, act_planes p, int image, hipStream_t st, bool q8k = false);
// kquants.hip (Q5_K as an unpinned extra: the published upstream format, no oracle in the reference)
// (r4: Q4_K beside it -- the same super-block without the fifth-bit bytes: `kq_type` = GGML_HIP_TYPE_Q5_K or _Q4_K; a weight's own ext_type)
static inline bool is_kquant(int t) { return t == GGML_HIP_TYPE_Q5_K || t == GGML_HIP_TYPE_Q4_K || t == GGML_HIP_TYPE_Q6_K; }
static inline size_t kquant_bytes(int t) { return t == GGML_HIP_TYPE_Q5_K ? 176 : t == GGML_HIP_TYPE_Q6_K ? 210 : 144; }   // per 256 weights
// the reference type whose resident planar form (and kernels) a k-quant weight lives in: Q5_K / Q4_K as Q5_1, Q6_K as Q4_2 (two scales per k-block)
static inline int kquant_resident_type(int t) { return t == GGML_HIP_TYPE_Q6_K ? GGML_TYPE_Q4_2 : GGML_TYPE_Q5_1; }
hipError_t launch_q5k_to_planar(int kq_type, const uint8_t *aos, uint64_t nb01, int64_t row_begin, int64_t rows, ggml_hip_weight *w, hipStream_t st);
hipError_t launch_planar_to_q5k(const ggml_hip_weight *w, uint8_t *aos, hipStream_t st);
hipError_t launch_dequantize_q5k(int kq_type, const void *blocks, int64_t nrows, int64_t k, float *y, hipStream_t st);
hipError_t launch_quantize_kq(int kq_type, const float *x, int64_t nrows, int64_t k, void *blocks, hipStream_t st);   // x: contiguous rows, 16-byte aligned
// (Q6_K: its own converters -- the planar Q4_2 form on int8 planes)
hipError_t launch_q6k_to_planar(const uint8_t *aos, uint64_t nb01, int64_t row_begin, int64_t rows, ggml_hip_weight *w, hipStream_t st);
hipError_t launch_planar_to_q6k(const ggml_hip_weight *w, uint8_t *aos, hipStream_t st);
hipError_t launch_dequantize_q6k(const void *blocks, int64_t nrows, int64_t k, float *y, hipStream_t st);
hipError_t launch_quantize_q6k(const float *x, int64_t nrows, int64_t k, void *blocks, hipStream_t st);
hipError_t launch_q8_aos_to_planes(int q8type, const void *blocks, int64_t N, int64_t K, act_planes p, hipStream_t st);
hipError_t launch_quantize_rows(int type, int src_type, const void *x, int64_t ld, int64_t nrows, int64_t k, void *blocks,
                                hipStream_t st);
hipError_t launch_add_q_f32(int type, const void *blocks_in, const float *x, int64_t nrows, int64_t k, void *blocks_out,
                            hipStream_t st);
hipError_t launch_dequantize_rows(int type, const void *blocks, int64_t nrows, int64_t k, float *y, hipStream_t st);
// dense16.hip: K padded to whole stages of 128, plus spare (zero) panels for the register look-ahead
#define DENSE16_SPARE_PANELS 8
static inline int64_t dense16_kpad(int64_t K) { return (K + 127) / 128 * 128; }
struct mm_plan;                      // plan.h: the one decision per product every launcher below consumes
hipError_t launch_f16_rows_to_panels(ggml_hip_weight *w, hipStream_t st);
hipError_t launch_dense16_init(const float *x, int64_t N, int64_t K, int64_t ld1, void *work, hipStream_t st);
hipError_t launch_dense16(const ggml_hip_weight *w, const mm_plan &pl, const void *work, int64_t N, float *dst, int64_t ldd, hipStream_t st);
// K10d (dense16.hip): F32 x F32 on the bf16 matrix cores, operands split into three bf16 pieces
#define DENSE32_SPARE_PANELS 24          // (one stage of look-ahead past the padded end: zero)
hipError_t launch_f32_rows_to_split_panels(ggml_hip_weight *w, hipStream_t st);
hipError_t launch_dense32_init(const float *x, int64_t N, int64_t K, int64_t ld1, void *work, hipStream_t st);
hipError_t launch_dense32(const ggml_hip_weight *w, const void *work, int64_t N, float *dst, int64_t ldd, hipStream_t st);
// eltwise.hip (op: 0 add, 1 mul; contiguous f32)
hipError_t launch_binary_f32(int op, const float *x, const float *y, float *z, int64_t n, hipStream_t st);
hipError_t launch_scale_f32(float *z, int64_t n, float v, hipStream_t st);
hipError_t launch_rms_norm_f32(const float *x, float *y, int64_t nr, int64_t nc, hipStream_t st);
hipError_t launch_silu_f32(const float *x, float *y, int64_t n, hipStream_t st);
// fused.hip: two nodes, one launch (both outputs written)
hipError_t launch_rms_norm_mul_f32(const float *x, const float *g, float *n_out, float *y_out, int64_t nr, int64_t nc, hipStream_t st);
hipError_t launch_silu_mul_f32(const float *a, const float *b, float *s_out, float *y_out, int64_t n, hipStream_t st);
// gemv.hip / gemm_q.hip / dense.hip
hipError_t launch_gemv_q(const ggml_hip_weight *w, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st);
// Up to four weight matrices of one type and K behind ONE activation matrix (the q / k / v or the gate / up projections of a
// transformer block): the fused mat-vec walks the row tiles of all of them in one launch.  Row tiles [tile_end[i-1], tile_end[i])
// belong to matrix i; every row's arithmetic is what the single-matrix call does.
struct mv_set {
    int n;
    int tile_end[4];
    const uint8_t *qs[4];
    const uint32_t *gs[4];
    float *dst[4];
    int64_t M[4], Mpad[4], ldd[4];
};
hipError_t launch_gemv_q_fused_multi(const ggml_hip_weight *const *w, int n_w, const float *x, int64_t ld1, const mm_prologue *pro, int64_t N,
                                     float *const *dst, const int64_t *ldd, hipStream_t st);
hipError_t launch_gemv_q_fused(const ggml_hip_weight *w, const float *x, int64_t ld1, int64_t N, float *dst, int64_t ldd,
                               hipStream_t st, const mm_epilogue *ep = nullptr);
bool gemv_fused_has_epilogue(int64_t N);       // the kernel form that serves N applies an mm_epilogue itself
// the same launch with the rms_norm -> mul prologue (N <= 4 only: gemv_fused_has_epilogue)
hipError_t launch_gemv_q_fused_pro(const ggml_hip_weight *w, const float *x, int64_t ld1, const mm_prologue &pro, int64_t N, float *dst,
                                   int64_t ldd, hipStream_t st, const mm_epilogue *ep = nullptr);
hipError_t launch_gemm_q(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st);
// Q8_0, 5 <= N <= 64, K >= 2048: the stage-free batched-decode form on the int8 matrix cores (gemm_q8s.hip; image 0 of K1); ep: add / scale in the store phase
hipError_t launch_gemm_q8_small(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue *ep);
hipError_t launch_gemm_q8_small_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st);
hipError_t launch_gemm_q16(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st);
// 2..4 matrices of one type and K behind one activation image in ONE launch (5 <= N <= 64, Q4_0 / Q4_1, K >= 2048: gemm_qmx.hip K3s);
// hipErrorNotSupported otherwise -- the caller computes them one after the other
hipError_t launch_gemm_qmx_multi(const ggml_hip_weight *const *w, int n_w, act_planes p, int64_t N, float *const *dst, const int64_t *ldd, hipStream_t st);
hipError_t launch_gemm_qmx(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st,
                           const mm_epilogue *ep = nullptr);
// Q4_0, prompt-sized batches, K >= 2048 (gemm_qmp.hip K3p); hipErrorNotSupported where the form does not apply
hipError_t launch_gemm_qmx_mid(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue &ep);
// the same form on the int8 matrix cores: Q8_0, image 0 (gemm_qmp.hip)
hipError_t launch_gemm_q8_mid(const ggml_hip_weight *w, const mm_plan &pl, act_planes p, int64_t N, float *dst, int64_t ldd, hipStream_t st, const mm_epilogue &ep);
hipError_t launch_dense(const ggml_hip_weight *w, const mm_plan &pl, const float *x, int64_t N, int64_t ld1, float *dst, int64_t ldd,
                        hipStream_t st);

// N at or below this goes to the wave-reduction mat-vec kernel, above it to the MFMA kernel.
#define GEMV_MAX_N 8
// ... and up to this many src1 rows the mat-vec kernel serves the type that has no small-batch MFMA configuration (Q4_2),
// in its two-step form (K1 planes, 16 columns per pass over the weights): 15 us at 4096 x 4096 x 16.  For the other
// types the 32-row, four-way-K-split tiles of gemm_qmx.hip / gemm_q16.hip take over at 9 rows (13 us / 16.5 us there).
#define GEMV_WIDE_MAX_N 16
