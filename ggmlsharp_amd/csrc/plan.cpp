// plan.cpp -- the ONE place where a product's kernel family, form and summation tree are decided (plan.h).  Host code only: no
// device is touched, so the whole decision surface is testable in a container without a GPU (tests/test_plan_cpu.py sweeps
// type x K x N x M through ggml_hip_mm_plan and asserts that tree_id never depends on M).
//
// The numbers behind every threshold are measurements on MI355X; they sit next to the launchers that consume the forms (gemm_qmx.hip,
// gemm_q16.hip, gemm_q.hip, gemm_qmp.hip, gemm_q8s.hip, gemv.hip, dense.hip, dense16.hip) and in DESIGN.md section 5.
#include "common.h"
#include "plan.h"

namespace {

thread_local int t_force_gemm = -1;   // -1: not set yet; 0 auto, 1 int8, 2 f16, 3 MX.  Per calling thread: a test hook never reaches another thread's calls

inline bool is_quant(int t) { return t >= GGML_TYPE_Q4_0 && t <= GGML_TYPE_Q8_1; }
inline bool min_type(int t) { return t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_1; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
constexpr uint64_t LIM32 = 0xFFFFFFFFull;                  // 32-bit buffer offsets

// src1 rows up to which the mat-vec kernel serves a type (two-step form above GEMV_MAX_N): the types with small-batch MFMA
// configurations leave it at 8; Q4_2 stays on it up to 16 where the batched-decode form does not reach (K < 2048, K > 32768: q8_small_serves)
int64_t gemv_rows_max(int t) { return t == GGML_TYPE_Q4_2 ? GEMV_WIDE_MAX_N : GEMV_MAX_N; }

// ---- K3p / K3s / q8s geometry of K: contiguous ranges of k-blocks, one per wave (KS = 8 waves) ----
constexpr int KS8 = 8;
int k3p_mx_nloc(int64_t K) { int n = (int)cdiv(pad_kblocks(K / QK), KS8); return n + (n & 1); }   // (r5: even, K3s-MX's rule -- pairs of blocks stay inside one wave: the two families are one tree)
int k3p_i8_nloc(int64_t K) { int n = (int)cdiv(pad_kblocks(K / QK), KS8); return n + (n & 1); }      // two k-blocks per trip (and the min term goes by pairs)
// the waves' row-scale tables: the whole range of a wave up to 80 k-blocks (K <= 20480), beyond that in refills inside the K loop (r4; it was
// a hard limit at K = 20480) -- up to four slices of 78, K <= 79872 (the 32-bit offsets of the planes end earlier for wide matrices)
bool k3p_lds_ok(int nloc) { return nloc <= K3P_MAX_SLICES * K3P_SLICE_ROWS; }

// ---- the stage-free int8 pair: K3s-int8 (gemm_q8s.hip, batched decode) and K3p-int8 (gemm_qmp.hip, prompt sizes) -- image 0 (+ the min-term piece planes) ----
// ONE summation tree (plan_mul_mat has the argument, tests/test_gpu_fullsize.py the proof), so where both serve the family follows M.  Weights: Q8_0's own planes;
// Q5_0 / Q5_1 / Q4_1 / Q4_2 and the k-quants on int8 operand planes built at upload.  2048 <= K <= 32768 for K3s (two rounds of table pieces per wave), <= 79872
// for K3p (sliced scale tables: k3p_lds_ok).
//     src1 rows     served by
//     (K from 1024: r5, k3s_kmin -- behind K < 2048 K3s serves 5 .. 64 rows and nothing else of this table applies)
//     5 .. 32       K3s   (Q5_1, Q4_2 and the k-quants of the Q5_1 form from 9: up to 8 rows their mat-vec is as fast; Q6_K, whose mat-vec ends at 4, from 5)
//     33 .. 512     both  -- plan_mul_mat picks by M (Q4_1 joins at 65, behind K >= 11008 at 129: below that its MX batched-decode form, another arithmetic)
//     513 .. top    K3p   (top: Q8_0 / Q5_0 3072 -- beyond, their staged f16 forms win; Q4_1 1024; Q5_1, Q4_2 and the k-quants: none)
// The measurements behind every bound: docs/NOTEBOOK_r4.md 10.2d (round 4's), DESIGN.md 12.2c / 12.2d and the tools/experiments/ab_*.sh scripts named there (round 5's):
//   * 33: up to 32 rows half of K3p's 64-column tile would be padding; a tall matrix at 33..64 rows is ONE round of K3p column tiles where K3s re-reads a weight tile per
//     32 columns (Q8_0 32000 x 4096 x 64 44.4 | 35.7 us, Q5_1 60.0 | 38.4; ab_dual_33.sh);
//   * 512: a short matrix (a grouped-query model's k / v projection, 1024 x 4096) at prompt sizes is 128 workgroups of K3p (K3p | K3s: Q8_0 1024 x 4096 x 512 13.7 | 11.4,
//     512 x 4096 x 512 13.5 | 7.8, 1024 x 11008 x 512 31.5 | 25.3; ab_dual_512.sh);
//   * Q4_1 at 65 / 129: MX K3s | the pair by M -- 32000 x 4096 x 128 97.5 | 65.4, 11008 x 4096 x 128 36.1 | 31.4, 1024 x 4096 x 256 15.0 | 9.0, the price 4096 x 4096 x 128 13.7 | 15.9;
//     behind a long K the min-term product's pieces grow with K (4096 x 11008 x 128 26.6 | 32.5), so its MX form keeps 65..128 rows there;
//   * the two-scale types (Q4_2, Q6_K in its form) ran the staged int8 kernel above 256 rows until K3p got its two-scale form (4096 x 11008 x 512 142 | 112; ab_k3p_two_scale.sh).
constexpr int64_t K3_DUAL_MIN_DEFAULT = 33, K3_DUAL_MAX = 512;
int64_t k3_dual_min() { static const int v = dev_env_int("GGML_HIP_K3_DUAL_NMIN", (int)K3_DUAL_MIN_DEFAULT); return v; }   // developer A/B switch
// r5: the batched-decode forms from K = 1024 (32 k-blocks; it was 2048) up to 64 src1 rows -- a small model's hidden size.  Behind such a K there is no K3p (its waves want 8 k-blocks each), so
// no partner on the same tree and no choice by M: only the range where K3s wins on short and tall matrices alike.  Staged | K3s, us per COMPUTE launch (tools/experiments/ab_k3s_kmin.sh):
// Q4_0 1536 x 1536 x 16 / 32 / 64 6.3 | 3.9, 6.0 | 4.5, 7.5 | 5.4, 2048 x 1792 x 32 6.8 | 4.6, 4096 x 1024 x 32 5.1 | 4.6, 8960 x 1536 x 32 7.7 | 7.2, Q8_0 1536 x 1536 x 32 6.5 | 4.9;
// the price: Q8_0 8960 x 1536 x 32 8.3 | 8.7, Q5_1 x 16 8.8 | 9.4.  Beyond 64 rows it wins on short matrices only (Q8_0 1536 x 1536 x 512 14.2 | 11.1 but 8960 x 1536 x 512 38.5 | 51.9): staged stays.
int64_t k3s_kmin() { static const int v = dev_env_int("GGML_HIP_K3S_KMIN", 32); return v; }   // developer A/B switch: the batched-decode forms from this many k-blocks (K / 32)
int64_t k3_dual_wgs(int64_t K) { return K >= 11008 ? 192 : 160; }   // K3p takes over from this many workgroups of 64-row tiles (plan_mul_mat has the measurements)
int64_t q41_pair_min(int64_t K) { static const int v = dev_env_int("GGML_HIP_Q41_PAIR_MIN", 65); return K >= 11008 ? 129 : v; }   // Q4_1: the first src1 row count served by the int8 pair
bool q8_small_serves(int type, int64_t K, int64_t N, bool i8_only = false, bool kq = false) {
    // developer A/B switches (product builds: the defaults): the lower bounds by type group, the upper bound
    static const int k3s_nmin = dev_env_int("GGML_HIP_K3S_NMIN", 5), k3s_nmin_kq = dev_env_int("GGML_HIP_K3S_NMIN_KQ", 9), k3s_nmin_2sc = dev_env_int("GGML_HIP_K3S_NMIN_2SC", GEMV_MAX_N + 1);
    static const int k3s_nmax_2sc = dev_env_int("GGML_HIP_K3S_NMAX_2SC", 0), k3s_nmax_dev = dev_env_int("GGML_HIP_K3S_NMAX", 0);
    const int64_t k3s_nmax = k3s_nmax_dev > 0 ? k3s_nmax_dev : K / QK < 64 ? 64 : K3_DUAL_MAX;   // (behind K < 2048: the batched-decode range only, see k3s_kmin)
    if (K / QK < k3s_kmin() || K / QK > 1024 || plan_force_gemm() != 0) return false;
    if (type == GGML_TYPE_Q4_1) return K / QK >= 64 && !kq && N >= q41_pair_min(K) && N <= k3s_nmax;
    if (type == GGML_TYPE_Q4_2) return N >= (i8_only ? 5 : k3s_nmin_2sc) && N <= (k3s_nmax_2sc > 0 ? k3s_nmax_2sc : k3s_nmax);
    if (type == GGML_TYPE_Q5_1) return N >= (kq ? k3s_nmin_kq : 9) && N <= k3s_nmax;
    return (type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q5_0) && N >= k3s_nmin && N <= k3s_nmax;
}
bool q8_mid_serves(int type, int64_t K, int64_t N) {
    static const int nmax = dev_env_int("GGML_HIP_K3P_NMAX", 0), nmin_dev = dev_env_int("GGML_HIP_K3P_NMIN", 0);   // developer A/B switches
    static const int two_min = dev_env_int("GGML_HIP_K3P_2SC_NMIN", 0), two_max = dev_env_int("GGML_HIP_K3P_2SC_NMAX", 0);   // ... for the two-scale types (NMIN 100000: the staged kernel)
    if (K / QK < 64 || !k3p_lds_ok(k3p_i8_nloc(K)) || plan_force_gemm() != 0) return false;
    const bool dual = K / QK <= 1024;                       // (K3s serves this K too: the shared range starts at K3_DUAL_MIN; beyond, K3p alone from 129 rows)
    const int64_t nmin = nmin_dev > 0 ? nmin_dev : !dual ? 129 : type == GGML_TYPE_Q4_1 ? q41_pair_min(K) : k3_dual_min();
    if (type == GGML_TYPE_Q4_2) return N >= (two_min > 0 ? two_min : nmin) && N <= (two_max > 0 ? two_max : INT64_MAX);
    const int64_t top = nmax > 0 ? nmax : type == GGML_TYPE_Q4_1 ? 1024 : type == GGML_TYPE_Q5_1 ? INT64_MAX : 3072;
    return (type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_1) && N >= nmin && N <= top;
}

int f16_image_kind(int type) { return type == GGML_TYPE_Q8_0 ? 2 : 1; }

// the arithmetic label of the stage-free int8 families (K3s-int8 and K3p-int8 share it, r5): Q8_0 fma(sumi, d1 * d0); the others fma(d0 * sumi, d1); min types:
// + the min terms of 16 k-blocks as five / six bf16-piece MFMAs in front of a wave's blocks; Q4_2 / Q6_K: two fma(d * yd, sumi) per k-block
int k3_i8_arith(int type) { return 300 + (min_type(type) ? 1 : 0) + (type == GGML_TYPE_Q8_0 ? 2 : 0) + (type == GGML_TYPE_Q4_2 ? 4 : 0); }

}  // namespace

int plan_force_gemm() {
    if (t_force_gemm < 0) {
        const char *e = dev_env_str("GGML_HIP_GEMM");
        t_force_gemm = !e ? 0 : (e[0] == 'i' ? 1 : (e[0] == 'f' ? 2 : (e[0] == 'm' ? 3 : 0)));
    }
    return t_force_gemm;
}
void plan_set_force_gemm(int which) { t_force_gemm = which < 0 || which > 3 ? 0 : which; }

// Which MFMA kernel (and so which activation image K1 writes) serves a quantized mat-mat (measured on MI355X):
//   gemm_qmx.hip (MX matrix path, bf6 digits, one exact MFMA per tile and block) -- Q4_0 / Q4_1,
//   gemm_q16.hip (f16 matrix cores, register-tile design) -- Q5_0 / Q5_1 / Q8_0 on prompt-sized batches (N <= 512, K split in the
//                workgroup) and from 1024 rows up,
//   gemm_q.hip   (int8 matrix cores, 64 x 64 / 128 x 128 tiles) -- Q4_2, and Q5_0 / Q5_1 / Q8_0 in between,
//   gemm_qmp.hip / gemm_q8s.hip -- the stage-free forms (K3p: q8_mid_serves / plan_mx; batched decode: q8_small_serves / plan_mx -- the
//                bounds live in those functions and nowhere else).
// ggml_hip_debug_force_gemm forces one (test / developer switch; the product library reads no environment variable: GGML_HIP_GEMM is
// honoured by -DGGML_HIP_DEV builds only).  Returns the K1 image kind: 0 = int8 planes, 1 / 2 = the f16 images, 3 = the bf6 image;
// 0 + ACT_IMAGE_MIN_PIECES (64) = the int8 planes and the min-term piece planes (K3p-int8 behind a Q5_1 / Q4_1 / Q5_K weight).
// NOT a function of the number of weight rows: the signature has no M to consult.
int plan_image_kind(int type, int64_t K, int64_t N) {
    const int force = plan_force_gemm();
    if (q8_mid_serves(type, K, N)) return min_type(type) ? ACT_IMAGE_MIN_PIECES : 0;   // (image 0 + the min-term piece planes)
    if (q8_small_serves(type, K, N)) return min_type(type) ? ACT_IMAGE_MIN_PIECES : 0;
    if (N <= 4 || force == 1) return 0;
    // 5..8 rows: the mat-vec kernel -- except where the batched-decode form of gemm_qmx.hip exists (Q4_0 / Q4_1, K >= 2048): from 5 rows
    // on it runs the same INIT image and kernel as 9..32 rows, with the epilogues and the projection groups that come with them
    // (r4, A/B of whole calls in one gpurun call, K3s from 5 | the fused mat-vec up to 8: 4096 x 4096 x 8 Q4_0 10.3 | 10.1 us, Q8_0 10.8 | 9.9 -- level --
    // 4096 x 11008 x 8 20.1 | 25.6, 11008 x 4096 x 8 15.6 | 18.4, 32000 x 4096 x 8 26.5 | 40.2: 5 stays)
    static const int k3s_nmin = dev_env_int("GGML_HIP_K3S_NMIN", 5);   // developer A/B switch: the batched-decode forms from this many rows
    if (N <= GEMV_MAX_N && !(force == 0 && (type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1) && K / QK >= k3s_kmin() && N >= k3s_nmin)) return 0;
    if (type == GGML_TYPE_Q4_2) return 0;   // served by the mat-vec and int8 kernels only (its k-block carries two scales)
    // the MX / f16 kernels address weights and the activation image through 32-bit buffer offsets
    const uint64_t nba = (uint64_t)pad_kblocks(K / QK);
    if (nba * 64 * (uint64_t)pad_act(N) > LIM32) return 0;
    if (force == 2) return f16_image_kind(type);
    if (type == GGML_TYPE_Q5_1 && force == 3) return 0;                 // no MX form: the forced choice falls back to the int8 kernel
    if (force == 3 || type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1) return 3;
    // Q5_0 / Q5_1 / Q8_0: the f16 kernel's K-split form wins on prompt-sized batches and its unsplit forms from 1024 rows up; in between
    // the int8 kernel's 64 x 64 tiles balance the chip better.  Decided from N and K only, like the K split itself.
    return ((N <= 512 || N >= 1024) && K / QK >= 8) ? f16_image_kind(type) : 0;
}

namespace {

// ---- family plans: the decisions that used to sit in each launcher, verbatim ----

void plan_gemv(mm_plan &p, int type, int64_t M, int64_t K, int64_t N, bool fused) {
    p.family = fused ? MMF_GEMV_FUSED : MMF_GEMV_ROWS;
    p.image = fused ? -1 : 0;
    // 16 rows per workgroup for every M (gemv.hip launch_typed: a choice by M would change the tree); 8 waves x 4 k-lanes = 32 k-workers,
    // worker u takes k-blocks u, u + 32, ...; two xor-shuffles + one LDS pass join them in a fixed order
    p.form = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : N <= 8 ? 8 : 16;                    // columns per pass (NC)
    p.arith = 100;   // (the wave-private fused form, N <= 4, is its own kernel with the SAME quants and tree as INIT + the mat-vec: bitwise equal,
                     //  tests/test_gpu_parity.py test_small_n_fused_path_equals_two_step_path -- so one label: ggml_hip_mm_plan's tree_id is what the COMPUTE-only entry runs too)
    p.ksplit = 32; p.kstyle = MMK_WORKERS; p.kunit = 1;
    p.tile_m = 16; p.tile_n = p.form; p.waves = 8; p.tiles_per_wave = 1;
    const int64_t ntiles = cdiv(M, 16);
    p.wgs = ntiles < 512 ? ntiles : 512;
    if (fused && (N <= 1 || N >= 3) && K / QK <= 128 && ntiles > 256) p.wgs = 256;       // the look-ahead form: one workgroup per CU
    p.flags |= MM_FLAG_PERSISTENT;
    if (fused && N >= 1 && N <= 4) p.flags |= MM_FLAG_EPILOGUE_FUSED;
}

bool plan_k3s_i8(mm_plan &p, int type, int64_t M, int64_t Mpad, int64_t K, int64_t N) {
    const int nbkp = (int)pad_kblocks(K / QK);
    // r5: K3p-int8's rule (an even number of k-blocks per wave; it was cdiv(nbkp, 8)): with the same ranges, the same statement per block and the same wave-order
    // sum the two families compute the SAME BITS per element -- one tree, see plan_mul_mat -- so between them the family may follow M
    const int nloc = k3p_i8_nloc(K);
    const int ncol = (int)cdiv(N, 32);
    if (nloc > 128) return false;                           // (two rounds of table pieces: K <= 32768; r4 -- it was one round, K <= 16384)
    if ((uint64_t)nbkp * 2 * (uint64_t)Mpad * 16 > LIM32 || (uint64_t)nbkp * 2 * (uint64_t)pad_act(N) * 16 > LIM32) return false;
    static const int geo = dev_env_int("GGML_HIP_Q8S_TILES", 0);   // developer A/B switch: 1 / 2 tiles per workgroup whatever M
    const int64_t t32 = cdiv(M, 32) * ncol;
    // (r5: four tiles per workgroup beyond 512 tile groups for Q8_0 / Q5_0, as the MX form has for Q4_0 -- 11008 x 4096 x 64 is 688 tile groups: 344
    // workgroups of two tiles ran a second, 34 %-full round (24.2 us where Q4_0's 172 workgroups of four took 17.0); the min-term and two-scale types keep two)
    const bool four_ok = (type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q5_0) && nloc <= 128;
    const int wmt = geo == 1 ? 1 : geo == 2 ? 2 : geo == 4 && four_ok ? 4 : t32 <= 256 ? 1 : (t32 <= 512 || !four_ok) ? 2 : 4;
    p.family = MMF_K3S_I8; p.image = 0;
    if (min_type(type)) p.flags |= MM_FLAG_MIN_PIECES;
    p.form = wmt == 4 ? 7 : wmt == 2 ? (nloc <= 8 ? 0 : 1) : nloc <= 8 ? 2 : nloc <= 16 ? 3 : 4;
    p.arith = k3_i8_arith(type); p.ksplit = KS8; p.kstyle = MMK_RANGES; p.kunit = nloc;   // Q8_0: fma(sumi, d1 * d0); Q5_0: fma(d0 * sumi, d1) -- K3p-int8's labels: the same arithmetic
    p.nloc = nloc; p.wmt = wmt;
    p.tile_m = 32 * wmt; p.tile_n = 32; p.waves = KS8; p.tiles_per_wave = wmt;
    p.wgs = cdiv(M, 32 * wmt) * ncol;
    p.flags |= MM_FLAG_EPILOGUE_FUSED;
    // r5: 16-row tiles on v_mfma_i32_16x16x32_i8 where the 32-row tiles leave CUs idle (gemm_q8s.hip gemm_q8_small16_kernel; first Q8_0 / Q5_0, then
    // the two-scale and the min-term types, see below).  GEOMETRY ONLY: the same eight K ranges, the same block order and statement, the same
    // wave-order sum -- tree_id does not move, so this choice MAY look at M.  A workgroup takes 16 rows x the (up to 32) src1 rows in one or two
    // 16-column slices: 4096 x 4096 x 32 is 256 workgroups instead of 128.  Measured (replayed graphs of 64 launches, 24 .. 32 weight copies in
    // turn, 32-row | 16-row tiles, us per COMPUTE launch): Q8_0 4096 x 4096 x 32 11.5 | 9.1, x 8 11.4 | 7.3, Q5_0 x 32 11.6 | 9.1.  Up to 32 rows and
    // one round of the chip only: four slices per workgroup at 33..64 rows lost to two 32-row workgroups sharing a weight tile through L2
    // (Q8_0 4096 x 4096 x 64 11.3 | 13.1, 4096 x 11008 x 64 25.9 | 30.5), more than 256 of them lost too (Q4_0 8192 x 8192 x 32 23 | 33).
    // (Q4_2 too, and Q6_K in its form: a product per 16-element half with the weight operand zero in the other half's lanes)
    // (Q5_1 too, and Q5_K / Q4_K in its form: the min-term product stays the 32-row form's instruction -- v_mfma_f32_32x32x16_bf16 with the workgroup's 16 rows in
    // half of the weight operand's lanes -- and its result is handed to the 16 x 16 tiles' lanes once per wave, so the bits are the 32-row form's.  32-row | 16-row tiles:
    // Q5_1 4096 x 4096 x 9 / 16 / 32 12.8 | 8.0, 12.9 | 8.0, 12.9 | 10.2 us, 4096 x 11008 x 16 / 32 27.7 | 16.2, 27.7 | 21.4, 2048 x 8192 x 32 16.7 | 13.1, 4096 x 28672 x 16 62.5 | 34.8,
    // Q5_K 4096 x 4096 x 16 13.0 | 8.1, Q4_K x 32 13.2 | 10.4, 4096 x 11008 x 16 28.4 | 16.4 -- tools/experiments/ab_q8s_16_min.sh)
    static const int min16 = dev_env_int("GGML_HIP_Q8S_16_MIN", 1);   // developer A/B switch: 0 = the min-term types keep the 32-row form
    // (up to 64 rows -- it was 32: on a short matrix 33..64 rows are column groups of 16 or 32 columns, 256 workgroups at most: Q4_0 1024 x 4096 x 64 7.8 -> 5.2 us, 2048 x 4096 x 64 8.9 -> 7.2,
    // Q8_0 9.6 -> 8.4 / 1024 rows 7.6 -> 5.5, Q5_1 9.8 -> 8.6, Q4_0 1536 x 1536 x 64 5.4 -> 4.6; 4096 rows and more keep two 32-row workgroups per weight tile.  tools/experiments/ab_tile16_n64.sh)
    // ... and beyond 64 rows wherever the 16-row workgroups still fit one round (1024 rows at 128 src1 rows, 512 at 256): Q4_0 1024 x 4096 x 128 7.7 -> 6.7, 512 x 4096 x 128 7.8 -> 5.2,
    // 1024 x 11008 x 128 16.0 -> 13.9, Q5_1 1024 x 4096 x 128 9.0 -> 7.9 (ab_tile16_n512.sh): the workgroup count decides, the row count does not
    static const int n16max = dev_env_int("GGML_HIP_K3S_16_NMAX", 512);   // developer A/B switch: 16-row tiles up to this many src1 rows
    if ((type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q4_2 || (type == GGML_TYPE_Q5_1 && min16)) && N <= n16max) {
        // (17..32 rows on a SHORT matrix: one 16-column slice per workgroup and two workgroups per weight tile while those fit one round -- 2048 rows are 256 workgroups that way,
        // 128 with both slices in one: Q4_0 2048 x 4096 x 32 7.5 -> 5.9 us, 1024 x 4096 x 32 6.7 -> 5.2, 2048 x 8192 x 32 12.6 -> 9.9, Q8_0 2048 x 4096 x 32 8.6 -> 6.6, Q5_1 8.3 -> 6.8;
        // tools/experiments/ab_tile16_tn.sh.  Geometry, like the tile height.)
        static const int tn16_dev = dev_env_int("GGML_HIP_K3S_16_TN", 0);   // developer A/B switch: 16 / 32 = that many columns per workgroup whatever M
        const int tn16 = N <= 16 || tn16_dev == 16 || (tn16_dev == 0 && cdiv(M, 16) * cdiv(N, 16) <= 256) ? 16 : 32;
        const int64_t wg16 = cdiv(M, 16) * cdiv(N, tn16);
        const int nb16 = nloc <= 8 ? 8 : 16, rows16 = (int)cdiv(nloc, nb16) * nb16;   // (the launcher's slots: gemm_q8s.hip)
        const bool fits = (int64_t)KS8 * rows16 * tn16 * 4 <= 160 * 1024;
        static const int lim16 = dev_env_int("GGML_HIP_Q8S_16_WGS", 256);   // developer A/B switch: 16-row tiles up to this many workgroups (0: never)
        if (fits && geo == 0 && wg16 <= lim16) {
            p.form = 5 + (tn16 == 16 ? 0 : 1);
            p.wmt = 1; p.tile_m = 16; p.tile_n = tn16; p.tiles_per_wave = tn16 / 16; p.wgs = wg16;
        }
    }
    return true;
}

bool plan_k3p_i8(mm_plan &p, int type, int64_t M, int64_t Mpad, int64_t K, int64_t N) {
    const int nloc = k3p_i8_nloc(K);
    if (nloc < 8 || !k3p_lds_ok(nloc)) return false;
    // (32-bit buffer offsets, the look-ahead past a wave's range included -- the SAME bound the exception test of plan_mul_mat uses)
    if ((uint64_t)(KS8 * nloc + 2) * 2 * (uint64_t)Mpad * 16 > LIM32 || (uint64_t)(KS8 * nloc + 2) * 2 * (uint64_t)pad_act(N) * 16 > LIM32) return false;
    p.family = MMF_K3P_I8; p.image = 0; p.form = 0;
    // Q8_0: fma(sumi, d1 * d0); others: fma(d0 * sumi, d1); min types: + the min terms of 16 k-blocks as six bf16-piece MFMAs per tile
    p.arith = k3_i8_arith(type);
    if (min_type(type)) p.flags |= MM_FLAG_MIN_PIECES;
    p.ksplit = KS8; p.kstyle = MMK_RANGES; p.kunit = nloc; p.nloc = nloc; p.wmt = 4;
    p.tile_m = 128; p.tile_n = 64; p.waves = KS8; p.tiles_per_wave = 8;
    p.wgs = cdiv(M, 128) * cdiv(N, 64);
    // r5: 64-row wave tiles (two m-tiles instead of four) where the 128-row tiles leave half of the chip idle -- 4096 rows at 129..256 src1 rows are
    // 96..128 workgroups on 256 CUs.  GEOMETRY ONLY: the same eight K ranges, block order, statement and wave-order sum per element, so the choice
    // may look at M (gemm_qmp.hip gemm_q8_mid_kernel<..., WM = 2>).
    static const int k3p_wmt = dev_env_int("GGML_HIP_K3P_WMT", 0);   // developer A/B switch: 2 / 4 = that wave tile whatever the grid
    const int64_t wg64 = cdiv(M, 64) * cdiv(N, 64), wg128 = p.wgs;
    // ... and on grids of a fractional number of rounds: a launch takes as long as its fullest CU, i.e. ceil(workgroups / 256) rounds (the hardware
    // hands a free CU the next workgroup; from four rounds on the kernel deals the tiles itself), and a 64-row workgroup costs 0.545 of a 128-row one
    // (config 4 forced onto 64-row tiles: 64.1 -> 69.7 us for exactly two rounds).  So 258 workgroups of 128 rows -- two rounds for two tiles -- are
    // three rounds of half the size.  Measured, 128 | 64 rows, us: Q8_0 11008 x 4096 x 192 (258 | 516 workgroups) 49.3 | 42.3, x 256 (344 | 688) 52.1 | 45.8,
    // x 384 (516 | 1032) 77.2 | 68.2, 5504 x 4096 x 512 48.8 | 43.3, 9000 x 4096 x 256 51.2 | 45.2; and where the model says no: 8192 x 8192 x 192 (192 | 384)
    // 45.8 | 53.1, 14336 x 4096 x 256 (448 | 896) 55.4 | 61.0 (tools/experiments/ab_k3p_wmt_frac.sh).  7 % of margin; the contract configs keep 128 rows.
    const double r128 = (double)cdiv(wg128, 256), r64 = (double)cdiv(wg64, 256) * 0.545;
    if (k3p_wmt == 2 || (k3p_wmt == 0 && (wg64 <= 256 || r64 <= 0.93 * r128))) {
        p.wmt = 2; p.tile_m = 64; p.tiles_per_wave = 4; p.wgs = wg64;
    }
    p.flags |= MM_FLAG_EPILOGUE_FUSED;
    return true;
}

bool plan_k3p_mx(mm_plan &p, int64_t M, int64_t Mpad, int64_t K, int64_t N) {
    const int nloc = k3p_mx_nloc(K);
    if (nloc < 8 || !k3p_lds_ok(nloc)) return false;
    if ((uint64_t)(KS8 * nloc + 2) * (uint64_t)Mpad * 16 > LIM32 || (uint64_t)(KS8 * nloc + 2) * 48 * (uint64_t)pad_act(N) > LIM32) return false;
    p.family = MMF_K3P_MX; p.image = 3; p.form = 0;
    p.arith = 310; p.ksplit = KS8; p.kstyle = MMK_RANGES; p.kunit = nloc; p.nloc = nloc; p.wmt = 4;
    p.tile_m = 128; p.tile_n = 64; p.waves = KS8; p.tiles_per_wave = 8;
    p.wgs = cdiv(M, 128) * cdiv(N, 64);
    // r5: 64-row wave tiles where the 128-row tiles leave half of the chip idle (2048 rows at 512 src1 rows are 128 workgroups) -- geometry only, as in
    // plan_k3p_i8 (same switch).  One round only: a 64-row workgroup of the MX kernel costs 0.63 of a 128-row one (Q4_0 11008 x 4096 x 320, 430 | 860
    // workgroups: 50.8 | 64.4 us), too much for the fractional-round rule of the int8 kernel to pay.
    static const int k3p_wmt = dev_env_int("GGML_HIP_K3P_WMT", 0);
    const int64_t wg64 = cdiv(M, 64) * cdiv(N, 64);
    if (k3p_wmt == 2 || (k3p_wmt == 0 && wg64 <= 256)) {
        p.wmt = 2; p.tile_m = 64; p.tiles_per_wave = 4; p.wgs = wg64;
    }
    p.flags |= MM_FLAG_EPILOGUE_FUSED;
    return true;
}

// K3s (gemm_qmx.hip launch_small): KS = 8 waves per 32-row tile, each a contiguous eighth of K in pairs of blocks
bool plan_k3s_mx(mm_plan &p, int type, int64_t M, int64_t Mpad, int64_t K, int64_t N) {
    const int nbkp = (int)pad_kblocks(K / QK);
    int nloc = (int)cdiv(nbkp, KS8);
    nloc += nloc & 1;                                       // pairs of blocks stay inside one wave
    if (nloc > 128) return false;                           // (the table pieces a lane holds, two rounds of eight: K <= 32768; r4 -- it was K <= 16384)
    if (((uint64_t)nbkp + K_LOOKAHEAD) * (uint64_t)Mpad * 16 > LIM32 || (uint64_t)nbkp * 48 * (uint64_t)pad_act(N) > LIM32) return false;
    static const int geo = dev_env_int("GGML_HIP_K3S_GEO", 0);   // developer A/B switch: 1 / 2 / 4 = that many tiles per workgroup whatever M
    const int ncol = (int)cdiv(N, 32);
    const int64_t t32 = cdiv(M, 32) * ncol;
    // (four tiles: Q4_0 only -- Q4_1's min-term registers do not fit beside four accumulator tiles)
    const int wmt = geo == 1 ? 1 : geo == 2 || type == GGML_TYPE_Q4_1 ? (geo == 2 || geo == 4 || t32 > 256 ? 2 : 1) : geo == 4 ? 4 : (t32 <= 256 ? 1 : t32 <= 512 ? 2 : 4);
    p.family = MMF_K3S_MX; p.image = 3; p.form = 0;
    p.arith = type == GGML_TYPE_Q4_1 ? 211 : 310;           // (Q4_0: K3p-MX's label -- the same arithmetic on the same ranges, r5)
    p.ksplit = KS8; p.kstyle = MMK_RANGES; p.kunit = nloc; p.nloc = nloc; p.wmt = wmt;
    p.tile_m = 32 * wmt; p.tile_n = 32; p.waves = KS8; p.tiles_per_wave = wmt;
    p.wgs = cdiv(M, 32 * wmt) * ncol;
    p.flags |= MM_FLAG_EPILOGUE_FUSED;
    // r5: 16-row tiles on v_mfma_scale_f32_16x16x128_f8f6f4 where the 32-row tiles leave CUs idle (gemm_qmx.hip gemm_qmx_small16_kernel, Q4_0;
    // Q4_1's min term keeps the 32-row form).  GEOMETRY ONLY, as in plan_k3s_i8: the same K ranges, pair and block order, statement and
    // wave-order sum -- tree_id does not move, so the choice may look at M.  Up to 32 src1 rows and one round of the chip (plan_k3s_i8 has the
    // reasons): Q4_0 4096 x 4096 x 32 10.6 | 8.6 us per COMPUTE launch (32-row | 16-row tiles, replayed graphs, 32 weight copies in turn), x 16 and
    // x 5 10.6 | 7.1, 4096 x 11008 x 32 22.6 | 19.0, 2048 x 8192 x 32 19.1 | 16.5; the whole call at 16 rows 13.8 | 10.3.
    // (Q4_1 too: its min term as k = 0 / 2 of one v_mfma_f32_16x16x4_f32 per pair -- the 32-row form's two fmaf in the same order)
    static const int n16max = dev_env_int("GGML_HIP_K3S_16_NMAX", 512);   // developer A/B switch: see plan_k3s_i8
    if ((type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1) && N <= n16max) {
        // (17..32 rows on a SHORT matrix: one 16-column slice per workgroup and two workgroups per weight tile while those fit one round -- 2048 rows are 256 workgroups that way,
        // 128 with both slices in one: Q4_0 2048 x 4096 x 32 7.5 -> 5.9 us, 1024 x 4096 x 32 6.7 -> 5.2, 2048 x 8192 x 32 12.6 -> 9.9, Q8_0 2048 x 4096 x 32 8.6 -> 6.6, Q5_1 8.3 -> 6.8;
        // tools/experiments/ab_tile16_tn.sh.  Geometry, like the tile height.)
        static const int tn16_dev = dev_env_int("GGML_HIP_K3S_16_TN", 0);   // developer A/B switch: 16 / 32 = that many columns per workgroup whatever M
        const int tn16 = N <= 16 || tn16_dev == 16 || (tn16_dev == 0 && cdiv(M, 16) * cdiv(N, 16) <= 256) ? 16 : 32;
        const int64_t wg16 = cdiv(M, 16) * cdiv(N, tn16);
        const int np16 = nloc <= 8 ? 4 : 8, rows16 = nloc > 2 * np16 ? nloc : 2 * np16;   // (the launcher's slots: gemm_qmx.hip launch_small)
        const bool fits = (int64_t)KS8 * rows16 * tn16 * 4 <= 160 * 1024;
        static const int lim16 = dev_env_int("GGML_HIP_K3S_16_WGS", 256);   // developer A/B switch: 16-row tiles up to this many workgroups (0: never)
        if (fits && geo == 0 && wg16 <= lim16) {
            p.form = 5 + (tn16 == 16 ? 0 : 1);
            p.wmt = 1; p.tile_m = 16; p.tile_n = tn16; p.tiles_per_wave = tn16 / 16; p.wgs = wg16;
        }
    }
    return true;
}

struct FormGeo { int wmt, wnt, wgm, wgn, ksp, vs; };
void set_staged_geo(mm_plan &p, const FormGeo &g, int64_t M, int64_t N) {
    p.tile_m = g.wgm * g.wmt * 32; p.tile_n = g.wgn * g.wnt * 32;
    p.waves = g.wgm * g.wgn * g.ksp; p.tiles_per_wave = g.wmt * g.wnt;
    p.ksplit = g.ksp * g.vs; p.kstyle = p.ksplit > 1 ? MMK_STAGE_SETS : MMK_CHAIN; p.kunit = 4;
    p.wgs = cdiv(M, p.tile_m) * cdiv(N, p.tile_n);
}

// the staged MX family (gemm_qmx.hip launch_typed), TYPE in {Q4_0, Q4_1} (+ Q5_0 / Q8_0 with two weight digits when MX is forced)
void plan_mx(mm_plan &p, int type, int64_t M, int64_t Mpad, int64_t K, int64_t N) {
    static const int var = dev_env_int("GGML_HIP_MX_TILE", 0);   // developer A/B switch
    const int64_t nbk = K / QK;
    const int64_t tm256 = cdiv(M, 256), tm128 = cdiv(M, 128), tn128 = cdiv(N, 128);
    const bool q40 = type == GGML_TYPE_Q4_0, q4 = q40 || type == GGML_TYPE_Q4_1;
    static const FormGeo geo[] = {{2, 4, 4, 1, 1, 1}, {4, 2, 2, 2, 1, 1}, {1, 1, 2, 1, 4, 1}, {1, 1, 1, 1, 4, 1}, {1, 2, 4, 1, 4, 1}, {1, 2, 2, 1, 4, 1},
                                  {1, 2, 1, 1, 4, 1}, {1, 2, 2, 1, 2, 2}, {1, 2, 4, 1, 2, 1}, {1, 2, 2, 1, 2, 1}, {2, 2, 2, 2, 1, 1}, {1, 1, 2, 2, 1, 1}, {1, 2, 4, 1, 1, 1}};
    auto take = [&](int f) {
        p.family = MMF_MX; p.image = 3; p.form = f;
        set_staged_geo(p, geo[f], M, N);
        // Q4_1: with ONE tile per wave the min-term MFMA lands between other scale-accumulates of its tile than in the multi-tile forms
        // (a different f32 addition order): the one-tile forms are chosen by N alone (up to 32 rows), the 64 x 64 form is Q4_0's only
        p.arith = 400 + (type == GGML_TYPE_Q4_1 ? 1 + (p.tiles_per_wave == 1 ? 1 : 0) : type == GGML_TYPE_Q4_0 ? 0 : 10 + type);
        p.flags |= MM_FLAG_EPILOGUE_FUSED;
    };
    // 8 tiles per wave only where the registers allow it: one weight digit and one scale plane (Q4_0); only above 512 rows: up to 512
    // rows every form splits K (two, four or eight ways by N and K) and this one does not
    // 256 x 128 tiles as soon as the 128 x 128 form would need MORE than one round of the chip (2 workgroups per CU: 512): r4 -- the bound
    // was 384 of these tiles (1.5 rounds of them), and between 257 and 383 the 128 x 128 form ran a second, nearly empty round:
    // 11008 x 4096 x 768 131 -> 114 us, 8192 x 8192 x 1025 257 -> 221, 4096 x 11008 x 2304 341 -> 294, 14336 x 4096 x 640 133 -> 113 (same tree: both unsplit)
    static const int t256 = dev_env_int("GGML_HIP_MX_T256", 257);   // developer A/B switch
    if (q40 && (tm256 * tn128 >= t256 || var == 30) && N > 512) return take(var == 2 ? MXF_256x128_ALT : MXF_256x128);
    // Batches up to 128 rows (Q4_1: 256): 32-row weight tiles with K split four ways inside the workgroup; the same four-way tree on
    // taller tiles where those cover the chip.  The choice of the SPLIT depends on N and K only; the tile height follows the tile count.
    // r5 (the MX side): Q4_0's stage-free forms K3s and K3p are ONE tree too -- the same eight K ranges (an even number of k-blocks per wave),
    // acc += (sumi * d1) * d0 block by block, the eight sums in wave order: the same bits (test_k3s_and_k3p_mx_compute_the_same_bits) -- so between 65 and 512 src1
    // rows the family follows M whatever K: K3p once its grid of 64-row tiles has 192 workgroups, K3s below; the staged four-way forms that served K < 11008 there lose to
    // one or the other at every size measured.  The r4 plan | K3s | K3p, us per COMPUTE launch (tools/experiments/ab_mx_dual.sh): 2048 x 4096 x 128 15.7 | 9.2 | 17.3,
    // 4096 x 4096 x 96 / 128 17.2 | 13.8 | 17.6, 17.2 | 13.7 | 17.5, 4096 x 11008 x 128 26.2 | 26.0 | 39.6 (K3s: 128 workgroups or fewer) -- 8192 x 4096 x 128 20.8 | 19.9 | 19.1,
    // 11008 x 4096 x 65 / 128 31.8 | 31.3 | 24.8, 31.9 | 35.3 | 25.3, 16384 x 4096 x 128 35.8 | 37.2 | 28.5, 32000 x 4096 x 96 / 128 69.1 | 57.3 | 54.9, 69.7 | 71.8 | 55.8,
    // 8192 x 8192 x 128 36.3 | 30.9 | 31.8, 11008 x 11008 x 128 71.8 | 72.0 | 53.6 (K3p: 256 and more).
    static const int mxdual_dev = dev_env_int("GGML_HIP_MX_DUAL_WGS", -1);     // developer A/B switch: K3p from this many 64-row workgroups (1: always, 1000000: never, 0: the r4 plan; -1: the rule)
    const int64_t mxdual = mxdual_dev >= 0 ? mxdual_dev : k3_dual_wgs(K);   // (160 behind K < 11008: Q4_0 2560 x 4096 x 256 16.6 | 14.8 us, 11008 x 4096 x 64 19.8 | 19.4; 192 behind a longer K)
    // ... and up to 256 rows (the staged two-/four-way forms served 129..256): the r4 plan | the family by M -- 1024 x 4096 x 256 15.3 | 8.0, 2048 x 4096 x 192 / 256 18.0 | 12.8, 17.5 | 12.6,
    // 4096 x 4096 x 129 / 192 / 256 19.1 | 17.6, 19.1 | 17.7, 19.5 | 17.8, 8192 x 4096 x 192 31.6 | 25.2, 32000 x 4096 x 192 / 256 85.5 | 79.7, 118 | 105, 4096 x 11008 x 192 / 256 45.2 | 40.0, 46.5 | 40.4,
    // 8192 x 8192 x 192 57.3 | 42.1, 1024 x 11008 x 256 35.2 | 17.1; the price is a grid just past a whole round -- 11008 x 4096 x 192 / 256 (258 / 344 workgroups of 128 rows) 37.7 | 46.5, 42.6 | 48.4
    // (tools/experiments/ab_mx_dual_256.sh)
    static const int mxdual_nmax = dev_env_int("GGML_HIP_MX_DUAL_NMAX", (int)K3_DUAL_MAX);   // (r5: 512 -- short matrices at prompt sizes, see K3_DUAL_MAX)
    if (mxdual > 0 && q40 && N >= k3_dual_min() && N <= mxdual_nmax && nbk >= 64 && var == 0) {
        if (cdiv(M, 64) * cdiv(N, 64) >= mxdual && plan_k3p_mx(p, M, Mpad, K, N)) return;
        if (plan_k3s_mx(p, type, M, Mpad, K, N)) return;
        if (plan_k3p_mx(p, M, Mpad, K, N)) return;
    }
    if (N <= (q40 || var == 20 ? 128 : 256) && nbk >= 16 && var != 3 && var != 9) {
        const int64_t t64 = cdiv(M, 64) * cdiv(N, 64);
        if (q4) {
            // up to 64 rows, K >= 2048: the stage-free form K3s (by N and K alone; GGML_HIP_MX_TILE=26: the staged form, A/B)
            // (r4: up to 128 rows behind K >= 11008, the rule and the reason of q8_small_serves -- staged | this form at 128 rows: 4096 x 11008 36.1 | 25.0 us,
            // Q4_1 49.5 | 28.3, 8192 x 28672 110 | 96.5; 11008 x 11008 66.4 | 83.5 is the price; behind a shorter K it pays by M: 4096 x 4096 16.1 | 13.1
            // but 11008 x 4096 29.0 | 37.3, 32000 x 4096 67.9 | 87.5)
            static const int ncmax_dev = dev_env_int("GGML_HIP_K3S_COLS", 0);   // developer A/B switch: 1 = K3s up to 32 rows only (0: the rule)
            // (with the XCD-aware tile order, staged | this form at 128 rows: Q4_1 4096 x 4096 20.2 | 12.8, 11008 x 4096 x 96 42.2 | 35.4 -- Q4_1 whatever K;
            // Q4_0 4096 x 4096 16.0 | 13.1 but 11008 x 4096 28.8 | 34.0, 32000 x 4096 62.5 | 68.6, 28672 x 8192 106 | 112: its staged forms are better, K >= 11008 stays)
            const int ncmax = ncmax_dev > 0 ? ncmax_dev : nbk < 64 ? 2 : (K >= 11008 || !q40) ? 4 : 2;   // (K < 2048: up to 64 rows, see k3s_kmin)
            if (N <= 32 * ncmax && nbk >= k3s_kmin() && var != 25 && var != 26 && plan_k3s_mx(p, type, M, Mpad, K, N)) return;
        }
        if (N <= 32 && var != 25) {
            const int h32 = var == 13 ? 32 : var == 15 || var == 12 ? 64 : (t64 < 160 ? 32 : 64);
            return take(h32 >= 64 ? MXF_N32_H64 : MXF_N32_H32);
        }
        const int h = var == 12 ? 128 : var == 15 ? 64 : var == 13 ? 32 : (t64 < 160 ? 32 : (t64 <= 320 || !q40) ? 64 : 128);
        if (q40 && h == 128) return take(MXF_S4_H128);
        return take(h >= 64 ? MXF_S4_H64 : MXF_S4_H32);
    }
    // 129 .. 256 rows (Q4_0): the four-way tree as well -- really split while the tiles are few, else two wave groups that each run two
    // of the four stage sets one after the other (same tree, same bits)
    if (q40 && N <= 256 && nbk >= 16 && var != 3 && var != 9 && var != 20) {
        const int64_t t64 = cdiv(M, 64) * cdiv(N, 64);
        return take(var == 23 || (var != 24 && t64 <= 256) ? MXF_S4_H64 : MXF_S2V2_H64);
    }
    // developer A/B switch.  r4, K3p-MX above 512 rows against the staged MX forms, staged | K3p: 4096 x 4096 x 1024 51.7 | 55.7 us, x 2048 84.5 |
    // 109.5, 4096 x 11008 x 1024 128.8 | 126.6, 11008 x 4096 x 1024 132.4 | 152.9, 8192 x 8192 x 1024 163 | 186: the staged forms keep that range
    // (the int8 types' staged forms do not: q8_mid_serves)
    static const int k3p_nmax = dev_env_int("GGML_HIP_K3P_MX_NMAX", 512);
    if (q40 && N > 512 && N <= k3p_nmax && nbk >= 64 && plan_k3p_mx(p, M, Mpad, K, N)) return;
    if (N <= 512 && nbk >= 8 && var != 3) {
        // 257 .. 512 rows, K >= 2048, Q4_0: K3p (gemm_qmp.hip).  By N and K alone (GGML_HIP_K3P=1: the staged form, A/B).
        if (q40) {
            static const int k3p = dev_env_int("GGML_HIP_K3P", 0);
            if (k3p != 1 && plan_k3p_mx(p, M, Mpad, K, N)) return;
        }
        return take(var == 17 || (var != 16 && tm128 * cdiv(N, 64) >= 1536) ? MXF_S2_H128 : MXF_S2_H64);
    }
    static const int t128 = dev_env_int("GGML_HIP_MX_T128", 384);   // developer A/B switch (r4: 257 and 129 measured -- 4096 x 4096 x 1280 67 | 73 | 72 us, 4096 x 11008 x 1280 170 | 183 | 172: 384 stays)
    if (tm128 * tn128 >= t128) return take(MXF_128x128);
    // a short, wide product (a row shard of a multi-GPU split): 64 x 64 tiles of four 1-tile waves wherever the 128 x 64 grid leaves
    // CUs idle -- the same unsplit K loop per element, the same bits, only the geometry follows M (Q4_0 only: see `arith` above)
    if (q40 && (var == 32 || (var != 31 && tm128 * cdiv(N, 64) <= 256))) return take(MXF_64x64);
    return take(MXF_128x64);
}

// the staged f16 family (gemm_q16.hip launch_typed)
void plan_f16(mm_plan &p, int type, int64_t M, int64_t K, int64_t N) {
    const int64_t nbk = K / QK;
    const bool mn = min_type(type);
    static const FormGeo geo[] = {{1, 1, 2, 1, 4, 1}, {1, 1, 1, 1, 4, 1}, {1, 2, 4, 1, 4, 1}, {1, 2, 2, 1, 4, 1}, {1, 2, 1, 1, 4, 1},
                                  {1, 2, 4, 1, 2, 1}, {2, 4, 4, 1, 1, 1}, {1, 1, 2, 2, 1, 1}, {1, 2, 4, 1, 1, 1}};
    auto take = [&](int f) {
        p.family = MMF_F16; p.image = f16_image_kind(type); p.form = f;
        set_staged_geo(p, geo[f], M, N);
        // the min term: on the matrix pipe per pair of k-blocks in the forms that split K, on the VALU per block in the unsplit ones --
        // by the SPLIT (i.e. by N and K), not by the tile count (r3)
        p.arith = 500 + type * 4 + (mn ? (p.ksplit > 1 ? 1 : 2) : 0);
    };
    const int64_t big = cdiv(M, 256) * cdiv(N, 128);
    static const bool old128 = dev_env_set("GGML_HIP_Q16_OLD128");   // developer A/B switch
    static const int n4 = dev_env_int("GGML_HIP_Q16_N4", 256);       // developer A/B switch (128 = the former bound)
    if (N <= n4 && nbk >= 16 && !old128) {
        static const int tile = dev_env_int("GGML_HIP_Q16_TILE", 0); // A/B: 1 = 128-row, 2 = 32-row, 3 = 64-row
        const int64_t t64 = cdiv(M, 64) * cdiv(N, 64);
        if (N <= 32 && tile != 4) {
            const int h32 = tile == 2 ? 32 : tile == 1 || tile == 3 ? 64 : (t64 < 160 ? 32 : 64);
            return take(h32 >= 64 ? F16F_N32_H64 : F16F_N32_H32);
        }
        const int h = tile == 1 ? 128 : tile == 3 ? 64 : tile == 2 ? 32 : (t64 < 160 ? 32 : (t64 <= 320 || mn) ? 64 : 128);
        if (!mn && h == 128) return take(F16F_S4_H128);
        return take(h >= 64 ? F16F_S4_H64 : F16F_S4_H32);
    }
    if (N <= 512 && nbk >= 8) return take(F16F_S2_H128);
    static const int t256 = dev_env_int("GGML_HIP_Q16_T256", 384);   // developer A/B switch
    if (big >= t256) return take(F16F_256x128);
    static const int tile64 = dev_env_int("GGML_HIP_Q16_T64", 0);   // developer A/B switch: 1 = never, 2 = always
    if (tile64 == 2 || (!mn && tile64 != 1 && cdiv(M, 128) * cdiv(N, 64) <= 256)) return take(F16F_64x64);
    return take(F16F_128x64);
}

// the staged int8 family (gemm_q.hip): 128 x 128 tiles when they already give >= 2 workgroups per CU, else 64 x 64; one chain over K either way
void plan_i8(mm_plan &p, int type, int64_t M, int64_t N) {
    static const char *force = dev_env_str("GGML_HIP_GEMM_TILE");  // developer override: "1" = 64x64, "2" = 128x128
    const int64_t big = cdiv(M, 128) * cdiv(N, 128);
    const bool b = force && force[0] == '1' ? false : force && force[0] == '2' ? true : big >= 512;
    p.family = MMF_I8; p.image = 0; p.form = b ? I8F_128x128 : I8F_64x64;
    p.arith = 600 + type; p.ksplit = 1; p.kstyle = MMK_CHAIN; p.kunit = 4;
    p.tile_m = p.tile_n = b ? 128 : 64; p.waves = 4; p.tiles_per_wave = b ? 4 : 1;
    p.wgs = cdiv(M, p.tile_m) * cdiv(N, p.tile_n);
}

void plan_dense(mm_plan &p, int type, int64_t M, int64_t Mpad, int64_t K, int64_t N) {
    const bool f16 = type == GGML_TYPE_F16;
    const uint64_t Kpad = (uint64_t)dense16_kpad(K), Npad = (uint64_t)pad_act(N);
    p.image = -1;
    // F16, more than 4 src1 rows: the f16 matrix cores (by N alone; K % 4: the INIT kernel reads src1 rows in 16-byte pieces, and a
    // contiguous row is K floats -- r4: the choice no longer depends on whether the CALLER brought a work buffer or an aligned src1:
    // ggml_hip_mul_mat_dev reports those as errors)
    if (f16 && N > 4 && K % 4 == 0) {
        if ((Kpad / 8 + DENSE16_SPARE_PANELS) * (uint64_t)Mpad * 16 > LIM32 || (Kpad / 8) * Npad * 16 > LIM32) {
            p.flags |= MM_FLAG_WIDE;
        } else {
            static const int shape = dev_env_int("GGML_HIP_D16_SHAPE", 0);   // developer A/B switch: 1 = the 32 x 32 x 16 forms everywhere
            static const int var = dev_env_int("GGML_HIP_D16_TILE", 0);      // developer A/B switch
            const int64_t nst = (int64_t)Kpad / (16 * 8);
            const int64_t tm128 = cdiv(M, 128), tn128 = cdiv(N, 128);
            p.family = MMF_DENSE16; p.image = 32; p.flags |= MM_FLAG_NEEDS_WORK;
            p.kunit = 8;
            auto take = [&](int f, int tm, int tn, int waves, int ks, int shape16) {
                p.form = f; p.tile_m = tm; p.tile_n = tn; p.waves = waves; p.ksplit = ks; p.kstyle = ks > 1 ? MMK_STAGE_SETS : MMK_CHAIN;
                p.arith = 700 + shape16; p.wgs = cdiv(M, tm) * cdiv(N, tn);
                p.tiles_per_wave = tm * tn / 1024 * ks / waves > 0 ? tm * tn / 1024 * ks / waves : 1;
            };
            static const int d16t = dev_env_int("GGML_HIP_D16_T256", 384);   // developer A/B switch
            if (N > 512 && shape != 1)   // (v_mfma_f32_16x16x32_f16, by N alone: the two tile sizes sum alike)
                return cdiv(M, 256) * tn128 >= d16t ? take(D16F_S_256x128, 256, 128, 4, 1, 1) : take(D16F_S_128x128, 128, 128, 4, 1, 1);
            if (N > 512 && cdiv(M, 256) * tn128 >= 384) return take(D16F_256x128, 256, 128, 4, 1, 0);
            static const int s4n = dev_env_int("GGML_HIP_D16_S4_NMAX", 128);   // developer A/B switch (r4: 256 measured -- 4096 x 4096 x 192 level, 11008 x 4096 x 160 44.8 | 65.9 us: 128 stays)
            if (N <= s4n && nst >= 8 && var != 9)
                return var == 1 || (var != 2 && tm128 * cdiv(N, 64) >= 80) ? take(D16F_S4_H128, 128, 64, 16, 4, 0) : take(D16F_S4_H32, 32, 64, 4, 4, 0);
            if (N <= 512 && nst >= 4 && var != 9 && var != 8) {
                // (r4: from 257 tiles on, it was 512 -- the 8-wave form below holds one workgroup per CU, so 257..511 tiles ran it in two rounds, the
                // second nearly empty: 11008 x 4096 x 512 87.9 -> 75.7 us, x 257 85.7 -> 69.9, 14336 x 4096 x 512 94.0 -> 84.9; 256 tiles: level)
                if (var == 5 || (var == 0 && tm128 * tn128 > 256)) return take(D16F_V2_128x128, 128, 128, 4, 2, 0);
                if (var == 7 || (var != 6 && tm128 * tn128 >= 160)) return take(D16F_S2_128x128, 128, 128, 8, 2, 0);
                return take(D16F_S2_128x64, 128, 64, 8, 2, 0);
            }
            return take(D16F_128x128, 128, 128, 4, 1, 0);
        }
    }
    // F32 above 256 src1 rows: the bf16 cores, every operand split exactly into three bf16 pieces (K10d).  By N alone.
    if (!f16 && N > 256 && K % 4 == 0) {
        static const int old = dev_env_int("GGML_HIP_D32_OLD", 0);   // developer A/B switch: 1 = dense.hip (f32 matrix instruction) everywhere
        if ((Kpad / 8 * 3 + DENSE32_SPARE_PANELS) * (uint64_t)Mpad * 16 > LIM32 || (Kpad / 8 * 3) * Npad * 16 > LIM32) {
            p.flags |= MM_FLAG_WIDE;
        } else if (old != 1) {
            p.family = MMF_DENSE32; p.image = 33; p.form = 0; p.flags |= MM_FLAG_NEEDS_WORK;
            p.arith = 710; p.ksplit = 1; p.kstyle = MMK_CHAIN; p.kunit = 1;
            p.tile_m = p.tile_n = 128; p.waves = 4; p.tiles_per_wave = 16; p.wgs = cdiv(M, 128) * cdiv(N, 128);
            return;
        }
    }
    static const bool oldd = dev_env_set("GGML_HIP_DENSE_OLD");   // developer A/B switch
    // r4: F32 weights, 5 .. 256 src1 rows, K a multiple of 256 from 1024 on: 32 x 32 tiles with K split over the workgroup's eight waves
    // (dense.hip dense_f32_ksplit_kernel: the tile kernels run a flat 120 us at 4096 x 4096 whatever N is).  By N and K alone.
    static const int ksn = dev_env_int("GGML_HIP_D32_KS_NMIN", 5);   // developer A/B switch (17: the mat-vec form keeps 5..16 rows)
    if (!f16 && N >= ksn && N <= 256 && K % 256 == 0 && K >= 1024 && !oldd) {
        p.family = MMF_DENSE; p.form = DNF_KSPLIT;
        p.arith = 740; p.ksplit = 8; p.kstyle = MMK_RANGES; p.kunit = (int)(K / 256);
        p.tile_m = p.tile_n = 32; p.waves = 8; p.tiles_per_wave = 1; p.wgs = cdiv(M, 32) * cdiv(N, 32);
        return;
    }
    // dense.hip.  Mat-vec form up to 16 rows (passes of 8 columns; rows of the resident copy are K elements apart: 16-byte pieces need
    // K % 8 (f16) / K % 4 (f32)); else the 64 x 64 tile kernel (F32: 128 x 128 tiles where they fill the chip -- bitwise the same result)
    if (N <= 16 && K % (f16 ? 8 : 4) == 0 && K >= 512) {
        p.family = MMF_DENSE_GEMV; p.form = 8;
        p.arith = 720 + (f16 ? 1 : 0); p.ksplit = 64; p.kstyle = MMK_WORKERS; p.kunit = 1;
        p.tile_m = 4; p.tile_n = 8; p.waves = 4; p.tiles_per_wave = 1; p.wgs = cdiv(M, 16);
        return;
    }
    const int64_t tm = cdiv(M, 128), tn = cdiv(N, 128);
    const bool bigt = !f16 && K % 32 == 0 && !oldd && tm * tn >= 256 && tm * tn < (1 << 30) && Mpad % 128 == 0;
    p.family = MMF_DENSE; p.form = bigt ? DNF_BIG : DNF_TILE;
    p.arith = 730 + (f16 ? 1 : 0); p.ksplit = 1; p.kstyle = MMK_CHAIN; p.kunit = 32;
    p.tile_m = p.tile_n = bigt ? 128 : 64; p.waves = 4; p.tiles_per_wave = 1; p.wgs = cdiv(M, p.tile_m) * cdiv(N, p.tile_n);
}

}  // namespace

mm_plan plan_mul_mat(int type, int ext_type, int64_t M, int64_t K, int64_t N, bool one_call) {
    mm_plan p = {};
    p.ksplit = 1;
    const int64_t Mpad = pad_rows(M > 0 ? M : 1);
    if (type == GGML_TYPE_F32 || type == GGML_TYPE_F16) { plan_dense(p, type, M, Mpad, K, N); return p; }
    if (!is_quant(type) || K <= 0 || K % QK != 0) return p;
    if (ext_type != 0) p.flags |= MM_FLAG_Q8K;
    // the stated exception: planes that do not fit 32-bit buffer offsets (> 4 GiB per plane) are served by the int8 family and its image,
    // whatever the type.  INIT and COMPUTE both come through here, so they agree.  (For the K3p-int8 types the bound is the form's own:
    // its waves' ranges reach up to 15 k-blocks past the padded end -- with the generic bound a matrix just under the limit fell to the
    // staged int8 kernel while its row shards ran K3p)
    const uint64_t nba = (uint64_t)pad_kblocks(K / QK);
    bool wide = (nba + K_LOOKAHEAD) * (uint64_t)Mpad * 32 > LIM32;
    int kind = wide ? 0 : plan_image_kind(type, K, N);
    bool no_fused = false;
    const bool i8_only = ext_type != 0 && type == GGML_TYPE_Q4_2;   // (Q6_K: no nibble plane -- the int8 forms only)
    const bool small = q8_small_serves(type, K, N, i8_only, ext_type != 0), mid = q8_mid_serves(type, K, N);
    // r5: K3s-int8 and K3p-int8 are ONE summation tree -- the same eight K ranges (k3p_i8_nloc), a range's min-term chunks and blocks in the
    // same order with the same statement, the eight sums added in wave order: the same bits per element (tests/test_gpu_fullsize.py
    // test_k3s_and_k3p_int8_compute_the_same_bits) and the same tree_id -- so where both serve (K3_DUAL_MIN .. K3_DUAL_MAX src1 rows) the FAMILY follows M: K3p once
    // its grid of 64-row tiles has k3_dual_wgs(K) workgroups, K3s below.  K3s | K3p, us per COMPUTE launch (tools/experiments/ab_k3s_k3p_overlap.sh): Q8_0 4096 x 4096 x 128 13.3 | 15.6,
    // 4096 x 11008 x 128 27.6 | 32.4, 1024 x 4096 x 256 7.8 | 13.8, 2048 x 4096 x 256 11.8 | 15.3, 1024 x 11008 x 256 16.9 | 29.2, 2048 x 8192 x 160 17.3 | 22.9 (128 workgroups
    // or fewer: K3s) -- 8192 x 4096 x 128 20.4 | 17.9, 11008 x 4096 x 65 / 128 31.1 | 27.0, 35.2 | 27.5, 16384 x 4096 x 128 39.7 | 31.4, 32000 x 4096 x 128 75.2 | 60.0, 8192 x 8192 x 128 33.0 | 30.0,
    // Q5_1 11008 x 4096 x 128 38.3 | 30.5, Q4_2 32000 x 4096 x 128 118 | 96.5, Q4_2 4096 x 4096 x 192 27.3 | 20.1 (192 workgroups and more: K3p).
    // (160 behind K < 11008, 192 behind a longer K -- ab_dual_wgs.sh, K3s | K3p: Q8_0 11008 x 4096 x 64 (172 workgroups) 20.0 | 17.2, Q5_1 27.5 | 19.5, Q4_2 31.3 | 21.3, 10240 x 4096 x 48 (160) 19.2 | 15.9,
    // 5120 x 4096 x 128 (160) 18.3 | 15.6, 2560 x 4096 x 256 (160) 16.4 | 13.9; but 5120 x 13824 x 96 (160) 32.5 | 41.4; at 128 workgroups K3s: 8192 x 4096 x 64 14.6 | 16.9, 4096 x 4096 x 128 13.4 | 15.7)
    static const int dual_dev = dev_env_int("GGML_HIP_K3_DUAL_WGS", 0);   // developer A/B switch: K3p from this many 64-row workgroups (1: always K3p, 1000000: never; 0: the rule)
    const int64_t dual = dual_dev > 0 ? dual_dev : k3_dual_wgs(K);
    const bool mid_first = small && mid && cdiv(M, 64) * cdiv(N, 64) >= dual;
    if (small && !mid_first) {                              // (r4: Q5_K weights too -- planar Q5_1 form, activations by the Q8_K rule)
        if (plan_k3s_i8(p, type, M, Mpad, K, N)) { if (ext_type != 0) p.flags &= ~MM_FLAG_EPILOGUE_FUSED; p.flags |= MM_FLAG_NEEDS_WORK; return p; }
        if (!mid) { wide = true; kind = 0; no_fused = true; }   // (its planes are beyond the form's offsets: the two-step forms below)
    }
    if (mid) {                                              // (Q5_K weights too: they live in the planar Q5_1 form, their activations in image 0 by the Q8_K rule)
        if (plan_k3p_i8(p, type, M, Mpad, K, N)) { if (ext_type != 0) p.flags &= ~MM_FLAG_EPILOGUE_FUSED; p.flags |= MM_FLAG_NEEDS_WORK; return p; }
        p = mm_plan{}; p.ksplit = 1; if (ext_type != 0) p.flags |= MM_FLAG_Q8K;
        if (mid_first && plan_k3s_i8(p, type, M, Mpad, K, N)) { if (ext_type != 0) p.flags &= ~MM_FLAG_EPILOGUE_FUSED; p.flags |= MM_FLAG_NEEDS_WORK; return p; }
        wide = true; kind = 0; if (small) no_fused = true;
    }
    if (wide) p.flags |= MM_FLAG_WIDE;
    if (kind == 0) {
        // (r4: the k-quants of the planar Q5_1 form have a fused mat-vec too, up to 4 rows and K = 32768 -- gemv.hip K8: the Q8_K rule in the kernel)
        const bool kq_fused = ext_type != 0 && (type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_2) && N <= 4 && K % 256 == 0 && K / 256 <= 128;
        if (one_call && N <= GEMV_MAX_N && (ext_type == 0 || kq_fused) && !no_fused) { plan_gemv(p, type, M, K, N, true); return p; }
        p.flags |= MM_FLAG_NEEDS_WORK;
        if (N <= (i8_only ? 4 : gemv_rows_max(type))) { plan_gemv(p, type, M, K, N, false); return p; }   // (Q6_K: the mat-vec on int8 planes up to 4 rows, then K3s / the staged form)
        plan_i8(p, type, M, N);
        return p;
    }
    if (kind == 3) plan_mx(p, type, M, Mpad, K, N);
    else plan_f16(p, type, M, K, N);
    p.flags |= MM_FLAG_NEEDS_WORK;
    if (ext_type != 0) p.flags &= ~MM_FLAG_EPILOGUE_FUSED;  // (the fused seams are for the reference's own types)
    return p;
}

uint32_t plan_tree_id(const mm_plan &p) {
    // what fixes an element's bits, and nothing of the geometry
    uint32_t h = 2166136261u;
    const int parts[] = {p.arith, p.ksplit, p.kstyle, p.kunit, p.flags & MM_FLAG_Q8K};
    for (int v : parts) { h ^= (uint32_t)v; h *= 16777619u; }
    return h;
}
