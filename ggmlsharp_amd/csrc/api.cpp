// api.cpp -- the C-ABI of include/ggml_hip.h: lifecycle, resident weights, the two seams.
// No CPU fallback anywhere: without a device every compute entry returns GGML_HIP_ERR_NO_DEVICE.
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace {

thread_local std::string g_err;
std::mutex g_mu;
int g_device = -1;
bool g_inited = false;
std::string g_arch;
hipStream_t g_stream = nullptr;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

const int BLCK[GGML_TYPE_COUNT] = {1, 1, 32, 32, 16, 16, 32, 32, 32, 32, 1, 1, 1};          // Ggml.cs:55-70
const size_t TSIZE[GGML_TYPE_COUNT] = {4, 2, 20, 24, 10, 12, 22, 24, 36, 44, 1, 2, 4};      // Ggml.cs:72-87

// quantized types with working row functions and dot products.  Q4_2 and Q5_1 follow SURVEY D7's intent (their half
// scales are IEEE bit patterns, as in the upstream scalar code the C# transcribes); Q4_3 / Q8_1 have null slots (D8).
bool wq_ok(int t) {
    return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q4_2 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 ||
           t == GGML_TYPE_Q8_0;
}
bool weight_type_ok(int t) { return wq_ok(t) || t == GGML_TYPE_F32 || t == GGML_TYPE_F16; }
// src1 rows up to which the mat-vec kernel serves a type (two-step form above GEMV_MAX_N): the types with small-batch MFMA
// configurations leave it at 8; Q4_2, which only has the int8 kernel's 64 x 64 tiles behind it, stays on it up to 16
int64_t gemv_rows_max(int t) { return t == GGML_TYPE_Q4_2 ? GEMV_WIDE_MAX_N : GEMV_MAX_N; }
bool has_min_plane(int t) { return t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_Q4_2; }   // Q4_2: its second scale
bool has_qh_plane(int t) { return t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1; }
size_t row_bytes_of(int t, int64_t k) { return TSIZE[t] * (size_t)(k / BLCK[t]); }
bool is_q(int t) { return t >= GGML_TYPE_Q4_0 && t <= GGML_TYPE_Q8_1; }
int vec_dot_type(int t) {  // Ggml.cs:219-290
    switch (t) {
    case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_2: case GGML_TYPE_Q5_0: case GGML_TYPE_Q8_0: return GGML_TYPE_Q8_0;
    case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_1: return GGML_TYPE_Q8_1;
    default: return -1;
    }
}

// Which MFMA kernel (and so which activation image K1 writes) serves a quantized mat-mat (measured on MI355X):
//   gemm_qmx.hip (MX matrix path, bf6 digits, one exact MFMA per tile and block) -- Q4_0 / Q4_1,
//   gemm_q16.hip (f16 matrix cores, register-tile design) -- Q5_0 / Q5_1 / Q8_0 on prompt-sized batches (N <= 512, K split in the
//                workgroup) and from 1024 rows up,
//   gemm_q.hip   (int8 matrix cores, 64 x 64 / 128 x 128 tiles) -- Q4_2, and Q5_0 / Q5_1 / Q8_0 in between.
// GGML_HIP_GEMM=mx|f16|i8 forces one (developer A/B switch).  The MX kernel also has a two-digit form for Q5_0 / Q8_0
// (two MFMAs per tile and block); it measured no faster than the kernels above, so its digit planes
// (1.5 B / weight) are only built for weights uploaded while "mx" is forced.
// Returns the K1 image kind: 0 = int8 planes, 1 / 2 = the f16 images of gemm_q16.hip, 3 = the bf6 image of gemm_qmx.hip.
std::atomic<int> g_force_gemm{-1};   // -1: read GGML_HIP_GEMM once; 0 auto, 1 int8, 2 f16, 3 MX (ggml_hip_debug_force_gemm)

int gemm_force() {
    int force = g_force_gemm.load();
    if (force < 0) {
        const char *e = getenv("GGML_HIP_GEMM");
        force = !e ? 0 : (e[0] == 'i' ? 1 : (e[0] == 'f' ? 2 : (e[0] == 'm' ? 3 : 0)));
        g_force_gemm.store(force);
    }
    return force;
}

int act_image_kind(int type, int64_t M, int64_t K, int64_t N) {
    const int force = gemm_force();
    if (N <= GEMV_MAX_N || force == 1) return 0;
    if (type == GGML_TYPE_Q4_2) return 0;   // served by the mat-vec and int8 kernels only (its k-block carries two scales)
    // the MX / f16 kernels address weights and the activation image through 32-bit buffer offsets
    const uint64_t nba = (uint64_t)pad_kblocks(K / QK);
    if ((nba + K_LOOKAHEAD) * (uint64_t)pad_rows(M) * 32 > 0xFFFFFFFFull || nba * 64 * (uint64_t)pad_act(N) > 0xFFFFFFFFull) return 0;
    if (force == 2) return gemm_q16_image_kind(type);
    if (type == GGML_TYPE_Q5_1 && force == 3) return 0;                 // no MX form: the forced choice falls back to the int8 kernel
    if (force == 3 || type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1) return 3;
    // Q5_0 / Q5_1 / Q8_0: the f16 kernel's K-split form wins on prompt-sized batches (4096 x 11008 x 512: 92 / 110 us against 108 /
    // 130 us for the int8 kernel) and its unsplit forms from 1024 rows up (Q8_0 4096^3 199 against 220 us, 32000 x 4096 x 1024 405
    // against 450, 4096 x 11008 x 1024 172 against 194; Q5_1 4096 x 4096 x 1024 89 against 115 -- 4096 x 4096 x 2048 is the
    // one shape measured the other way, 124 against 116); in between (4096 x 4096 x 640: 67 against 57 us) the int8 kernel's
    // 64 x 64 tiles balance the chip better.  Decided from N and K only, like the K split itself: never from M, so a row shard
    // runs the kernel form of the unsplit matrix.
    (void)M;
    return ((N <= 512 || N >= 1024) && K / QK >= 8) ? gemm_q16_image_kind(type) : 0;
}

int ensure_init() {
    if (g_inited) {
        hipError_t e = hipSetDevice(g_device);  // calls arrive on arbitrary threads (SURVEY 8(b) "Threading")
        if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "hipSetDevice: %s", hipGetErrorString(e));
        return GGML_HIP_OK;
    }
    return ggml_hip_init(g_device < 0 ? 0 : g_device);
}

// ---- host-level scratch (Seam 1 / Seam 2 host forms) ----
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, n) != hipSuccess) return -1;
        cap = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
Scratch g_src1, g_dst, g_work, g_stage;

// ---- graph-level residency: between ggml_hip_graph_begin / _end the device copy of every offloaded
// node's dst stays alive, keyed by the host pointer of the tensor data.  A later node whose src1 IS that tensor reads it
// from HBM instead of taking it back over PCIe, and the device -> host copies (the reference's contract: every node's
// data is in host memory when ggml_graph_compute returns) are synchronised once, at graph end, behind the compute. ----
struct Resident { void *p; size_t bytes; };
int g_graph_depth = 0;
std::map<const void *, Resident> g_resident;          // host data pointer -> device copy (graph scope)
std::vector<Resident> g_pool;                          // device buffers free for reuse
uint64_t g_h2d_bytes = 0, g_d2h_bytes = 0, g_resident_hits = 0;

void *pool_take(size_t n) {
    for (size_t i = 0; i < g_pool.size(); ++i)
        if (g_pool[i].bytes >= n && g_pool[i].bytes <= 2 * n + 4096) {
            void *p = g_pool[i].p;
            g_resident[nullptr] = g_pool[i];   // placeholder slot, replaced by the caller's key
            g_pool.erase(g_pool.begin() + (long)i);
            return p;
        }
    void *p = nullptr;
    if (hipMalloc(&p, n) != hipSuccess) return nullptr;
    g_resident[nullptr] = Resident{p, n};
    return p;
}
// ---- operands / results of the f32 element-wise seams: contiguous tensors, device-resident inside a graph scope ----
bool contiguous_f32(const ggml_tensor *t) {
    return t->type == GGML_TYPE_F32 && t->nb[0] == 4 && t->nb[1] == (uint64_t)t->ne[0] * 4 && t->nb[2] == t->nb[1] * (uint64_t)t->ne[1] &&
           t->nb[3] == t->nb[2] * (uint64_t)t->ne[2];
}
int64_t nelem(const ggml_tensor *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
// device pointer of a contiguous f32 operand: its resident copy if an earlier node of this graph produced it, else an upload
int operand_f32(const ggml_tensor *t, Scratch &scratch, const float **out, hipStream_t st);
int operand_f32(const ggml_tensor *t, Scratch &scratch, const float **out, hipStream_t st) {
    const size_t bytes = (size_t)nelem(t) * 4;
    if (g_graph_depth > 0) {
        auto r = g_resident.find(t->data);
        if (r != g_resident.end() && r->second.bytes >= bytes) { *out = (const float *)r->second.p; ++g_resident_hits; return 0; }
        if (hipStreamSynchronize(st) != hipSuccess) return -1;   // host memory may still be receiving an earlier node's result
    }
    if (scratch.ensure(bytes)) return -1;
    if (hipMemcpyAsync(scratch.p, t->data, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
    g_h2d_bytes += bytes;
    *out = (const float *)scratch.p;
    return 0;
}
// device buffer for a contiguous f32 result: kept resident inside a graph scope (in place when dst shares src's data)
float *result_f32(const ggml_tensor *t, Scratch &scratch) {
    const size_t bytes = (size_t)nelem(t) * 4;
    if (g_graph_depth > 0) {
        auto old = g_resident.find(t->data);
        if (old != g_resident.end() && old->second.bytes >= bytes) return (float *)old->second.p;
        if (old != g_resident.end()) { g_pool.push_back(old->second); g_resident.erase(old); }
        void *p = pool_take(bytes);
        if (!p) return nullptr;
        g_resident[t->data] = g_resident[nullptr];
        g_resident.erase(nullptr);
        return (float *)p;
    }
    return scratch.ensure(bytes) ? nullptr : (float *)scratch.p;
}
int finish_f32(ggml_tensor *t, const float *dev, hipStream_t st) {
    const size_t bytes = (size_t)nelem(t) * 4;
    if (hipMemcpyAsync(t->data, dev, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    g_d2h_bytes += bytes;
    if (g_graph_depth == 0 && hipStreamSynchronize(st) != hipSuccess) return -1;
    return 0;
}

void pool_drain_locked(bool free_all) {
    for (auto &kv : g_resident) g_pool.push_back(kv.second);
    g_resident.clear();
    if (free_all) {
        for (Resident &r : g_pool) (void)hipFree(r.p);
        g_pool.clear();
    }
}

// ---- weight cache for Seam 1, keyed by the host pointer + shape ----
using CacheKey = std::tuple<const void *, int, int64_t, int64_t, int64_t, int64_t, uint64_t, uint64_t, uint64_t>;
std::map<CacheKey, std::vector<ggml_hip_weight *>> g_cache;

int alloc_weight(int type, int64_t K, int64_t M, ggml_hip_weight **out) {
    ggml_hip_weight *w = new ggml_hip_weight();
    memset(w, 0, sizeof *w);
    w->type = type; w->M = M; w->K = K; w->Mpad = pad_rows(M > 0 ? M : 1); w->device = g_device;
    size_t off_qs = 0, off_d = 0, off_m = 0, off_qh = 0, off_6a = 0, off_6b = 0, total = 0;
    bool with6 = false;
    size_t off_p16 = 0;
    if (type == GGML_TYPE_F32 || type == GGML_TYPE_F16) {
        total = ((size_t)w->Mpad * K * (type == GGML_TYPE_F32 ? 4 : 2) + 255) / 256 * 256;
        if (type == GGML_TYPE_F16) {   // k-panel copy for the f16 MFMA kernel (dense16.hip)
            off_p16 = total;
            total += (size_t)(dense16_kpad(K) / 8 + DENSE16_SPARE_PANELS) * w->Mpad * 16;
        }
    } else {
        w->nbk = K / QK;
        const int64_t nba = pad_kblocks(w->nbk) + K_LOOKAHEAD;   // allocated k-blocks (zero past the real end)
        const size_t qs_bytes = (size_t)nba * w->Mpad * (type == GGML_TYPE_Q8_0 ? 32 : 16);
        const size_t plane = (size_t)nba * w->Mpad * 4;
        off_qs = 0; total = qs_bytes;
        off_d = total; total += plane;
        if (has_min_plane(type)) { off_m = total; total += plane; }
        if (has_qh_plane(type)) { off_qh = total; total += plane; }
        const bool q4 = type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1;
        with6 = q4 || (gemm_force() == 3 && (type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q8_0));
        if (with6) {   // bf6 operand planes of the MX mat-mat kernel: 0.75 B / weight and digit (Q5_0, Q8_0: two digits)
            const size_t nf = q4 ? 1 : 2;
            off_6a = total; total += (size_t)nba * nf * w->Mpad * 16;
            off_6b = total; total += (size_t)nba * nf * w->Mpad * 8;
        }
    }
    if (total == 0) total = 16;
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, total);
    if (e != hipSuccess) { delete w; return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc(%zu): %s", total, hipGetErrorString(e)); }
    w->bytes = total;
    if (type == GGML_TYPE_F32 || type == GGML_TYPE_F16) {
        w->dense = base;
        if (type == GGML_TYPE_F16) w->p16 = (uint8_t *)base + off_p16;
    } else {
        w->qs = (uint8_t *)base + off_qs;
        w->d = (float *)((uint8_t *)base + off_d);
        if (has_min_plane(type)) w->m = (float *)((uint8_t *)base + off_m);
        if (has_qh_plane(type)) w->qh = (uint32_t *)((uint8_t *)base + off_qh);
        if (with6) { w->q6a = (uint8_t *)base + off_6a; w->q6b = (uint8_t *)base + off_6b; }
    }
    *out = w;
    return GGML_HIP_OK;
}

void *weight_base(const ggml_hip_weight *w) { return w->dense ? w->dense : (void *)w->qs; }

int make_weight(int type, const void *rows, bool rows_on_host, int64_t ne00, int64_t ne01, uint64_t nb01,
                int64_t row_begin, int64_t row_end, hipStream_t st, ggml_hip_weight **out) {
    if (!out) return fail(GGML_HIP_ERR_ARG, "out is null");
    *out = nullptr;
    if (type < 0 || type >= GGML_TYPE_COUNT || !weight_type_ok(type))
        return fail(GGML_HIP_ERR_TYPE, "type %d is not a supported weight type (Q4_3/Q8_1 have null slots, Ggml.cs:248,278-282)", type);
    if (!rows || ne00 <= 0 || ne01 < 0 || row_begin < 0 || row_end < row_begin || row_end > ne01)
        return fail(GGML_HIP_ERR_ARG, "bad weight arguments");
    if (ne00 % BLCK[type] != 0 || (is_q(type) && ne00 % QK != 0))   // every dot product runs against 32-element Q8 blocks
        return fail(GGML_HIP_ERR_SHAPE, "ne00 %% %d != 0 (Ggml.cs:6694)", is_q(type) ? QK : BLCK[type]);
    const uint64_t row_bytes = (uint64_t)TSIZE[type] * (uint64_t)(ne00 / BLCK[type]);
    if (nb01 < row_bytes) return fail(GGML_HIP_ERR_SHAPE, "nb01 smaller than a row (transposed src0, Ggml.cs:8229)");
    int rc = ensure_init();
    if (rc) return rc;
    const int64_t rows_n = row_end - row_begin;
    ggml_hip_weight *w = nullptr;
    rc = alloc_weight(type, ne00, rows_n, &w);
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(weight_base(w), 0, w->bytes, st);
    const uint8_t *dev_rows = (const uint8_t *)rows;
    void *staging = nullptr;
    if (e == hipSuccess && rows_on_host && rows_n > 0) {
        e = hipMalloc(&staging, (size_t)rows_n * row_bytes);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(staging, row_bytes, (const uint8_t *)rows + (uint64_t)row_begin * nb01, nb01, row_bytes,
                                 (size_t)rows_n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = launch_repack_to_planar(type, (const uint8_t *)staging, row_bytes, 0, rows_n, w, st);
    } else if (e == hipSuccess) {
        e = launch_repack_to_planar(type, dev_rows, nb01, row_begin, rows_n, w, st);
    }
    if (e == hipSuccess) e = launch_nibbles_to_bf6(w, st);
    if (e == hipSuccess) e = launch_f16_rows_to_panels(w, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (staging) (void)hipFree(staging);
    if (e != hipSuccess) {
        (void)hipFree(weight_base(w));
        delete w;
        return fail(GGML_HIP_ERR_RUNTIME, "weight upload: %s", hipGetErrorString(e));
    }
    *out = w;
    return GGML_HIP_OK;
}

void invalidate_locked(const void *host_ptr) {
    for (auto it = g_cache.begin(); it != g_cache.end();) {
        if (std::get<0>(it->first) == host_ptr) {
            for (ggml_hip_weight *w : it->second) ggml_hip_weight_free(w);
            it = g_cache.erase(it);
        } else {
            ++it;
        }
    }
}

void free_cache_locked() {
    for (auto &kv : g_cache)
        for (ggml_hip_weight *w : kv.second) ggml_hip_weight_free(w);
    g_cache.clear();
}

}  // namespace

extern "C" {

int ggml_hip_blck_size(int type) { return (type >= 0 && type < GGML_TYPE_COUNT) ? BLCK[type] : 0; }
size_t ggml_hip_type_size(int type) { return (type >= 0 && type < GGML_TYPE_COUNT) ? TSIZE[type] : 0; }

int ggml_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ggml_hip_init(int device) {
    int n = ggml_hip_device_count();
    if (n <= 0) return fail(GGML_HIP_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
    if (device < 0 || device >= n) return fail(GGML_HIP_ERR_ARG, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_inited && g_device == device) return GGML_HIP_OK;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    g_arch = prop.gcnArchName;
    if (g_stream == nullptr) HIP_TRY(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    g_device = device;
    g_inited = true;
    return GGML_HIP_OK;
}

void ggml_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_inited) return;
    (void)hipSetDevice(g_device);
    free_cache_locked();
    g_src1.release(); g_dst.release(); g_work.release(); g_stage.release();
    pool_drain_locked(true); g_graph_depth = 0;
    if (g_stream) (void)hipStreamDestroy(g_stream);
    g_stream = nullptr;
    g_inited = false;
}

const char *ggml_hip_last_error(void) { return g_err.c_str(); }
const char *ggml_hip_arch(void) { return g_arch.c_str(); }

int ggml_hip_weight_upload(int type, const void *host_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                           int64_t row_begin, int64_t row_end, void *stream, ggml_hip_weight **out) {
    return make_weight(type, host_rows, true, ne00, ne01, nb01, row_begin, row_end, (hipStream_t)stream, out);
}

int ggml_hip_weight_from_device(int type, const void *dev_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                                int64_t row_begin, int64_t row_end, void *stream, ggml_hip_weight **out) {
    return make_weight(type, dev_rows, false, ne00, ne01, nb01, row_begin, row_end, (hipStream_t)stream, out);
}

int ggml_hip_weight_download(const ggml_hip_weight *w, void *host_rows, void *stream) {
    if (!w || !host_rows) return fail(GGML_HIP_ERR_ARG, "null argument");
    int rc = ensure_init();
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t row_bytes = TSIZE[w->type] * (size_t)(w->K / BLCK[w->type]);
    const size_t total = row_bytes * (size_t)w->M;
    if (total == 0) return GGML_HIP_OK;
    void *staging = nullptr;
    HIP_TRY(hipMalloc(&staging, total));
    hipError_t e = launch_planar_to_aos(w, (uint8_t *)staging, st);
    if (e == hipSuccess) e = hipMemcpyAsync(host_rows, staging, total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "weight download: %s", hipGetErrorString(e));
    return GGML_HIP_OK;
}

void ggml_hip_weight_free(ggml_hip_weight *w) {
    if (!w) return;
    (void)hipFree(weight_base(w));
    delete w;
}
int64_t ggml_hip_weight_rows(const ggml_hip_weight *w) { return w ? w->M : 0; }
int64_t ggml_hip_weight_cols(const ggml_hip_weight *w) { return w ? w->K : 0; }
int ggml_hip_weight_type(const ggml_hip_weight *w) { return w ? w->type : -1; }

size_t ggml_hip_mul_mat_work_size(int type, int64_t K, int64_t N) {
    if (K <= 0 || N <= 0) return 0;
    if (type == GGML_TYPE_F16) return (size_t)dense16_kpad(K) * (size_t)pad_act(N) * 2;   // src1 as Half (Ggml.cs:3356-3357), padded
    if (!is_q(type)) return 0;
    return act_bytes(K, pad_act(N));
}

int ggml_hip_mul_mat_init_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, void *d_work,
                              size_t work_bytes, void *stream) {
    if (!w || !d_src1) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (N <= 0) return GGML_HIP_OK;
    if (!is_q(w->type)) return GGML_HIP_OK;  // dense: INIT is a no-op for f32 (Ggml.cs:6117-6120); f16 rounds in-kernel
    if (ld1 < w->K) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K");
    if (!d_work || work_bytes < ggml_hip_mul_mat_work_size(w->type, w->K, N))
        return fail(GGML_HIP_ERR_ARG, "work buffer too small: need %zu", ggml_hip_mul_mat_work_size(w->type, w->K, N));
    act_planes p = act_carve(d_work, w->K, pad_act(N));
    HIP_TRY(launch_quantize_act(d_src1, N, w->K, ld1, p, act_image_kind(w->type, w->M, w->K, N), (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_act_image_kind(int type, int64_t M, int64_t K, int64_t N) { return act_image_kind(type, M, K, N); }
void ggml_hip_debug_force_gemm(int which) { g_force_gemm.store(which < 0 || which > 3 ? 0 : which); }

int ggml_hip_quantize_act_dev(const float *d_src1, int64_t N, int64_t K, int64_t ld1, void *d_work, size_t work_bytes,
                              int image_kind, void *stream) {
    if (N <= 0) return GGML_HIP_OK;
    if (!d_src1 || !d_work) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (K <= 0 || K % QK != 0 || ld1 < K) return fail(GGML_HIP_ERR_SHAPE, "K %% 32 != 0 or ld1 < K");
    if (image_kind < 0 || image_kind > 3) return fail(GGML_HIP_ERR_ARG, "image kind %d", image_kind);
    if (work_bytes < act_bytes(K, pad_act(N))) return fail(GGML_HIP_ERR_ARG, "work buffer too small: need %zu", act_bytes(K, pad_act(N)));
    // the MFMA images are written (and read) through 32-bit buffer offsets: 64 image bytes per row and k-block
    if (image_kind != 0 && (uint64_t)pad_kblocks(K / QK) * 64 * (uint64_t)pad_act(N) > 0xFFFFFFFFull)
        return fail(GGML_HIP_ERR_SHAPE, "image kind %d needs K/32 * 64 * Npad < 4 GiB (ggml_hip_act_image_kind never selects it beyond that)", image_kind);
    HIP_TRY(launch_quantize_act(d_src1, N, K, ld1, act_carve(d_work, K, pad_act(N)), image_kind, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_mul_mat_compute_dev(const ggml_hip_weight *w, int64_t N, float *d_dst, int64_t ldd, const void *d_work,
                                 size_t work_bytes, void *stream) {
    if (!w || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (!is_q(w->type)) return fail(GGML_HIP_ERR_TYPE, "compute_dev is the quantized COMPUTE phase; use ggml_hip_mul_mat_dev for dense");
    if (ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "ldd < M");
    if (!d_work || work_bytes < ggml_hip_mul_mat_work_size(w->type, w->K, N)) return fail(GGML_HIP_ERR_ARG, "work buffer too small");
    act_planes p = act_carve((void *)d_work, w->K, pad_act(N));
    if (N <= gemv_rows_max(w->type))
        HIP_TRY(launch_gemv_q(w, p, N, d_dst, ldd, (hipStream_t)stream));
    else if (act_image_kind(w->type, w->M, w->K, N) == 3)
        HIP_TRY(launch_gemm_qmx(w, p, N, d_dst, ldd, (hipStream_t)stream));
    else if (act_image_kind(w->type, w->M, w->K, N) != 0)
        HIP_TRY(launch_gemm_q16(w, p, N, d_dst, ldd, (hipStream_t)stream));
    else
        HIP_TRY(launch_gemm_q(w, p, N, d_dst, ldd, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_mul_mat_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, float *d_dst,
                         int64_t ldd, void *d_work, size_t work_bytes, void *stream) {
    if (!w) return fail(GGML_HIP_ERR_ARG, "null weight");
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (!d_src1 || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (ld1 < w->K || ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K or ldd < M");
    if (!is_q(w->type)) {
        if (dense16_serves(w, N) && d_work && work_bytes >= ggml_hip_mul_mat_work_size(w->type, w->K, N) && ld1 % 4 == 0 &&
            ((uintptr_t)d_src1 & 15) == 0) {      // (the INIT kernel reads src1 rows in 16-byte pieces)
            HIP_TRY(launch_dense16_init(d_src1, N, w->K, ld1, d_work, (hipStream_t)stream));       // INIT: src1 -> Half (Ggml.cs:6362-6379)
            HIP_TRY(launch_dense16(w, d_work, N, d_dst, ldd, (hipStream_t)stream));
            return GGML_HIP_OK;
        }
        HIP_TRY(launch_dense(w, d_src1, N, ld1, d_dst, ldd, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
    if (N <= GEMV_MAX_N) {  // small N: INIT and COMPUTE fused in one launch, no scratch needed
        HIP_TRY(launch_gemv_q_fused(w, d_src1, ld1, N, d_dst, ldd, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
    int rc = ggml_hip_mul_mat_init_dev(w, d_src1, N, ld1, d_work, work_bytes, stream);
    if (rc) return rc;
    return ggml_hip_mul_mat_compute_dev(w, N, d_dst, ldd, d_work, work_bytes, stream);
}

int ggml_hip_quantize_rows_dev(int type, const float *d_x, int64_t nrows, int64_t k, void *d_blocks, void *stream) {
    if (nrows <= 0) return GGML_HIP_OK;  // empty input: nothing to do (buffers may be null)
    if (!d_x || !d_blocks) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (!(wq_ok(type) || type == GGML_TYPE_Q8_1))
        return fail(GGML_HIP_ERR_TYPE, "quantize: unsupported type %d", type);
    if (k % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "k %% 32 != 0 (Ggml.cs:336)");
    HIP_TRY(launch_quantize_rows(type, GGML_TYPE_F32, d_x, k, nrows, k, d_blocks, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_dequantize_rows_dev(int type, const void *d_blocks, int64_t nrows, int64_t k, float *d_y, void *stream) {
    if (nrows <= 0) return GGML_HIP_OK;
    if (!d_y || !d_blocks) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (!wq_ok(type))
        return fail(GGML_HIP_ERR_TYPE, "dequantize: unsupported type %d (Q8_1 slot is null, Ggml.cs:278)", type);
    if (k % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "k %% 32 != 0 (Ggml.cs:839)");
    HIP_TRY(launch_dequantize_rows(type, d_blocks, nrows, k, d_y, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_quantize_row(int type, const float *x, void *y, int k) {
    if (!x || !y || k <= 0) return fail(GGML_HIP_ERR_ARG, "bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    if (type < 0 || type >= GGML_TYPE_COUNT || !is_q(type)) return fail(GGML_HIP_ERR_TYPE, "not a quantized type");
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t xb = (size_t)k * 4, yb = TSIZE[type] * (size_t)(k / BLCK[type]);
    if (g_src1.ensure(xb) || g_dst.ensure(yb)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(g_src1.p, x, xb, hipMemcpyHostToDevice, g_stream));
    rc = ggml_hip_quantize_rows_dev(type, (const float *)g_src1.p, 1, k, g_dst.p, g_stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(y, g_dst.p, yb, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return GGML_HIP_OK;
}

int ggml_hip_dequantize_row(int type, const void *x, float *y, int k) {
    if (!x || !y || k <= 0) return fail(GGML_HIP_ERR_ARG, "bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    if (type < 0 || type >= GGML_TYPE_COUNT || !is_q(type)) return fail(GGML_HIP_ERR_TYPE, "not a quantized type");
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t yb = (size_t)k * 4, xb = TSIZE[type] * (size_t)(k / BLCK[type]);
    if (g_src1.ensure(xb) || g_dst.ensure(yb)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(g_src1.p, x, xb, hipMemcpyHostToDevice, g_stream));
    rc = ggml_hip_dequantize_rows_dev(type, g_src1.p, 1, k, (float *)g_dst.p, g_stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(y, g_dst.p, yb, hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return GGML_HIP_OK;
}

int ggml_hip_vec_dot(int type, int n, float *s, const void *vx, const void *vy) {
    if (!s || !vx || !vy || n <= 0) return fail(GGML_HIP_ERR_ARG, "bad argument");
    const int vt = (type >= 0 && type < GGML_TYPE_COUNT) ? vec_dot_type(type) : -1;
    if (vt < 0) return fail(GGML_HIP_ERR_TYPE, "vec_dot: type %d has no usable slot (D7/D8)", type);
    if (n % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "n %% 32 != 0 (Ggml.cs:1129)");
    int rc = ensure_init();
    if (rc) return rc;
    ggml_hip_weight *w = nullptr;
    const uint64_t xb = (uint64_t)row_bytes_of(type, n);
    rc = ggml_hip_weight_upload(type, vx, n, 1, xb, 0, 1, g_stream, &w);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t yb = TSIZE[vt] * (size_t)(n / QK);
    const size_t wb = ggml_hip_mul_mat_work_size(type, n, 1);
    rc = GGML_HIP_OK;
    if (g_src1.ensure(yb) || g_work.ensure(wb) || g_dst.ensure(4)) rc = fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed");
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(g_src1.p, vy, yb, hipMemcpyHostToDevice, g_stream);
        if (e == hipSuccess) e = launch_q8_aos_to_planes(vt, g_src1.p, 1, n, act_carve(g_work.p, n, pad_act(1)), g_stream);
        if (e == hipSuccess) rc = ggml_hip_mul_mat_compute_dev(w, 1, (float *)g_dst.p, 1, g_work.p, wb, g_stream);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(s, g_dst.p, 4, hipMemcpyDeviceToHost, g_stream);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(g_stream);
    }
    ggml_hip_weight_free(w);
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "vec_dot: %s", hipGetErrorString(e));
    return rc;
}

int ggml_hip_relayout_gathered_dev(const float *d_gathered, int G, int64_t N, int64_t Ms, float *d_dst, int64_t M,
                                   int64_t ldd, void *stream) {
    if (!d_gathered || !d_dst || G <= 0 || Ms <= 0) return fail(GGML_HIP_ERR_ARG, "bad argument");
    if (ldd < M) return fail(GGML_HIP_ERR_SHAPE, "ldd < M");
    HIP_TRY(launch_relayout_gathered(d_gathered, G, N, Ms, d_dst, M, ldd, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_quantize_rows_src_dev(int type, int src_type, const void *d_x, int64_t ld, int64_t nrows, int64_t k,
                                   void *d_blocks, void *stream) {
    if (nrows <= 0) return GGML_HIP_OK;
    if (!d_x || !d_blocks) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (!(wq_ok(type) || type == GGML_TYPE_Q8_1))
        return fail(GGML_HIP_ERR_TYPE, "quantize: unsupported type %d", type);
    if (src_type != GGML_TYPE_F32 && src_type != GGML_TYPE_F16) return fail(GGML_HIP_ERR_TYPE, "quantize: source must be F32 or F16");
    if (k % QK != 0 || ld < k) return fail(GGML_HIP_ERR_SHAPE, "k %% 32 != 0 or ld < k");
    if (src_type == GGML_TYPE_F16 && ld % 8 != 0) return fail(GGML_HIP_ERR_SHAPE, "f16 rows must be 16-byte aligned (ld %% 8)");
    if (src_type == GGML_TYPE_F32 && ld % 4 != 0) return fail(GGML_HIP_ERR_SHAPE, "f32 rows must be 16-byte aligned (ld %% 4)");
    HIP_TRY(launch_quantize_rows(type, src_type, d_x, ld, nrows, k, d_blocks, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_add_q_f32_rows_dev(int type, const void *d_blocks_in, const float *d_x, int64_t nrows, int64_t k,
                                void *d_blocks_out, void *stream) {
    if (nrows <= 0) return GGML_HIP_OK;
    if (!d_blocks_in || !d_x || !d_blocks_out) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (!wq_ok(type))
        return fail(GGML_HIP_ERR_TYPE, "add_q_f32: unsupported type %d", type);
    if (k % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "k %% 32 != 0 (Ggml.cs:4893)");
    HIP_TRY(launch_add_q_f32(type, d_blocks_in, d_x, nrows, k, d_blocks_out, (hipStream_t)stream));
    return GGML_HIP_OK;
}

/* ggml_compute_forward_cpy, quantizing branch of dup_f32 / dup_f16 (Ggml.cs:4339-4363, 3935-3966) */
int ggml_hip_compute_forward_cpy(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 struct ggml_tensor *dst) {
    if (!params || !src0 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    const int st = src0->type, dt = dst->type;
    if (st != GGML_TYPE_F32 && st != GGML_TYPE_F16) return fail(GGML_HIP_ERR_TYPE, "cpy: src0 must be F32 or F16 (Ggml.cs:4602-4619)");
    if (!wq_ok(dt))
        return fail(GGML_HIP_ERR_TYPE, "cpy: only the quantizing branch is on this path (dst type %d)", dt);
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    const int64_t n_src = ne00 * ne01 * ne02 * ne03, n_dst = dst->ne[0] * dst->ne[1] * dst->ne[2] * dst->ne[3];
    if (n_src != n_dst) return fail(GGML_HIP_ERR_SHAPE, "cpy: element counts differ (Ggml.cs:8281)");
    const size_t es = st == GGML_TYPE_F32 ? 4 : 2;
    if (src0->nb[0] != es) return fail(GGML_HIP_ERR_SHAPE, "cpy: src0 rows must be contiguous");
    if (dst->nb[0] != TSIZE[dt] || dst->nb[1] != dst->nb[0] * (uint64_t)(dst->ne[0] / BLCK[dt]) || dst->nb[2] != dst->nb[1] * (uint64_t)dst->ne[1] ||
        dst->nb[3] != dst->nb[2] * (uint64_t)dst->ne[2])
        return fail(GGML_HIP_ERR_SHAPE, "cpy: dst must be contiguous (Ggml.cs:4290)");
    if (ne00 % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "cpy: ne00 %% 32 != 0");
    if (src0->nb[1] % 16 != 0 && ne01 * ne02 * ne03 > 1) return fail(GGML_HIP_ERR_SHAPE, "cpy: src0 row stride must be a multiple of 16 bytes");
    if (!src0->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (n_src == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_graph_depth > 0) HIP_TRY(hipStreamSynchronize(g_stream));   // operands may be dst of an earlier node still on its way to the host
    const size_t row_in = (size_t)ne00 * es, rs = row_bytes_of(dt, ne00);   // rs as in Ggml.cs:4345
    if (g_src1.ensure(row_in * ne01) || g_dst.ensure(rs * ne01)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
    size_t id = 0;
    for (int64_t i03 = 0; i03 < ne03; ++i03)
        for (int64_t i02 = 0; i02 < ne02; ++i02) {
            const uint8_t *src = (const uint8_t *)src0->data + i02 * src0->nb[2] + i03 * src0->nb[3];
            HIP_TRY(hipMemcpy2DAsync(g_src1.p, row_in, src, src0->nb[1], row_in, (size_t)ne01, hipMemcpyHostToDevice, g_stream));
            rc = ggml_hip_quantize_rows_src_dev(dt, st, g_src1.p, ne00, ne01, ne00, g_dst.p, g_stream);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync((uint8_t *)dst->data + id, g_dst.p, rs * ne01, hipMemcpyDeviceToHost, g_stream));
            HIP_TRY(hipStreamSynchronize(g_stream));
            id += rs * ne01;
        }
    invalidate_locked(dst->data);   // dst is usually a future src0: its cached device copy (if any) is now stale
    return GGML_HIP_OK;
}

/* ggml_compute_forward_add_f32 / _mul_f32 (Ggml.cs:4622-4682, 5007-5035): same-shape contiguous f32 operands */
static int binary_f32_seam(int op, const struct ggml_tensor *src0, const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    const char *name = op == 0 ? "add_f32" : "mul_f32";
    if (src0->type != GGML_TYPE_F32 || src1->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "%s: F32 operands only (Ggml.cs:5043-5056)", name);
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != src1->ne[i] || src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "%s: shapes differ (Ggml.cs:4628, 5014)", name);
    if (!contiguous_f32(src0) || !contiguous_f32(src1) || !contiguous_f32(dst)) return fail(GGML_HIP_ERR_SHAPE, "%s: contiguous operands only", name);
    if (!src0->data || !src1->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (nelem(src0) == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    const float *a = nullptr, *b = nullptr;
    if (operand_f32(src0, g_src1, &a, g_stream) || operand_f32(src1, g_stage, &b, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "%s: operand staging failed", name);
    float *z = result_f32(dst, g_dst);
    if (!z) return fail(GGML_HIP_ERR_RUNTIME, "%s: hipMalloc failed", name);
    HIP_TRY(launch_binary_f32(op, a, b, z, nelem(src0), g_stream));
    if (finish_f32(dst, z, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "%s: copy back failed", name);
    return GGML_HIP_OK;
}

/* ggml_compute_forward_add_q_f32 (Ggml.cs:4797-4906) */
int ggml_hip_compute_forward_add(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    const int t = src0->type;
    if (t == GGML_TYPE_F32) return binary_f32_seam(0, src0, src1, dst);      // ggml_compute_forward_add_f32 (Ggml.cs:4622-4682)
    if (!wq_ok(t))
        return fail(GGML_HIP_ERR_TYPE, "add: src0 must be F32 or quantized (add_q_f32), got type %d", t);
    if (dst->type != t || src1->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "add_q_f32: dst must have src0's type, src1 F32 (Ggml.cs:4863-4865)");
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != src1->ne[i] || src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "add_q_f32: shapes differ (Ggml.cs:4803)");
    if (src0->nb[0] != TSIZE[t] || dst->nb[0] != TSIZE[t] || src1->nb[0] != 4) return fail(GGML_HIP_ERR_SHAPE, "add_q_f32: permuted operand (Ggml.cs:4853-4854)");
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    if (ne00 % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "ne00 %% 32 != 0 (Ggml.cs:4893)");
    if (src1->nb[1] % 16 != 0 && ne01 > 1) return fail(GGML_HIP_ERR_SHAPE, "add_q_f32: src1 row stride must be a multiple of 16 bytes");
    if (!src0->data || !src1->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (ne00 * ne01 * ne02 * ne03 == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_graph_depth > 0) HIP_TRY(hipStreamSynchronize(g_stream));   // operands may be dst of an earlier node still on its way to the host
    const size_t rs = row_bytes_of(t, ne00), rx = (size_t)ne00 * 4;
    if (g_stage.ensure(rs * ne01) || g_src1.ensure(rx * ne01) || g_dst.ensure(rs * ne01)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
    for (int64_t i03 = 0; i03 < ne03; ++i03)
        for (int64_t i02 = 0; i02 < ne02; ++i02) {
            const uint8_t *a = (const uint8_t *)src0->data + i02 * src0->nb[2] + i03 * src0->nb[3];
            const uint8_t *b = (const uint8_t *)src1->data + i02 * src1->nb[2] + i03 * src1->nb[3];
            // the reference offsets dst rows by i3*nb0 (Ggml.cs:4891), an upstream typo for nb3; intent is followed
            uint8_t *d = (uint8_t *)dst->data + i02 * dst->nb[2] + i03 * dst->nb[3];
            HIP_TRY(hipMemcpy2DAsync(g_stage.p, rs, a, src0->nb[1], rs, (size_t)ne01, hipMemcpyHostToDevice, g_stream));
            HIP_TRY(hipMemcpy2DAsync(g_src1.p, rx, b, src1->nb[1], rx, (size_t)ne01, hipMemcpyHostToDevice, g_stream));
            rc = ggml_hip_add_q_f32_rows_dev(t, g_stage.p, (const float *)g_src1.p, ne01, ne00, g_dst.p, g_stream);
            if (rc) return rc;
            HIP_TRY(hipMemcpy2DAsync(d, dst->nb[1], g_dst.p, rs, rs, (size_t)ne01, hipMemcpyDeviceToHost, g_stream));
            HIP_TRY(hipStreamSynchronize(g_stream));
        }
    invalidate_locked(dst->data);
    return GGML_HIP_OK;
}

/* ggml_compute_forward_mul (Ggml.cs:5037-5056) */
int ggml_hip_compute_forward_mul(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    return binary_f32_seam(1, src0, src1, dst);
}

/* ggml_compute_forward_scale_f32 (Ggml.cs:6746-6778): dst (a view of src0) *= *(float *)src1->data */
int ggml_hip_compute_forward_scale(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                   const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (src0->type != GGML_TYPE_F32 || src1->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "scale: F32 only (Ggml.cs:6786-6799)");
    if (nelem(src1) != 1) return fail(GGML_HIP_ERR_SHAPE, "scale: src1 must be a scalar (Ggml.cs:6755)");
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "scale: shapes differ (Ggml.cs:6754)");
    if (!contiguous_f32(src0) || !contiguous_f32(dst)) return fail(GGML_HIP_ERR_SHAPE, "scale: contiguous operands only (Ggml.cs:6752-6753)");
    if (!src0->data || !src1->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (nelem(src0) == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_graph_depth > 0) HIP_TRY(hipStreamSynchronize(g_stream));           // the scalar is read from host memory
    const float v = *(const float *)src1->data;
    // the reference scales dst's own memory: when dst is not a view of src0 that memory is whatever it held before, and
    // so it is here (the device form of "dst" is then an upload of dst->data, not of src0->data)
    const float *cur = nullptr;
    if (operand_f32(dst, g_src1, &cur, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "scale: operand staging failed");
    float *z = result_f32(dst, g_dst);
    if (!z) return fail(GGML_HIP_ERR_RUNTIME, "scale: hipMalloc failed");
    if (z != cur) HIP_TRY(hipMemcpyAsync(z, cur, (size_t)nelem(dst) * 4, hipMemcpyDeviceToDevice, g_stream));
    HIP_TRY(launch_scale_f32(z, nelem(dst), v, g_stream));
    if (finish_f32(dst, z, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "scale: copy back failed");
    return GGML_HIP_OK;
}

/* ggml_compute_forward_rms_norm_f32 (Ggml.cs:5858-5920) */
int ggml_hip_compute_forward_rms_norm(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                      struct ggml_tensor *dst) {
    if (!params || !src0 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (src0->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "rms_norm: F32 only (Ggml.cs:5927-5940)");
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "rms_norm: shapes differ (Ggml.cs:5863)");
    if (!contiguous_f32(src0) || !contiguous_f32(dst)) return fail(GGML_HIP_ERR_SHAPE, "rms_norm: contiguous operands only");
    if (!src0->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (nelem(src0) == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    const float *a = nullptr;
    if (operand_f32(src0, g_src1, &a, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "rms_norm: operand staging failed");
    float *z = result_f32(dst, g_dst);
    if (!z) return fail(GGML_HIP_ERR_RUNTIME, "rms_norm: hipMalloc failed");
    HIP_TRY(launch_rms_norm_f32(a, z, nelem(src0) / src0->ne[0], src0->ne[0], g_stream));
    if (finish_f32(dst, z, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "rms_norm: copy back failed");
    return GGML_HIP_OK;
}

/* ggml_compute_forward_silu_f32 (Ggml.cs:5705-5748) */
int ggml_hip_compute_forward_silu(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                  struct ggml_tensor *dst) {
    if (!params || !src0 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (src0->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "silu: F32 only (Ggml.cs:5755-5768)");
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "silu: shapes differ (Ggml.cs:5712)");
    if (!contiguous_f32(src0) || !contiguous_f32(dst)) return fail(GGML_HIP_ERR_SHAPE, "silu: contiguous operands only (Ggml.cs:5710-5711)");
    if (!src0->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (nelem(src0) == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    const float *a = nullptr;
    if (operand_f32(src0, g_src1, &a, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "silu: operand staging failed");
    float *z = result_f32(dst, g_dst);
    if (!z) return fail(GGML_HIP_ERR_RUNTIME, "silu: hipMalloc failed");
    HIP_TRY(launch_silu_f32(a, z, nelem(src0), g_stream));
    if (finish_f32(dst, z, g_stream)) return fail(GGML_HIP_ERR_RUNTIME, "silu: copy back failed");
    return GGML_HIP_OK;
}

void ggml_hip_invalidate(const void *host_ptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    invalidate_locked(host_ptr);
}

void ggml_hip_invalidate_all(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    free_cache_locked();
}

/* Seam 1.  Checks mirror the Debug.Asserts of the three drivers (Ggml.cs:6026-6046, 6222-6241, 6477-6504) and of
 * ggml_mul_mat_impl (Ggml.cs:8228-8229); the reference silently drops them in Release, here they are errors. */
int ggml_hip_compute_forward_mul_mat(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                     const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    // offload convention of the reference's own dead GPU blocks (Ggml.cs:6510-6521)
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    const int type = src0->type;
    if (type < 0 || type >= GGML_TYPE_COUNT || !weight_type_ok(type))
        return fail(GGML_HIP_ERR_TYPE, "src0 type %d unsupported (Q4_3/Q8_1 null slots; Q4_2/Q5_1 broken storage, SURVEY D7/D8)", type);
    if (src1->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "src1 and dst must be F32");
    const int64_t ne00 = src0->ne[0], ne01 = src0->ne[1], ne02 = src0->ne[2], ne03 = src0->ne[3];
    const int64_t ne10 = src1->ne[0], ne11 = src1->ne[1], ne12 = src1->ne[2], ne13 = src1->ne[3];
    if (ne00 != ne10 || ne02 != ne12 || ne03 != ne13) return fail(GGML_HIP_ERR_SHAPE, "!ggml_can_mul_mat (Ggml.cs:8345-8353)");
    if (dst->ne[0] != ne01 || dst->ne[1] != ne11 || dst->ne[2] != ne02 || dst->ne[3] != ne03)
        return fail(GGML_HIP_ERR_SHAPE, "dst shape (Ggml.cs:6488-6491)");
    if (src0->nb[0] != TSIZE[type]) return fail(GGML_HIP_ERR_SHAPE, "permuted src0 (Ggml.cs:6477)");
    if (src0->nb[0] > src0->nb[1]) return fail(GGML_HIP_ERR_SHAPE, "transposed src0 (Ggml.cs:8229)");
    if (src1->nb[0] != 4) return fail(GGML_HIP_ERR_SHAPE, "permuted src1 (Ggml.cs:6478)");
    if (dst->nb[0] != 4 || dst->nb[0] > dst->nb[1] || dst->nb[1] > dst->nb[2] || dst->nb[2] > dst->nb[3])
        return fail(GGML_HIP_ERR_SHAPE, "dst transposed or permuted (Ggml.cs:6481-6484)");
    if (ne00 % BLCK[type] != 0) return fail(GGML_HIP_ERR_SHAPE, "ne00 %% 32 != 0 (Ggml.cs:6694)");
    if (src1->nb[1] % 4 != 0 || dst->nb[1] % 4 != 0) return fail(GGML_HIP_ERR_SHAPE, "row strides must be multiples of 4");
    if (!src0->data || !src1->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (ne01 == 0 || ne11 == 0 || ne02 * ne03 == 0) return GGML_HIP_OK;
    int rc = ensure_init();
    if (rc) return rc;

    std::lock_guard<std::mutex> lk(g_mu);
    const CacheKey key{src0->data, type, ne00, ne01, ne02, ne03, src0->nb[1], src0->nb[2], src0->nb[3]};
    auto it = g_cache.find(key);
    if (it == g_cache.end()) {
        std::vector<ggml_hip_weight *> slices;
        for (int64_t i03 = 0; i03 < ne03; ++i03)
            for (int64_t i02 = 0; i02 < ne02; ++i02) {
                ggml_hip_weight *w = nullptr;
                const uint8_t *base = (const uint8_t *)src0->data + i02 * src0->nb[2] + i03 * src0->nb[3];
                rc = ggml_hip_weight_upload(type, base, ne00, ne01, src0->nb[1], 0, ne01, g_stream, &w);
                if (rc) {
                    for (ggml_hip_weight *x : slices) ggml_hip_weight_free(x);
                    return rc;
                }
                slices.push_back(w);
            }
        it = g_cache.emplace(key, std::move(slices)).first;
    }
    const size_t x_bytes = (size_t)ne11 * ne10 * 4, d_bytes = (size_t)ne11 * ne01 * 4;
    const size_t w_bytes = ggml_hip_mul_mat_work_size(type, ne00, ne11);
    const int64_t nslice = ne02 * ne03;
    // graph scope: is src1 the (contiguous) dst of an earlier offloaded node?  will dst be kept?
    const bool src1_contig = src1->nb[1] == (uint64_t)ne10 * 4 && src1->nb[2] == src1->nb[1] * (uint64_t)ne11 &&
                             src1->nb[3] == src1->nb[2] * (uint64_t)ne12;
    const bool dst_contig = dst->nb[1] == (uint64_t)ne01 * 4 && dst->nb[2] == dst->nb[1] * (uint64_t)ne11 &&
                            dst->nb[3] == dst->nb[2] * (uint64_t)ne02;
    const uint8_t *x_res = nullptr;
    if (g_graph_depth > 0 && src1_contig) {
        auto r = g_resident.find(src1->data);
        if (r != g_resident.end() && r->second.bytes >= x_bytes * (size_t)nslice) { x_res = (const uint8_t *)r->second.p; ++g_resident_hits; }
    }
    uint8_t *d_res = nullptr;
    if (g_graph_depth > 0 && dst_contig) {
        auto old = g_resident.find(dst->data);           // the same tensor computed again: reuse its buffer
        if (old != g_resident.end() && old->second.bytes >= d_bytes * (size_t)nslice) {
            d_res = (uint8_t *)old->second.p;
        } else {
            if (old != g_resident.end()) { g_pool.push_back(old->second); g_resident.erase(old); }
            d_res = (uint8_t *)pool_take(d_bytes * (size_t)nslice);
            if (!d_res) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for a resident dst");
            g_resident[dst->data] = g_resident[nullptr];
            g_resident.erase(nullptr);
        }
    }
    if ((!x_res && g_src1.ensure(x_bytes)) || (!d_res && g_dst.ensure(d_bytes)) || g_work.ensure(w_bytes ? w_bytes : 16))
        return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
    for (int64_t i03 = 0; i03 < ne03; ++i03)
        for (int64_t i02 = 0; i02 < ne02; ++i02) {  // slice offsets as in Ggml.cs:6566-6570
            const int64_t sl = i03 * ne02 + i02;
            const ggml_hip_weight *w = it->second[(size_t)sl];
            const uint8_t *x = (const uint8_t *)src1->data + i02 * src1->nb[2] + i03 * src1->nb[3];
            uint8_t *d = (uint8_t *)dst->data + i02 * dst->nb[2] + i03 * dst->nb[3];
            const float *xd = x_res ? (const float *)(x_res + (size_t)sl * x_bytes) : (const float *)g_src1.p;
            float *dd = d_res ? (float *)(d_res + (size_t)sl * d_bytes) : (float *)g_dst.p;
            if (!x_res) {
                // src1 comes from host memory: inside a graph scope an earlier node's device -> host copy into that
                // very memory may still be in flight, and a pageable source is read when the copy is enqueued
                if (g_graph_depth > 0) HIP_TRY(hipStreamSynchronize(g_stream));
                HIP_TRY(hipMemcpy2DAsync(g_src1.p, (size_t)ne10 * 4, x, src1->nb[1], (size_t)ne10 * 4, (size_t)ne11,
                                         hipMemcpyHostToDevice, g_stream));
                g_h2d_bytes += x_bytes;
            }
            rc = ggml_hip_mul_mat_dev(w, xd, ne11, ne10, dd, ne01, g_work.p, g_work.cap, g_stream);
            if (rc) return rc;
            HIP_TRY(hipMemcpy2DAsync(d, dst->nb[1], dd, (size_t)ne01 * 4, (size_t)ne01 * 4, (size_t)ne11,
                                     hipMemcpyDeviceToHost, g_stream));
            g_d2h_bytes += d_bytes;
            // outside a graph scope the call returns with dst on the host; inside, ggml_hip_graph_end waits once
            // (the scratch buffers are reused in stream order)
            if (g_graph_depth == 0) HIP_TRY(hipStreamSynchronize(g_stream));
        }
    return GGML_HIP_OK;
}

/* Graph scope for ggml_graph_compute's node loop (Ggml.cs:3539-3704): see "graph-level residency" above. */
int ggml_hip_graph_begin(void) {
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    ++g_graph_depth;
    return GGML_HIP_OK;
}
int ggml_hip_graph_end(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_graph_depth <= 0) return fail(GGML_HIP_ERR_ARG, "ggml_hip_graph_end without ggml_hip_graph_begin");
    hipError_t e = g_stream ? hipStreamSynchronize(g_stream) : hipSuccess;   // every node's dst is on the host from here on
    if (--g_graph_depth == 0) pool_drain_locked(false);
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "graph_end: %s", hipGetErrorString(e));
    return GGML_HIP_OK;
}
void ggml_hip_debug_transfer_counters(uint64_t *h2d_bytes, uint64_t *d2h_bytes, uint64_t *resident_hits) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (h2d_bytes) *h2d_bytes = g_h2d_bytes;
    if (d2h_bytes) *d2h_bytes = g_d2h_bytes;
    if (resident_hits) *resident_hits = g_resident_hits;
}

}  // extern "C"
