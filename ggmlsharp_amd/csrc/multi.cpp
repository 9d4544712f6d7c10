// multi.cpp -- the C-ABI of include/ggml_hip.h, part 3: one process, several devices (SURVEY 8(b) "n_devices", 8(e)).
//
// The weight matrix is row-split over the device slots with the reference's own partition (Ggml.cs:6665-6672: dr =
// ceil(M / G), part g owns rows [dr*g, min(dr*(g+1), M))).  Every slot runs INIT + COMPUTE for its rows on its own stream
// and writes its columns straight into its copy of the reference-layout dst [N][M] (the kernels take a row stride, so no
// [G][N][Ms] intermediate exists); the exchange then completes every slot's copy.  Two exchange forms:
//   0 (default) peer DMA: each slot pushes its column range into every peer's dst with one strided 2-D copy per peer --
//       xGMI is point-to-point, so the G-1 pushes of a slot run on different links; no re-layout pass, no staging buffer;
//   1 RCCL: ncclAllGather (librccl, loaded at run time) of contiguous [N][Ms] shards + the re-layout kernel;
//   2 (r4) no exchange pass: every slot's GEMM stores its rows into EVERY slot's dst from its own store phase (mm_epilogue mode 3,
//       ggml_hip_mul_mat_push_dev) -- ggml_hip_mul_mat_split_dev; the graph-scope row split of seams.cpp keeps form 0 behind its products.
// Same bits whichever runs: the exchange only moves data.
#include "ctx.h"

#include <dlfcn.h>
#include <memory>

namespace ghip {

namespace {

std::atomic<int> g_exchange_mode{0};

// ---- RCCL through dlopen: the product library does not link librccl; only this exchange form needs it ----
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::vector<ncclComm_t> comms;
    std::vector<int> devices;
    bool load() {
        if (h) return true;
        h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
    }
    void destroy() {
        for (ncclComm_t c : comms) if (c) CommDestroy(c);
        comms.clear(); devices.clear();
    }
    // one communicator per slot, created once per device set
    int ensure(int G, DeviceCtx *const *ctxs) {
        if (!load()) return fail(GGML_HIP_ERR_RUNTIME, "librccl.so not found: %s", dlerror());
        std::vector<int> devs((size_t)G);
        for (int g = 0; g < G; ++g) devs[(size_t)g] = ctxs[g]->device;
        if (devs == devices) return GGML_HIP_OK;
        for (int a = 0; a < G; ++a)
            for (int b = a + 1; b < G; ++b)
                if (devs[(size_t)a] == devs[(size_t)b]) return fail(GGML_HIP_ERR_ARG, "RCCL exchange needs distinct devices (slots %d and %d share device %d)", a, b, devs[(size_t)a]);
        destroy();
        comms.assign((size_t)G, nullptr);
        const int r = CommInitAll(comms.data(), G, devs.data());
        if (r != 0) { comms.clear(); return fail(GGML_HIP_ERR_RUNTIME, "ncclCommInitAll: %s", GetErrorString ? GetErrorString(r) : "error"); }
        devices = devs;
        return GGML_HIP_OK;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int exchange_peer(int G, DeviceCtx *const *ctxs, float *const *bufs, int64_t N, int64_t ldd, const int64_t *r0, const int64_t *r1) {
    // 1. every slot: "everything issued so far on my stream is done" -- its own columns are written, and whatever read the
    //    (possibly recycled) buffer before is finished, so peers may write into it
    for (int g = 0; g < G; ++g) {
        int rc = ctxs[g]->make_current();
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ctxs[g]->ev_ready, ctxs[g]->stream));
    }
    // 2. each slot pushes its column range into every peer's buffer: one strided 2-D copy per peer, on the slot's own stream
    for (int g = 0; g < G; ++g) {
        const int64_t Ms = r1[g] - r0[g];
        int rc = ctxs[g]->make_current();
        if (rc) return rc;
        for (int k = 1; k < G && Ms > 0; ++k) {
            const int p = (g + k) % G;                       // staggered start: slot g's first push goes to g+1, not to slot 0
            HIP_TRY(hipStreamWaitEvent(ctxs[g]->stream, ctxs[p]->ev_ready, 0));
            HIP_TRY(hipMemcpy2DAsync(bufs[p] + r0[g], (size_t)ldd * 4, bufs[g] + r0[g], (size_t)ldd * 4, (size_t)Ms * 4, (size_t)N,
                                     hipMemcpyDefault, ctxs[g]->stream));
        }
        HIP_TRY(hipEventRecord(ctxs[g]->ev_xchg, ctxs[g]->stream));
    }
    // 3. every slot's stream continues only when all pushes into its buffer have landed
    for (int p = 0; p < G; ++p) {
        int rc = ctxs[p]->make_current();
        if (rc) return rc;
        for (int g = 0; g < G; ++g)
            if (g != p) HIP_TRY(hipStreamWaitEvent(ctxs[p]->stream, ctxs[g]->ev_xchg, 0));
    }
    return GGML_HIP_OK;
}

int exchange_rccl(int G, DeviceCtx *const *ctxs, float *const *bufs, int64_t N, int64_t ldd, const int64_t *r0, const int64_t *r1) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    int rc = g_rccl.ensure(G, ctxs);
    if (rc) return rc;
    const int64_t Ms = r1[0] - r0[0];                        // the partition's shard width dr (the last shard may be shorter)
    const size_t shard_elems = (size_t)N * (size_t)Ms;
    std::vector<float *> send((size_t)G), recv((size_t)G);
    for (int g = 0; g < G; ++g) {
        DeviceCtx *c = ctxs[g];
        rc = c->make_current();
        if (rc) return rc;
        if (c->stage.ensure(shard_elems * 4 * (size_t)(G + 1))) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for the all-gather buffers");
        send[(size_t)g] = (float *)c->stage.p;
        recv[(size_t)g] = (float *)c->stage.p + shard_elems;
        // own columns -> contiguous [N][Ms] shard (zero-padded when this shard is shorter than dr)
        const int64_t mine = r1[g] - r0[g];
        if (mine < Ms) HIP_TRY(hipMemsetAsync(send[(size_t)g], 0, shard_elems * 4, c->stream));
        if (mine > 0)
            HIP_TRY(hipMemcpy2DAsync(send[(size_t)g], (size_t)Ms * 4, bufs[g] + r0[g], (size_t)ldd * 4, (size_t)mine * 4, (size_t)N,
                                     hipMemcpyDeviceToDevice, c->stream));
    }
    int r = g_rccl.GroupStart();
    for (int g = 0; g < G && r == 0; ++g) {
        (void)hipSetDevice(ctxs[g]->device);
        r = g_rccl.AllGather(send[(size_t)g], recv[(size_t)g], shard_elems, /* ncclFloat32 */ 7, g_rccl.comms[(size_t)g], ctxs[g]->stream);
    }
    const int r2 = g_rccl.GroupEnd();
    if (r != 0 || r2 != 0) return fail(GGML_HIP_ERR_RUNTIME, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r ? r : r2) : "error");
    int64_t M = 0;
    for (int g = 0; g < G; ++g) M = r1[g] > M ? r1[g] : M;
    for (int g = 0; g < G; ++g) {                            // [G][N][Ms] -> the reference layout [N][M] (SURVEY 8(e) layout catch)
        rc = ctxs[g]->make_current();
        if (rc) return rc;
        HIP_TRY(launch_relayout_gathered(recv[(size_t)g], G, N, Ms, bufs[g], M, ldd, ctxs[g]->stream));
    }
    return GGML_HIP_OK;
}

}  // namespace

// can a kernel on any slot's device store into any other slot's memory?  (same device, or peer access that is or can be enabled)
bool slots_reach_each_other(int G, DeviceCtx *const *ctxs) {
    for (int a = 0; a < G; ++a)
        for (int b = 0; b < G; ++b) {
            if (ctxs[a]->device == ctxs[b]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, ctxs[a]->device, ctxs[b]->device) != hipSuccess || !can) { (void)hipGetLastError(); return false; }
            if (hipSetDevice(ctxs[a]->device) != hipSuccess) { (void)hipGetLastError(); return false; }
            const hipError_t e = hipDeviceEnablePeerAccess(ctxs[b]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return false; }
            (void)hipGetLastError();
        }
    return true;
}

int exchange_columns(int G, DeviceCtx *const *ctxs, float *const *bufs, int64_t N, int64_t ldd, const int64_t *r0, const int64_t *r1) {
    if (G <= 1 || N <= 0) return GGML_HIP_OK;
    if (g_exchange_mode.load() == 1) return exchange_rccl(G, ctxs, bufs, N, ldd, r0, r1);
    return exchange_peer(G, ctxs, bufs, N, ldd, r0, r1);
}

void rccl_shutdown() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) g_rccl.destroy();
}

}  // namespace ghip

using namespace ghip;

// row shards of one 2-D weight matrix, one per device slot
struct ggml_hip_split_weight {
    int type = 0;
    int64_t M = 0, K = 0;
    int G = 0;
    std::vector<ggml_hip_weight *> shard;
    std::vector<int64_t> r0, r1;
    // can every slot's device store into every other slot's memory (exchange mode 2)?  -1: not asked yet.  Asked ONCE per slot set, at the
    // first mode-2 product (the slots of a split weight never change: G == n_slots() is checked per call) -- it is O(G^2) runtime calls
    // (hipDeviceCanAccessPeer + hipSetDevice + hipDeviceEnablePeerAccess, ~170 at G = 8) that have no place on the hot path.
    mutable std::atomic<int> reach{-1};
};

extern "C" {

int ggml_hip_set_exchange(int mode) {
    if (mode < 0 || mode > 2) return fail(GGML_HIP_ERR_ARG, "exchange mode %d (0 = peer DMA, 1 = RCCL all-gather, 2 = the GEMM's store phase)", mode);
    g_exchange_mode.store(mode);
    return GGML_HIP_OK;
}

/* Exercises the RCCL exchange form on ONE slot (librccl loaded at run time, ncclCommInitAll over slot 0's device, an
 * all-gather of one rank, the re-layout kernel) and checks the bytes: what a one-GPU box can verify of that form. */
int ggml_hip_debug_rccl_selftest(void) {
    int rc = ensure_init();
    if (rc) return rc;
    DeviceCtx *c = slot(0);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    rc = c->make_current();
    if (rc) return rc;
    const int64_t N = 5, M = 96, ldd = 128;
    std::vector<float> h((size_t)N * ldd), back((size_t)N * ldd, -1.0f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i * 0.25f - 3.0f;
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, h.size() * 4));
    hipError_t e = hipMemcpyAsync(d, h.data(), h.size() * 4, hipMemcpyHostToDevice, c->stream);
    const int64_t r0 = 0, r1 = M;
    DeviceCtx *ctxs[1] = {c};
    float *bufs[1] = {d};
    if (e == hipSuccess) rc = exchange_rccl(1, ctxs, bufs, N, ldd, &r0, &r1);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(back.data(), d, h.size() * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "rccl selftest: %s", hipGetErrorString(e));
    if (rc) return rc;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t m = 0; m < M; ++m)
            if (back[(size_t)(n * ldd + m)] != h[(size_t)(n * ldd + m)]) return fail(GGML_HIP_ERR_RUNTIME, "rccl selftest: element (%lld, %lld) differs", (long long)n, (long long)m);
    return GGML_HIP_OK;
}

int ggml_hip_split_weight_upload(int type, const void *host_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                                 ggml_hip_split_weight **out) {
    if (!out) return fail(GGML_HIP_ERR_ARG, "out is null");
    *out = nullptr;
    int rc = ensure_init();
    if (rc) return rc;
    const int G = n_slots();
    std::unique_ptr<ggml_hip_split_weight> sw(new ggml_hip_split_weight());
    sw->type = type; sw->M = ne01; sw->K = ne00; sw->G = G;
    sw->shard.assign((size_t)G, nullptr); sw->r0.assign((size_t)G, 0); sw->r1.assign((size_t)G, 0);
    for (int g = 0; g < G && !rc; ++g) {
        DeviceCtx *c = slot(g);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        shard_rows(ne01, G, g, &sw->r0[(size_t)g], &sw->r1[(size_t)g]);
        rc = make_weight(c, type, host_rows, true, ne00, ne01, nb01, sw->r0[(size_t)g], sw->r1[(size_t)g], c->stream, &sw->shard[(size_t)g]);
    }
    if (rc) {
        for (ggml_hip_weight *w : sw->shard) ggml_hip_weight_free(w);
        return rc;
    }
    *out = sw.release();
    return GGML_HIP_OK;
}

void ggml_hip_split_weight_free(ggml_hip_split_weight *w) {
    if (!w) return;
    for (ggml_hip_weight *s : w->shard) ggml_hip_weight_free(s);
    delete w;
}

int ggml_hip_split_weight_rows(const ggml_hip_split_weight *w, int slot_index, int64_t *row_begin, int64_t *row_end) {
    if (!w || slot_index < 0 || slot_index >= w->G) return fail(GGML_HIP_ERR_ARG, "bad argument");
    if (row_begin) *row_begin = w->r0[(size_t)slot_index];
    if (row_end) *row_end = w->r1[(size_t)slot_index];
    return GGML_HIP_OK;
}

int ggml_hip_mul_mat_split_dev(const ggml_hip_split_weight *w, const float *const *d_src1, int64_t N, int64_t ld1,
                               float *const *d_dst, int64_t ldd) {
    if (!w || !d_src1 || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (w->G != n_slots()) return fail(GGML_HIP_ERR_ARG, "the weight was split over %d slots, the library now has %d", w->G, n_slots());
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (ld1 < w->K || ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K or ldd < M");
    const int G = w->G;
    std::vector<DeviceCtx *> ctxs((size_t)G);
    std::vector<std::unique_lock<std::recursive_mutex>> locks;
    for (int g = 0; g < G; ++g) {
        ctxs[(size_t)g] = slot(g);
        if (!d_src1[g] || !d_dst[g]) return fail(GGML_HIP_ERR_ARG, "null buffer for slot %d", g);
        locks.emplace_back(ctxs[(size_t)g]->mu);
    }
    const size_t wb = ggml_hip_mul_mat_work_size(w->type, w->K, N);
    // r4, exchange mode 2: every slot's GEMM stores its rows, as they leave the accumulators, into EVERY slot's dst (mm_epilogue mode 3
    // through ggml_hip_mul_mat_push_dev: up to eight destinations; kernel forms without the store phase push their columns with the
    // column kernel behind the product) -- no exchange pass at all.  Needs every slot's device to reach every other's memory.
    bool reach = false;
    if (g_exchange_mode.load() == 2 && G > 1 && G <= 16) {
        int r = w->reach.load(std::memory_order_acquire);
        if (r < 0) {
            r = slots_reach_each_other(G, ctxs.data()) ? 1 : 0;       // (leaves another device current: every use below makes its own current)
            w->reach.store(r, std::memory_order_release);
        }
        reach = r == 1;
    }
    if (reach) {
        // An error on one slot must not leave the others' streams unordered against stores already in flight: the first error is
        // kept, every ev_xchg is still recorded and every wait still issued, and only then is the error returned.
        int first = GGML_HIP_OK;
        auto keep = [&](int rc) { if (rc && !first) first = rc; return rc; };
        auto keep_hip = [&](hipError_t e, const char *what) {
            if (e != hipSuccess) { (void)hipGetLastError(); if (!first) first = fail(GGML_HIP_ERR_RUNTIME, "%s: %s", what, hipGetErrorString(e)); }
        };
        for (int g = 0; g < G; ++g) {                        // "everything issued so far on my stream is done": peers may write into my dst
            if (keep(ctxs[(size_t)g]->make_current())) continue;
            keep_hip(hipEventRecord(ctxs[(size_t)g]->ev_ready, ctxs[(size_t)g]->stream), "hipEventRecord(ready)");
        }
        for (int g = 0; g < G; ++g) {
            DeviceCtx *c = ctxs[(size_t)g];
            if (keep(c->make_current())) continue;
            if (!first && w->r1[(size_t)g] > w->r0[(size_t)g]) {
                if (c->work.ensure(wb ? wb : 16)) keep(fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch"));
                else {
                    for (int p = 0; p < G; ++p)
                        if (p != g) keep_hip(hipStreamWaitEvent(c->stream, ctxs[(size_t)p]->ev_ready, 0), "hipStreamWaitEvent(ready)");
                    if (!first)
                        keep(ggml_hip_mul_mat_push_dev(w->shard[(size_t)g], d_src1[g], N, ld1, d_dst, G, g, ldd, w->r0[(size_t)g], c->work.p, c->work.cap, c->stream));
                }
            }
            keep_hip(hipEventRecord(c->ev_xchg, c->stream), "hipEventRecord(xchg)");
        }
        for (int p = 0; p < G; ++p) {                        // a slot's stream continues only when every slot's stores into its dst have landed
            if (keep(ctxs[(size_t)p]->make_current())) continue;
            for (int g = 0; g < G; ++g)
                if (g != p) keep_hip(hipStreamWaitEvent(ctxs[(size_t)p]->stream, ctxs[(size_t)g]->ev_xchg, 0), "hipStreamWaitEvent(xchg)");
        }
        return first;
    }
    for (int g = 0; g < G; ++g) {
        DeviceCtx *c = ctxs[(size_t)g];
        int rc = c->make_current();
        if (rc) return rc;
        if (w->r1[(size_t)g] <= w->r0[(size_t)g]) continue;
        if (c->work.ensure(wb ? wb : 16)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
        // this slot's rows of W against all of src1, written as columns [r0, r1) of its [N][ldd] dst
        rc = ggml_hip_mul_mat_dev(w->shard[(size_t)g], d_src1[g], N, ld1, d_dst[g] + w->r0[(size_t)g], ldd, c->work.p, c->work.cap, c->stream);
        if (rc) return rc;
    }
    return exchange_columns(G, ctxs.data(), const_cast<float *const *>(d_dst), N, ldd, w->r0.data(), w->r1.data());
}

/* ---- one process PER device: direct exchange through IPC-shared dst buffers (ggmlsharp_amd/dist.py, exchange "push") ---- */
int ggml_hip_ipc_alloc(size_t bytes, void **d_ptr, uint8_t *handle64) {
    if (!d_ptr || !handle64) return fail(GGML_HIP_ERR_ARG, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI carries an IPC handle as 64 opaque bytes");
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); (void)hipGetLastError(); return fail(GGML_HIP_ERR_RUNTIME, "hipIpcGetMemHandle: %s", hipGetErrorString(e)); }
    memcpy(handle64, &h, 64);
    *d_ptr = p;
    return GGML_HIP_OK;
}
int ggml_hip_ipc_open(const uint8_t *handle64, void **d_ptr) {
    if (!d_ptr || !handle64) return fail(GGML_HIP_ERR_ARG, "null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    hipError_t e = hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(GGML_HIP_ERR_RUNTIME, "hipIpcOpenMemHandle: %s", hipGetErrorString(e)); }
    return GGML_HIP_OK;
}
int ggml_hip_ipc_close(void *d_ptr) {
    if (!d_ptr) return GGML_HIP_OK;
    HIP_TRY(hipIpcCloseMemHandle(d_ptr));
    return GGML_HIP_OK;
}
int ggml_hip_ipc_free(void *d_ptr) {
    if (!d_ptr) return GGML_HIP_OK;
    HIP_TRY(hipFree(d_ptr));
    return GGML_HIP_OK;
}
int ggml_hip_push_columns_dev(const float *d_shard, int64_t lds, int64_t N, int64_t Ms, float *const *d_peers, int n_peers,
                              int64_t ldd, int64_t col0, void *stream) {
    if (!d_shard || !d_peers || n_peers <= 0 || n_peers > 16) return fail(GGML_HIP_ERR_ARG, "bad argument (1..16 peers)");
    if (lds < Ms || ldd < col0 + Ms) return fail(GGML_HIP_ERR_SHAPE, "lds < Ms or ldd < col0 + Ms");
    HIP_TRY(launch_push_columns(d_shard, lds, N, Ms, d_peers, n_peers, ldd, col0, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_sync_slots(void) {
    int rc = GGML_HIP_OK;
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        int r = c->make_current();
        if (!r) r = c->sync_all();
        if (r && !rc) rc = r;
    }
    return rc;
}

void *ggml_hip_slot_malloc(int s, size_t bytes) {
    DeviceCtx *c = slot(s);
    if (!c || c->make_current()) return nullptr;
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void ggml_hip_slot_free(int s, void *p) {
    DeviceCtx *c = slot(s);
    if (!c || !p || c->make_current()) return;
    (void)hipFree(p);
}
int ggml_hip_slot_upload(int s, void *d_dst, const void *host_src, size_t bytes) {
    DeviceCtx *c = slot(s);
    if (!c || !d_dst || !host_src) return fail(GGML_HIP_ERR_ARG, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    int rc = c->make_current();
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GGML_HIP_OK;
}
int ggml_hip_slot_download(int s, void *host_dst, const void *d_src, size_t bytes) {
    DeviceCtx *c = slot(s);
    if (!c || !host_dst || !d_src) return fail(GGML_HIP_ERR_ARG, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    int rc = c->make_current();
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GGML_HIP_OK;
}

}  // extern "C"
