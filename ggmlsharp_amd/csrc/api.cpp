// api.cpp -- the C-ABI of include/ggml_hip.h, part 1: lifecycle and device slots, resident weights, the hot path on
// device-resident data, the row functions (Seam 2).  seams.cpp holds the host-pointer seams, multi.cpp the row split
// over several devices.  No CPU fallback anywhere: without a device every compute entry returns GGML_HIP_ERR_NO_DEVICE.
#include "ctx.h"
#include "plan.h"
#include <algorithm>
#include <vector>

namespace ghip {

thread_local std::string t_err;
thread_local int t_slot = -1;                     // ggml_hip_bind_thread
std::mutex g_table_mu;
DeviceCtx *g_slots[MAX_SLOTS] = {};
std::atomic<int> g_nslots{0};

// An error reported while the calling thread has a NAMED graph scope open: the host falls back to its own code for
// that node, so the scope must never be captured -- a replay would skip the seams, the fallback with them, and
// the node's data would never be produced.  The key is refused for good; a capture in progress is ended and issued live.
static void poison_open_scope() {
    const int b = bound_slot();
    DeviceCtx *c = slot(b >= 0 ? b : 0);
    if (!c) return;
    std::unique_lock<std::recursive_mutex> lk(c->mu, std::try_to_lock);
    if (!lk.owns_lock()) return;
    if ((c->scope_mode != 1 && c->scope_mode != 2) || c->scope_owner != std::this_thread::get_id()) return;
    const uint64_t key = c->scope_key;
    if (c->scope_mode == 2) c->scope_dirty();
    else c->scope_clean = false;
    DeviceCtx::Captured &e = c->captured[key];
    e.seen = 0;
    if (!e.refused) { e.refused = true; ++c->n_refused; }
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_err = buf;
    if (code != GGML_HIP_OK) poison_open_scope();
    return code;
}

const int BLCK[GGML_TYPE_COUNT] = {1, 1, 32, 32, 16, 16, 32, 32, 32, 32, 1, 1, 1};          // Ggml.cs:55-70
const size_t TSIZE[GGML_TYPE_COUNT] = {4, 2, 20, 24, 10, 12, 22, 24, 36, 44, 1, 2, 4};      // Ggml.cs:72-87

// Q4_2 and Q5_1 follow SURVEY D7's intent (their half scales are IEEE bit patterns, as in the upstream scalar code the C#
// transcribes); Q4_3 / Q8_1 have null slots (D8).
bool wq_ok(int t) {
    return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q4_2 || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 ||
           t == GGML_TYPE_Q8_0;
}
bool weight_type_ok(int t) { return wq_ok(t) || t == GGML_TYPE_F32 || t == GGML_TYPE_F16; }
bool is_q(int t) { return t >= GGML_TYPE_Q4_0 && t <= GGML_TYPE_Q8_1; }
int vec_dot_type(int t) {  // Ggml.cs:219-290
    switch (t) {
    case GGML_TYPE_Q4_0: case GGML_TYPE_Q4_2: case GGML_TYPE_Q5_0: case GGML_TYPE_Q8_0: return GGML_TYPE_Q8_0;
    case GGML_TYPE_Q4_1: case GGML_TYPE_Q5_1: return GGML_TYPE_Q8_1;
    default: return -1;
    }
}
bool contiguous_f32(const ggml_tensor *t) {
    return t->type == GGML_TYPE_F32 && t->nb[0] == 4 && t->nb[1] == (uint64_t)t->ne[0] * 4 && t->nb[2] == t->nb[1] * (uint64_t)t->ne[1] &&
           t->nb[3] == t->nb[2] * (uint64_t)t->ne[2];
}

namespace {
bool has_min_plane(int t) { return t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_Q4_2; }   // Q4_2: its second scale
bool has_qh_plane(int t) { return t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1; }
}  // namespace

// Kernel selection lives in plan.cpp (plan.h): ONE decision per product -- family, form, summation tree -- that every launcher consumes.
// The K1 image for (type, K, N): NOT a function of the number of weight rows (a row shard runs the kernel form of the unsplit matrix).
int act_image_kind(int type, int64_t K, int64_t N) { return plan_image_kind(type, K, N); }
mm_plan weight_plan(const ggml_hip_weight *w, int64_t N, bool one_call) { return plan_mul_mat(w->type, w->ext_type, w->M, w->K, N, one_call); }
// ... for ONE weight, as the two-phase entries see it (INIT writes it, COMPUTE reads it): the stated exception included -- planes beyond
// 32-bit buffer offsets are served by the int8 family and its image
int weight_image_kind(const ggml_hip_weight *w, int64_t N) {
    const mm_plan pl = weight_plan(w, N, false);
    return (pl.image >= 0 && pl.image <= 3 ? pl.image : 0) | ((pl.flags & MM_FLAG_MIN_PIECES) ? ACT_IMAGE_MIN_PIECES : 0);
}

// ---------------- DeviceCtx ----------------
int DeviceCtx::make_current() const {
    hipError_t e = hipSetDevice(device);   // calls arrive on arbitrary host threads (SURVEY 8(b) "Threading")
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    return GGML_HIP_OK;
}
uint64_t DeviceCtx::scratch_sig() const {
    uint64_t h = 1469598103934665603ull;
    const void *ps[] = {src1.p, dst.p, dst2.p, work.p, stage.p, aux[0].p, aux[1].p, aux[2].p, aux[3].p};
    for (const void *q : ps) { h ^= (uint64_t)(uintptr_t)q; h *= 1099511628211ull; }
    return h;
}
void DeviceCtx::scope_dirty() {
    if (scope_mode == 1) { scope_clean = false; return; }
    if (scope_mode != 2) return;
    // capturing, and the next thing cannot be captured: end the capture, run what it holds, go on live
    scope_mode = 0;
    {   // (a weight re-upload after an invalidation lands here once; a scope that keeps doing it is left alone)
        Captured &e = captured[scope_key];
        e.seen = 0;
        if (++e.strikes >= 3) { e.refused = true; ++n_refused; }
    }
    // (the capture is begun in relaxed mode, so the thread that ends it need not be the one that began it: a foreign thread's seam
    // on this slot lands here too, seams.cpp Call::begin)
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(stream, &g);
    hipGraphExec_t ex = nullptr;
    if (e == hipSuccess && !g) e = hipErrorStreamCaptureInvalidated;
    if (e == hipSuccess) e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e == hipSuccess) e = hipGraphLaunch(ex, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (ex) (void)hipGraphExecDestroy(ex);
    if (g) (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        // what the scope had issued so far did not run and cannot be issued again from here: its owner is told at the scope's end
        // (ggml_hip_graph_end returns the error), the key is never captured again
        (void)hipGetLastError();
        scope_lost = true;
        Captured &c = captured[scope_key];
        if (!c.refused) { c.refused = true; ++n_refused; }
        fail(GGML_HIP_ERR_RUNTIME, "a graph scope's capture could not be ended and issued live (%s): its nodes so far did not run", hipGetErrorString(e));
    }
}
void DeviceCtx::drop_captured() {
    for (auto &kv : captured) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
        for (Resident &r : kv.second.buffers) pool.push_back(r);
    }
    captured.clear();
}
int DeviceCtx::sync_all() {
    scope_dirty();
    d2h_busy = false;
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s_h2d);
    if (e == hipSuccess) e = hipStreamSynchronize(s_d2h);
    return e == hipSuccess ? GGML_HIP_OK : fail(GGML_HIP_ERR_RUNTIME, "stream synchronize: %s", hipGetErrorString(e));
}
void *DeviceCtx::take(size_t n) {
    // best fit: a graph that is computed again asks for the same sizes again and must find every one of them (a first fit
    // hands a larger buffer to a smaller request and sends the larger request to hipMalloc -- four computes of a decoder layer
    // until the pool had settled, and an allocation keeps a named scope from being captured)
    size_t best = pool.size();
    for (size_t i = 0; i < pool.size(); ++i)
        if (pool[i].bytes >= n && pool[i].bytes <= 2 * n + 4096 && (best == pool.size() || pool[i].bytes < pool[best].bytes)) best = i;
    if (best != pool.size()) {
        void *p = pool[best].p;
        pool.erase(pool.begin() + (long)best);
        return p;
    }
    scope_dirty();                                 // (an allocation is not something a capture may contain)
    void *p = nullptr;
    if (hipMalloc(&p, n) != hipSuccess) return nullptr;
    return p;
}
const void *DeviceCtx::resident_lookup(const void *host, size_t bytes) {
    auto r = resident.find(host);
    if (r != resident.end() && r->second.bytes >= bytes) return r->second.p;
    return nullptr;
}
// a write to host range [host, host + bytes) makes every OTHER resident copy that overlaps it stale (a view with an
// offset gets its own entry; the parent's device copy must not be served afterwards)
void DeviceCtx::owe(void *host, const void *dev, size_t bytes) {
    if (bytes >= OWE_EARLY_BYTES && wanted(host)) {
        // a large result goes home at once, by DMA on the copy stream, beside the kernels that follow (a prompt-sized batch
        // owes tens of MB per decoder layer: PCIe time that has to overlap the compute, not follow it); such a scope is
        // issued live (a captured graph runs its branches one after the other)
        scope_dirty();
        for (size_t i = 0; i < owed.size(); ++i)
            if (owed[i].host == host) { owed.erase(owed.begin() + (long)i); break; }
        hipError_t e = hipEventRecord(ev_ready, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_d2h, ev_ready, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s_d2h);
        if (e == hipSuccess) { d2h_busy = true; d2h_bytes += bytes; return; }
        (void)hipGetLastError();                   // could not: owe it like a small one
    }
    for (Owed &o : owed)
        if (o.host == host) { o.dev = dev; o.bytes = bytes; return; }
    owed.push_back(Owed{host, dev, bytes});
}
// a buffer is about to be reused while a copy on the copy stream may still read it: later kernels wait for that stream
void DeviceCtx::join_copies() {
    if (!d2h_busy) return;
    if (hipEventRecord(ev_d2h, s_d2h) != hipSuccess || hipStreamWaitEvent(stream, ev_d2h, 0) != hipSuccess) (void)hipGetLastError();
}
int DeviceCtx::pay(const void *only_dev, size_t only_bytes) {
    if (owed.empty()) return GGML_HIP_OK;
    std::vector<Owed> now, keep;
    // (a product with ne02 > 1 owes one entry per 2-D slice, all inside one resident buffer: every entry that lies in
    // [only_dev, only_dev + only_bytes) goes out before that buffer is recycled, not just the one at its base)
    const uint8_t *lo = (const uint8_t *)only_dev, *hi = lo + (only_bytes ? only_bytes : 1);
    for (const Owed &o : owed) (only_dev == nullptr || ((const uint8_t *)o.dev >= lo && (const uint8_t *)o.dev < hi) ? now : keep).push_back(o);
    owed.swap(keep);
    if (now.empty()) return GGML_HIP_OK;
    d2h_busy = true;
    // through the device mapping of the registered pool: one kernel for up to 32 copies; anything else: a DMA each
    std::vector<const void *> src; std::vector<void *> dst; std::vector<size_t> nb;
    hipError_t e = hipSuccess;
    for (const Owed &o : now) {
        d2h_bytes += o.bytes;
        void *m = ((uintptr_t)o.host % 4 == 0 && (uintptr_t)o.dev % 4 == 0 && o.bytes % 4 == 0) ? host_range_device_ptr(o.host, o.bytes) : nullptr;
        if (m) { src.push_back(o.dev); dst.push_back(m); nb.push_back(o.bytes); }
        else if (e == hipSuccess) e = hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, stream);
    }
    for (size_t i = 0; i < src.size() && e == hipSuccess; i += 32) {
        const int n = (int)(src.size() - i < 32 ? src.size() - i : 32);
        e = launch_scatter_copy(src.data() + i, dst.data() + i, nb.data() + i, n, stream);
    }
    return e == hipSuccess ? GGML_HIP_OK : fail(GGML_HIP_ERR_RUNTIME, "device -> host copy of graph results: %s", hipGetErrorString(e));
}
// Only what may still be on its way matters: a range that is owed a result is paid first; copies already issued are waited
// for; otherwise (the common case in a graph scope: leaves and nothing in flight) the upload can go out at once.
int DeviceCtx::before_host_read(const void *host, size_t bytes) {
    const uint8_t *a = (const uint8_t *)host, *b = a + bytes;
    for (const Owed &o : owed) {
        const uint8_t *x = (const uint8_t *)o.host, *y = x + o.bytes;
        if (x < b && a < y) return pay_and_sync();
    }
    return d2h_busy ? sync_all() : GGML_HIP_OK;
}
int DeviceCtx::pay_and_sync() {
    int rc = pay();
    const int r2 = sync_all();
    return rc ? rc : r2;
}
void DeviceCtx::drop_overlapping(const void *host, size_t bytes, bool keep_exact) {
    const uint8_t *a = (const uint8_t *)host, *b = a + bytes;
    for (auto it = resident.begin(); it != resident.end();) {
        const uint8_t *x = (const uint8_t *)it->first, *y = x + it->second.bytes;
        const bool overlap = x < b && a < y;
        if (overlap && !(keep_exact && it->first == host)) {
            (void)pay(it->second.p, it->second.bytes);   // its host copies go out (stream-ordered) before the buffer is reused
            join_copies();
            pool.push_back(it->second);
            it = resident.erase(it);
        } else {
            ++it;
        }
    }
}
void *DeviceCtx::resident_buffer(const void *host, size_t bytes) {
    drop_overlapping(host, bytes, true);
    auto old = resident.find(host);
    if (old != resident.end()) {
        if (old->second.bytes >= bytes) return old->second.p;      // the same tensor computed again: reuse its buffer
        (void)pay(old->second.p, old->second.bytes);
        join_copies();
        pool.push_back(old->second);
        resident.erase(old);
    }
    void *p = take(bytes);
    if (!p) return nullptr;
    // the entry records the bytes the TENSOR occupies on the host, not the (possibly larger) capacity of a recycled buffer
    resident[host] = Resident{p, bytes};
    return p;
}
void DeviceCtx::drain(bool free_all) {
    if (!owed.empty()) (void)pay_and_sync();       // (graph end has paid already; a shutdown inside a scope has not)
    for (auto &kv : resident) pool.push_back(kv.second);
    resident.clear();
    for (ggml_hip_weight *w : transient) ggml_hip_weight_free(w);
    transient.clear();
    if (free_all) {
        drop_captured();
        for (Resident &r : pool) (void)hipFree(r.p);
        pool.clear();
    }
}
void DeviceCtx::invalidate(const void *host, size_t bytes) {
    const uint8_t *a = (const uint8_t *)host, *b = a + (bytes ? bytes : 1);
    bool synced = false;
    for (auto it = cache.begin(); it != cache.end();) {
        const uint8_t *x = (const uint8_t *)it->second.host, *y = x + (it->second.host_bytes ? it->second.host_bytes : 1);
        if (x < b && a < y) {
            if (!synced) { (void)sync_all(); drop_captured(); synced = true; }     // kernels of an open graph scope may still read the entry
            for (ggml_hip_weight *w : it->second.slices) ggml_hip_weight_free(w);
            cache_bytes -= it->second.dev_bytes < cache_bytes ? it->second.dev_bytes : cache_bytes;
            it = cache.erase(it);
        } else {
            ++it;
        }
    }
}
void DeviceCtx::free_cache() {
    drop_captured();
    for (auto &kv : cache)
        for (ggml_hip_weight *w : kv.second.slices) ggml_hip_weight_free(w);
    cache.clear();
    cache_bytes = 0;
}
size_t DeviceCtx::evict_lru(size_t need, size_t budget, const CacheKey *keep) {
    size_t freed = 0;
    bool synced = false;
    for (;;) {
        const bool over = budget ? cache_bytes + need > budget : freed < need;
        if (!over) break;
        auto victim = cache.end();
        for (auto it = cache.begin(); it != cache.end(); ++it) {
            if (keep && it->first == *keep) continue;
            if (victim == cache.end() || it->second.last_use < victim->second.last_use) victim = it;
        }
        if (victim == cache.end()) break;                   // nothing left that may go
        if (!synced) { (void)sync_all(); drop_captured(); synced = true; }     // kernels in flight (an open graph scope) may still read the entry
        for (ggml_hip_weight *w : victim->second.slices) ggml_hip_weight_free(w);
        freed += victim->second.dev_bytes;
        cache_bytes -= victim->second.dev_bytes < cache_bytes ? victim->second.dev_bytes : cache_bytes;
        cache.erase(victim);
        ++cache_evictions;
    }
    return freed;
}

int n_slots() { return g_nslots.load(std::memory_order_acquire); }
DeviceCtx *slot(int i) { return (i >= 0 && i < n_slots()) ? g_slots[i] : nullptr; }
int bound_slot() { return (t_slot >= 0 && t_slot < n_slots()) ? t_slot : -1; }

namespace {

int create_slot_locked(int i, int device) {
    DeviceCtx *c = new DeviceCtx();
    c->slot = i;
    c->device = device;
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) c->arch = prop.gcnArchName;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking);
    for (int k = 0; k < PIPE_EVENTS && e == hipSuccess; ++k) {
        e = hipEventCreateWithFlags(&c->ev_in[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_k[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_d2h, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_xchg, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete c;
        return fail(GGML_HIP_ERR_RUNTIME, "device %d: %s", device, hipGetErrorString(e));
    }
    g_slots[i] = c;
    return GGML_HIP_OK;
}

std::vector<DeviceCtx *> g_graveyard;              // contexts of shut-down slots: resources released, the object kept for late lockers

void release_slot_resources_locked(DeviceCtx *c) {
    (void)hipSetDevice(c->device);
    (void)c->sync_all();
    c->free_cache();
    c->src1.release(); c->dst.release(); c->dst2.release(); c->work.release(); c->stage.release();
    for (Scratch &a : c->aux) a.release();
    c->drain(true);
    for (int k = 0; k < PIPE_EVENTS; ++k) {
        if (c->ev_in[k]) (void)hipEventDestroy(c->ev_in[k]);
        if (c->ev_k[k]) (void)hipEventDestroy(c->ev_k[k]);
    }
    for (hipEvent_t ev : {c->ev_compute, c->ev_d2h, c->ev_xchg, c->ev_ready})
        if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t s : {c->stream, c->s_h2d, c->s_d2h})
        if (s) (void)hipStreamDestroy(s);
    c->stream = c->s_h2d = c->s_d2h = nullptr;
}
void destroy_slot_locked(DeviceCtx *c) {            // (a slot that never became visible: failed initialisation)
    release_slot_resources_locked(c);
    delete c;
}

int init_slots(int n, const int *ids) {
    const int ndev = ggml_hip_device_count();
    if (ndev <= 0) return fail(GGML_HIP_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
    if (n <= 0 || n > MAX_SLOTS) return fail(GGML_HIP_ERR_ARG, "n_devices %d out of range [1,%d]", n, MAX_SLOTS);
    for (int i = 0; i < n; ++i) {
        const int d = ids ? ids[i] : i;
        if (d < 0 || d >= ndev) return fail(GGML_HIP_ERR_ARG, "device %d out of range [0,%d)", d, ndev);
    }
    std::lock_guard<std::mutex> lk(g_table_mu);
    if (g_nslots.load() > 0) {
        bool same = g_nslots.load() == n;
        for (int i = 0; same && i < n; ++i) same = g_slots[i]->device == (ids ? ids[i] : i);
        if (same) return GGML_HIP_OK;
        // streams, scratch, caches and kernel attributes belong to the devices they were created on: no silent re-targeting
        return fail(GGML_HIP_ERR_ARG, "already initialised on a different device set; call ggml_hip_shutdown first");
    }
    for (int i = 0; i < n; ++i) {
        int rc = create_slot_locked(i, ids ? ids[i] : i);
        if (rc) {
            for (int k = 0; k < i; ++k) { destroy_slot_locked(g_slots[k]); g_slots[k] = nullptr; }
            return rc;
        }
    }
    // peer access between distinct devices (xGMI on an MI355X node): kernels and DMA engines of one slot may then address
    // another slot's buffers directly.  Failure is not fatal -- peer copies then stage through the host.
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k) {
            if (g_slots[i]->device == g_slots[k]->device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g_slots[i]->device, g_slots[k]->device) == hipSuccess && can) {
                (void)hipSetDevice(g_slots[i]->device);
                hipError_t e = hipDeviceEnablePeerAccess(g_slots[k]->device, 0);
                if (e != hipSuccess) (void)hipGetLastError();     // already enabled, or refused: clear the sticky error
            }
        }
    (void)hipSetDevice(g_slots[0]->device);
    g_nslots.store(n, std::memory_order_release);
    return GGML_HIP_OK;
}

}  // namespace

int ensure_init() {
    if (n_slots() > 0) return GGML_HIP_OK;
    return ggml_hip_init(0);
}

// the slot a device-level or row-function call runs on: the calling thread's bound slot, else slot 0
static DeviceCtx *call_slot() {
    const int b = bound_slot();
    return slot(b >= 0 ? b : 0);
}

static int alloc_weight(DeviceCtx *c, int type, int64_t K, int64_t M, ggml_hip_weight **out, int kq_type = 0) {
    ggml_hip_weight *w = new ggml_hip_weight();
    memset(w, 0, sizeof *w);
    const bool kq = kq_type != 0;                           // (a k-quant weight: Q5_K / Q4_K in the planar Q5_1 form, Q6_K in the planar Q4_2 form)
    w->ext_type = kq_type;
    static std::atomic<uint64_t> next_uid{1};
    w->type = type; w->M = M; w->K = K; w->Mpad = pad_rows(M > 0 ? M : 1); w->device = c->device; w->uid = next_uid.fetch_add(1);
    size_t off_qs = 0, off_d = 0, off_m = 0, off_qh = 0, off_6a = 0, off_6b = 0, off_kh = 0, off_gs = 0, off_i8 = 0, off_mp = 0, total = 0;
    bool with6 = false;
    size_t off_p16 = 0;
    if (type == GGML_TYPE_F32 || type == GGML_TYPE_F16) {
        total = ((size_t)w->Mpad * K * (type == GGML_TYPE_F32 ? 4 : 2) + 255) / 256 * 256;
        if (type == GGML_TYPE_F16) {   // k-panel copy for the f16 MFMA kernel (dense16.hip)
            off_p16 = total;
            total += (size_t)(dense16_kpad(K) / 8 + DENSE16_SPARE_PANELS) * w->Mpad * 16;
        } else {                       // the rows as three bf16 pieces per element (dense16.hip K10d)
            off_p16 = total;
            total += (size_t)(dense16_kpad(K) / 8 * 3 + DENSE32_SPARE_PANELS) * w->Mpad * 16;
        }
    } else {
        w->nbk = K / QK;
        const int64_t nba = pad_kblocks(w->nbk) + K_LOOKAHEAD;   // allocated k-blocks (zero past the real end)
        // (Q6_K lives in the planar Q4_2 form on its int8 planes alone -- no kernel it is planned onto reads the nibble plane: a stub)
        const size_t qs_bytes = kq_type == GGML_HIP_TYPE_Q6_K ? 256 : (size_t)nba * w->Mpad * (type == GGML_TYPE_Q8_0 ? 32 : 16);
        const size_t plane = (size_t)nba * w->Mpad * 4;
        off_qs = 0; total = qs_bytes;
        off_d = total; total += plane;
        if (has_min_plane(type)) { off_m = total; total += plane; }
        if (has_qh_plane(type)) { off_qh = total; total += plane; }
        off_gs = total; total += (size_t)w->nbk * w->Mpad * 4 * gemv_side_planes(type);       // the mat-vec's tile-major copy of d / m / qh
        const bool q4 = type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1;
        with6 = q4 || (plan_force_gemm() == 3 && (type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q8_0));
        if (type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q4_2) { off_i8 = total; total += (size_t)nba * w->Mpad * 32; }   // int8 operand planes (gemm_qmp.hip), zero past the end of K
        // the min plane as three bf16 pieces (K3p-int8's min-term product): whole pairs of k-groups, zero past the end of K
        if (type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_1) { off_mp = total; total += (size_t)((w->nbk + 15) / 16 * 2 * 3) * w->Mpad * 16; }
        if (kq) { off_kh = total; total += (size_t)(w->nbk / 8 + 1) * w->Mpad * (kq_type == GGML_HIP_TYPE_Q6_K ? 32 : 16); }   // super-block headers, for the byte-exact download
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
        else w->p32 = (uint8_t *)base + off_p16;
    } else {
        w->qs = (uint8_t *)base + off_qs;
        w->d = (float *)((uint8_t *)base + off_d);
        if (has_min_plane(type)) w->m = (float *)((uint8_t *)base + off_m);
        if (has_qh_plane(type)) w->qh = (uint32_t *)((uint8_t *)base + off_qh);
        w->gs = (uint32_t *)((uint8_t *)base + off_gs);
        if (with6) { w->q6a = (uint8_t *)base + off_6a; w->q6b = (uint8_t *)base + off_6b; }
        if (kq) w->khdr = (uint8_t *)base + off_kh;
        if (type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q4_2) w->i8p = (uint8_t *)base + off_i8;
        if (type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q4_1) w->mp3 = (uint8_t *)base + off_mp;
    }
    *out = w;
    return GGML_HIP_OK;
}

static void *weight_base(const ggml_hip_weight *w) { return w->dense ? w->dense : (void *)w->qs; }

int make_weight(DeviceCtx *c, int type, const void *rows, bool rows_on_host, int64_t ne00, int64_t ne01, uint64_t nb01,
                int64_t row_begin, int64_t row_end, hipStream_t st, ggml_hip_weight **out) {
    if (!out) return fail(GGML_HIP_ERR_ARG, "out is null");
    *out = nullptr;
    // Q5_K (unpinned extra, kquants.hip): super-blocks of 256 are re-laid-out as eight k-blocks of the planar Q5_1 form
    // (r4: Q4_K the same way -- its super-block is Q5_K's without the fifth-bit bytes; the fifth-bit plane stays zero)
    const bool kq = is_kquant(type);
    if (kq) {
        if (!rows || ne00 <= 0 || ne01 < 0 || row_begin < 0 || row_end < row_begin || row_end > ne01) return fail(GGML_HIP_ERR_ARG, "bad weight arguments");
        if (ne00 % 256 != 0) return fail(GGML_HIP_ERR_SHAPE, "Q5_K / Q4_K: ne00 %% 256 != 0 (QK_K)");
        if (nb01 < (uint64_t)(ne00 / 256) * kquant_bytes(type)) return fail(GGML_HIP_ERR_SHAPE, "nb01 smaller than a row");
        int rc = c ? GGML_HIP_OK : ensure_init();
        if (rc) return rc;
        if (!c) c = call_slot();
        rc = c->make_current();
        if (rc) return rc;
        const int64_t rows_n = row_end - row_begin;
        const uint64_t rb = (uint64_t)(ne00 / 256) * kquant_bytes(type);
        ggml_hip_weight *w = nullptr;
        const bool q6k = type == GGML_HIP_TYPE_Q6_K;           // (the planar Q4_2 form on int8 planes: kquants.hip)
        rc = alloc_weight(c, kquant_resident_type(type), ne00, rows_n, &w, type);
        if (rc) return rc;
        hipError_t e = hipMemsetAsync(w->qs, 0, w->bytes, st);
        void *staging = nullptr;
        if (e == hipSuccess && rows_on_host && rows_n > 0) {
            e = hipMalloc(&staging, (size_t)rows_n * rb);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(staging, rb, (const uint8_t *)rows + (uint64_t)row_begin * nb01, nb01, rb, (size_t)rows_n, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = q6k ? launch_q6k_to_planar((const uint8_t *)staging, rb, 0, rows_n, w, st) : launch_q5k_to_planar(type, (const uint8_t *)staging, rb, 0, rows_n, w, st);
        } else if (e == hipSuccess) {
            e = q6k ? launch_q6k_to_planar((const uint8_t *)rows, nb01, row_begin, rows_n, w, st) : launch_q5k_to_planar(type, (const uint8_t *)rows, nb01, row_begin, rows_n, w, st);
        }
        if (e == hipSuccess && !q6k) e = launch_q5_to_i8(w, st);      // (the planar Q5_1 form's int8 operand planes: gemm_qmp.hip serves prompt-sized batches; Q6_K's converter writes them itself)
        if (e == hipSuccess && !q6k) e = launch_min_pieces(w, st);
        if (e == hipSuccess) e = launch_gemv_side_image(w, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (staging) (void)hipFree(staging);
        if (e != hipSuccess) { (void)hipFree(w->qs); delete w; return fail(GGML_HIP_ERR_RUNTIME, "k-quant weight upload: %s", hipGetErrorString(e)); }
        *out = w;
        return GGML_HIP_OK;
    }
    if (type < 0 || type >= GGML_TYPE_COUNT || !weight_type_ok(type))
        return fail(GGML_HIP_ERR_TYPE, "type %d is not a supported weight type (Q4_3/Q8_1 have null slots, Ggml.cs:248,278-282)", type);
    if (!rows || ne00 <= 0 || ne01 < 0 || row_begin < 0 || row_end < row_begin || row_end > ne01)
        return fail(GGML_HIP_ERR_ARG, "bad weight arguments");
    if (ne00 % BLCK[type] != 0 || (is_q(type) && ne00 % QK != 0))   // every dot product runs against 32-element Q8 blocks
        return fail(GGML_HIP_ERR_SHAPE, "ne00 %% %d != 0 (Ggml.cs:6694)", is_q(type) ? QK : BLCK[type]);
    const uint64_t row_bytes = (uint64_t)TSIZE[type] * (uint64_t)(ne00 / BLCK[type]);
    if (nb01 < row_bytes) return fail(GGML_HIP_ERR_SHAPE, "nb01 smaller than a row (transposed src0, Ggml.cs:8229)");
    if (!rows_on_host && (type == GGML_TYPE_F32 || type == GGML_TYPE_F16) && (nb01 % 2 != 0 || ((uintptr_t)rows & 1)))
        return fail(GGML_HIP_ERR_SHAPE, "dense device rows must be 2-byte aligned");
    int rc = c ? GGML_HIP_OK : ensure_init();
    if (rc) return rc;
    if (!c) c = call_slot();
    rc = c->make_current();
    if (rc) return rc;
    const int64_t rows_n = row_end - row_begin;
    ggml_hip_weight *w = nullptr;
    rc = alloc_weight(c, type, ne00, rows_n, &w);
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
    if (e == hipSuccess) e = launch_q5_to_i8(w, st);
    if (e == hipSuccess) e = launch_min_pieces(w, st);
    if (e == hipSuccess) e = launch_gemv_side_image(w, st);
    if (e == hipSuccess) e = launch_f16_rows_to_panels(w, st);
    if (e == hipSuccess) e = launch_f32_rows_to_split_panels(w, st);
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

// device-level entries take the caller's stream; the device the weight lives on must be current for the launch
static int weight_device_current(const ggml_hip_weight *w) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == w->device) return GGML_HIP_OK;
    hipError_t e = hipSetDevice(w->device);
    return e == hipSuccess ? GGML_HIP_OK : fail(GGML_HIP_ERR_RUNTIME, "hipSetDevice(%d): %s", w->device, hipGetErrorString(e));
}

}  // namespace ghip

using namespace ghip;

extern "C" {

int ggml_hip_blck_size(int type) { return is_kquant(type) ? 256 : (type >= 0 && type < GGML_TYPE_COUNT) ? BLCK[type] : 0; }
size_t ggml_hip_type_size(int type) { return is_kquant(type) ? kquant_bytes(type) : (type >= 0 && type < GGML_TYPE_COUNT) ? TSIZE[type] : 0; }

int ggml_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ggml_hip_init(int device) {
    const int n = ggml_hip_device_count();
    if (n <= 0) return fail(GGML_HIP_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
    if (device < 0 || device >= n) return fail(GGML_HIP_ERR_ARG, "device %d out of range [0,%d)", device, n);
    if (n_slots() > 0) {
        // a process that already drives a device set keeps it; selecting one of ITS devices for the calling thread is fine
        for (int i = 0; i < n_slots(); ++i)
            if (slot(i)->device == device) { HIP_TRY(hipSetDevice(device)); return GGML_HIP_OK; }
        return fail(GGML_HIP_ERR_ARG, "already initialised on another device (%d); call ggml_hip_shutdown first", slot(0)->device);
    }
    return init_slots(1, &device);
}

int ggml_hip_init_devices(int n_devices, const int *device_ids) { return init_slots(n_devices, device_ids); }
int ggml_hip_n_slots(void) { return n_slots(); }
int ggml_hip_slot_device(int s) { DeviceCtx *c = slot(s); return c ? c->device : -1; }

int ggml_hip_bind_thread(int s) {
    if (s < -1 || s >= n_slots()) return fail(GGML_HIP_ERR_ARG, "slot %d out of range [-1,%d)", s, n_slots());
    t_slot = s;
    if (s >= 0) return slot(s)->make_current();
    return GGML_HIP_OK;
}

void ggml_hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_table_mu);
    const int n = g_nslots.load();
    if (n == 0) return;
    g_nslots.store(0, std::memory_order_release);
    for (int i = 0; i < n; ++i) {
        DeviceCtx *c = g_slots[i];
        {
            // a seam that is still inside finishes first; one that fetched the pointer before n_slots dropped to 0 but has not
            // locked yet finds `dead` set when it does (Call::begin, scope_replaying) and returns "not initialised"
            std::lock_guard<std::recursive_mutex> l2(c->mu);
            c->dead = true;
            g_slots[i] = nullptr;
            release_slot_resources_locked(c);
        }
        g_graveyard.push_back(c);          // the object (its mutex) outlives late lockers; freed at the next shutdown / process end
    }
    for (size_t k = 0; k + 64 < g_graveyard.size(); ++k) { delete g_graveyard[k]; g_graveyard[k] = nullptr; }
    g_graveyard.erase(std::remove(g_graveyard.begin(), g_graveyard.end(), nullptr), g_graveyard.end());
    rccl_shutdown();
}

const char *ggml_hip_last_error(void) { return t_err.c_str(); }
const char *ggml_hip_arch(void) { DeviceCtx *c = slot(0); return c ? c->arch.c_str() : ""; }

int ggml_hip_weight_upload(int type, const void *host_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                           int64_t row_begin, int64_t row_end, void *stream, ggml_hip_weight **out) {
    return make_weight(nullptr, type, host_rows, true, ne00, ne01, nb01, row_begin, row_end, (hipStream_t)stream, out);
}

int ggml_hip_weight_from_device(int type, const void *dev_rows, int64_t ne00, int64_t ne01, uint64_t nb01,
                                int64_t row_begin, int64_t row_end, void *stream, ggml_hip_weight **out) {
    return make_weight(nullptr, type, dev_rows, false, ne00, ne01, nb01, row_begin, row_end, (hipStream_t)stream, out);
}

int ggml_hip_weight_download(const ggml_hip_weight *w, void *host_rows, void *stream) {
    if (!w || !host_rows) return fail(GGML_HIP_ERR_ARG, "null argument");
    int rc = ensure_init();
    if (rc) return rc;
    rc = weight_device_current(w);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t row_bytes = w->ext_type != 0 ? (size_t)(w->K / 256) * kquant_bytes(w->ext_type) : TSIZE[w->type] * (size_t)(w->K / BLCK[w->type]);
    const size_t total = row_bytes * (size_t)w->M;
    if (total == 0) return GGML_HIP_OK;
    void *staging = nullptr;
    HIP_TRY(hipMalloc(&staging, total));
    hipError_t e = w->ext_type == GGML_HIP_TYPE_Q6_K ? launch_planar_to_q6k(w, (uint8_t *)staging, st)
                   : w->ext_type != 0 ? launch_planar_to_q5k(w, (uint8_t *)staging, st) : launch_planar_to_aos(w, (uint8_t *)staging, st);
    if (e == hipSuccess) e = hipMemcpyAsync(host_rows, staging, total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "weight download: %s", hipGetErrorString(e));
    return GGML_HIP_OK;
}

void ggml_hip_weight_free(ggml_hip_weight *w) {
    if (!w) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != w->device) (void)hipSetDevice(w->device);
    (void)hipFree(weight_base(w));
    if (cur >= 0 && cur != w->device) (void)hipSetDevice(cur);
    delete w;
}
int64_t ggml_hip_weight_rows(const ggml_hip_weight *w) { return w ? w->M : 0; }
int64_t ggml_hip_weight_cols(const ggml_hip_weight *w) { return w ? w->K : 0; }
int ggml_hip_weight_type(const ggml_hip_weight *w) { return w ? (w->ext_type ? w->ext_type : w->type) : -1; }

size_t ggml_hip_mul_mat_work_size(int type, int64_t K, int64_t N) {
    if (K <= 0 || N <= 0) return 0;
    if (is_kquant(type)) type = kquant_resident_type(type);     // same operand images
    if (type == GGML_TYPE_F16) return (size_t)dense16_kpad(K) * (size_t)pad_act(N) * 2;   // src1 as Half (Ggml.cs:3356-3357), padded
    if (type == GGML_TYPE_F32) return N > 256 ? (size_t)dense16_kpad(K) * (size_t)pad_act(N) * 6 : 0;   // src1 as three bf16 pieces (dense16.hip K10d; the reference needs none)
    if (!is_q(type)) return 0;
    return act_bytes(K, pad_act(N));
}

// the INIT kernels read src1 rows in 16-byte pieces (float4 loads, LDS-DMA): base and row stride must allow that
static int check_src1_alignment(const float *d_src1, int64_t ld1) {
    if (((uintptr_t)d_src1 & 15) != 0 || ld1 % 4 != 0)
        return fail(GGML_HIP_ERR_SHAPE, "src1 must be 16-byte aligned with a row stride that is a multiple of 4 elements");
    return GGML_HIP_OK;
}

int ggml_hip_mul_mat_init_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, void *d_work,
                              size_t work_bytes, void *stream) {
    if (!w || !d_src1) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (N <= 0) return GGML_HIP_OK;
    if (!is_q(w->type)) return GGML_HIP_OK;  // dense: INIT is a no-op for f32 (Ggml.cs:6117-6120); f16 rounds in-kernel
    if (ld1 < w->K) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K");
    int rc = check_src1_alignment(d_src1, ld1);
    if (rc) return rc;
    if (!d_work || work_bytes < ggml_hip_mul_mat_work_size(w->type, w->K, N))
        return fail(GGML_HIP_ERR_ARG, "work buffer too small: need %zu", ggml_hip_mul_mat_work_size(w->type, w->K, N));
    rc = weight_device_current(w);
    if (rc) return rc;
    act_planes p = act_carve(d_work, w->K, pad_act(N));
    HIP_TRY(launch_quantize_act(d_src1, N, w->K, ld1, p, weight_image_kind(w, N), (hipStream_t)stream,
                                w->ext_type != 0));                      // k-quant weights: the Q8_K rule (one scale per 256)
    return GGML_HIP_OK;
}

int ggml_hip_act_image_kind(int type, int64_t K, int64_t N) { return act_image_kind(is_kquant(type) ? kquant_resident_type(type) : type, K, N); }
void ggml_hip_debug_force_gemm(int which) { plan_set_force_gemm(which); }

// the plan of mul_mat(type, M, K, N) as ggml_hip_mul_mat_dev will run it; no device is needed (tests/test_plan_cpu.py)
int ggml_hip_mm_plan(int type, int64_t M, int64_t K, int64_t N, ggml_hip_mm_plan_t *out) {
    if (!out) return fail(GGML_HIP_ERR_ARG, "out is null");
    const bool kq = is_kquant(type);
    const int t = kq ? kquant_resident_type(type) : type;
    if (t < 0 || t >= GGML_TYPE_COUNT || !weight_type_ok(t)) return fail(GGML_HIP_ERR_TYPE, "type %d is not a supported weight type", type);
    if (M <= 0 || K <= 0 || N <= 0 || K % BLCK[t] != 0 || (is_q(t) && K % QK != 0) || (kq && K % 256 != 0)) return fail(GGML_HIP_ERR_SHAPE, "bad shape");
    const mm_plan p = plan_mul_mat(t, kq ? type : 0, M, K, N, true);
    out->family = p.family; out->image_kind = p.image | ((p.flags & MM_FLAG_MIN_PIECES) ? ACT_IMAGE_MIN_PIECES : 0); out->form = p.form; out->tree_id = plan_tree_id(p);
    out->ksplit = p.ksplit; out->kstyle = p.kstyle; out->kunit = p.kunit; out->arith = p.arith;
    out->tile_m = p.tile_m; out->tile_n = p.tile_n; out->waves = p.waves; out->tiles_per_wave = p.tiles_per_wave;
    out->workgroups = p.wgs; out->flags = p.flags;
    return GGML_HIP_OK;
}

int ggml_hip_quantize_act_dev(const float *d_src1, int64_t N, int64_t K, int64_t ld1, void *d_work, size_t work_bytes,
                              int image_kind, void *stream) {
    if (N <= 0) return GGML_HIP_OK;
    if (!d_src1 || !d_work) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (K <= 0 || K % QK != 0 || ld1 < K) return fail(GGML_HIP_ERR_SHAPE, "K %% 32 != 0 or ld1 < K");
    int rc = check_src1_alignment(d_src1, ld1);
    if (rc) return rc;
    const bool pieces = image_kind >= 0 && (image_kind & ACT_IMAGE_MIN_PIECES) != 0;   // + 64 (kind 0 only): the min-term piece planes as well
    if (pieces) image_kind &= ~ACT_IMAGE_MIN_PIECES;
    const bool q8k = image_kind >= 16;                         // + 16: the Q8_K rule of the k-quants (K % 256 == 0; kinds 0..2)
    if (q8k) image_kind -= 16;
    if (image_kind < 0 || image_kind > 3 || (q8k && (image_kind == 3 || K % 256 != 0)) || (pieces && (image_kind != 0 || K / QK < 8)))
        return fail(GGML_HIP_ERR_ARG, "image kind %d", image_kind);
    if (work_bytes < act_bytes(K, pad_act(N))) return fail(GGML_HIP_ERR_ARG, "work buffer too small: need %zu", act_bytes(K, pad_act(N)));
    // the MFMA images are written (and read) through 32-bit buffer offsets: 64 image bytes per row and k-block
    if (image_kind != 0 && (uint64_t)pad_kblocks(K / QK) * 64 * (uint64_t)pad_act(N) > 0xFFFFFFFFull)
        return fail(GGML_HIP_ERR_SHAPE, "image kind %d needs K/32 * 64 * Npad < 4 GiB (ggml_hip_act_image_kind never selects it beyond that)", image_kind);
    HIP_TRY(launch_quantize_act(d_src1, N, K, ld1, act_carve(d_work, K, pad_act(N)), image_kind | (pieces ? ACT_IMAGE_MIN_PIECES : 0), (hipStream_t)stream, q8k));
    return GGML_HIP_OK;
}

int ggml_hip_mul_mat_compute_dev(const ggml_hip_weight *w, int64_t N, float *d_dst, int64_t ldd, const void *d_work,
                                 size_t work_bytes, void *stream) {
    if (!w || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (!is_q(w->type)) return fail(GGML_HIP_ERR_TYPE, "compute_dev is the quantized COMPUTE phase; use ggml_hip_mul_mat_dev for dense");
    if (ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "ldd < M");
    if (!d_work || work_bytes < ggml_hip_mul_mat_work_size(w->type, w->K, N)) return fail(GGML_HIP_ERR_ARG, "work buffer too small");
    int rc = weight_device_current(w);
    if (rc) return rc;
    act_planes p = act_carve((void *)d_work, w->K, pad_act(N));
    const mm_plan pl = weight_plan(w, N, false);            // (the COMPUTE-only entry: INIT wrote this plan's image)
    const mm_epilogue none{0, nullptr, 0, nullptr, 0, 1.0f};
    switch (pl.family) {
    case MMF_K3S_I8:    HIP_TRY(launch_gemm_q8_small(w, pl, p, N, d_dst, ldd, (hipStream_t)stream, nullptr)); break;
    case MMF_K3P_I8:    HIP_TRY(launch_gemm_q8_mid(w, pl, p, N, d_dst, ldd, (hipStream_t)stream, none)); break;
    case MMF_GEMV_ROWS: HIP_TRY(launch_gemv_q(w, p, N, d_dst, ldd, (hipStream_t)stream)); break;
    case MMF_K3S_MX: case MMF_K3P_MX: case MMF_MX:
                        HIP_TRY(launch_gemm_qmx(w, pl, p, N, d_dst, ldd, (hipStream_t)stream)); break;
    case MMF_F16:       HIP_TRY(launch_gemm_q16(w, pl, p, N, d_dst, ldd, (hipStream_t)stream)); break;
    case MMF_I8:        HIP_TRY(launch_gemm_q(w, pl, p, N, d_dst, ldd, (hipStream_t)stream)); break;
    default:            return fail(GGML_HIP_ERR_RUNTIME, "no kernel family for this product (plan family %d)", pl.family);
    }
    return GGML_HIP_OK;
}

int ggml_hip_mul_mat_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, float *d_dst,
                         int64_t ldd, void *d_work, size_t work_bytes, void *stream) {
    if (!w) return fail(GGML_HIP_ERR_ARG, "null weight");
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (!d_src1 || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (ld1 < w->K || ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K or ldd < M");
    int rc = weight_device_current(w);
    if (rc) return rc;
    const mm_plan pl = weight_plan(w, N, true);
    if (!is_q(w->type)) {
        if (pl.family == MMF_DENSE16 || pl.family == MMF_DENSE32) {
            // The matrix-core forms need INIT scratch (the reference's own wdata for F16, Ggml.cs:6362-6379; three bf16 pieces per element for
            // F32 above 256 rows) and read src1 rows in 16-byte pieces.  r4: a missing buffer or a misaligned src1 is an ERROR -- it
            // used to select dense.hip silently, whose arithmetic differs in the last bits, so the same product could differ between callers.
            if (!d_work || work_bytes < ggml_hip_mul_mat_work_size(w->type, w->K, N))
                return fail(GGML_HIP_ERR_ARG, "work buffer too small: this shape needs %zu bytes (ggml_hip_mul_mat_work_size)", ggml_hip_mul_mat_work_size(w->type, w->K, N));
            if (ld1 % 4 != 0 || ((uintptr_t)d_src1 & 15) != 0) return fail(GGML_HIP_ERR_SHAPE, "src1 rows must be 16-byte aligned (base and row stride) for this shape");
            if (pl.family == MMF_DENSE16) {
                HIP_TRY(launch_dense16_init(d_src1, N, w->K, ld1, d_work, (hipStream_t)stream));       // INIT: src1 -> Half (Ggml.cs:6362-6379)
                HIP_TRY(launch_dense16(w, pl, d_work, N, d_dst, ldd, (hipStream_t)stream));
            } else {
                HIP_TRY(launch_dense32_init(d_src1, N, w->K, ld1, d_work, (hipStream_t)stream));
                HIP_TRY(launch_dense32(w, d_work, N, d_dst, ldd, (hipStream_t)stream));
            }
            return GGML_HIP_OK;
        }
        HIP_TRY(launch_dense(w, pl, d_src1, N, ld1, d_dst, ldd, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
    if (pl.family == MMF_GEMV_FUSED) {                      // small N: INIT and COMPUTE fused in one launch, no scratch needed
        rc = check_src1_alignment(d_src1, ld1);     // (float4 loads of the activation rows)
        if (rc) return rc;
        HIP_TRY(launch_gemv_q_fused(w, d_src1, ld1, N, d_dst, ldd, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
    rc = ggml_hip_mul_mat_init_dev(w, d_src1, N, ld1, d_work, work_bytes, stream);
    if (rc) return rc;
    return ggml_hip_mul_mat_compute_dev(w, N, d_dst, ldd, d_work, work_bytes, stream);
}

// does a kernel form with a fused store-phase epilogue serve this (weight, N)?  The fused mat-vec (N <= 4), the MX mat-mat (Q4_0 / Q4_1),
// K3p and Q8_0's batched-decode form; every other form runs the epilogue as its own launch behind the mat-mul.  (The plan's flag: selection,
// this report and the launch agree by construction -- ADVICE r3.)
static bool epilogue_is_fused(const ggml_hip_weight *w, int64_t N) {
    return is_q(w->type) && w->ext_type == 0 && (weight_plan(w, N, true).flags & MM_FLAG_EPILOGUE_FUSED) != 0;
}

int ggml_hip_mul_mat_epilogue_fused(const ggml_hip_weight *w, int64_t N) { return w && epilogue_is_fused(w, N) ? 1 : 0; }

int ggml_hip_mul_mat_epilogue_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, float *d_dst, int64_t ldd,
                                  void *d_work, size_t work_bytes, int mode, const float *d_addend, int64_t ld_add, float *d_dst2,
                                  int64_t ldd2, float scale, void *stream) {
    if (mode == 0) return ggml_hip_mul_mat_dev(w, d_src1, N, ld1, d_dst, ldd, d_work, work_bytes, stream);
    if (!w) return fail(GGML_HIP_ERR_ARG, "null weight");
    if (mode != 1 && mode != 2) return fail(GGML_HIP_ERR_ARG, "epilogue mode %d (1 = add, 2 = scale)", mode);
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (mode == 1 && (!d_addend || !d_dst2 || ld_add < w->M || ldd2 < w->M)) return fail(GGML_HIP_ERR_ARG, "add epilogue: addend / dst2 missing or too narrow");
    const mm_epilogue ep = {mode, d_addend, ld_add, d_dst2, ldd2, scale};
    if (epilogue_is_fused(w, N)) {
        if (!d_src1 || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
        if (ld1 < w->K || ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K or ldd < M");
        int rc = weight_device_current(w);
        if (rc) return rc;
        const mm_plan plan = weight_plan(w, N, true);
        if (plan.family == MMF_GEMV_FUSED) {
            rc = check_src1_alignment(d_src1, ld1);
            if (rc) return rc;
            HIP_TRY(launch_gemv_q_fused(w, d_src1, ld1, N, d_dst, ldd, (hipStream_t)stream, &ep));
            return GGML_HIP_OK;
        }
        rc = ggml_hip_mul_mat_init_dev(w, d_src1, N, ld1, d_work, work_bytes, stream);
        if (rc) return rc;
        const act_planes pl = act_carve(d_work, w->K, pad_act(N));
        const hipError_t e = plan.family == MMF_K3S_I8 ? launch_gemm_q8_small(w, plan, pl, N, d_dst, ldd, (hipStream_t)stream, &ep)
                             : plan.family == MMF_K3P_I8 ? launch_gemm_q8_mid(w, plan, pl, N, d_dst, ldd, (hipStream_t)stream, ep)
                                                         : launch_gemm_qmx(w, plan, pl, N, d_dst, ldd, (hipStream_t)stream, &ep);
        if (e == hipSuccess) return GGML_HIP_OK;
        // not supported = operands of the EPILOGUE beyond the 32-bit offsets of these kernels (ld_add / ld2 past 4 GiB): the product itself
        // succeeds, so it does with an epilogue -- the unfused path below
        if (e != hipErrorNotSupported) HIP_TRY(e);
        (void)hipGetLastError();
    }
    // no fused form for this kernel: the product, then the node's own kernel row by row (same values)
    int rc = ggml_hip_mul_mat_dev(w, d_src1, N, ld1, d_dst, ldd, d_work, work_bytes, stream);
    if (rc) return rc;
    if (mode == 2) {
        if (ldd == w->M) HIP_TRY(launch_scale_f32(d_dst, N * w->M, scale, (hipStream_t)stream));
        else for (int64_t n = 0; n < N; ++n) HIP_TRY(launch_scale_f32(d_dst + n * ldd, w->M, scale, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
    if (ldd == w->M && ld_add == w->M && ldd2 == w->M) HIP_TRY(launch_binary_f32(0, d_dst, d_addend, d_dst2, N * w->M, (hipStream_t)stream));
    else for (int64_t n = 0; n < N; ++n) HIP_TRY(launch_binary_f32(0, d_dst + n * ldd, d_addend + n * ld_add, d_dst2 + n * ldd2, w->M, (hipStream_t)stream));
    return GGML_HIP_OK;
}

/* ---- the row split's product + exchange ---- */
// Which kernel forms store to the peers themselves: the staged MX forms and K3p (plan.h).  The others compute into this rank's own buffer
// and hand their columns to the peers with the column-push kernel (layout.hip) behind the product: the same bytes, one more launch.
static bool push_is_fused(const ggml_hip_weight *w, int64_t N, int n_peers) {
    if (!is_q(w->type) || n_peers - 1 > MM_PUSH_MAX) return false;
    const int f = weight_plan(w, N, true).family;
    return f == MMF_MX || f == MMF_K3P_MX || f == MMF_K3P_I8 || f == MMF_K3S_MX || f == MMF_K3S_I8;   // (r5: the batched-decode forms too -- a short shard at prompt sizes runs them)
}
int ggml_hip_mul_mat_push_fused(const ggml_hip_weight *w, int64_t N, int n_peers) { return w && push_is_fused(w, N, n_peers) ? 1 : 0; }

int ggml_hip_mul_mat_push_dev(const ggml_hip_weight *w, const float *d_src1, int64_t N, int64_t ld1, float *const *d_peers, int n_peers, int own,
                              int64_t ld_total, int64_t col0, void *d_work, size_t work_bytes, void *stream) {
    if (!w || !d_peers) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (n_peers < 1 || n_peers > 16 || own < 0 || own >= n_peers || !d_peers[own]) return fail(GGML_HIP_ERR_ARG, "1..16 destination buffers, this rank's own among them");
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (col0 < 0 || ld_total < col0 + w->M) return fail(GGML_HIP_ERR_SHAPE, "ld_total < col0 + M");
    float *mine = d_peers[own] + col0;
    if (push_is_fused(w, N, n_peers)) {
        if (!d_src1) return fail(GGML_HIP_ERR_ARG, "null argument");
        if (ld1 < w->K) return fail(GGML_HIP_ERR_SHAPE, "ld1 < K");
        mm_epilogue ep = {3, nullptr, 0, nullptr, 0, 1.0f, 0, {}};
        for (int g = 0; g < n_peers; ++g)
            if (g != own && d_peers[g]) ep.push[ep.npush++] = d_peers[g] + col0;
        int rc = weight_device_current(w);
        if (rc) return rc;
        rc = ggml_hip_mul_mat_init_dev(w, d_src1, N, ld1, d_work, work_bytes, stream);
        if (rc) return rc;
        const mm_plan plan = weight_plan(w, N, true);
        const act_planes pl = act_carve(d_work, w->K, pad_act(N));
        const hipError_t e = plan.family == MMF_K3P_I8 ? launch_gemm_q8_mid(w, plan, pl, N, mine, ld_total, (hipStream_t)stream, ep)
                             : plan.family == MMF_K3S_I8 ? launch_gemm_q8_small(w, plan, pl, N, mine, ld_total, (hipStream_t)stream, &ep)
                                                         : launch_gemm_qmx(w, plan, pl, N, mine, ld_total, (hipStream_t)stream, &ep);
        if (e == hipSuccess) return GGML_HIP_OK;
        // not supported = ld_total beyond what the store phase addresses (a tile's rows past the 32-bit buffer offsets of the staged MX forms,
        // a row stride past K3p's int): the product itself still runs, as ggml_hip_mul_mat_epilogue_dev's unfused path does -- the plain
        // product (the one-call entry routes such a dst to a kernel that can address it) and the column-push kernel behind it
        if (e != hipErrorNotSupported) HIP_TRY(e);
        (void)hipGetLastError();
    }
    int rc = ggml_hip_mul_mat_dev(w, d_src1, N, ld1, mine, ld_total, d_work, work_bytes, stream);
    if (rc) return rc;
    std::vector<float *> others((size_t)n_peers);
    for (int g = 0; g < n_peers; ++g) others[(size_t)g] = g == own ? nullptr : d_peers[g];
    HIP_TRY(launch_push_columns(mine, ld_total, N, w->M, others.data(), n_peers, ld_total, col0, (hipStream_t)stream));
    return GGML_HIP_OK;
}

// rms_norm -> mul -> mul_mat [-> add | scale] on device-resident rows.  One launch where the fused mat-vec serves N (the
// prologue computes and quantizes y = (x * rms_scale) * g in-kernel and writes both nodes' results), else the pair kernel of
// fused.hip followed by the mat-mul with its epilogue.  d_x, d_g: [N] rows, ld_x / ld_g elements apart; d_norm, d_y: [N][K].
static bool prologue_is_fused(const ggml_hip_weight *w, int64_t N, const float *d_x, int64_t ld_x, const float *d_g, int64_t ld_g,
                              const float *d_norm, const float *d_y) {
    return is_q(w->type) && w->ext_type == 0 && gemv_fused_has_epilogue(N) && ld_x % 4 == 0 && ld_g % 4 == 0 &&
           (((uintptr_t)d_x | (uintptr_t)d_g | (uintptr_t)d_norm | (uintptr_t)d_y) & 15) == 0;
}

int ggml_hip_norm_mul_mat_fused(const ggml_hip_weight *w, int64_t N) {
    return w && is_q(w->type) && w->ext_type == 0 && gemv_fused_has_epilogue(N) ? 1 : 0;
}

int ggml_hip_norm_mul_mat_dev(const ggml_hip_weight *w, const float *d_x, int64_t ld_x, const float *d_g, int64_t ld_g, int64_t N,
                              float *d_norm, float *d_y, float *d_dst, int64_t ldd, void *d_work, size_t work_bytes, int mode,
                              const float *d_addend, int64_t ld_add, float *d_dst2, int64_t ldd2, float scale, void *stream) {
    if (!w || !d_x || !d_g || !d_norm || !d_y || !d_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (N <= 0 || w->M <= 0) return GGML_HIP_OK;
    if (ld_x < w->K || ld_g < w->K || ldd < w->M) return fail(GGML_HIP_ERR_SHAPE, "row stride smaller than a row");
    if (mode < 0 || mode > 2) return fail(GGML_HIP_ERR_ARG, "epilogue mode %d", mode);
    if (mode == 1 && (!d_addend || !d_dst2 || ld_add < w->M || ldd2 < w->M)) return fail(GGML_HIP_ERR_ARG, "add epilogue: addend / dst2 missing or too narrow");
    int rc = weight_device_current(w);
    if (rc) return rc;
    if (prologue_is_fused(w, N, d_x, ld_x, d_g, ld_g, d_norm, d_y)) {
        const mm_prologue pro = {d_g, ld_g, d_norm, d_y};
        const mm_epilogue ep = {mode, d_addend, ld_add, d_dst2, ldd2, scale};
        HIP_TRY(launch_gemv_q_fused_pro(w, d_x, ld_x, pro, N, d_dst, ldd, (hipStream_t)stream, mode ? &ep : nullptr));
        return GGML_HIP_OK;
    }
    if (ld_x == w->K && ld_g == w->K) HIP_TRY(launch_rms_norm_mul_f32(d_x, d_g, d_norm, d_y, N, w->K, (hipStream_t)stream));
    else for (int64_t n = 0; n < N; ++n) HIP_TRY(launch_rms_norm_mul_f32(d_x + n * ld_x, d_g + n * ld_g, d_norm + n * w->K, d_y + n * w->K, 1, w->K, (hipStream_t)stream));
    return ggml_hip_mul_mat_epilogue_dev(w, d_y, N, w->K, d_dst, ldd, d_work, work_bytes, mode, d_addend, ld_add, d_dst2, ldd2, scale, stream);
}

/* ---- several weight matrices behind ONE activation matrix (q / k / v, gate / up): one launch for N <= 4 ---- */
static bool multi_ok(const ggml_hip_weight *const *w, int n_w, int64_t N) {
    if (!w || n_w < 2 || n_w > 4 || !gemv_fused_has_epilogue(N)) return false;
    for (int i = 0; i < n_w; ++i)
        if (!w[i] || !is_q(w[i]->type) || w[i]->ext_type != 0 || w[i]->type != w[0]->type || w[i]->K != w[0]->K || w[i]->device != w[0]->device || w[i]->M <= 0)
            return false;
    return true;
}
int ggml_hip_mul_mat_multi_fused(const ggml_hip_weight *const *w, int n_w, int64_t N) { return multi_ok(w, n_w, N) ? 1 : 0; }

int ggml_hip_mul_mat_multi_dev(const ggml_hip_weight *const *w, int n_w, const float *d_src1, int64_t ld1, int64_t N, float *const *d_dst,
                               const int64_t *ldd, const float *d_g, int64_t ld_g, float *d_norm, float *d_y, void *stream) {
    if (!w || !d_src1 || !d_dst || !ldd) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (!multi_ok(w, n_w, N)) return fail(GGML_HIP_ERR_SHAPE, "multi mul_mat: 2..4 quantized matrices of one type and K on one device, N <= 4");
    for (int i = 0; i < n_w; ++i)
        if (!d_dst[i] || ldd[i] < w[i]->M) return fail(GGML_HIP_ERR_SHAPE, "multi mul_mat: dst %d missing or its row stride smaller than a row", i);
    if (ld1 < w[0]->K) return fail(GGML_HIP_ERR_SHAPE, "row stride smaller than a row");
    int rc = weight_device_current(w[0]);
    if (rc) return rc;
    rc = check_src1_alignment(d_src1, ld1);
    if (rc) return rc;
    const bool with_pro = d_g != nullptr;
    if (with_pro) {
        if (!d_norm || !d_y || ld_g < w[0]->K) return fail(GGML_HIP_ERR_ARG, "prologue: norm / mul outputs missing or g too narrow");
        if (!prologue_is_fused(w[0], N, d_src1, ld1, d_g, ld_g, d_norm, d_y)) return fail(GGML_HIP_ERR_SHAPE, "prologue: operands must be 16-byte aligned with row strides that are multiples of 4");
    }
    const mm_prologue pro = {d_g, ld_g, d_norm, d_y};
    HIP_TRY(launch_gemv_q_fused_multi(w, n_w, d_src1, ld1, with_pro ? &pro : nullptr, N, d_dst, ldd, (hipStream_t)stream));
    return GGML_HIP_OK;
}

/* the same for any N, with the work buffer a batch needs: src1 is quantized ONCE (the INIT phase, Ggml.cs:6641-6654, is the same for
 * every matrix of one type and K) and the matrices follow -- in one launch where gemm_qmx.hip has the form (5 <= N <= 64), else one
 * COMPUTE after the other behind the shared image.  Every row is bit for bit what ggml_hip_mul_mat_dev gives for that matrix. */
int ggml_hip_mul_mat_multi_work_dev(const ggml_hip_weight *const *w, int n_w, const float *d_src1, int64_t ld1, int64_t N, float *const *d_dst,
                                    const int64_t *ldd, void *d_work, size_t work_bytes, void *stream) {
    if (!w || !d_src1 || !d_dst || !ldd) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (n_w < 1 || n_w > 4) return fail(GGML_HIP_ERR_ARG, "multi mul_mat: 1..4 matrices");
    if (N <= 0) return GGML_HIP_OK;
    for (int i = 0; i < n_w; ++i) {
        if (!w[i] || !d_dst[i]) return fail(GGML_HIP_ERR_ARG, "multi mul_mat: matrix or dst %d missing", i);
        if (!is_q(w[i]->type) || w[i]->type != w[0]->type || w[i]->ext_type != w[0]->ext_type || w[i]->K != w[0]->K || w[i]->device != w[0]->device)
            return fail(GGML_HIP_ERR_SHAPE, "multi mul_mat: quantized matrices of one type and K on one device");
        if (ldd[i] < w[i]->M) return fail(GGML_HIP_ERR_SHAPE, "multi mul_mat: dst %d row stride smaller than a row", i);
    }
    if (n_w >= 2 && multi_ok(w, n_w, N)) return ggml_hip_mul_mat_multi_dev(w, n_w, d_src1, ld1, N, d_dst, ldd, nullptr, 0, nullptr, nullptr, stream);
    // one image for all of them?  (the image kind of a type follows N and K; M enters for shapes beyond the 32-bit offsets only)
    const mm_plan pl0 = weight_plan(w[0], N, false);
    const int kind = weight_image_kind(w[0], N);
    bool shared = pl0.family != MMF_GEMV_ROWS && w[0]->ext_type == 0;
    for (int i = 1; i < n_w && shared; ++i) shared = weight_image_kind(w[i], N) == kind;
    if (!shared) {
        for (int i = 0; i < n_w; ++i) {
            const int rc = ggml_hip_mul_mat_dev(w[i], d_src1, N, ld1, d_dst[i], ldd[i], d_work, work_bytes, stream);
            if (rc) return rc;
        }
        return GGML_HIP_OK;
    }
    int rc = ggml_hip_mul_mat_init_dev(w[0], d_src1, N, ld1, d_work, work_bytes, stream);
    if (rc) return rc;
    if (kind == 3 && n_w >= 2) {
        for (int i = 1; i < n_w; ++i) { rc = weight_device_current(w[i]); if (rc) return rc; }
        const hipError_t e = launch_gemm_qmx_multi(w, n_w, act_carve(d_work, w[0]->K, pad_act(N)), N, d_dst, ldd, (hipStream_t)stream);
        if (e == hipSuccess) return GGML_HIP_OK;
        if (e != hipErrorNotSupported) { (void)hipGetLastError(); return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: %s", hipGetErrorString(e)); }
    }
    if (kind == 0 && n_w >= 2 && pl0.family == MMF_K3S_I8) {
        for (int i = 1; i < n_w; ++i) { rc = weight_device_current(w[i]); if (rc) return rc; }
        const hipError_t e = launch_gemm_q8_small_multi(w, n_w, act_carve(d_work, w[0]->K, pad_act(N)), N, d_dst, ldd, (hipStream_t)stream);
        if (e == hipSuccess) return GGML_HIP_OK;
        if (e != hipErrorNotSupported) { (void)hipGetLastError(); return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: %s", hipGetErrorString(e)); }
    }
    for (int i = 0; i < n_w; ++i) {
        rc = ggml_hip_mul_mat_compute_dev(w[i], N, d_dst[i], ldd[i], d_work, work_bytes, stream);
        if (rc) return rc;
    }
    return GGML_HIP_OK;
}

int ggml_hip_rms_norm_mul_rows_dev(const float *d_x, const float *d_g, float *d_norm, float *d_y, int64_t nrows, int64_t k, void *stream) {
    if (nrows <= 0 || k <= 0) return GGML_HIP_OK;
    if (!d_x || !d_g || !d_norm || !d_y) return fail(GGML_HIP_ERR_ARG, "null argument");
    HIP_TRY(launch_rms_norm_mul_f32(d_x, d_g, d_norm, d_y, nrows, k, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_quantize_rows_dev(int type, const float *d_x, int64_t nrows, int64_t k, void *d_blocks, void *stream) {
    if (nrows <= 0) return GGML_HIP_OK;  // empty input: nothing to do (buffers may be null)
    if (!d_x || !d_blocks) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (is_kquant(type)) {                                      // unpinned extra (kquants.hip, r4)
        if (k % 256 != 0) return fail(GGML_HIP_ERR_SHAPE, "k-quants: k %% 256 != 0");
        if (((uintptr_t)d_x & 15) != 0) return fail(GGML_HIP_ERR_SHAPE, "k-quants: the rows must be 16-byte aligned");
        if (type == GGML_HIP_TYPE_Q6_K) HIP_TRY(launch_quantize_q6k(d_x, nrows, k, d_blocks, (hipStream_t)stream));
        else HIP_TRY(launch_quantize_kq(type, d_x, nrows, k, d_blocks, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
    if (!(wq_ok(type) || type == GGML_TYPE_Q8_1))
        return fail(GGML_HIP_ERR_TYPE, "quantize: unsupported type %d", type);
    if (k % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "k %% 32 != 0 (Ggml.cs:336)");
    HIP_TRY(launch_quantize_rows(type, GGML_TYPE_F32, d_x, k, nrows, k, d_blocks, (hipStream_t)stream));
    return GGML_HIP_OK;
}

int ggml_hip_dequantize_rows_dev(int type, const void *d_blocks, int64_t nrows, int64_t k, float *d_y, void *stream) {
    if (nrows <= 0) return GGML_HIP_OK;
    if (!d_y || !d_blocks) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (is_kquant(type)) {                                      // unpinned extra (kquants.hip)
        if (k % 256 != 0) return fail(GGML_HIP_ERR_SHAPE, "k-quants: k %% 256 != 0");
        if (type == GGML_HIP_TYPE_Q6_K) HIP_TRY(launch_dequantize_q6k(d_blocks, nrows, k, d_y, (hipStream_t)stream));
        else HIP_TRY(launch_dequantize_q5k(type, d_blocks, nrows, k, d_y, (hipStream_t)stream));
        return GGML_HIP_OK;
    }
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
    DeviceCtx *c = call_slot();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    rc = c->make_current();
    if (rc) return rc;
    const size_t xb = (size_t)k * 4, yb = TSIZE[type] * (size_t)(k / BLCK[type]);
    if (c->src1.ensure(xb) || c->dst.ensure(yb)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(c->src1.p, x, xb, hipMemcpyHostToDevice, c->stream));
    rc = ggml_hip_quantize_rows_dev(type, (const float *)c->src1.p, 1, k, c->dst.p, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(y, c->dst.p, yb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GGML_HIP_OK;
}

int ggml_hip_dequantize_row(int type, const void *x, float *y, int k) {
    if (!x || !y || k <= 0) return fail(GGML_HIP_ERR_ARG, "bad argument");
    int rc = ensure_init();
    if (rc) return rc;
    if (type < 0 || type >= GGML_TYPE_COUNT || !is_q(type)) return fail(GGML_HIP_ERR_TYPE, "not a quantized type");
    DeviceCtx *c = call_slot();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    rc = c->make_current();
    if (rc) return rc;
    const size_t yb = (size_t)k * 4, xb = TSIZE[type] * (size_t)(k / BLCK[type]);
    if (c->src1.ensure(xb) || c->dst.ensure(yb)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed");
    HIP_TRY(hipMemcpyAsync(c->src1.p, x, xb, hipMemcpyHostToDevice, c->stream));
    rc = ggml_hip_dequantize_rows_dev(type, c->src1.p, 1, k, (float *)c->dst.p, c->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(y, c->dst.p, yb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GGML_HIP_OK;
}

int ggml_hip_vec_dot(int type, int n, float *s, const void *vx, const void *vy) {
    if (!s || !vx || !vy || n <= 0) return fail(GGML_HIP_ERR_ARG, "bad argument");
    const int vt = (type >= 0 && type < GGML_TYPE_COUNT) ? vec_dot_type(type) : -1;
    if (vt < 0) return fail(GGML_HIP_ERR_TYPE, "vec_dot: type %d has no usable slot (D7/D8)", type);
    if (n % QK != 0) return fail(GGML_HIP_ERR_SHAPE, "n %% 32 != 0 (Ggml.cs:1129)");
    int rc = ensure_init();
    if (rc) return rc;
    DeviceCtx *c = call_slot();
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    ggml_hip_weight *w = nullptr;
    const uint64_t xb = (uint64_t)row_bytes_of(type, n);
    rc = make_weight(c, type, vx, true, n, 1, xb, 0, 1, c->stream, &w);
    if (rc) return rc;
    const size_t yb = TSIZE[vt] * (size_t)(n / QK);
    const size_t wb = ggml_hip_mul_mat_work_size(type, n, 1);
    rc = GGML_HIP_OK;
    if (c->src1.ensure(yb) || c->work.ensure(wb) || c->dst.ensure(4)) rc = fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed");
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(c->src1.p, vy, yb, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_q8_aos_to_planes(vt, c->src1.p, 1, n, act_carve(c->work.p, n, pad_act(1)), c->stream);
        if (e == hipSuccess) rc = ggml_hip_mul_mat_compute_dev(w, 1, (float *)c->dst.p, 1, c->work.p, wb, c->stream);
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(s, c->dst.p, 4, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(c->stream);
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

}  // extern "C"
