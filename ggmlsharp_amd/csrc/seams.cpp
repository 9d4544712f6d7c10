// seams.cpp -- the C-ABI of include/ggml_hip.h, part 2: the host-pointer seams.
//   Seam 1  ggml_compute_forward_mul_mat (Ggml.cs:6714-6744): pipelined host -> device / kernels / device -> host on
//           three streams per device slot, row-split over the slots the caller is not bound away from, graph-scope
//           residency, weight cache with invalidation on every write;
//   its neighbours: cpy -> Q, add (q + f32, f32 + f32), mul, scale, rms_norm, silu;
//   host pool registration (pinned DMA), graph scope, counters.
#include "ctx.h"

#include <memory>

namespace ghip {

// ---------------- host memory registered for DMA ----------------
namespace {
struct HostRange { const uint8_t *p; size_t n; uint8_t *dev; };   // dev: the range's device-visible address (nullptr: not mapped)
std::mutex g_pin_mu;
std::vector<HostRange> g_pinned;
}  // namespace

bool host_range_pinned(const void *p, size_t n) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    const uint8_t *a = (const uint8_t *)p;
    for (const HostRange &r : g_pinned)
        if (a >= r.p && a + n <= r.p + r.n) return true;
    return false;
}

void *host_range_device_ptr(const void *p, size_t n) {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    const uint8_t *a = (const uint8_t *)p;
    for (const HostRange &r : g_pinned)
        if (r.dev && a >= r.p && a + n <= r.p + r.n) return r.dev + (a - r.p);
    return nullptr;
}

namespace {

// ---------------- which slots serve a call ----------------
// A thread bound to a slot (ggml_hip_bind_thread) runs on that slot alone; an unbound caller gets every slot, i.e. the
// reference's row partition (Ggml.cs:6665-6672) for mul_mat and replicated execution for the cheap element-wise nodes.
struct Call {
    std::vector<DeviceCtx *> ctxs;
    std::vector<std::unique_lock<std::recursive_mutex>> locks;
    int begin() {
        int rc = ensure_init();
        if (rc) return rc;
        const int b = bound_slot();
        if (b >= 0) ctxs.push_back(slot(b));
        else for (int i = 0; i < n_slots(); ++i) ctxs.push_back(slot(i));
        for (DeviceCtx *c : ctxs) {
            if (!c) return fail(GGML_HIP_ERR_RUNTIME, "library was shut down during the call");
            locks.emplace_back(c->mu);                 // slot order: no lock-order inversion between callers
            if (c->dead) return fail(GGML_HIP_ERR_RUNTIME, "library was shut down during the call");   // (ggml_hip_shutdown won the race for this slot's lock)
            // another thread is capturing a named scope on this slot's stream: this call's launches are none of that graph's
            // business -- the capture is ended and issued live (its owner carries on live), then this call runs
            if (c->scope_mode == 2 && c->scope_owner != std::this_thread::get_id()) c->scope_dirty();
        }
        return GGML_HIP_OK;
    }
    int G() const { return (int)ctxs.size(); }
    bool in_graph() const { return ctxs[0]->graph_depth_ > 0; }
};

size_t tensor_host_bytes(const ggml_tensor *t) {   // bytes from data to the end of the last element (strided tensors included)
    size_t end = 0;
    const int blck = BLCK[t->type];
    end = (size_t)((t->ne[0] + blck - 1) / blck) * TSIZE[t->type];
    for (int i = 1; i < 4; ++i)
        if (t->ne[i] > 1) end += (size_t)(t->ne[i] - 1) * t->nb[i];
    return end;
}

// every writer of host tensor memory: weight-cache entries built from it and other resident copies overlapping it are stale
void note_host_write(Call &call, const ggml_tensor *dst, bool keep_exact_resident) {
    const size_t bytes = tensor_host_bytes(dst);
    for (DeviceCtx *c : call.ctxs) {
        c->invalidate(dst->data, bytes);
        c->drop_overlapping(dst->data, bytes, keep_exact_resident);
    }
}

// device pointer of a contiguous f32 operand on slot c: its resident copy if an earlier node of this graph produced it,
// else an upload from host memory into `scratch` on the slot's compute stream
int operand_f32(DeviceCtx *c, bool in_graph, const ggml_tensor *t, Scratch &scratch, const float **out) {
    const size_t bytes = (size_t)nelem(t) * 4;
    if (in_graph) {
        const void *r = c->resident_lookup(t->data, bytes);
        if (r) { *out = (const float *)r; ++c->resident_hits; return 0; }
        if (c->before_host_read(t->data, bytes)) return -1;   // host memory may be owed an earlier node's result, or still receiving it
        // the upload stays resident for the rest of the scope: a leaf used by several nodes (the residual stream) moves once
        void *keep = c->resident_buffer(t->data, bytes);
        if (!keep) return -1;
        if (!host_range_pinned(t->data, bytes)) c->scope_dirty();   // a copy from pageable memory is staged by the runtime: not something a capture may hold
        c->note_leaf(t->data, bytes);
        if (hipMemcpyAsync(keep, t->data, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1;
        c->h2d_bytes += bytes;
        *out = (const float *)keep;
        return 0;
    }
    if (scratch.ensure(bytes)) return -1;
    if (hipMemcpyAsync(scratch.p, t->data, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1;
    c->h2d_bytes += bytes;
    *out = (const float *)scratch.p;
    return 0;
}
// device buffer for a contiguous f32 result: kept resident inside a graph scope (in place when dst shares src's data)
float *result_f32(DeviceCtx *c, bool in_graph, const ggml_tensor *t, Scratch &scratch) {
    const size_t bytes = (size_t)nelem(t) * 4;
    if (in_graph) return (float *)c->resident_buffer(t->data, bytes);
    return scratch.ensure(bytes) ? nullptr : (float *)scratch.p;
}
int finish_f32(DeviceCtx *c, bool in_graph, bool to_host, ggml_tensor *t, const float *dev) {
    const size_t bytes = (size_t)nelem(t) * 4;
    if (to_host && in_graph) {
        c->owe(t->data, dev, bytes);        // (dev is the tensor's resident buffer: result_f32)
    } else if (to_host) {
        if (hipMemcpyAsync(t->data, dev, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
        c->d2h_bytes += bytes;
    }
    if (!in_graph && hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    return 0;
}
// the slots an element-wise node runs on: all of them inside a graph scope (every slot then holds every resident
// tensor, so a later row-split mul_mat finds its src1 in its own HBM), the first one otherwise
int eltwise_slots(const Call &call) { return call.in_graph() ? call.G() : 1; }

// Seam-1 pipeline: src1 rows per chunk.  A function of N and K only (never of M or the slot count), so a row shard and the
// unsplit matrix run the same kernel forms on the same chunks: bit-identical results for any split.
#ifndef SEAM_MIN_CHUNK
#define SEAM_MIN_CHUNK 128            // (A/B 256 / 512 rows: 4096 x 4096 x 512 0.349 -> 0.365 / 0.373 ms, x 4096 1.71 -> 1.76 / 1.73: no gain)
#endif
int64_t seam_chunk_rows(int64_t N, int64_t K, bool pinned) {
    if (!pinned) return N;                              // pageable memory: the runtime stages synchronously, nothing overlaps
    const int64_t x_bytes = N * K * 4;
    if (N < 128 || x_bytes < (2ll << 20)) return N;
    // what the pipeline cannot hide is its fill and drain -- the first chunk's upload and the last chunk's download -- but
    // every chunk costs the host ~10 runtime calls and each DMA its launch latency: N / 8 rows per chunk, at least 128
    // (measured on MI355X / PCIe 5, 4096 x 4096 x {4096, 512}: 8 chunks 1.71 / 0.36 ms (two chunks at 512), 16 chunks 1.75 / 0.39 ms;
    // the box moves 94 GB/s with both directions busy: 1.42 ms for the 134 MB of the first shape)
    int64_t rows = ((N + 7) / 8 + 31) / 32 * 32;
    if (rows < SEAM_MIN_CHUNK) rows = SEAM_MIN_CHUNK;
    return rows;
}

// ---------------- the Seam-1 pipeline of one 2-D slice on one slot ----------------
struct PipeArgs {
    const ggml_hip_weight *w;
    const uint8_t *x_host; uint64_t nb11;          // src1 rows on the host (ignored when x is resident)
    uint8_t *d_host; uint64_t nb1;                 // dst rows on the host
    const float *xd; bool upload;                  // src1 on the device (+ whether it has to be uploaded first)
    float *dd; int64_t ldd;                        // dst on the device: base (already at this slot's column) and row stride
    int64_t N, K, Ms, col0, chunk;
    void *work; size_t work_cap;
    // the node that follows the mul_mat, fused into its kernels (common.h mm_epilogue); mode 0 = none
    int epi_mode = 0;
    const float *addend = nullptr;                 // [N][Ms] on the device (mode 1)
    float *dst2 = nullptr; uint8_t *d2_host = nullptr; uint64_t nb1_2 = 0;   // the add node's result, device and host
    float scale = 1.0f;                            // mode 2
    bool owe_dst = false;                          // graph scope, outputs-only: dst is resident and the caller does not want it -- owed, not copied
    // the rms_norm -> mul pair in front of the mul_mat, computed by the mat-vec's prologue (common.h mm_prologue): src1 IS the
    // mul node; x / g are its inputs on the device, pro_n / pro_y receive both nodes' results ([N][K]), n_host / y_host their tensors
    const float *pro_x = nullptr, *pro_g = nullptr;
    float *pro_n = nullptr, *pro_y = nullptr;
    uint8_t *n_host = nullptr, *y_host = nullptr;
};

// H2D of chunk k on s_h2d | INIT + COMPUTE of chunk k on stream | D2H of chunk k on s_d2h, chained by events.
int issue_chunks(DeviceCtx *c, const PipeArgs &a) {
    hipError_t e = hipSuccess;
    // fork: uploads wait for everything issued so far on the compute stream (earlier kernels may still read the src1 scratch)
    e = hipEventRecord(c->ev_compute, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->s_h2d, c->ev_compute, 0);
    int k = 0;
    for (int64_t r = 0; r < a.N && e == hipSuccess; r += a.chunk, ++k) {
        const int64_t n = a.N - r < a.chunk ? a.N - r : a.chunk;
        const int ke = k % PIPE_EVENTS;
        if (a.upload) {
            if (a.nb11 == (uint64_t)a.K * 4)      // contiguous rows: one linear DMA
                e = hipMemcpyAsync((uint8_t *)a.xd + (size_t)r * a.K * 4, a.x_host + (size_t)r * a.nb11, (size_t)n * a.K * 4, hipMemcpyHostToDevice, c->s_h2d);
            else
                e = hipMemcpy2DAsync((uint8_t *)a.xd + (size_t)r * a.K * 4, (size_t)a.K * 4, a.x_host + (size_t)r * a.nb11, a.nb11, (size_t)a.K * 4,
                                     (size_t)n, hipMemcpyHostToDevice, c->s_h2d);
            if (e == hipSuccess) e = hipEventRecord(c->ev_in[ke], c->s_h2d);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_in[ke], 0);
            if (e != hipSuccess) break;
        }
        int rc;
        if (a.pro_x)        // (a prologue call is a single chunk: N <= 4)
            rc = ggml_hip_norm_mul_mat_dev(a.w, a.pro_x, a.K, a.pro_g, a.K, n, a.pro_n, a.pro_y, a.dd, a.ldd, a.work, a.work_cap, a.epi_mode, a.addend, a.Ms,
                                           a.dst2, a.Ms, a.scale, c->stream);
        else
            rc = ggml_hip_mul_mat_epilogue_dev(a.w, a.xd + (size_t)r * a.K, n, a.K, a.dd + (size_t)r * a.ldd, a.ldd, a.work, a.work_cap, a.epi_mode,
                                               a.addend ? a.addend + (size_t)r * a.Ms : nullptr, a.Ms, a.dst2 ? a.dst2 + (size_t)r * a.Ms : nullptr, a.Ms,
                                               a.scale, c->stream);
        if (rc) return rc;
        if (a.owe_dst) continue;
        e = hipEventRecord(c->ev_k[ke], c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->s_d2h, c->ev_k[ke], 0);
        if (e == hipSuccess) {
            if (a.nb1 == (uint64_t)a.Ms * 4 && a.ldd == a.Ms)    // whole contiguous rows (one slot): one linear DMA
                e = hipMemcpyAsync(a.d_host + (size_t)r * a.nb1, a.dd + (size_t)r * a.ldd, (size_t)n * a.Ms * 4, hipMemcpyDeviceToHost, c->s_d2h);
            else
                e = hipMemcpy2DAsync(a.d_host + (size_t)r * a.nb1 + (size_t)a.col0 * 4, a.nb1, a.dd + (size_t)r * a.ldd, (size_t)a.ldd * 4,
                                     (size_t)a.Ms * 4, (size_t)n, hipMemcpyDeviceToHost, c->s_d2h);
            if (e == hipSuccess && a.epi_mode == 1)             // the add node's data goes to the host too (contiguous rows)
                e = hipMemcpyAsync(a.d2_host + (size_t)r * a.nb1_2, a.dst2 + (size_t)r * a.Ms, (size_t)n * a.Ms * 4, hipMemcpyDeviceToHost, c->s_d2h);
            if (e == hipSuccess && a.pro_x) {                    // ... and so do the norm and the mul node's
                e = hipMemcpyAsync(a.n_host, a.pro_n, (size_t)n * a.K * 4, hipMemcpyDeviceToHost, c->s_d2h);
                if (e == hipSuccess) e = hipMemcpyAsync(a.y_host, a.pro_y, (size_t)n * a.K * 4, hipMemcpyDeviceToHost, c->s_d2h);
            }
        }
    }
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "seam 1 pipeline: %s", hipGetErrorString(e));
    return GGML_HIP_OK;
}

// (Capturing a recurring pipeline into a hipGraph and replaying it with one launch was built and measured: the replay runs
// the copy and kernel branches one after the other -- 4096 x 4096 x 4096 3.10 ms against 1.75 ms for the three live
// streams, x 512 0.66 against 0.39 -- so the pipeline is always issued directly.  Results going home by kernel stores through
// the pool's device mapping instead of the DMA engine were measured as well: a copy kernel per chunk 2.08 / 0.45 ms, the
// mat-mat epilogue storing straight into the mapping 1.92 / 0.40 ms, against 1.72 / 0.35 ms -- the DMA engines stay.)
int run_pipeline(DeviceCtx *c, const PipeArgs &a) {
    // order against work issued earlier on the copy streams (they read / write the scratch buffers this call reuses)
    hipError_t e = hipEventRecord(c->ev_d2h, c->s_d2h);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_d2h, 0);
    if (e == hipSuccess) e = hipEventRecord(c->ev_xchg, c->s_h2d);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_xchg, 0);
    if (e != hipSuccess) return fail(GGML_HIP_ERR_RUNTIME, "seam 1 pipeline: %s", hipGetErrorString(e));
    if (a.upload) c->h2d_bytes += (size_t)a.N * a.K * 4;
    c->d2h_busy = true;
    c->d2h_bytes += (size_t)a.N * a.Ms * 4 * (a.epi_mode == 1 ? 2 : 1) + (a.pro_x ? (size_t)a.N * a.K * 8 : 0);
    return issue_chunks(c, a);
}

}  // namespace
}  // namespace ghip

using namespace ghip;

extern "C" {

int ggml_hip_register_host_pool(void *ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(GGML_HIP_ERR_ARG, "null pool");
    int rc = ensure_init();
    if (rc) return rc;
    rc = slot(0)->make_current();
    if (rc) return rc;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (const HostRange &r : g_pinned)
            if (r.p == (const uint8_t *)ptr) return r.n == bytes ? GGML_HIP_OK : fail(GGML_HIP_ERR_ARG, "pool already registered with another size");
    }
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(GGML_HIP_ERR_RUNTIME, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    // the device-visible address of the pool: graph scopes write their results through it (DeviceCtx::pay).  One slot only:
    // with several devices every one would need its own mapping, and their element-wise results come from slot 0 anyway.
    void *dev = nullptr;
    if (n_slots() != 1 || hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess) { (void)hipGetLastError(); dev = nullptr; }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.push_back(HostRange{(const uint8_t *)ptr, bytes, (uint8_t *)dev});
    return GGML_HIP_OK;
}

int ggml_hip_unregister_host_pool(void *ptr) {
    if (!ptr) return fail(GGML_HIP_ERR_ARG, "null pool");
    bool found = false;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].p == (const uint8_t *)ptr) { bytes = g_pinned[i].n; g_pinned.erase(g_pinned.begin() + (long)i); found = true; break; }
    }
    if (!found) return GGML_HIP_OK;                      // never registered (no device at ggml_init time): nothing to undo
    // no DMA may still target the pool, and nothing cached from it may outlive it
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        if (c->make_current() == GGML_HIP_OK) (void)c->pay_and_sync();
        c->drop_captured();                             // captured scopes hold the pool's device mapping
        c->invalidate(ptr, bytes);
        c->drop_overlapping(ptr, bytes, false);
    }
    hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(GGML_HIP_ERR_RUNTIME, "hipHostUnregister: %s", hipGetErrorString(e)); }
    return GGML_HIP_OK;
}

void ggml_hip_invalidate(const void *host_ptr) {
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        (void)c->make_current();
        c->invalidate(host_ptr, 1);
        c->drop_overlapping(host_ptr, 1, false);
    }
}

void ggml_hip_invalidate_range(const void *host_ptr, size_t bytes) {
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        (void)c->make_current();
        c->invalidate(host_ptr, bytes);
        c->drop_overlapping(host_ptr, bytes, false);
    }
}

/* A CPU node inside a graph scope is about to READ host memory [host_ptr, host_ptr + bytes) (a source produced by an
 * offloaded node): whatever the scope still owes that range is copied and waited for.  Cheap when nothing is owed. */
int ggml_hip_host_read(const void *host_ptr, size_t bytes) {
    if (!host_ptr || bytes == 0) return GGML_HIP_OK;
    int rc = GGML_HIP_OK;
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        if (c->owed.empty() && !c->d2h_busy) continue;
        int r = c->make_current();
        if (!r) r = c->before_host_read(host_ptr, bytes);
        if (r && !rc) rc = r;
    }
    return rc;
}

void ggml_hip_invalidate_all(void) {
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        (void)c->make_current();
        (void)c->sync_all();
        c->free_cache();
    }
}

// (Round 3 replayed short scopes -- up to eight kernel nodes in one chain -- by re-issuing the captured graph's kernel nodes with
// hipLaunchKernel on the graph's own argument blocks: 75.1 against 77.3 us for a one-layer decode graph.  It leaned on the lifetime of the
// pointers hipGraphKernelNodeGetParams hands out, which HIP does not promise across hipGraphExec updates; 2 us are not worth that
// (VERDICT r3).  r4: a captured scope is always replayed through hipGraphLaunch.)

/* Graph scope for ggml_graph_compute's node loop (Ggml.cs:3539-3704): see ctx.h "graph scope". */
static int graph_begin(uint64_t key) {
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    for (DeviceCtx *c : call.ctxs) ++c->graph_depth_;
    DeviceCtx *c = call.ctxs[0];
    if (key == 0 || call.G() != 1 || c->graph_depth_ != 1) return GGML_HIP_OK;
    rc = c->make_current();
    if (rc) { --c->graph_depth_; return rc; }            // (no scope was opened)
    c->scope_key = key;
    c->scope_owner = std::this_thread::get_id();
    c->scope_clean = true;
    c->scope_lost = false;
    c->scope_leaves.clear();
    c->scope_mode = 1;
    auto it = c->captured.find(key);
    if (it == c->captured.end()) {                                   // first sight: observe
        if (c->captured.size() >= 256) c->drop_captured();           // (a host that never repeats a graph: do not collect its keys)
        return GGML_HIP_OK;
    }
    DeviceCtx::Captured &e = it->second;
    if (e.refused) { c->scope_mode = 0; return GGML_HIP_OK; }
    if (!e.exec && e.seen == 0) return GGML_HIP_OK;                  // not observed clean yet: keep observing
    if (e.exec) {
        if (e.scratch_sig == c->scratch_sig()) { c->scope_mode = 3; return GGML_HIP_OK; }
        // a scratch buffer was reallocated since: the captured launches hold stale pointers
        (void)hipGraphExecDestroy(e.exec); (void)hipGraphDestroy(e.graph);
        for (Resident &r : e.buffers) c->pool.push_back(r);
        c->captured.erase(it);
        return GGML_HIP_OK;
    }
    // observed clean before: capture this run.  Relaxed mode: other threads' HIP calls are none of the capture's business, and a
    // foreign thread whose seam arrives on this slot (Call::begin) must be able to END the capture -- in the other modes only the
    // thread that began a capture may end it (hipErrorStreamCaptureWrongThread), and the stream would stay capturing.
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) c->scope_mode = 2;
    else { (void)hipGetLastError(); e.refused = true; c->scope_mode = 0; return GGML_HIP_OK; }
    // the leaves the scope uploaded one by one when it was observed come down TOGETHER, as the first node of the captured
    // graph (one launch through the device mapping of the pool instead of a copy node each); the seams then find them resident
    std::vector<const void *> src; std::vector<void *> dst; std::vector<size_t> nb;
    for (const auto &lf : e.leaves) {
        if (lf.second % 4 != 0 || (uintptr_t)lf.first % 4 != 0 || lf.second > (8u << 20)) continue;
        const void *m = host_range_device_ptr(lf.first, lf.second);
        if (!m || c->resident_lookup(lf.first, lf.second)) continue;
        void *d = c->resident_buffer(lf.first, lf.second);
        if (!d) break;
        src.push_back(m); dst.push_back(d); nb.push_back(lf.second);
        c->h2d_bytes += lf.second;
    }
    // (the launches are good live as well: should one of the buffer requests above have ended the capture -- an allocation --
    // the leaves still come down; a leaf whose copy could not be issued loses its resident entry again and is uploaded by the
    // node that reads it)
    for (size_t i = 0; i < src.size(); i += 32) {
        const int n = (int)(src.size() - i < 32 ? src.size() - i : 32);
        if (launch_scatter_copy(src.data() + i, dst.data() + i, nb.data() + i, n, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            c->scope_dirty();
            for (size_t k = i; k < src.size(); ++k)
                for (auto it = c->resident.begin(); it != c->resident.end(); ++it)
                    if (it->second.p == dst[k]) { c->pool.push_back(it->second); c->resident.erase(it); break; }
            break;
        }
    }
    return GGML_HIP_OK;
}
int ggml_hip_graph_begin(void) { return graph_begin(0); }
int ggml_hip_graph_begin_keyed(uint64_t key) { return graph_begin(key); }
int ggml_hip_graph_end(void) {
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    if (call.ctxs[0]->graph_depth_ <= 0) return fail(GGML_HIP_ERR_ARG, "ggml_hip_graph_end without ggml_hip_graph_begin");
    for (DeviceCtx *c : call.ctxs) {
        int r = c->make_current();
        const bool outer = c->graph_depth_ == 1;
        if (outer && c->outputs_only) {                  // the caller asked for these tensors only (ggml_hip_graph_outputs)
            std::vector<DeviceCtx::Owed> keep;
            for (const DeviceCtx::Owed &o : c->owed) if (c->wanted(o.host)) keep.push_back(o);
            c->owed.swap(keep);
        }
        const int mode = outer ? c->scope_mode : 0;
        if (!r && mode == 3) {                           // replay: the whole scope is one launch
            DeviceCtx::Captured &e = c->captured[c->scope_key];
            c->scope_mode = 0;
            ++c->n_replayed;
            const hipError_t he = hipGraphLaunch(e.exec, c->stream);
            if (he != hipSuccess) r = fail(GGML_HIP_ERR_RUNTIME, "replay of a captured scope: %s", hipGetErrorString(he));
            if (!r) r = c->sync_all();
        } else if (!r && mode == 2) {                    // capture: close it, keep it, run it
            r = c->pay();                                // (captured like everything else of the scope)
            c->scope_mode = 0;
            DeviceCtx::Captured &e = c->captured[c->scope_key];
            hipGraph_t g = nullptr;
            hipError_t he = hipStreamEndCapture(c->stream, &g);
            if (he == hipSuccess && g) he = hipGraphInstantiate(&e.exec, g, nullptr, nullptr, 0);
            if (he != hipSuccess || !g) {
                (void)hipGetLastError();
                if (g) (void)hipGraphDestroy(g);
                e.exec = nullptr; e.refused = true; ++c->n_refused;
                if (!r) r = fail(GGML_HIP_ERR_RUNTIME, "graph scope capture failed (%s): the scope's nodes did not run", hipGetErrorString(he));
            } else {
                e.graph = g;
                ++c->n_captured;
                e.scratch_sig = c->scratch_sig();
                for (auto &kv : c->resident) e.buffers.push_back(kv.second);     // owned by the entry from here on
                c->resident.clear();
                he = hipGraphLaunch(e.exec, c->stream);
                if (he != hipSuccess && !r) r = fail(GGML_HIP_ERR_RUNTIME, "hipGraphLaunch: %s", hipGetErrorString(he));
                const int r2 = c->sync_all();
                if (!r) r = r2;
            }
        } else if (!r) {
            const bool clean = c->scope_clean;           // (the wait below is the scope's end, not part of it)
            if (mode == 1) c->scope_mode = 0;
            r = c->pay_and_sync();                       // every node's dst is on the host from here on
            if (mode == 1) {
                DeviceCtx::Captured &e = c->captured[c->scope_key];
                if (clean && !r) { e.seen = 1; e.leaves = c->scope_leaves; ++c->n_observed; }     // (a scope that was not clean is simply observed again next time)
            }
        }
        if (outer && c->scope_lost) {                    // (DeviceCtx::scope_dirty could not end a capture of this scope)
            c->scope_lost = false;
            if (!r) r = fail(GGML_HIP_ERR_RUNTIME, "a capture of this graph scope could not be ended: nodes issued before that point did not run");
        }
        if (outer) { c->scope_mode = 0; c->outputs_only = false; c->outputs.clear(); }
        if (r && !rc) rc = r;
        if (--c->graph_depth_ == 0) c->drain(false);
    }
    return rc;
}
/* Opt-in, and a DEVIATION from the reference's contract: inside the open graph scope, declare that only these tensors (by
 * data pointer) have to be in host memory when the scope ends.  Every other result stays on the device: its host memory is
 * NOT updated (the reference leaves every node's data there).  What the library itself needs on the host is still copied. */
int ggml_hip_graph_outputs(const void *const *host_ptrs, int n) {
    if (n < 0 || (n > 0 && !host_ptrs)) return fail(GGML_HIP_ERR_ARG, "bad argument");
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    if (!call.in_graph()) return fail(GGML_HIP_ERR_ARG, "ggml_hip_graph_outputs outside a graph scope");
    for (DeviceCtx *c : call.ctxs) {
        c->outputs_only = true;
        c->outputs.assign(host_ptrs, host_ptrs + n);
    }
    return GGML_HIP_OK;
}
void ggml_hip_debug_scope_counters(uint64_t *observed, uint64_t *captured, uint64_t *replayed, uint64_t *refused) {
    uint64_t v[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        v[0] += c->n_observed; v[1] += c->n_captured; v[2] += c->n_replayed; v[3] += c->n_refused;
    }
    if (observed) *observed = v[0];
    if (captured) *captured = v[1];
    if (replayed) *replayed = v[2];
    if (refused) *refused = v[3];
}
/* The Seam-1 weight cache: a budget per slot in bytes (0 = the device's memory is the limit), and what the cache holds / has evicted so far. */
static std::atomic<size_t> g_weight_cache_budget{0};
static size_t weight_cache_budget() { return g_weight_cache_budget.load(std::memory_order_relaxed); }
void ggml_hip_debug_weight_cache_budget(size_t bytes_per_slot) { g_weight_cache_budget.store(bytes_per_slot, std::memory_order_relaxed); }
void ggml_hip_debug_weight_cache_stats(uint64_t *entries, uint64_t *bytes, uint64_t *evictions) {
    uint64_t n = 0, b = 0, e = 0;
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        n += c->cache.size(); b += c->cache_bytes; e += c->cache_evictions;
    }
    if (entries) *entries = n;
    if (bytes) *bytes = b;
    if (evictions) *evictions = e;
}
void ggml_hip_debug_transfer_counters(uint64_t *h2d_bytes, uint64_t *d2h_bytes, uint64_t *resident_hits) {
    uint64_t a = 0, b = 0, h = 0;
    for (int i = 0; i < n_slots(); ++i) {
        DeviceCtx *c = slot(i);
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        a += c->h2d_bytes; b += c->d2h_bytes; h += c->resident_hits;
    }
    if (h2d_bytes) *h2d_bytes = a;
    if (d2h_bytes) *d2h_bytes = b;
    if (resident_hits) *resident_hits = h;
}

/* Seam 1.  Checks mirror the Debug.Asserts of the three drivers (Ggml.cs:6026-6046, 6222-6241, 6477-6504) and of
 * ggml_mul_mat_impl (Ggml.cs:8228-8229); the reference silently drops them in Release, here they are errors. */
// inside a replayed graph scope the seams have nothing to do: ggml_hip_graph_end launches the captured scope
static bool scope_replaying() {
    const int b = bound_slot();
    DeviceCtx *c = slot(b >= 0 ? b : 0);
    if (!c) return false;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    return !c->dead && c->scope_mode == 3 && c->scope_owner == std::this_thread::get_id();
}

static bool src1_contig_early(const ggml_tensor *t) { return t->nb[0] == 4 && t->nb[1] == (uint64_t)t->ne[0] * 4; }

struct SeamEpi {                                    // the node after the mul_mat (fused seams below); mode as in mm_epilogue
    int mode;
    const struct ggml_tensor *addend;               // mode 1: the other operand of the add node
    struct ggml_tensor *add_dst;                    //         and that node
    float scale;                                    // mode 2
    // prologue (rms_norm -> mul in front): src1 of the call is the mul node; these are the pair's operands and the norm node
    const struct ggml_tensor *pro_x = nullptr, *pro_g = nullptr;
    struct ggml_tensor *pro_norm = nullptr;
};
static const int SEAM_NOT_FUSABLE = 1;              // (internal) the caller runs the two seams one after the other

static int seam1(const struct ggml_compute_params *params, const struct ggml_tensor *src0, const struct ggml_tensor *src1,
                 struct ggml_tensor *dst, const SeamEpi *epi) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    // offload convention of the reference's own dead GPU blocks (Ggml.cs:6510-6521)
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
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

    Call call;
    int rc = call.begin();
    if (rc) return rc;
    const int G = call.G();
    const bool in_graph = call.in_graph();
    const int64_t nslice = ne02 * ne03;
    if (epi) {
        // fused when one slot serves one 2-D slice with contiguous f32 tensors of dst's shape; anything else: two seams
        bool ok = G == 1 && nslice == 1 && dst->nb[1] == (uint64_t)ne01 * 4;
        if (ok && epi->mode == 1) {
            const ggml_tensor *a = epi->addend, *d2 = epi->add_dst;
            ok = a && d2 && a->data && d2->data && contiguous_f32(a) && contiguous_f32(d2) && a->ne[0] == ne01 && a->ne[1] == ne11 &&
                 a->ne[2] == 1 && a->ne[3] == 1 && d2->ne[0] == ne01 && d2->ne[1] == ne11 && d2->ne[2] == 1 && d2->ne[3] == 1 && d2->data != dst->data;
        }
        if (ok && epi->pro_x) {    // the prologue form: the fused mat-vec only (N <= 4), contiguous [K, N] operands
            const ggml_tensor *ts[3] = {epi->pro_x, epi->pro_g, epi->pro_norm};
            ok = ne11 <= 4 && is_q(type) && src1_contig_early(src1) && contiguous_f32(src1);
            for (const ggml_tensor *t : ts)
                ok = ok && t && t->data && contiguous_f32(t) && t->ne[0] == ne10 && t->ne[1] == ne11 && t->ne[2] == 1 && t->ne[3] == 1;
            ok = ok && epi->pro_norm->data != src1->data && epi->pro_norm->data != dst->data;
        }
        if (!ok) return SEAM_NOT_FUSABLE;
    }
    const size_t x_bytes = (size_t)ne11 * ne10 * 4, d_full = (size_t)ne11 * ne01 * 4;
    const size_t w_bytes = ggml_hip_mul_mat_work_size(type, ne00, ne11);
    const bool src1_contig = src1->nb[1] == (uint64_t)ne10 * 4 && src1->nb[2] == src1->nb[1] * (uint64_t)ne11 &&
                             src1->nb[3] == src1->nb[2] * (uint64_t)ne12;
    const bool dst_contig = dst->nb[1] == (uint64_t)ne01 * 4 && dst->nb[2] == dst->nb[1] * (uint64_t)ne11 &&
                            dst->nb[3] == dst->nb[2] * (uint64_t)ne02;
    const bool pinned = host_range_pinned(src1->data, tensor_host_bytes(src1)) && host_range_pinned(dst->data, tensor_host_bytes(dst));
    const int64_t chunk = seam_chunk_rows(ne11, ne10, pinned);
    // a src0 that some node computes (or that is a leaf the host rewrites between computes without telling us) must not be
    // served from the cache: only leaves are cached, and every seam that writes host memory invalidates what overlaps it
    const bool cacheable = src0->op == GGML_OP_NONE;

    // dst is about to be written: stale weight-cache entries / resident copies of anything overlapping it go first
    note_host_write(call, dst, true);

    std::vector<std::vector<ggml_hip_weight *>> W((size_t)G);       // [slot][slice]
    std::vector<ggml_hip_weight *> to_free;                         // transient weights of a call outside a graph scope
    std::vector<const uint8_t *> x_res((size_t)G, nullptr);
    std::vector<uint8_t *> d_res((size_t)G, nullptr);
    std::vector<int64_t> r0((size_t)G), r1((size_t)G);
    for (int g = 0; g < G; ++g) {
        DeviceCtx *c = call.ctxs[(size_t)g];
        rc = c->make_current();
        if (rc) return rc;
        shard_rows(ne01, G, g, &r0[(size_t)g], &r1[(size_t)g]);
        const CacheKey key{src0->data, type, ne00, ne01, ne02, ne03, src0->nb[1], src0->nb[2], src0->nb[3], r0[(size_t)g], r1[(size_t)g]};
        auto it = cacheable ? c->cache.find(key) : c->cache.end();
        if (it != c->cache.end()) {
            W[(size_t)g] = it->second.slices;
            it->second.last_use = ++c->cache_clock;
        } else {
            // the source rows: a device copy left by the node that computed them (graph scope), else host memory -- which an
            // earlier node's device -> host copy may still be filling, so wait for this slot's streams first
            const uint8_t *dev_src = nullptr;
            if (!cacheable && in_graph && ggml_hip_type_size(type) == src0->nb[0] && (type == GGML_TYPE_F32) && contiguous_f32(src0))
                dev_src = (const uint8_t *)c->resident_lookup(src0->data, (size_t)nelem(src0) * 4);
            if (!dev_src && in_graph) { rc = c->pay_and_sync(); if (rc) return rc; }
            c->scope_dirty();                             // a weight upload (allocation, staging, a wait) is not replayable
            std::vector<ggml_hip_weight *> slices;
            // r5: the cache is bounded by the device's memory (and by the budget of ggml_hip_debug_weight_cache_budget, a test hook): when an
            // upload cannot be allocated, the least recently used leaf entries go and the upload is tried ONCE more (it used
            // to be an error return for good -- resident images are 1.5-2.6 x the file format)
            for (int attempt = 0; attempt < 2; ++attempt) {
                rc = GGML_HIP_OK;
                for (int64_t i03 = 0; i03 < ne03 && !rc; ++i03)
                    for (int64_t i02 = 0; i02 < ne02 && !rc; ++i02) {
                        ggml_hip_weight *w = nullptr;
                        const size_t off = (size_t)(i02 * src0->nb[2] + i03 * src0->nb[3]);
                        rc = make_weight(c, type, (dev_src ? dev_src : (const uint8_t *)src0->data) + off, dev_src == nullptr, ne00, ne01,
                                         src0->nb[1], r0[(size_t)g], r1[(size_t)g], c->stream, &w);
                        if (!rc) slices.push_back(w);
                    }
                if (rc != GGML_HIP_ERR_RUNTIME || attempt == 1 || c->cache.empty()) break;
                size_t have = 0;
                for (ggml_hip_weight *x : slices) { have += x->bytes; ggml_hip_weight_free(x); }
                slices.clear();
                (void)hipGetLastError();
                // (how much the failed upload wanted is not known here: at least what its slices so far took, and no less than a quarter of the cache)
                const size_t want = have > c->cache_bytes / 4 ? have : c->cache_bytes / 4;
                if (c->evict_lru(want ? want : 1, 0, nullptr) == 0) break;
            }
            if (rc) {
                for (ggml_hip_weight *x : slices) ggml_hip_weight_free(x);
                for (ggml_hip_weight *x : to_free) ggml_hip_weight_free(x);
                return rc;
            }
            W[(size_t)g] = slices;
            if (cacheable) {
                CachedWeight cw;
                cw.host = src0->data; cw.host_bytes = tensor_host_bytes(src0); cw.slices = slices;
                cw.row_begin = r0[(size_t)g]; cw.row_end = r1[(size_t)g];
                cw.last_use = ++c->cache_clock;
                for (ggml_hip_weight *x : slices) cw.dev_bytes += x->bytes;
                c->cache_bytes += cw.dev_bytes;
                c->cache.emplace(key, std::move(cw));
                const size_t budget = weight_cache_budget();
                if (budget && c->cache_bytes > budget) c->evict_lru(0, budget, &key);   // (the entry this call runs on stays, whatever its size)
            } else if (in_graph) {
                for (ggml_hip_weight *x : slices) c->transient.push_back(x);     // alive until graph end (kernels in flight)
            } else {
                for (ggml_hip_weight *x : slices) to_free.push_back(x);
            }
        }
        // graph scope: is src1 the (contiguous) dst of an earlier offloaded node?  will dst be kept?
        if (in_graph && src1_contig) {
            x_res[(size_t)g] = (const uint8_t *)c->resident_lookup(src1->data, x_bytes * (size_t)nslice);
            if (x_res[(size_t)g]) ++c->resident_hits;
        }
        if (in_graph && dst_contig) {
            d_res[(size_t)g] = (uint8_t *)c->resident_buffer(dst->data, d_full * (size_t)nslice);
            if (!d_res[(size_t)g]) rc = fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for a resident dst");
        }
        const int64_t Ms = r1[(size_t)g] - r0[(size_t)g];
        if (!rc && ((!x_res[(size_t)g] && c->src1.ensure(x_bytes)) || (!d_res[(size_t)g] && c->dst.ensure((size_t)ne11 * (size_t)(Ms > 0 ? Ms : 1) * 4)) ||
                    c->work.ensure(w_bytes ? w_bytes : 16)))
            rc = fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
        if (rc) { for (ggml_hip_weight *x : to_free) ggml_hip_weight_free(x); return rc; }
    }

    const float *epi_addend = nullptr;
    float *epi_dst2 = nullptr;
    if (epi && epi->mode == 1) {                   // (G == 1) the add node's other operand and its result buffer
        DeviceCtx *c = call.ctxs[0];
        if (operand_f32(c, in_graph, epi->addend, c->stage, &epi_addend)) rc = fail(GGML_HIP_ERR_RUNTIME, "mul_mat + add: operand staging failed");
        if (!rc) {
            note_host_write(call, epi->add_dst, true);
            epi_dst2 = result_f32(c, in_graph, epi->add_dst, c->dst2);
            if (!epi_dst2) rc = fail(GGML_HIP_ERR_RUNTIME, "mul_mat + add: hipMalloc failed");
        }
        if (rc) { for (ggml_hip_weight *x : to_free) ggml_hip_weight_free(x); return rc; }
    }
    const float *pro_x = nullptr, *pro_g = nullptr;
    float *pro_n = nullptr, *pro_y = nullptr;
    if (epi && epi->pro_x) {                        // (G == 1) the pair's operands and both nodes' result buffers
        DeviceCtx *c = call.ctxs[0];
        if (operand_f32(c, in_graph, epi->pro_x, c->aux[0], &pro_x) || operand_f32(c, in_graph, epi->pro_g, c->aux[1], &pro_g))
            rc = fail(GGML_HIP_ERR_RUNTIME, "norm + mul + mul_mat: operand staging failed");
        if (!rc) {
            note_host_write(call, epi->pro_norm, true);
            note_host_write(call, src1, true);
            pro_n = result_f32(c, in_graph, epi->pro_norm, c->aux[2]);
            pro_y = result_f32(c, in_graph, src1, c->aux[3]);
            if (!pro_n || !pro_y) rc = fail(GGML_HIP_ERR_RUNTIME, "norm + mul + mul_mat: hipMalloc failed");
        }
        if (rc) { for (ggml_hip_weight *x : to_free) ggml_hip_weight_free(x); return rc; }
        x_res[0] = nullptr;                         // src1 is produced by this call, not looked up
    }
    hipError_t e = hipSuccess;
    for (int64_t i03 = 0; i03 < ne03 && !rc && e == hipSuccess; ++i03)
        for (int64_t i02 = 0; i02 < ne02 && !rc && e == hipSuccess; ++i02) {  // slice offsets as in Ggml.cs:6566-6570
            const int64_t sl = i03 * ne02 + i02;
            const uint8_t *x_host = (const uint8_t *)src1->data + i02 * src1->nb[2] + i03 * src1->nb[3];
            uint8_t *d_host = (uint8_t *)dst->data + i02 * dst->nb[2] + i03 * dst->nb[3];
            for (int g = 0; g < G && !rc && e == hipSuccess; ++g) {
                DeviceCtx *c = call.ctxs[(size_t)g];
                const int64_t Ms = r1[(size_t)g] - r0[(size_t)g];
                if (Ms <= 0) continue;
                rc = c->make_current();
                if (rc) break;
                const ggml_hip_weight *w = W[(size_t)g][(size_t)sl];
                const float *xd = x_res[(size_t)g] ? (const float *)(x_res[(size_t)g] + (size_t)sl * x_bytes) : (const float *)c->src1.p;
                // device dst: this slot's columns of the resident [N][M] copy, or a [N][Ms] scratch shard
                float *dd = d_res[(size_t)g] ? (float *)(d_res[(size_t)g] + (size_t)sl * d_full) + r0[(size_t)g] : (float *)c->dst.p;
                const int64_t ldd = d_res[(size_t)g] ? ne01 : Ms;
                if (!x_res[(size_t)g] && in_graph && !(epi && epi->pro_x)) {
                    // src1 comes from host memory: inside a graph scope an earlier node's device -> host copy into that
                    // very memory may still be in flight
                    rc = c->before_host_read(x_host, (size_t)(ne11 - 1) * src1->nb[1] + (size_t)ne10 * 4);
                    if (rc) break;
                }
                // Graph scope, one slot, one chunk, dst kept resident: the node is its kernels on the compute stream and
                // nothing else -- no fork / join over the copy streams, no device -> host copy of its own (DeviceCtx::owe:
                // the host copies of a scope's results go out together).  17 nodes of a 7B decoder layer at batch 1:
                // 439 -> see DESIGN 8 us per graph compute.
                // (a batch of any size when src1 is already in HBM: there is no upload for chunks to overlap; the result's way
                // home overlaps the NEXT nodes' kernels -- DeviceCtx::owe sends a large one at once on the copy stream)
                if (in_graph && G == 1 && d_res[0] && (x_res[0] || (epi && epi->pro_x) || (ne11 <= chunk && src1->nb[1] == (uint64_t)ne10 * 4))) {
                    const bool pro = epi && epi->pro_x;
                    if (!x_res[0] && !pro) {
                        if (!pinned) c->scope_dirty();
                        if (nslice == 1) c->note_leaf(src1->data, (size_t)ne11 * ne10 * 4);
                        e = hipMemcpyAsync(c->src1.p, x_host, (size_t)ne11 * ne10 * 4, hipMemcpyHostToDevice, c->stream);
                        if (e != hipSuccess) break;
                        c->h2d_bytes += (size_t)ne11 * ne10 * 4;
                    }
                    const int mode = epi ? epi->mode : 0;
                    if (pro)
                        rc = ggml_hip_norm_mul_mat_dev(w, pro_x, ne10, pro_g, ne10, ne11, pro_n, pro_y, dd, ldd, c->work.p, c->work.cap, mode, epi_addend, Ms,
                                                       epi_dst2, Ms, epi->scale, c->stream);
                    else
                        rc = ggml_hip_mul_mat_epilogue_dev(w, xd, ne11, ne10, dd, ldd, c->work.p, c->work.cap, mode, epi_addend, Ms, epi_dst2, Ms,
                                                           epi ? epi->scale : 1.0f, c->stream);
                    if (rc) break;
                    c->owe(d_host, dd, (size_t)ne11 * ne01 * 4);
                    if (mode == 1) c->owe(epi->add_dst->data, epi_dst2, (size_t)ne11 * ne01 * 4);
                    if (pro) {
                        c->owe(epi->pro_norm->data, pro_n, (size_t)ne11 * ne10 * 4);
                        c->owe(src1->data, pro_y, (size_t)ne11 * ne10 * 4);
                    }
                    continue;
                }
                c->scope_dirty();                         // the three-stream pipeline is issued live
                PipeArgs pa;
                pa.w = w; pa.x_host = x_host; pa.nb11 = src1->nb[1]; pa.d_host = d_host; pa.nb1 = dst->nb[1];
                pa.xd = xd; pa.upload = !x_res[(size_t)g]; pa.dd = dd; pa.ldd = ldd;
                pa.N = ne11; pa.K = ne10; pa.Ms = Ms; pa.col0 = r0[(size_t)g]; pa.chunk = chunk;
                pa.work = c->work.p; pa.work_cap = c->work.cap;
                if (epi && epi->pro_x) {
                    pa.upload = false;
                    pa.pro_x = pro_x; pa.pro_g = pro_g; pa.pro_n = pro_n; pa.pro_y = pro_y;
                    pa.n_host = (uint8_t *)epi->pro_norm->data; pa.y_host = (uint8_t *)src1->data;
                }
                if (in_graph && G == 1 && d_res[0] && !epi && !c->wanted(dst->data) && dst_contig) {
                    pa.owe_dst = true;                     // (kept for the library's own needs, dropped at scope end)
                    c->owe(d_host, dd, (size_t)ne11 * ne01 * 4);
                }
                if (epi) {
                    pa.epi_mode = epi->mode; pa.scale = epi->scale;
                    if (epi->mode == 1) { pa.addend = epi_addend; pa.dst2 = epi_dst2; pa.d2_host = (uint8_t *)epi->add_dst->data; pa.nb1_2 = epi->add_dst->nb[1]; }
                }
                rc = run_pipeline(c, pa);
            }
            // row split inside a graph scope: every slot's resident copy of dst gets the other slots' columns, so the next
            // node finds its operand whole in its own HBM (peer DMA over xGMI, or an in-process RCCL all-gather: multi.cpp)
            if (!rc && e == hipSuccess && G > 1 && d_res[0]) {
                std::vector<float *> bufs((size_t)G);
                for (int g = 0; g < G; ++g) bufs[(size_t)g] = (float *)(d_res[(size_t)g] + (size_t)sl * d_full);
                rc = exchange_columns(G, call.ctxs.data(), bufs.data(), ne11, ne01, r0.data(), r1.data());
            }
        }
    if (e != hipSuccess && !rc) rc = fail(GGML_HIP_ERR_RUNTIME, "seam 1: %s", hipGetErrorString(e));
    // outside a graph scope the call returns with dst on the host; inside, ggml_hip_graph_end waits once
    if (!in_graph || rc || !to_free.empty())
        for (DeviceCtx *c : call.ctxs) {
            int r = c->make_current();
            if (!r) r = c->sync_all();
            if (r && !rc) rc = r;
        }
    for (ggml_hip_weight *x : to_free) ggml_hip_weight_free(x);
    return rc;
}

int ggml_hip_compute_forward_mul_mat(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                     const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    return seam1(params, src0, src1, dst, nullptr);
}

/* mul_mat node + the ADD node that consumes it (epilogue side): mm_dst = mul_mat(src0, src1),
 * add_dst = mm_dst + addend, the add applied in the store phase of the mat-mul kernels where they have the form
 * (ggml_hip_mul_mat_epilogue_fused), by the add kernel behind them otherwise; both nodes' data reach host memory. */
int ggml_hip_compute_forward_mul_mat_add(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                         const struct ggml_tensor *src1, struct ggml_tensor *mm_dst,
                                         const struct ggml_tensor *addend, struct ggml_tensor *add_dst) {
    if (!params || !src0 || !src1 || !mm_dst || !addend || !add_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    const SeamEpi epi = {1, addend, add_dst, 1.0f};
    int rc = seam1(params, src0, src1, mm_dst, &epi);
    if (rc != SEAM_NOT_FUSABLE) return rc;
    rc = seam1(params, src0, src1, mm_dst, nullptr);
    if (rc) return rc;
    return ggml_hip_compute_forward_add(params, mm_dst, addend, add_dst);
}

/* mul_mat node + the SCALE node on it (in place: the scale node is a view of the product, Ggml.cs:8265): the data both
 * nodes share ends as product * scalar. */
int ggml_hip_compute_forward_mul_mat_scale(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                           const struct ggml_tensor *src1, struct ggml_tensor *mm_dst,
                                           const struct ggml_tensor *scalar, struct ggml_tensor *scale_dst) {
    if (!params || !src0 || !src1 || !mm_dst || !scalar || !scale_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    if (scalar->type != GGML_TYPE_F32 || nelem(scalar) != 1 || !scalar->data) return fail(GGML_HIP_ERR_SHAPE, "scale: src1 must be an F32 scalar (Ggml.cs:6755)");
    bool fusable = scale_dst->data == mm_dst->data && scalar->op == GGML_OP_NONE;   // a view of the product; the scalar a leaf in host memory
    for (int i = 0; i < 4; ++i) fusable = fusable && scale_dst->ne[i] == mm_dst->ne[i];
    int rc = SEAM_NOT_FUSABLE;
    if (fusable) {
        const SeamEpi epi = {2, nullptr, nullptr, *(const float *)scalar->data};
        rc = seam1(params, src0, src1, mm_dst, &epi);
    }
    if (rc != SEAM_NOT_FUSABLE) return rc;
    rc = seam1(params, src0, src1, mm_dst, nullptr);
    if (rc) return rc;
    return ggml_hip_compute_forward_scale(params, mm_dst, scalar, scale_dst);
}

/* rms_norm node, the MUL node that consumes it, the MUL_MAT node whose src1 that is [, the ADD node on the product]: the
 * whole pre-projection chain of a transformer block as ONE launch for decode-sized batches (N <= 4: the fused mat-vec computes
 * y = (x * rms_scale) * g in its prologue, quantizes it in-kernel and applies the add in its store phase); larger batches, several
 * slots or strided tensors run the pair seam and the mul_mat(+add) seam one after the other.  Every node's data is produced. */
int ggml_hip_compute_forward_norm_mul_mat(const struct ggml_compute_params *params, const struct ggml_tensor *x, const struct ggml_tensor *g,
                                          struct ggml_tensor *norm_dst, struct ggml_tensor *mul_dst, const struct ggml_tensor *src0,
                                          struct ggml_tensor *mm_dst, const struct ggml_tensor *addend, struct ggml_tensor *add_dst) {
    if (!params || !x || !g || !norm_dst || !mul_dst || !src0 || !mm_dst || ((addend == nullptr) != (add_dst == nullptr))) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    SeamEpi epi = {addend ? 1 : 0, addend, add_dst, 1.0f};
    epi.pro_x = x; epi.pro_g = g; epi.pro_norm = norm_dst;
    int rc = seam1(params, src0, mul_dst, mm_dst, &epi);
    if (rc != SEAM_NOT_FUSABLE) return rc;
    rc = ggml_hip_compute_forward_rms_norm_mul(params, x, g, norm_dst, mul_dst);
    if (rc) return rc;
    if (addend) return ggml_hip_compute_forward_mul_mat_add(params, src0, mul_dst, mm_dst, addend, add_dst);
    return seam1(params, src0, mul_dst, mm_dst, nullptr);
}

/* Several MUL_MAT nodes with the SAME src1 (q / k / v, gate / up; optionally the rms_norm -> mul pair that produces that src1
 * in front) as one call: one launch inside a graph scope on one slot for N <= 4 when every weight is a cached leaf of one
 * quantized type and K and every tensor is contiguous; anything else runs the nodes through their own seams, one after the
 * other (which also fills the weight cache, so the next compute of the graph takes the fused form).  pro_x == NULL: no pair. */
int ggml_hip_compute_forward_mul_mat_multi(const struct ggml_compute_params *params, int n, const struct ggml_tensor *const *src0,
                                           const struct ggml_tensor *src1, struct ggml_tensor *const *dst, const struct ggml_tensor *pro_x,
                                           const struct ggml_tensor *pro_g, struct ggml_tensor *pro_norm) {
    if (!params || !src0 || !src1 || !dst || n < 1 || n > 4) return fail(GGML_HIP_ERR_ARG, "bad argument (1..4 nodes)");
    for (int i = 0; i < n; ++i)
        if (!src0[i] || !dst[i]) return fail(GGML_HIP_ERR_ARG, "null argument");
    if ((pro_x == nullptr) != (pro_g == nullptr) || (pro_x == nullptr) != (pro_norm == nullptr)) return fail(GGML_HIP_ERR_ARG, "prologue: x, g and the norm node go together");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    auto one_by_one = [&]() {
        int rc = pro_x ? ggml_hip_compute_forward_norm_mul_mat(params, pro_x, pro_g, pro_norm, const_cast<ggml_tensor *>(src1), src0[0], dst[0], nullptr, nullptr)
                       : seam1(params, src0[0], src1, dst[0], nullptr);
        for (int i = 1; i < n && !rc; ++i) rc = seam1(params, src0[i], src1, dst[i], nullptr);
        return rc;
    };
    if (n < 2) return one_by_one();
    const int type = src0[0]->type;
    const int64_t K = src1->ne[0], N = src1->ne[1];
    bool ok = type >= 0 && type < GGML_TYPE_COUNT && is_q(type) && weight_type_ok(type) && src1->type == GGML_TYPE_F32 && N >= 1 && N <= 64 &&
              src1->ne[2] == 1 && src1->ne[3] == 1 && src1->data && contiguous_f32(src1) && K % QK == 0;
    for (int i = 0; i < n && ok; ++i) {
        const ggml_tensor *w = src0[i], *d = dst[i];
        ok = w->type == type && w->op == GGML_OP_NONE && w->data && w->ne[0] == K && w->ne[2] == 1 && w->ne[3] == 1 && w->ne[1] > 0 &&
             w->nb[0] == TSIZE[type] && w->nb[0] <= w->nb[1] && d->type == GGML_TYPE_F32 && d->data && contiguous_f32(d) && d->ne[0] == w->ne[1] &&
             d->ne[1] == N && d->ne[2] == 1 && d->ne[3] == 1;
        for (int j = 0; j < i && ok; ++j) ok = dst[j]->data != d->data;
    }
    if (ok && pro_x) {
        const ggml_tensor *ts[3] = {pro_x, pro_g, pro_norm};
        for (const ggml_tensor *t : ts)
            ok = ok && t->data && t->type == GGML_TYPE_F32 && contiguous_f32(t) && t->ne[0] == K && t->ne[1] == N && t->ne[2] == 1 && t->ne[3] == 1;
        ok = ok && pro_norm->data != src1->data;
    }
    if (!ok) return one_by_one();
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    if (call.G() != 1 || !call.in_graph()) return one_by_one();
    DeviceCtx *c = call.ctxs[0];
    rc = c->make_current();
    if (rc) return rc;
    const ggml_hip_weight *W[4];
    for (int i = 0; i < n; ++i) {
        const ggml_tensor *w = src0[i];
        const CacheKey key{w->data, type, w->ne[0], w->ne[1], w->ne[2], w->ne[3], w->nb[1], w->nb[2], w->nb[3], (int64_t)0, w->ne[1]};
        auto it = c->cache.find(key);
        if (it == c->cache.end() || it->second.slices.size() != 1) return one_by_one();      // (uploads and caches them: fused next time)
        W[i] = it->second.slices[0];
    }
    // up to 4 rows: the fused mat-vec takes them (and the prologue) in one launch; 5..64 rows (Q4_0 / Q4_1): one INIT for all of them and one launch
    // where gemm_qmx.hip has the form (ggml_hip_mul_mat_multi_work_dev), the norm -> mul pair in front as its own kernel
    const bool fused_rows = ggml_hip_mul_mat_multi_fused(W, n, N) != 0;
    if (!fused_rows && (N <= 4 || !ggml_hip_mul_mat_epilogue_fused(W[0], N))) return one_by_one();
    for (int i = 0; i < n; ++i) note_host_write(call, dst[i], true);
    const float *xd = nullptr, *gd = nullptr;
    float *nd = nullptr, *yd = nullptr;
    if (pro_x) {
        if (operand_f32(c, true, pro_x, c->aux[0], &xd) || operand_f32(c, true, pro_g, c->aux[1], &gd)) return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: operand staging failed");
        note_host_write(call, pro_norm, true);
        note_host_write(call, src1, true);
        nd = result_f32(c, true, pro_norm, c->aux[2]);
        yd = result_f32(c, true, src1, c->aux[3]);
        if (!nd || !yd) return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: hipMalloc failed");
    } else if (operand_f32(c, true, src1, c->src1, &xd)) {
        return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: operand staging failed");
    }
    float *dd[4];
    int64_t ldd[4];
    for (int i = 0; i < n; ++i) {
        dd[i] = result_f32(c, true, dst[i], c->dst);
        if (!dd[i]) return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: hipMalloc failed for a resident dst");
        ldd[i] = dst[i]->ne[0];
    }
    if (fused_rows) {
        rc = ggml_hip_mul_mat_multi_dev(W, n, xd, K, N, dd, ldd, gd, K, nd, yd, c->stream);
    } else {
        const size_t w_bytes = ggml_hip_mul_mat_work_size(type, K, N);
        if (c->work.ensure(w_bytes ? w_bytes : 16)) return fail(GGML_HIP_ERR_RUNTIME, "multi mul_mat: hipMalloc failed for scratch");
        if (pro_x) rc = ggml_hip_rms_norm_mul_rows_dev(xd, gd, nd, yd, N, K, c->stream);
        if (!rc) rc = ggml_hip_mul_mat_multi_work_dev(W, n, pro_x ? yd : xd, K, N, dd, ldd, c->work.p, c->work.cap, c->stream);
    }
    if (rc) return rc;
    for (int i = 0; i < n; ++i) c->owe(dst[i]->data, dd[i], (size_t)N * dst[i]->ne[0] * 4);
    if (pro_x) {
        c->owe(pro_norm->data, nd, (size_t)N * K * 4);
        c->owe(src1->data, yd, (size_t)N * K * 4);
    }
    return GGML_HIP_OK;
}

/* ggml_compute_forward_cpy, quantizing branch of dup_f32 / dup_f16 (Ggml.cs:4339-4363, 3935-3966) */
int ggml_hip_compute_forward_cpy(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 struct ggml_tensor *dst) {
    if (!params || !src0 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
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
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    DeviceCtx *c = call.ctxs[0];                      // the result goes to host memory only: one slot does it
    rc = c->make_current();
    if (rc) return rc;
    if (call.in_graph()) { rc = c->pay_and_sync(); if (rc) return rc; }   // operands may be dst of an earlier node owed to / on its way to the host
    note_host_write(call, dst, false);                // dst is usually a future src0: its cached device copy (if any) is now stale
    const size_t row_in = (size_t)ne00 * es, rs = row_bytes_of(dt, ne00);   // rs as in Ggml.cs:4345
    if (c->src1.ensure(row_in * ne01) || c->dst.ensure(rs * ne01)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
    size_t id = 0;
    for (int64_t i03 = 0; i03 < ne03; ++i03)
        for (int64_t i02 = 0; i02 < ne02; ++i02) {
            const uint8_t *src = (const uint8_t *)src0->data + i02 * src0->nb[2] + i03 * src0->nb[3];
            HIP_TRY(hipMemcpy2DAsync(c->src1.p, row_in, src, src0->nb[1], row_in, (size_t)ne01, hipMemcpyHostToDevice, c->stream));
            rc = ggml_hip_quantize_rows_src_dev(dt, st, c->src1.p, ne00, ne01, ne00, c->dst.p, c->stream);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync((uint8_t *)dst->data + id, c->dst.p, rs * ne01, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            id += rs * ne01;
        }
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
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    const bool in_graph = call.in_graph();
    const int ns = eltwise_slots(call);
    for (int g = 0; g < ns; ++g) {
        DeviceCtx *c = call.ctxs[(size_t)g];
        rc = c->make_current();
        if (rc) return rc;
        const float *a = nullptr, *b = nullptr;
        if (operand_f32(c, in_graph, src0, c->src1, &a) || operand_f32(c, in_graph, src1, c->stage, &b)) return fail(GGML_HIP_ERR_RUNTIME, "%s: operand staging failed", name);
        if (g == 0) note_host_write(call, dst, true);      // (after the operands were resolved: dst may alias one of them)
        float *z = result_f32(c, in_graph, dst, c->dst);
        if (!z) return fail(GGML_HIP_ERR_RUNTIME, "%s: hipMalloc failed", name);
        HIP_TRY(launch_binary_f32(op, a, b, z, nelem(src0), c->stream));
        if (finish_f32(c, in_graph, g == 0, dst, z)) return fail(GGML_HIP_ERR_RUNTIME, "%s: copy back failed", name);
    }
    return GGML_HIP_OK;
}

/* ggml_compute_forward_add_q_f32 (Ggml.cs:4797-4906) */
int ggml_hip_compute_forward_add(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
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
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    DeviceCtx *c = call.ctxs[0];
    rc = c->make_current();
    if (rc) return rc;
    if (call.in_graph()) { rc = c->pay_and_sync(); if (rc) return rc; }   // operands may be dst of an earlier node owed to / on its way to the host
    note_host_write(call, dst, false);
    const size_t rs = row_bytes_of(t, ne00), rx = (size_t)ne00 * 4;
    if (c->stage.ensure(rs * ne01) || c->src1.ensure(rx * ne01) || c->dst.ensure(rs * ne01)) return fail(GGML_HIP_ERR_RUNTIME, "hipMalloc failed for scratch");
    for (int64_t i03 = 0; i03 < ne03; ++i03)
        for (int64_t i02 = 0; i02 < ne02; ++i02) {
            const uint8_t *a = (const uint8_t *)src0->data + i02 * src0->nb[2] + i03 * src0->nb[3];
            const uint8_t *b = (const uint8_t *)src1->data + i02 * src1->nb[2] + i03 * src1->nb[3];
            // the reference offsets dst rows by i3*nb0 (Ggml.cs:4891), an upstream typo for nb3; intent is followed
            uint8_t *d = (uint8_t *)dst->data + i02 * dst->nb[2] + i03 * dst->nb[3];
            HIP_TRY(hipMemcpy2DAsync(c->stage.p, rs, a, src0->nb[1], rs, (size_t)ne01, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpy2DAsync(c->src1.p, rx, b, src1->nb[1], rx, (size_t)ne01, hipMemcpyHostToDevice, c->stream));
            rc = ggml_hip_add_q_f32_rows_dev(t, c->stage.p, (const float *)c->src1.p, ne01, ne00, c->dst.p, c->stream);
            if (rc) return rc;
            HIP_TRY(hipMemcpy2DAsync(d, dst->nb[1], c->dst.p, rs, rs, (size_t)ne01, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    return GGML_HIP_OK;
}

/* ggml_compute_forward_mul (Ggml.cs:5037-5056) */
int ggml_hip_compute_forward_mul(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                 const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    return binary_f32_seam(1, src0, src1, dst);
}

/* ggml_compute_forward_scale_f32 (Ggml.cs:6746-6778): dst (a view of src0) *= *(float *)src1->data */
int ggml_hip_compute_forward_scale(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                   const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    if (!params || !src0 || !src1 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    if (src0->type != GGML_TYPE_F32 || src1->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "scale: F32 only (Ggml.cs:6786-6799)");
    if (nelem(src1) != 1) return fail(GGML_HIP_ERR_SHAPE, "scale: src1 must be a scalar (Ggml.cs:6755)");
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "scale: shapes differ (Ggml.cs:6754)");
    if (!contiguous_f32(src0) || !contiguous_f32(dst)) return fail(GGML_HIP_ERR_SHAPE, "scale: contiguous operands only (Ggml.cs:6752-6753)");
    if (!src0->data || !src1->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (nelem(src0) == 0) return GGML_HIP_OK;
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    const bool in_graph = call.in_graph();
    const int ns = eltwise_slots(call);
    float v = 0.0f;
    for (int g = 0; g < ns; ++g) {
        DeviceCtx *c = call.ctxs[(size_t)g];
        rc = c->make_current();
        if (rc) return rc;
        if (g == 0) {
            if (in_graph) { rc = c->before_host_read(src1->data, 4); if (rc) return rc; }      // the scalar is read from host memory (and is part of a named scope's key)
            v = *(const float *)src1->data;
        }
        // the reference scales dst's own memory: when dst is not a view of src0 that memory is whatever it held before, and
        // so it is here (the device form of "dst" is then an upload of dst->data, not of src0->data)
        const float *cur = nullptr;
        if (operand_f32(c, in_graph, dst, c->src1, &cur)) return fail(GGML_HIP_ERR_RUNTIME, "scale: operand staging failed");
        if (g == 0) note_host_write(call, dst, true);
        float *z = result_f32(c, in_graph, dst, c->dst);
        if (!z) return fail(GGML_HIP_ERR_RUNTIME, "scale: hipMalloc failed");
        if (z != cur) HIP_TRY(hipMemcpyAsync(z, cur, (size_t)nelem(dst) * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(launch_scale_f32(z, nelem(dst), v, c->stream));
        if (finish_f32(c, in_graph, g == 0, dst, z)) return fail(GGML_HIP_ERR_RUNTIME, "scale: copy back failed");
    }
    return GGML_HIP_OK;
}

// unary f32 seams share one body: which = 0 rms_norm (Ggml.cs:5858-5920), 1 silu (Ggml.cs:5705-5748)
static int unary_f32_seam(int which, const struct ggml_tensor *src0, struct ggml_tensor *dst) {
    const char *name = which == 0 ? "rms_norm" : "silu";
    if (src0->type != GGML_TYPE_F32 || dst->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "%s: F32 only (Ggml.cs:5927-5940, 5755-5768)", name);
    for (int i = 0; i < 4; ++i)
        if (src0->ne[i] != dst->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "%s: shapes differ (Ggml.cs:5863, 5712)", name);
    if (!contiguous_f32(src0) || !contiguous_f32(dst)) return fail(GGML_HIP_ERR_SHAPE, "%s: contiguous operands only (Ggml.cs:5710-5711)", name);
    if (!src0->data || !dst->data) return fail(GGML_HIP_ERR_ARG, "null data");
    if (nelem(src0) == 0) return GGML_HIP_OK;
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    const bool in_graph = call.in_graph();
    const int ns = eltwise_slots(call);
    for (int g = 0; g < ns; ++g) {
        DeviceCtx *c = call.ctxs[(size_t)g];
        rc = c->make_current();
        if (rc) return rc;
        const float *a = nullptr;
        if (operand_f32(c, in_graph, src0, c->src1, &a)) return fail(GGML_HIP_ERR_RUNTIME, "%s: operand staging failed", name);
        if (g == 0) note_host_write(call, dst, true);
        float *z = result_f32(c, in_graph, dst, c->dst);
        if (!z) return fail(GGML_HIP_ERR_RUNTIME, "%s: hipMalloc failed", name);
        if (which == 0) HIP_TRY(launch_rms_norm_f32(a, z, nelem(src0) / src0->ne[0], src0->ne[0], c->stream));
        else HIP_TRY(launch_silu_f32(a, z, nelem(src0), c->stream));
        if (finish_f32(c, in_graph, g == 0, dst, z)) return fail(GGML_HIP_ERR_RUNTIME, "%s: copy back failed", name);
    }
    return GGML_HIP_OK;
}

// two element-wise nodes, one launch (fused.hip): which = 0 rms_norm -> mul, 1 silu -> mul.  `first` is the operand of the
// first node, `other` the second operand of the mul node; both nodes' results are written and reach host memory.
static int fused_pair_seam(int which, const struct ggml_tensor *first, const struct ggml_tensor *other, struct ggml_tensor *mid_dst,
                           struct ggml_tensor *mul_dst) {
    const char *name = which == 0 ? "rms_norm + mul" : "silu + mul";
    const ggml_tensor *all[4] = {first, other, mid_dst, mul_dst};
    for (const ggml_tensor *t : all) {
        if (t->type != GGML_TYPE_F32) return fail(GGML_HIP_ERR_TYPE, "%s: F32 only", name);
        if (!contiguous_f32(t)) return fail(GGML_HIP_ERR_SHAPE, "%s: contiguous operands only", name);
        if (!t->data) return fail(GGML_HIP_ERR_ARG, "null data");
        for (int i = 0; i < 4; ++i)
            if (t->ne[i] != first->ne[i]) return fail(GGML_HIP_ERR_SHAPE, "%s: shapes differ", name);
    }
    if (mid_dst->data == mul_dst->data) return fail(GGML_HIP_ERR_SHAPE, "%s: the two nodes share their data", name);
    if (nelem(first) == 0) return GGML_HIP_OK;
    Call call;
    int rc = call.begin();
    if (rc) return rc;
    const bool in_graph = call.in_graph();
    const int ns = eltwise_slots(call);
    for (int g = 0; g < ns; ++g) {
        DeviceCtx *c = call.ctxs[(size_t)g];
        rc = c->make_current();
        if (rc) return rc;
        const float *a = nullptr, *b = nullptr;
        if (operand_f32(c, in_graph, first, c->src1, &a) || operand_f32(c, in_graph, other, c->stage, &b)) return fail(GGML_HIP_ERR_RUNTIME, "%s: operand staging failed", name);
        if (g == 0) { note_host_write(call, mid_dst, true); note_host_write(call, mul_dst, true); }
        float *z1 = result_f32(c, in_graph, mid_dst, c->dst), *z2 = result_f32(c, in_graph, mul_dst, c->dst2);
        if (!z1 || !z2) return fail(GGML_HIP_ERR_RUNTIME, "%s: hipMalloc failed", name);
        if (which == 0) HIP_TRY(launch_rms_norm_mul_f32(a, b, z1, z2, nelem(first) / first->ne[0], first->ne[0], c->stream));
        else HIP_TRY(launch_silu_mul_f32(a, b, z1, z2, nelem(first), c->stream));
        // (both copies on the compute stream; outside a graph scope the second finish waits for both)
        if (g == 0) {
            const size_t bytes = (size_t)nelem(first) * 4;
            if (in_graph) {
                c->owe(mid_dst->data, z1, bytes);
            } else {
                HIP_TRY(hipMemcpyAsync(mid_dst->data, z1, bytes, hipMemcpyDeviceToHost, c->stream));
                c->d2h_bytes += bytes;
            }
        }
        if (finish_f32(c, in_graph, g == 0, mul_dst, z2)) return fail(GGML_HIP_ERR_RUNTIME, "%s: copy back failed", name);
    }
    return GGML_HIP_OK;
}

/* rms_norm node + the MUL node that consumes it (prologue side of a mul_mat): norm_dst = rms_norm(x),
 * mul_dst = norm_dst * g, one launch, the unfused kernels' bits */
int ggml_hip_compute_forward_rms_norm_mul(const struct ggml_compute_params *params, const struct ggml_tensor *x,
                                          const struct ggml_tensor *g, struct ggml_tensor *norm_dst, struct ggml_tensor *mul_dst) {
    if (!params || !x || !g || !norm_dst || !mul_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    return fused_pair_seam(0, x, g, norm_dst, mul_dst);
}

/* silu node + the MUL node that consumes it (the SwiGLU gate): silu_dst = silu(a), mul_dst = silu_dst * b */
int ggml_hip_compute_forward_silu_mul(const struct ggml_compute_params *params, const struct ggml_tensor *a,
                                      const struct ggml_tensor *b, struct ggml_tensor *silu_dst, struct ggml_tensor *mul_dst) {
    if (!params || !a || !b || !silu_dst || !mul_dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    return fused_pair_seam(1, a, b, silu_dst, mul_dst);
}

/* ggml_compute_forward_rms_norm_f32 (Ggml.cs:5858-5920) */
int ggml_hip_compute_forward_rms_norm(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                      struct ggml_tensor *dst) {
    if (!params || !src0 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    return unary_f32_seam(0, src0, dst);
}

/* ggml_compute_forward_silu_f32 (Ggml.cs:5705-5748) */
int ggml_hip_compute_forward_silu(const struct ggml_compute_params *params, const struct ggml_tensor *src0,
                                  struct ggml_tensor *dst) {
    if (!params || !src0 || !dst) return fail(GGML_HIP_ERR_ARG, "null argument");
    if (params->ith != 0 || params->type != GGML_TASK_COMPUTE) return GGML_HIP_OK;
    if (scope_replaying()) return GGML_HIP_OK;
    return unary_f32_seam(1, src0, dst);
}

}  // extern "C"
