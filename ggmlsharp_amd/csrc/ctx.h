// ctx.h -- host-side state of the C-ABI (internal): one DeviceCtx per device slot instead of process globals.
//
// A *slot* is one GPU the library drives from this process (ggml_hip_init_devices).  Every slot owns its streams,
// scratch, graph-scope residency table and weight cache, so two host threads bound to different slots
// (ggml_hip_bind_thread) run the seams concurrently, and an unbound caller gets the reference's own row partition
// (Ggml.cs:6665-6672) over all slots.  Two slots may sit on the same physical device: that is how the split is
// rehearsed -- and tested bit for bit -- on a one-GPU box.
#pragma once
#include "common.h"
#include "plan.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <tuple>
#include <vector>

namespace ghip {

int fail(int code, const char *fmt, ...);          // sets the calling thread's ggml_hip_last_error text, returns code
#define HIP_TRY(expr)                                                                                               \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return ghip::fail(GGML_HIP_ERR_RUNTIME, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

extern const int BLCK[GGML_TYPE_COUNT];            // Ggml.cs:55-70
extern const size_t TSIZE[GGML_TYPE_COUNT];        // Ggml.cs:72-87
bool wq_ok(int t);                                 // quantized types with working row functions and dot products
bool weight_type_ok(int t);
bool is_q(int t);
int vec_dot_type(int t);                           // Ggml.cs:219-290
inline size_t row_bytes_of(int t, int64_t k) { return TSIZE[t] * (size_t)(k / BLCK[t]); }
inline int64_t nelem(const ggml_tensor *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
bool contiguous_f32(const ggml_tensor *t);
int act_image_kind(int type, int64_t K, int64_t N);              // by type, K and N alone (plan.cpp)
int weight_image_kind(const ggml_hip_weight *w, int64_t N);         // ... for one weight (kind 0 when its planes exceed 32-bit offsets)
mm_plan weight_plan(const ggml_hip_weight *w, int64_t N, bool one_call);   // plan.h: the one decision per product

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {   // (the owning slot's device is current)
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(&p, n) != hipSuccess) return -1;
        cap = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Resident { void *p; size_t bytes; };

// cached device form of a src0 (Seam 1): its 2-D slices, each one row shard per participating slot
struct CachedWeight {
    const void *host = nullptr;
    size_t host_bytes = 0;                        // [host, host + host_bytes) is what the entry was built from
    std::vector<ggml_hip_weight *> slices;        // ne02 * ne03 entries (this slot's rows of each)
    int64_t row_begin = 0, row_end = 0;
    uint64_t last_use = 0;                        // DeviceCtx::cache_clock at the entry's last hit (least recently used goes first)
    size_t dev_bytes = 0;                         // resident bytes of the slices (the operand images are 1.5-2.6 x the file format)
};
using CacheKey = std::tuple<const void *, int, int64_t, int64_t, int64_t, int64_t, uint64_t, uint64_t, uint64_t, int64_t, int64_t>;

constexpr int MAX_SLOTS = 16;
constexpr int PIPE_EVENTS = 32;                   // event pairs of the Seam-1 pipeline (chunks of one call reuse them round-robin)

struct DeviceCtx {
    int slot = -1;
    int device = -1;                               // HIP device ordinal
    std::string arch;
    std::recursive_mutex mu;                       // serialises the seams that use this slot
    hipStream_t stream = nullptr;                  // kernels (+ the small copies of the element-wise seams)
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;  // Seam-1 pipeline: host -> device, device -> host
    hipEvent_t ev_in[PIPE_EVENTS] = {}, ev_k[PIPE_EVENTS] = {};
    hipEvent_t ev_compute = nullptr, ev_d2h = nullptr, ev_xchg = nullptr, ev_ready = nullptr;
    Scratch src1, dst, dst2, work, stage, aux[4];
    // graph scope: host data pointer -> device copy kept while the graph runs
    std::map<const void *, Resident> resident;
    std::vector<Resident> pool;                    // device buffers free for reuse
    std::vector<ggml_hip_weight *> transient;      // weights built for one node from a computed src0, freed at graph end
    std::map<CacheKey, CachedWeight> cache;        // Seam-1 weight cache
    uint64_t cache_clock = 0;                      // ticks once per cache hit / insertion
    size_t cache_bytes = 0;                        // sum of the entries' dev_bytes
    uint64_t cache_evictions = 0;                  // entries evicted so far (tests)
    // Least-recently-used entries go until `need` bytes are free under `budget` (budget 0: until `need` bytes have been freed at all);
    // the entry with key `keep` (the one the running call uses) stays.  Returns the bytes freed.  Waits for this slot's streams first and
    // drops the captured graphs (their launches hold the entries' device pointers) -- only when something is evicted.
    size_t evict_lru(size_t need, size_t budget, const CacheKey *keep);
    uint64_t h2d_bytes = 0, d2h_bytes = 0, resident_hits = 0;
    int graph_depth_ = 0;                          // ggml_hip_graph_begin / _end nesting on this slot
    // graph scope: results whose host copy is still owed.  A node of a graph scope costs the host ONE kernel launch; the
    // device -> host copies every node's data needs (the reference leaves every node's data in host memory) are issued
    // together -- one scatter kernel writing through the device mapping of the registered pool -- when the scope ends, when a
    // buffer is about to be recycled, or before anything reads host memory that may be owed.
    struct Owed { void *host; const void *dev; size_t bytes; };
    // opt-in (ggml_hip_graph_outputs): only the listed tensors are owed to the caller at scope end; the others are still
    // copied when the LIBRARY needs them in host memory (a recycled buffer, an upload of an overlapping range), never otherwise
    bool outputs_only = false;
    std::vector<const void *> outputs;
    bool wanted(const void *host) const {
        if (!outputs_only) return true;
        for (const void *p : outputs) if (p == host) return true;
        return false;
    }
    std::vector<Owed> owed;
    static constexpr size_t OWE_EARLY_BYTES = 1u << 20;      // results this large are copied at once, on the copy stream (7B layer, batch 32 = 512-KB results: 468 us captured and paid at the end, 550 us with early copies; batch 512: 4.97 -> 4.26 ms)
    void owe(void *host, const void *dev, size_t bytes);      // (replaces an entry for the same host pointer)
    void join_copies();
    int pay(const void *only_dev = nullptr, size_t only_bytes = 0);   // issue the copies on `stream` (all, or those reading [only_dev, only_dev + only_bytes))
    int pay_and_sync();                                       // ... and wait: host memory is current afterwards
    // Keyed graph scopes (ggml_hip_graph_begin_keyed): the caller names the graph it is about to run.  A scope that needed
    // nothing but launches on the compute stream (observed once) is captured into a hipGraph the second time and REPLAYED with one
    // launch from the third time on: the seams return at once, leaf data is re-read from host memory by the captured copies.
    struct Captured {
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        std::vector<Resident> buffers;             // the scope's resident buffers, owned by the entry while it can be replayed
        uint64_t scratch_sig = 0;                  // the scratch pointers baked into the captured launches
        std::vector<std::pair<const void *, size_t>> leaves;   // host tensors the scope uploaded when it was observed: prefetched together
        int seen = 0, strikes = 0; bool refused = false;   // refused: a capture failed outright, or was cut short three times
    };
    std::map<uint64_t, Captured> captured;
    uint64_t scope_key = 0;
    std::thread::id scope_owner;                   // the thread that opened the named scope: only ITS seam calls are the scope's
    bool dead = false;                             // ggml_hip_shutdown released this slot while the caller was waiting for its lock
    int scope_mode = 0;                            // 0 plain, 1 observing, 2 capturing, 3 replaying
    bool scope_lost = false;                       // a capture of the open scope could not be ended: ggml_hip_graph_end reports it
    bool scope_clean = true;
    std::vector<std::pair<const void *, size_t>> scope_leaves;   // (observing) whole host tensors uploaded so far in this scope
    void note_leaf(const void *host, size_t bytes) { if (scope_mode == 1) scope_leaves.emplace_back(host, bytes); }
    uint64_t n_observed = 0, n_captured = 0, n_replayed = 0, n_refused = 0;   // keyed scopes by what became of them (tests, tuning)
    uint64_t scratch_sig() const;
    void scope_dirty();                            // something other than a compute-stream launch is about to happen
    void drop_captured();                          // device pointers the captured graphs hold are about to go away
    bool d2h_busy = false;                                    // a device -> host copy was issued since the last sync_all
    int before_host_read(const void *host, size_t bytes);     // host memory [host, host + bytes) is about to be read (uploaded)

    int make_current() const;                      // hipSetDevice(device) as a status
    int sync_all();                                // every stream of this slot
    // graph-scope buffers
    void *take(size_t n);                          // a device buffer of >= n bytes (recycled or new); nullptr on failure
    void *resident_buffer(const void *host, size_t bytes);   // the buffer kept for `host` (existing, recycled or new)
    const void *resident_lookup(const void *host, size_t bytes);
    void drop_overlapping(const void *host, size_t bytes, bool keep_exact);
    void drain(bool free_all);
    void invalidate(const void *host, size_t bytes);          // weight-cache entries built from an overlapping host range
    void free_cache();
};

// ---- the slot table ----
int n_slots();
DeviceCtx *slot(int i);
int ensure_init();                                 // lazily ggml_hip_init(0), as before
int bound_slot();                                  // the calling thread's slot, -1 = unbound (all slots)
extern std::mutex g_table_mu;

// host memory registered for DMA (ggml_hip_register_host_pool): true when [p, p + n) lies inside a registered range
bool host_range_pinned(const void *p, size_t n);
// device-visible address of registered host memory [p, p + n) (hipHostRegisterMapped), nullptr when it is not registered
void *host_range_device_ptr(const void *p, size_t n);

// ---- weights ----
int make_weight(DeviceCtx *c, int type, const void *rows, bool rows_on_host, int64_t ne00, int64_t ne01, uint64_t nb01,
                int64_t row_begin, int64_t row_end, hipStream_t st, ggml_hip_weight **out);

// reference partition of M rows over G parts (Ggml.cs:6665-6672)
inline void shard_rows(int64_t M, int G, int g, int64_t *r0, int64_t *r1) {
    const int64_t dr = (M + G - 1) / G;
    const int64_t a = dr * g < M ? dr * g : M;
    *r0 = a;
    *r1 = a + dr < M ? a + dr : M;
}

// ---- exchange of dst column ranges between slots (multi.cpp) ----
// every slot g has written columns [r0[g], r1[g]) of its own [N][ldd] buffer bufs[g]; afterwards every buffer holds all
// columns.  Stream-ordered on each slot's compute stream.
int exchange_columns(int G, DeviceCtx *const *ctxs, float *const *bufs, int64_t N, int64_t ldd, const int64_t *r0,
                     const int64_t *r1);
void rccl_shutdown();

}  // namespace ghip
