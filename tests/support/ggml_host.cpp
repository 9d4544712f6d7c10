// ggml_host.cpp -- host mirror of the GGMLSharp public API for the mul_mat path (include/ggml.h).
// Same pool arithmetic, struct layouts and graph walk as the reference; MUL_MAT nodes are dispatched to the
// HIP path (Seam 1).  There is deliberately no CPU compute here.
//
// TEST SUPPORT, not product: this is the stand-in for the reference's C# host (no .NET toolchain in the image).  It is
// built into its own library, libggml_hostmirror.so, which links libggml_hip.so exactly as the C# host would bind it --
// through the exported ggml_hip_* C-ABI only.  The real host never loads it, and libggml_hip.so exports no ggml_* name
// that could collide with a native ggml in the same process.
#include "ggml.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

static_assert(sizeof(ggml_object) == 32, "TypeDefinitions.cs:48-56");
static_assert(sizeof(ggml_context) == 88, "TypeDefinitions.cs:32-46");
static_assert(sizeof(ggml_cgraph) == 98360, "TypeDefinitions.cs:102-121");

namespace {

constexpr int GGML_MAX_CONTEXTS = 64;  // TypeDefinitions.cs:224
struct Container { bool used; ggml_context context; };
Container g_contexts[GGML_MAX_CONTEXTS];
std::mutex g_ctx_mu;  // ggml_critical_section_start/end (Ggml.cs:8451-8470)

const int BLCK[GGML_TYPE_COUNT] = {1, 1, 32, 32, 16, 16, 32, 32, 32, 32, 1, 1, 1};
const size_t TSIZE[GGML_TYPE_COUNT] = {4, 2, 20, 24, 10, 12, 22, 24, 36, 44, 1, 2, 4};

bool type_ok(int t) { return t >= 0 && t < GGML_TYPE_COUNT; }

// Ggml.cs:7722-7866
ggml_tensor *new_tensor_impl(ggml_context *ctx, int type, int n_dims, const int64_t *ne, void *data) {
    if (!ctx || !type_ok(type) || n_dims < 1 || n_dims > GGML_MAX_DIMS) return nullptr;
    ggml_object *obj_cur = ctx->objects_end;
    const uint64_t cur_offs = obj_cur == nullptr ? 0 : obj_cur->offs;
    const uint64_t cur_size = obj_cur == nullptr ? 0 : obj_cur->size;
    const uint64_t cur_end = cur_offs + cur_size;

    uint64_t size_needed = 0;
    if (data == nullptr && !ctx->no_alloc) {
        size_needed += TSIZE[type] * (uint64_t)(ne[0] / BLCK[type]);
        for (int i = 1; i < n_dims; i++) size_needed *= (uint64_t)ne[i];
        size_needed = ((size_needed + GGML_MEM_ALIGN - 1) / GGML_MEM_ALIGN) * GGML_MEM_ALIGN;
    }
    uint8_t *mem_buffer = (uint8_t *)ctx->mem_buffer;
    ggml_object *obj_new = (ggml_object *)(mem_buffer + cur_end);

    if (ctx->scratch.data == nullptr || data != nullptr) {
        size_needed += sizeof(ggml_tensor);
        if (cur_end + size_needed + sizeof(ggml_object) > ctx->mem_size) {
            fprintf(stderr, "ggml_new_tensor_impl: not enough space in the context's memory pool (needed %llu, available %llu)\n",
                    (unsigned long long)(cur_end + size_needed + sizeof(ggml_object)), (unsigned long long)ctx->mem_size);
            return nullptr;  // Ggml.cs:7757-7763
        }
        obj_new->offs = cur_end + sizeof(ggml_object);
        obj_new->size = size_needed;
        obj_new->next = nullptr;
    } else {
        if (ctx->scratch.offs + size_needed > ctx->scratch.size) {
            fprintf(stderr, "ggml_new_tensor_impl: not enough space in the scratch memory\n");
            return nullptr;
        }
        if (cur_end + sizeof(ggml_tensor) + sizeof(ggml_object) > ctx->mem_size) {
            fprintf(stderr, "ggml_new_tensor_impl: not enough space in the context's memory pool\n");
            return nullptr;
        }
        data = (uint8_t *)ctx->scratch.data + ctx->scratch.offs;
        obj_new->offs = cur_end + sizeof(ggml_object);
        obj_new->size = sizeof(ggml_tensor);
        obj_new->next = nullptr;
        ctx->scratch.offs += size_needed;
    }
    if (obj_cur != nullptr) obj_cur->next = obj_new; else ctx->objects_begin = obj_new;
    ctx->objects_end = obj_new;

    ggml_tensor *result = (ggml_tensor *)(mem_buffer + obj_new->offs);
    memset(result, 0, sizeof *result);
    result->type = type;
    result->n_dims = n_dims;
    result->op = GGML_OP_NONE;
    result->data = (data == nullptr && !ctx->no_alloc) ? (void *)(result + 1) : data;  // Ggml.cs:7839
    for (int i = 0; i < GGML_MAX_DIMS; i++) result->ne[i] = 1;
    for (int i = 0; i < n_dims; i++) result->ne[i] = ne[i];
    result->nb[0] = TSIZE[type];                                                   // Ggml.cs:7856-7861
    result->nb[1] = result->nb[0] * (uint64_t)(result->ne[0] / BLCK[type]);
    for (int i = 2; i < GGML_MAX_DIMS; i++) result->nb[i] = result->nb[i - 1] * (uint64_t)result->ne[i - 1];
    ctx->n_objects++;
    return result;
}

// Ggml.cs:7559-7619
bool visit_parents(ggml_cgraph *g, ggml_tensor *node) {
    for (int i = 0; i < g->n_nodes; i++) if (g->nodes[i] == node) return true;
    for (int i = 0; i < g->n_leafs; i++) if (g->leafs[i] == node) return true;
    if (node->src0 && !visit_parents(g, node->src0)) return false;
    if (node->src1 && !visit_parents(g, node->src1)) return false;
    for (int i = 0; i < GGML_MAX_OPT; ++i)
        if (node->opt[i] != 0 && !visit_parents(g, (ggml_tensor *)(intptr_t)node->opt[i])) return false;
    if (node->op == GGML_OP_NONE && node->grad == nullptr) {
        if (g->n_leafs >= GGML_MAX_NODES) return false;
        g->leafs[g->n_leafs++] = node;
    } else {
        if (g->n_nodes >= GGML_MAX_NODES) return false;
        g->nodes[g->n_nodes] = node;
        g->grads[g->n_nodes] = node->grad;
        g->n_nodes++;
    }
    return true;
}

}  // namespace

extern "C" {

ggml_context *ggml_init(const ggml_init_params *params) {
    if (!params) return nullptr;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    ggml_context *ctx = nullptr;
    for (int i = 0; i < GGML_MAX_CONTEXTS; i++)
        if (!g_contexts[i].used) { g_contexts[i].used = true; ctx = &g_contexts[i].context; break; }
    if (!ctx) return nullptr;  // Ggml.cs:1529-1536
    const uint64_t mem_size = (params->mem_size + GGML_MEM_ALIGN - 1) & ~(uint64_t)(GGML_MEM_ALIGN - 1);
    memset(ctx, 0, sizeof *ctx);
    ctx->mem_size = mem_size;
    ctx->mem_buffer_owned = params->mem_buffer ? 0 : 1;
    ctx->mem_buffer = params->mem_buffer ? params->mem_buffer : aligned_alloc(GGML_MEM_ALIGN, mem_size ? mem_size : GGML_MEM_ALIGN);
    ctx->no_alloc = params->no_alloc;
    if (!ctx->mem_buffer) {
        for (int i = 0; i < GGML_MAX_CONTEXTS; i++) if (&g_contexts[i].context == ctx) g_contexts[i].used = false;
        return nullptr;
    }
    // the pool is ONE allocation that every tensor's data lives in (Ggml.cs:1545): register it for DMA so that Seam 1 can
    // overlap its copies with the kernels.  Best effort: without a device, or if pinning fails, the pool stays pageable.
    if (mem_size >= (64u << 10) && ggml_hip_device_count() > 0) (void)ggml_hip_register_host_pool(ctx->mem_buffer, mem_size);
    return ctx;
}

void ggml_free(ggml_context *ctx) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (int i = 0; i < GGML_MAX_CONTEXTS; i++)
        if (&g_contexts[i].context == ctx) {
            g_contexts[i].used = false;
            // the reference's ggml_free gives the device layer no callback; this mirror does: in-flight copies are waited
            // for, the pool is unpinned, and every cached / resident device copy made from it is dropped
            (void)ggml_hip_unregister_host_pool(ctx->mem_buffer);
            ggml_hip_invalidate_range(ctx->mem_buffer, ctx->mem_size);
            if (ctx->mem_buffer_owned) free(ctx->mem_buffer);
            break;
        }
}

size_t ggml_used_mem(const ggml_context *ctx) {
    return (!ctx || !ctx->objects_end) ? 0 : (size_t)(ctx->objects_end->offs + ctx->objects_end->size);
}

ggml_tensor *ggml_new_tensor(ggml_context *ctx, int type, int n_dims, const int64_t *ne) {
    return new_tensor_impl(ctx, type, n_dims, ne, nullptr);
}
ggml_tensor *ggml_new_tensor_1d(ggml_context *ctx, int type, int64_t ne0) { return ggml_new_tensor(ctx, type, 1, &ne0); }
ggml_tensor *ggml_new_tensor_2d(ggml_context *ctx, int type, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return ggml_new_tensor(ctx, type, 2, ne);
}
ggml_tensor *ggml_new_tensor_3d(ggml_context *ctx, int type, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return ggml_new_tensor(ctx, type, 3, ne);
}
ggml_tensor *ggml_new_tensor_4d(ggml_context *ctx, int type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    return ggml_new_tensor(ctx, type, 4, ne);
}

int64_t ggml_nelements(const ggml_tensor *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
int64_t ggml_nrows(const ggml_tensor *t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
size_t ggml_nbytes(const ggml_tensor *t) { return ((size_t)ggml_nelements(t) * TSIZE[t->type]) / (size_t)BLCK[t->type]; }
int ggml_blck_size(int type) { return type_ok(type) ? BLCK[type] : 0; }
size_t ggml_type_size(int type) { return type_ok(type) ? TSIZE[type] : 0; }
int ggml_is_quantized(int type) { return type >= GGML_TYPE_Q4_0 && type <= GGML_TYPE_Q8_1; }
int ggml_is_contiguous(const ggml_tensor *t) {
    return t->nb[0] == TSIZE[t->type] && t->nb[1] == (t->nb[0] * (uint64_t)t->ne[0]) / (uint64_t)BLCK[t->type] &&
           t->nb[2] == t->nb[1] * (uint64_t)t->ne[1] && t->nb[3] == t->nb[2] * (uint64_t)t->ne[2];
}
int ggml_can_mul_mat(const ggml_tensor *t0, const ggml_tensor *t1) {
    return t0->ne[0] == t1->ne[0] && t0->ne[2] == t1->ne[2] && t0->ne[3] == t1->ne[3];
}

ggml_tensor *ggml_set_f32(ggml_tensor *t, float value) {
    if (!t || t->type != GGML_TYPE_F32) return t;  // the integer/f16 cases of Ggml.cs:2501-2565 are not on this path
    const int64_t n = ggml_nrows(t), nc = t->ne[0];
    for (int64_t i = 0; i < n; i++) {
        float *row = (float *)((uint8_t *)t->data + i * t->nb[1]);
        for (int64_t j = 0; j < nc; j++) row[j] = value;
    }
    // a host write to tensor memory: a device copy cached from it (an F32 weight leaf) would be stale
    if (n > 0 && nc > 0) ggml_hip_invalidate_range(t->data, (size_t)((n - 1) * (int64_t)t->nb[1] + nc * 4));
    return t;
}
float ggml_get_f32_1d(const ggml_tensor *t, int i) { return ((const float *)t->data)[i]; }
void ggml_set_f32_1d(ggml_tensor *t, int i, float value) {
    ((float *)t->data)[i] = value;
    ggml_hip_invalidate_range((const float *)t->data + i, 4);
}

ggml_tensor *ggml_mul_mat(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b) {
    if (!ctx || !a || !b) return nullptr;
    if (!ggml_can_mul_mat(a, b)) return nullptr;   // Debug.Assert, Ggml.cs:8228
    if (a->nb[0] > a->nb[1]) return nullptr;       // !ggml_is_transposed(a), Ggml.cs:8229
    const bool is_node = a->grad != nullptr || b->grad != nullptr;
    const int64_t ne[4] = {a->ne[1], b->ne[1], a->ne[2], b->ne[3]};
    ggml_tensor *result = ggml_new_tensor(ctx, GGML_TYPE_F32, a->n_dims < b->n_dims ? a->n_dims : b->n_dims, ne);
    if (!result) return nullptr;
    result->op = GGML_OP_MUL_MAT;
    result->grad = is_node ? ggml_new_tensor(ctx, GGML_TYPE_F32, result->n_dims, result->ne) : nullptr;  // ggml_dup_tensor
    result->src0 = a;
    result->src1 = b;
    return result;
}

ggml_tensor *ggml_view_tensor(ggml_context *ctx, ggml_tensor *src) {
    if (!ctx || !src) return nullptr;
    ggml_tensor *result = new_tensor_impl(ctx, src->type, src->n_dims, src->ne, src->data);
    if (!result) return nullptr;
    for (int i = 0; i < GGML_MAX_DIMS; ++i) result->nb[i] = src->nb[i];
    return result;
}

ggml_tensor *ggml_dup_tensor(ggml_context *ctx, const ggml_tensor *src) {
    if (!ctx || !src) return nullptr;
    return new_tensor_impl(ctx, src->type, src->n_dims, src->ne, nullptr);
}

ggml_tensor *ggml_cpy(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b) {
    if (!ctx || !a || !b) return nullptr;
    if (ggml_nelements(a) != ggml_nelements(b)) return nullptr;   // Debug.Assert, Ggml.cs:8281
    if (a->grad != nullptr || b->grad != nullptr) return nullptr;  // "TODO: implement backward", Ggml.cs:8284-8288
    ggml_tensor *result = ggml_view_tensor(ctx, b);
    if (!result) return nullptr;
    result->op = GGML_OP_CPY;
    result->src0 = a;
    result->src1 = b;
    return result;
}

ggml_tensor *ggml_add(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b) {
    if (!ctx || !a || !b) return nullptr;
    for (int i = 0; i < GGML_MAX_DIMS; ++i)
        if (a->ne[i] != b->ne[i]) return nullptr;                   // ggml_are_same_shape, Ggml.cs:7874
    const bool is_node = a->grad != nullptr || b->grad != nullptr;
    ggml_tensor *result = ggml_dup_tensor(ctx, a);
    if (!result) return nullptr;
    result->op = GGML_OP_ADD;
    result->grad = is_node ? ggml_dup_tensor(ctx, result) : nullptr;
    result->src0 = a;
    result->src1 = b;
    return result;
}

// Ggml.cs:6878-6884 -> ggml_mul_impl 7918-7946 (not in place): result = dup(a); NULL unless same shape (:7924)
ggml_tensor *ggml_mul(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b) {
    if (!ctx || !a || !b) return nullptr;
    for (int i = 0; i < GGML_MAX_DIMS; ++i)
        if (a->ne[i] != b->ne[i]) return nullptr;
    const bool is_node = a->grad != nullptr || b->grad != nullptr;
    ggml_tensor *result = ggml_dup_tensor(ctx, a);
    if (!result) return nullptr;
    result->op = GGML_OP_MUL;
    result->grad = is_node ? ggml_dup_tensor(ctx, result) : nullptr;
    result->src0 = a;
    result->src1 = b;
    return result;
}

// Ggml.cs:7153-7159 -> ggml_scale_impl 8248-8273: b must be a scalar (:8254); the result is a VIEW of a (:8265): in place
ggml_tensor *ggml_scale(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b) {
    if (!ctx || !a || !b) return nullptr;
    if (ggml_nelements(b) != 1) return nullptr;
    const bool is_node = a->grad != nullptr || b->grad != nullptr;
    ggml_tensor *result = ggml_view_tensor(ctx, a);
    if (!result) return nullptr;
    result->op = GGML_OP_SCALE;
    result->grad = is_node ? ggml_dup_tensor(ctx, result) : nullptr;
    result->src0 = a;
    result->src1 = b;
    return result;
}

// Ggml.cs:7123-7128 -> ggml_rms_norm_impl 8199-8220 (not in place): result = dup(a); backward is not implemented (:8208)
ggml_tensor *ggml_rms_norm(ggml_context *ctx, ggml_tensor *a) {
    if (!ctx || !a) return nullptr;
    if (a->grad != nullptr) return nullptr;
    ggml_tensor *result = ggml_dup_tensor(ctx, a);
    if (!result) return nullptr;
    result->op = GGML_OP_RMS_NORM;
    result->src0 = a;
    result->src1 = nullptr;
    return result;
}

// Ggml.cs:7095-7107 -> ggml_silu_impl 8154-8174
static ggml_tensor *silu_impl(ggml_context *ctx, ggml_tensor *a, bool inplace) {
    if (!ctx || !a) return nullptr;
    const bool is_node = !inplace && a->grad != nullptr;
    ggml_tensor *result = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    if (!result) return nullptr;
    result->op = GGML_OP_SILU;
    result->grad = is_node ? ggml_dup_tensor(ctx, result) : nullptr;
    result->src0 = a;
    result->src1 = nullptr;
    return result;
}
ggml_tensor *ggml_silu(ggml_context *ctx, ggml_tensor *a) { return silu_impl(ctx, a, false); }
ggml_tensor *ggml_silu_inplace(ggml_context *ctx, ggml_tensor *a) { return silu_impl(ctx, a, true); }

void ggml_build_forward_expand(ggml_cgraph *cgraph, ggml_tensor *tensor) { visit_parents(cgraph, tensor); }

void ggml_build_forward(ggml_cgraph *out, ggml_tensor *tensor) {
    out->n_nodes = 0; out->n_leafs = 0;
    out->n_threads = GGML_DEFAULT_N_THREADS;
    out->work_size = 0; out->work = nullptr;
    out->perf_runs = 0; out->perf_cycles = 0; out->perf_time_us = 0;
    visit_parents(out, tensor);
}

int ggml_graph_compute(ggml_context *ctx, ggml_cgraph *cgraph) {
    (void)ctx;
    if (!cgraph) return GGML_HIP_ERR_ARG;
    // plan (Ggml.cs:3260-3519): offloaded MUL_MAT nodes take the reference's own "n_tasks = 1, no host work buffer"
    // slot (:3368-3370); the Q8 scratch lives on the device (ggml_hip_mul_mat_work_size).
    for (int i = 0; i < cgraph->n_nodes; i++) {
        ggml_tensor *node = cgraph->nodes[i];
        if (node->op == GGML_OP_NONE) continue;
        if (node->op != GGML_OP_MUL_MAT && node->op != GGML_OP_CPY && node->op != GGML_OP_ADD && node->op != GGML_OP_MUL &&
            node->op != GGML_OP_SCALE && node->op != GGML_OP_RMS_NORM && node->op != GGML_OP_SILU) {
            fprintf(stderr, "ggml_graph_compute: op %d is outside the MI355X mul_mat path (SURVEY.md 2.2)\n", node->op);
            return GGML_HIP_ERR_TYPE;
        }
        node->n_tasks = 1;
    }
    cgraph->work_size = 0;
    // graph scope: results of offloaded nodes stay in HBM for the nodes that consume them.  The scope is
    // named by a hash over everything that decides the calls made below (include/ggml_hip.h ggml_hip_graph_begin_keyed): a
    // graph that is computed again and again -- a decoder's token loop -- is replayed as one captured launch.
    uint64_t key = 1469598103934665603ull;
    auto mix = [&key](uint64_t v) { key ^= v; key *= 1099511628211ull; };
    auto mix_tensor = [&mix](const ggml_tensor *t) {
        if (!t) { mix(0x9e3779b97f4a7c15ull); return; }
        mix((uint64_t)(uintptr_t)t); mix((uint64_t)(uintptr_t)t->data); mix((uint64_t)t->type); mix((uint64_t)t->op);
        for (int d = 0; d < 4; ++d) { mix((uint64_t)t->ne[d]); mix((uint64_t)t->nb[d]); }
    };
    mix((uint64_t)cgraph->n_nodes);
    for (int i = 0; i < cgraph->n_nodes; i++) {
        const ggml_tensor *node = cgraph->nodes[i];
        mix_tensor(node); mix_tensor(node->src0); mix_tensor(node->src1);
        if (node->op == GGML_OP_SCALE && node->src1 && node->src1->data && node->src1->type == GGML_TYPE_F32) {   // read on the host at issue time
            uint32_t bits; memcpy(&bits, node->src1->data, 4); mix(bits);
        }
    }
#ifdef GGML_MIRROR_PLAIN_SCOPE                 // (experiment build, tools/experiments/node_cost.py: every graph issued live, never captured)
    int grc = ggml_hip_graph_begin();
#else
    int grc = ggml_hip_graph_begin_keyed(key ? key : 1);
#endif
    if (grc != GGML_HIP_OK) return grc;
    // Execution order: the graph's own (Ggml.cs:7559-7619 is a depth-first walk), except that a SILU whose consumer -- the MUL of
    // a gated feed-forward, mul(silu(u), g) -- sits a few nodes further on is run right in front of it, so that the pair goes
    // down as one launch.  The walk puts g's producers between the two; a node may move past them when none of them reads what
    // it writes or writes what it reads or writes.
    static thread_local ggml_tensor *order[GGML_MAX_NODES];     // (a plain array: the test-support library exports ggml_* only)
    const int n_order = cgraph->n_nodes;
    for (int i = 0; i < n_order; ++i) order[i] = cgraph->nodes[i];
    auto range_of = [](const ggml_tensor *t, const uint8_t **a, const uint8_t **b) {
        *a = (const uint8_t *)t->data; *b = *a + ggml_nbytes(t);
    };
    auto overlaps = [&](const ggml_tensor *x, const ggml_tensor *y) {
        if (!x || !y || !x->data || !y->data) return false;
        const uint8_t *a0, *a1, *b0, *b1;
        range_of(x, &a0, &a1); range_of(y, &b0, &b1);
        return a0 < b1 && b0 < a1;
    };
    for (int i = 0; i + 2 < n_order; ++i) {
        ggml_tensor *s = order[i];
        if (s->op != GGML_OP_SILU || !s->src0 || s->data == s->src0->data) continue;
        int j = -1;
        for (int k = i + 2; k < n_order && k <= i + 4; ++k) {
            ggml_tensor *m = order[k];
            if (m->op == GGML_OP_MUL && (m->src0 == s || m->src1 == s) && m->src0 != m->src1 && m->data != s->data) { j = k; break; }
        }
        if (j < 0) continue;
        bool ok = true;
        for (int k = i + 1; k < j && ok; ++k) {
            ggml_tensor *t = order[k];
            ok = t->op != GGML_OP_NONE && t->src0 != s && t->src1 != s && !overlaps(t->src0, s) && !overlaps(t->src1, s) && !overlaps(t, s) &&
                 !overlaps(t, s->src0);
        }
        if (!ok) continue;
        for (int k = i; k + 1 < j; ++k) order[k] = order[k + 1];      // s moves to j - 1
        order[j - 1] = s;
    }
    // MUL_MAT nodes that read the SAME src1 through quantized leaves of one type (q / k / v, gate / up) are brought together
    // behind the first of them -- they go down as one call (ggml_hip_compute_forward_mul_mat_multi).  A node moves up past
    // the nodes in between when it does not depend on them and nobody there touches what it reads or writes.
    auto groupable = [](const ggml_tensor *t, const ggml_tensor *first) {
        if (!t || t->op != GGML_OP_MUL_MAT || !t->src0 || !t->src1 || t->src0->op != GGML_OP_NONE) return false;
        const int ty = t->src0->type;
        if (!(ty == GGML_TYPE_Q4_0 || ty == GGML_TYPE_Q4_1 || ty == GGML_TYPE_Q5_0 || ty == GGML_TYPE_Q5_1 || ty == GGML_TYPE_Q8_0)) return false;
        // (up to 32 rows: beyond, every result is a megabyte going home by DMA beside the next node's kernel -- a 7B layer at batch 64:
        // 800 us node by node, 862 us with q / k / v finishing together; the library's group call itself takes up to 64 rows)
        if (t->src1->ne[1] > 32 || t->src1->ne[2] != 1 || t->src1->ne[3] != 1 || t->src0->ne[2] != 1 || t->src0->ne[3] != 1) return false;
        return !first || (t->src1 == first->src1 && ty == first->src0->type && t->src0->ne[0] == first->src0->ne[0] && t != first);
    };
    for (int i = 0; i + 1 < n_order; ++i) {
        ggml_tensor *first = order[i];
        if (!groupable(first, nullptr)) continue;
        int cnt = 1, pos = i + 1;
        for (int j = i + 1; j < n_order && j <= i + 16 && cnt < 4; ++j) {
            ggml_tensor *t = order[j];
            if (!groupable(t, first)) continue;
            bool ok = true;
            for (int k = pos; k < j && ok; ++k) {
                ggml_tensor *u = order[k];
                ok = u->op != GGML_OP_NONE && t->src0 != u && t->src1 != u && !overlaps(u, t->src0) && !overlaps(u, t->src1) && !overlaps(u, t) &&
                     !overlaps(u->src0, t) && !overlaps(u->src1, t);
            }
            if (!ok) continue;
            for (int k = j; k > pos; --k) order[k] = order[k - 1];
            order[pos++] = t;
            ++cnt;
        }
        i = pos - 1;
    }
    auto group_at = [&](int at, const ggml_tensor *src1, const ggml_tensor **w, ggml_tensor **d) {   // members from order[at] on
        int g = 0;
        while (g < 4 && at + g < n_order && groupable(order[at + g], nullptr) && order[at + g]->src1 == src1 &&
               (g == 0 || (order[at + g]->src0->type == order[at]->src0->type && order[at + g]->src0->ne[0] == order[at]->src0->ne[0]))) {
            w[g] = order[at + g]->src0; d[g] = order[at + g]; ++g;
        }
        return g;
    };
    for (int i = 0; i < cgraph->n_nodes; i++) {
        ggml_tensor *node = order[i];
        if (node->op == GGML_OP_NONE) continue;
        ggml_compute_params params;
        params.ith = 0; params.nth = node->n_tasks; params.wsize = 0; params.wdata = nullptr;
        // a node and the node right behind it that consumes it go to the device as ONE call when the
        // library has the fused form (both nodes' data are still produced; the unfused seams are the fallback inside).
        ggml_tensor *next = i + 1 < cgraph->n_nodes ? order[i + 1] : nullptr;
        // the longest form first: rms_norm, mul, mul_mat [, add] -- the pre-projection chain of a transformer block -- goes down
        // as one call (one LAUNCH for decode-sized batches; the library splits it into pair + mul_mat otherwise)
        if (node->op == GGML_OP_RMS_NORM && i + 2 < cgraph->n_nodes) {
            ggml_tensor *mul = order[i + 1], *mm = order[i + 2];
            ggml_tensor *add = i + 3 < cgraph->n_nodes ? order[i + 3] : nullptr;
            const bool pair = mul->op == GGML_OP_MUL && (mul->src0 == node || mul->src1 == node) && mul->src0 != mul->src1 && mul->data != node->data;
            if (pair && mm->op == GGML_OP_MUL_MAT && mm->src1 == mul && mm->src0 != mul && mm->src0 != node) {
                const ggml_tensor *gw[4];
                ggml_tensor *gd[4];
                const int gn = group_at(i + 2, mul, gw, gd);
                if (gn >= 2) {                        // the pair and every projection that reads it: one call
                    ggml_tensor *g = mul->src0 == node ? mul->src1 : mul->src0;
                    for (int phase = GGML_TASK_INIT; phase <= GGML_TASK_FINALIZE; ++phase) {
                        params.type = phase;
                        int rc = ggml_hip_compute_forward_mul_mat_multi(&params, gn, gw, mul, gd, node->src0, g, node);
                        if (rc != GGML_HIP_OK) { (void)ggml_hip_graph_end(); return rc; }
                    }
                    node->perf_runs++; mul->perf_runs++;
                    for (int k = 0; k < gn; ++k) gd[k]->perf_runs++;
                    i += 1 + gn;
                    continue;
                }
                const bool with_add = add && add->op == GGML_OP_ADD && add->src0 == mm && add->src1 != mm && add->src1 != mul && add->src1 != node &&
                                      add->src1->type == GGML_TYPE_F32;
                ggml_tensor *g = mul->src0 == node ? mul->src1 : mul->src0;
                for (int phase = GGML_TASK_INIT; phase <= GGML_TASK_FINALIZE; ++phase) {
                    params.type = phase;
                    int rc = ggml_hip_compute_forward_norm_mul_mat(&params, node->src0, g, node, mul, mm->src0, mm, with_add ? add->src1 : nullptr,
                                                                   with_add ? add : nullptr);
                    if (rc != GGML_HIP_OK) { (void)ggml_hip_graph_end(); return rc; }
                }
                node->perf_runs++; mul->perf_runs++; mm->perf_runs++;
                if (with_add) add->perf_runs++;
                i += with_add ? 3 : 2;
                continue;
            }
        }
        if (node->op == GGML_OP_MUL_MAT) {
            const ggml_tensor *gw[4];
            ggml_tensor *gd[4];
            const int gn = group_at(i, node->src1, gw, gd);
            if (gn >= 2) {
                for (int phase = GGML_TASK_INIT; phase <= GGML_TASK_FINALIZE; ++phase) {
                    params.type = phase;
                    int rc = ggml_hip_compute_forward_mul_mat_multi(&params, gn, gw, node->src1, gd, nullptr, nullptr, nullptr);
                    if (rc != GGML_HIP_OK) { (void)ggml_hip_graph_end(); return rc; }
                }
                for (int k = 0; k < gn; ++k) gd[k]->perf_runs++;
                i += gn - 1;
                continue;
            }
        }
        int fused = 0;   // 1 rms_norm + mul, 2 silu + mul, 3 mul_mat + add, 4 mul_mat + scale
        if (next && next->op == GGML_OP_MUL && (next->src0 == node || next->src1 == node) && next->src0 != next->src1 && next->data != node->data) {
            if (node->op == GGML_OP_RMS_NORM) fused = 1;
            else if (node->op == GGML_OP_SILU && node->data != node->src0->data) fused = 2;
        } else if (next && node->op == GGML_OP_MUL_MAT && next->op == GGML_OP_ADD && next->src0 == node && next->src1 != node &&
                   next->src1->type == GGML_TYPE_F32) {
            fused = 3;
        } else if (next && node->op == GGML_OP_MUL_MAT && next->op == GGML_OP_SCALE && next->src0 == node && next->data == node->data) {
            fused = 4;
        }
        if (fused) {
            ggml_tensor *other = next->src0 == node ? next->src1 : next->src0;
            for (int phase = GGML_TASK_INIT; phase <= GGML_TASK_FINALIZE; ++phase) {
                params.type = phase;
                int rc;
                if (fused == 1) rc = ggml_hip_compute_forward_rms_norm_mul(&params, node->src0, other, node, next);
                else if (fused == 2) rc = ggml_hip_compute_forward_silu_mul(&params, node->src0, other, node, next);
                else if (fused == 3) rc = ggml_hip_compute_forward_mul_mat_add(&params, node->src0, node->src1, node, next->src1, next);
                else rc = ggml_hip_compute_forward_mul_mat_scale(&params, node->src0, node->src1, node, next->src1, next);
                if (rc != GGML_HIP_OK) { (void)ggml_hip_graph_end(); return rc; }
            }
            node->perf_runs++;
            next->perf_runs++;
            ++i;
            continue;
        }
        for (int phase = GGML_TASK_INIT; phase <= GGML_TASK_FINALIZE; ++phase) {  // Ggml.cs:3553-3670
            params.type = phase;
            int rc;
            if (node->op == GGML_OP_MUL_MAT) rc = ggml_hip_compute_forward_mul_mat(&params, node->src0, node->src1, node);
            else if (node->op == GGML_OP_CPY) rc = ggml_hip_compute_forward_cpy(&params, node->src0, node);   // Ggml.cs:8659-8663
            else if (node->op == GGML_OP_MUL) rc = ggml_hip_compute_forward_mul(&params, node->src0, node->src1, node);       // Ggml.cs:8569-8573
            else if (node->op == GGML_OP_SCALE) rc = ggml_hip_compute_forward_scale(&params, node->src0, node->src1, node);   // Ggml.cs:8654-8658
            else if (node->op == GGML_OP_SILU) rc = ggml_hip_compute_forward_silu(&params, node->src0, node);                 // Ggml.cs:8634-8638
            else if (node->op == GGML_OP_RMS_NORM) rc = ggml_hip_compute_forward_rms_norm(&params, node->src0, node);         // Ggml.cs:8644-8648
            else rc = ggml_hip_compute_forward_add(&params, node->src0, node->src1, node);                    // Ggml.cs:8566-8570
            if (rc != GGML_HIP_OK) { (void)ggml_hip_graph_end(); return rc; }
        }
        node->perf_runs++;
    }
    cgraph->perf_runs++;
    return ggml_hip_graph_end();
}

}  // extern "C"
