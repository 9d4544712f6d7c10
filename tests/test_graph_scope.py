"""GPU tests (-m gpu) of the graph scope's host side (include/ggml_hip.h ggml_hip_graph_begin / _begin_keyed): results owed to the
host and paid together, named scopes captured and replayed -- every run must leave every node's data in host memory, equal bit
for bit to the node-by-node seams, whatever the scope did behind the scenes."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ggml_mirror as G
from ggmlsharp_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    from ggmlsharp_amd import device
    device.init(0)
    return device


def _counters():
    v = [C.c_uint64() for _ in range(4)]
    _lib.lib().ggml_hip_debug_scope_counters(*[C.byref(x) for x in v])
    return tuple(int(x.value) for x in v)        # observed, captured, replayed, refused


def _layer(ctx, rng, K, M, F, N):
    """a decoder-layer-shaped graph of the ops on this path; returns (graph, leaves, every node in order)"""
    def f32(k, n, scale=1.0):
        t = G.ggml_new_tensor_2d(ctx, G.F32, k, n)
        G.tensor_f32(t)[:] = (rng.standard_normal((n, k)) * scale).astype(np.float32).reshape(1, 1, n, k)
        return t

    def q(ty, k, m):
        t = G.ggml_new_tensor_2d(ctx, ty, k, m)
        G.tensor_bytes(t)[:] = O.quantize_row(ty, rng.standard_normal((m, k)).astype(np.float32)).reshape(-1)
        return t

    x, g1, g2 = f32(K, N, 2.0), f32(K, N), f32(K, N)
    S = G.ggml_new_tensor_1d(ctx, G.F32, 1)
    G.tensor_f32(S)[:] = 0.25
    wq, wk, wo = q(G.Q4_0, K, M), q(G.Q8_0, K, M), q(G.Q4_0, M, K)
    w1, w3, w2 = q(G.Q4_0, K, F), q(G.Q4_0, K, F), q(G.Q5_0, F, K)
    cur = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, x), g1)
    qq, kk = G.ggml_mul_mat(ctx, wq, cur), G.ggml_mul_mat(ctx, wk, cur)
    a = G.ggml_scale(ctx, G.ggml_add(ctx, qq, kk), S)
    h = G.ggml_add(ctx, G.ggml_mul_mat(ctx, wo, a), x)
    cur2 = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, h), g2)
    u, gt = G.ggml_mul_mat(ctx, w1, cur2), G.ggml_mul_mat(ctx, w3, cur2)
    s = G.ggml_mul(ctx, G.ggml_silu(ctx, u), gt)
    out = G.ggml_add(ctx, G.ggml_mul_mat(ctx, w2, s), h)
    gf = G.ggml_build_forward(out)
    nodes = [gf.nodes[i] for i in range(gf.n_nodes)]
    return gf, (x, g1, g2, S), nodes


def _node_by_node(gf):
    """the same nodes through the single seams, each call on its own (no scope): the reference for every bit"""
    L = _lib.lib()
    p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
    for i in range(gf.n_nodes):
        n = gf.nodes[i].contents
        op = n.op
        if op == _lib.GGML_OP_MUL_MAT:
            rc = L.ggml_hip_compute_forward_mul_mat(C.byref(p), n.src0, n.src1, gf.nodes[i])
        elif op == _lib.GGML_OP_ADD:
            rc = L.ggml_hip_compute_forward_add(C.byref(p), n.src0, n.src1, gf.nodes[i])
        elif op == _lib.GGML_OP_MUL:
            rc = L.ggml_hip_compute_forward_mul(C.byref(p), n.src0, n.src1, gf.nodes[i])
        elif op == _lib.GGML_OP_SCALE:
            rc = L.ggml_hip_compute_forward_scale(C.byref(p), n.src0, n.src1, gf.nodes[i])
        elif op == _lib.GGML_OP_RMS_NORM:
            rc = L.ggml_hip_compute_forward_rms_norm(C.byref(p), n.src0, gf.nodes[i])
        elif op == _lib.GGML_OP_SILU:
            rc = L.ggml_hip_compute_forward_silu(C.byref(p), n.src0, gf.nodes[i])
        else:
            raise AssertionError(op)
        _lib.check(rc, f"node {i}")


def _snapshot(nodes):
    return [np.array(G.tensor_f32(t), copy=True) for t in nodes]


@pytest.mark.parametrize("N", [1, 3, 20])
def test_named_scope_is_captured_replayed_and_tracks_its_leaves(dev, N):
    """six computes of one graph with new leaf CONTENTS each time: observed, captured, then replayed -- every node's host
    data equals the node-by-node seams' bit for bit every time; then a weight is rewritten (invalidate) and it still does."""
    rng = np.random.default_rng(7 + N)
    K, M, F = 256, 192, 320
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        gf, (x, g1, g2, S), nodes = _layer(ctx, rng, K, M, F, N)
        c0 = _counters()
        for it in range(6):
            G.tensor_f32(x)[:] = (rng.standard_normal((N, K)) * (1 + it)).astype(np.float32).reshape(1, 1, N, K)
            _lib.lib().ggml_hip_invalidate_range(x.contents.data, N * K * 4)     # (what ggml_set_f32 does; x is no weight: nothing is dropped)
            G.ggml_graph_compute(ctx, gf)
            got = _snapshot(nodes)
            _node_by_node(gf)
            ref = _snapshot(nodes)
            for i, (a, b) in enumerate(zip(got, ref)):
                assert np.array_equal(a, b), (it, i)
        c1 = _counters()
        # (the first compute uploads the weights, the node-by-node calls in between grow a scratch buffer once: observed and
        # captured after that, replayed from then on)
        assert c1[0] - c0[0] >= 1 and c1[1] - c0[1] >= 1 and c1[2] - c0[2] >= 2 and c1[3] == c0[3], (c0, c1)
        # the scale factor is read on the host: a new value is a new key (observed again, not replayed with the old factor)
        G.tensor_f32(S)[:] = 0.5
        G.ggml_graph_compute(ctx, gf)
        got = _snapshot(nodes)
        _node_by_node(gf)
        for i, (a, b) in enumerate(zip(got, _snapshot(nodes))):
            assert np.array_equal(a, b), ("scale", i)
        c2 = _counters()
        assert c2[2] == c1[2], "a graph with another scale factor must not replay the captured one"
        G.tensor_f32(S)[:] = 0.25
        # a weight rewritten in place + invalidate: the captured scope (which holds the old device copy) is dropped
        w = nodes[2].contents.src0                                   # wq
        wb = G.tensor_bytes(w)
        wb[:] = O.quantize_row(G.Q4_0, rng.standard_normal((M, K)).astype(np.float32)).reshape(-1)
        _lib.lib().ggml_hip_invalidate(w.contents.data)
        for it in range(4):
            G.ggml_graph_compute(ctx, gf)
            got = _snapshot(nodes)
            _node_by_node(gf)
            for i, (a, b) in enumerate(zip(got, _snapshot(nodes))):
                assert np.array_equal(a, b), ("rewritten weight", it, i)
        c3 = _counters()
        assert c3[1] - c2[1] == 1 and c3[2] - c2[2] >= 1, (c2, c3)
    finally:
        G.ggml_free(ctx)


def test_plain_scope_pays_everything_it_owes(dev):
    """ggml_hip_graph_begin (no key): one launch per node, the host copies together at the end -- including a tensor that is
    overwritten in place (scale) and a leaf that two nodes read."""
    rng = np.random.default_rng(3)
    ctx = G.ggml_init(64 * 1024 * 1024)
    try:
        gf, leaves, nodes = _layer(ctx, rng, 128, 96, 160, 2)
        L = _lib.lib()
        p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
        _lib.check(L.ggml_hip_graph_begin(), "begin")
        for i in range(gf.n_nodes):
            n = gf.nodes[i].contents
            f = {_lib.GGML_OP_MUL_MAT: L.ggml_hip_compute_forward_mul_mat, _lib.GGML_OP_ADD: L.ggml_hip_compute_forward_add,
                 _lib.GGML_OP_MUL: L.ggml_hip_compute_forward_mul, _lib.GGML_OP_SCALE: L.ggml_hip_compute_forward_scale}.get(n.op)
            if f is not None:
                _lib.check(f(C.byref(p), n.src0, n.src1, gf.nodes[i]), f"node {i}")
            elif n.op == _lib.GGML_OP_RMS_NORM:
                _lib.check(L.ggml_hip_compute_forward_rms_norm(C.byref(p), n.src0, gf.nodes[i]), f"node {i}")
            else:
                _lib.check(L.ggml_hip_compute_forward_silu(C.byref(p), n.src0, gf.nodes[i]), f"node {i}")
        _lib.check(L.ggml_hip_graph_end(), "end")
        got = _snapshot(nodes)
        _node_by_node(gf)
        for i, (a, b) in enumerate(zip(got, _snapshot(nodes))):
            assert np.array_equal(a, b), i
    finally:
        G.ggml_free(ctx)


def test_host_read_inside_a_scope_pays_what_is_owed(dev):
    """a node the host computes itself between offloaded nodes: ggml_hip_host_read makes its source current, and what it
    writes (followed by ggml_hip_invalidate_range) is what the next offloaded node uploads."""
    rng = np.random.default_rng(11)
    K, N = 256, 3
    ctx = G.ggml_init(16 * 1024 * 1024)
    try:
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        Y = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        x = rng.standard_normal((N, K)).astype(np.float32)
        y = rng.standard_normal((N, K)).astype(np.float32)
        G.tensor_f32(X)[:] = x.reshape(1, 1, N, K)
        G.tensor_f32(Y)[:] = y.reshape(1, 1, N, K)
        t_add = G.ggml_add(ctx, X, Y)
        t_mul = G.ggml_mul(ctx, t_add, Y)
        G.tensor_f32(t_add)[:] = -7.0                       # poison: the host copy must really be written
        L = _lib.lib()
        p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
        _lib.check(L.ggml_hip_graph_begin(), "begin")
        a = t_add.contents
        _lib.check(L.ggml_hip_compute_forward_add(C.byref(p), a.src0, a.src1, t_add), "add")
        _lib.check(L.ggml_hip_host_read(a.data, N * K * 4), "host read")
        got = np.array(G.tensor_f32(t_add)[0, 0], copy=True)
        assert np.array_equal(got, x + y)
        G.tensor_f32(t_add)[:] = (got * 2).reshape(1, 1, N, K)          # the "CPU node": in place
        L.ggml_hip_invalidate_range(a.data, N * K * 4)
        m = t_mul.contents
        _lib.check(L.ggml_hip_compute_forward_mul(C.byref(p), m.src0, m.src1, t_mul), "mul")
        _lib.check(L.ggml_hip_graph_end(), "end")
        assert np.array_equal(G.tensor_f32(t_mul)[0, 0], (x + y) * 2 * y)
    finally:
        G.ggml_free(ctx)


def test_a_freed_pool_takes_its_captured_scopes_with_it(dev):
    """ggml_free + ggml_init usually hand out the same addresses again, i.e. the same key: the captured scope of the old pool (its
    device mapping, its resident buffers) must be gone -- the new graph is observed and captured afresh and is right."""
    rng = np.random.default_rng(21)
    for round_ in range(3):
        ctx = G.ggml_init(32 * 1024 * 1024)
        try:
            gf, (x, g1, g2, S), nodes = _layer(ctx, rng, 128, 96, 160, 1)
            for it in range(4):
                G.tensor_f32(x)[:] = rng.standard_normal((1, 128)).astype(np.float32).reshape(1, 1, 1, 128)
                G.ggml_graph_compute(ctx, gf)
                got = _snapshot(nodes)
                _node_by_node(gf)
                for i, (a, b) in enumerate(zip(got, _snapshot(nodes))):
                    assert np.array_equal(a, b), (round_, it, i)
        finally:
            G.ggml_free(ctx)


def test_two_stacked_layers_with_shared_projections(dev):
    """two decoder-layer-shaped blocks in one graph (the second reads the first's output), three projections reading one
    tensor in each: the host mirror's reordering (projections brought together, SILU moved to its consumer) and the fused
    calls it makes must leave every node's data equal to the node-by-node seams', through observe / capture / replay."""
    rng = np.random.default_rng(31)
    K, M, F, N = 256, 256, 384, 2
    ctx = G.ggml_init(96 * 1024 * 1024)
    try:
        def f32(k, n, scale=1.0):
            t = G.ggml_new_tensor_2d(ctx, G.F32, k, n)
            G.tensor_f32(t)[:] = (rng.standard_normal((n, k)) * scale).astype(np.float32).reshape(1, 1, n, k)
            return t

        def q(ty, k, m):
            t = G.ggml_new_tensor_2d(ctx, ty, k, m)
            G.tensor_bytes(t)[:] = O.quantize_row(ty, rng.standard_normal((m, k)).astype(np.float32)).reshape(-1)
            return t

        x = f32(K, N, 2.0)
        h = x
        for layer in range(2):
            g1, g2 = f32(K, N), f32(K, N)
            wq, wk, wv, wo = q(G.Q4_0, K, M), q(G.Q4_0, K, M), q(G.Q4_0, K, M), q(G.Q4_0, M, K)
            w1, w3, w2 = q(G.Q8_0, K, F), q(G.Q8_0, K, F), q(G.Q4_1, F, K)
            cur = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, h), g1)
            qq, kk, vv = G.ggml_mul_mat(ctx, wq, cur), G.ggml_mul_mat(ctx, wk, cur), G.ggml_mul_mat(ctx, wv, cur)
            a = G.ggml_add(ctx, G.ggml_mul(ctx, qq, kk), vv)             # (stand-in for the attention: the projections' consumers interleave)
            h = G.ggml_add(ctx, G.ggml_mul_mat(ctx, wo, a), h)
            cur2 = G.ggml_mul(ctx, G.ggml_rms_norm(ctx, h), g2)
            u, gt = G.ggml_mul_mat(ctx, w1, cur2), G.ggml_mul_mat(ctx, w3, cur2)
            s = G.ggml_mul(ctx, G.ggml_silu(ctx, u), gt)
            h = G.ggml_add(ctx, G.ggml_mul_mat(ctx, w2, s), h)
        gf = G.ggml_build_forward(h)
        nodes = [gf.nodes[i] for i in range(gf.n_nodes)]
        assert gf.n_nodes == 34
        for it in range(5):
            G.tensor_f32(x)[:] = (rng.standard_normal((N, K)) * (1 + it)).astype(np.float32).reshape(1, 1, N, K)
            G.ggml_graph_compute(ctx, gf)
            got = _snapshot(nodes)
            _node_by_node(gf)
            for i, (a_, b_) in enumerate(zip(got, _snapshot(nodes))):
                assert np.array_equal(a_, b_), (it, i)
    finally:
        G.ggml_free(ctx)


@pytest.mark.parametrize("N", [2, 300])
def test_outputs_only_scope_copies_what_was_asked_for(dev, N):
    """ggml_hip_graph_outputs (opt-in, not the reference's contract): the named tensor is right, the others' host memory is
    left alone, and far fewer bytes cross PCIe."""
    rng = np.random.default_rng(41)
    ctx = G.ggml_init(256 * 1024 * 1024)
    try:
        gf, leaves, nodes = _layer(ctx, rng, 256, 192, 320, N)
        _node_by_node(gf)
        ref = _snapshot(nodes)
        for t in nodes:
            G.tensor_f32(t)[:] = -9.0
        L = _lib.lib()
        p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
        c0 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
        c1 = [C.c_uint64(), C.c_uint64(), C.c_uint64()]
        L.ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c0])
        _lib.check(L.ggml_hip_graph_begin(), "begin")
        want = (C.c_void_p * 1)(nodes[-1].contents.data)
        _lib.check(L.ggml_hip_graph_outputs(want, 1), "outputs")
        for i in range(gf.n_nodes):
            n = gf.nodes[i].contents
            f = {_lib.GGML_OP_MUL_MAT: L.ggml_hip_compute_forward_mul_mat, _lib.GGML_OP_ADD: L.ggml_hip_compute_forward_add,
                 _lib.GGML_OP_MUL: L.ggml_hip_compute_forward_mul, _lib.GGML_OP_SCALE: L.ggml_hip_compute_forward_scale}.get(n.op)
            if f is not None:
                _lib.check(f(C.byref(p), n.src0, n.src1, gf.nodes[i]), f"node {i}")
            elif n.op == _lib.GGML_OP_RMS_NORM:
                _lib.check(L.ggml_hip_compute_forward_rms_norm(C.byref(p), n.src0, gf.nodes[i]), f"node {i}")
            else:
                _lib.check(L.ggml_hip_compute_forward_silu(C.byref(p), n.src0, gf.nodes[i]), f"node {i}")
        _lib.check(L.ggml_hip_graph_end(), "end")
        L.ggml_hip_debug_transfer_counters(*[C.byref(c) for c in c1])
        got = _snapshot(nodes)
        assert np.array_equal(got[-1], ref[-1])
        untouched = sum(bool(np.all(g == -9.0)) for g in got[:-1])
        assert untouched >= len(nodes) - 3, untouched          # (an in-place view shares its parent's memory)
        assert c1[1].value - c0[1].value <= 2 * got[-1].nbytes
        # and the scope after it is an ordinary one again
        G.ggml_graph_compute(ctx, gf)
        for i, (a, b) in enumerate(zip(_snapshot(nodes), ref)):
            assert np.array_equal(a, b), i
    finally:
        G.ggml_free(ctx)


def test_an_unregistered_pool_is_never_captured(dev):
    """a context pool too small to be registered for DMA (pageable host memory): the graph is computed live every time."""
    rng = np.random.default_rng(51)
    ctx = G.ggml_init(48 * 1024)                      # below the mirror's registration threshold
    try:
        K, M, N = 64, 32, 2
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        Y = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        W = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
        G.tensor_bytes(W)[:] = O.quantize_row(G.Q4_0, rng.standard_normal((M, K)).astype(np.float32)).reshape(-1)
        G.tensor_f32(Y)[:] = rng.standard_normal((N, K)).astype(np.float32).reshape(1, 1, N, K)
        out = G.ggml_mul_mat(ctx, W, G.ggml_mul(ctx, X, Y))
        gf = G.ggml_build_forward(out)
        nodes = [gf.nodes[i] for i in range(gf.n_nodes)]
        c0 = _counters()
        for it in range(5):
            G.tensor_f32(X)[:] = rng.standard_normal((N, K)).astype(np.float32).reshape(1, 1, N, K)
            G.ggml_graph_compute(ctx, gf)
            got = _snapshot(nodes)
            _node_by_node(gf)
            for i, (a, b) in enumerate(zip(got, _snapshot(nodes))):
                assert np.array_equal(a, b), (it, i)
        c1 = _counters()
        assert c1[1] == c0[1] and c1[2] == c0[2], (c0, c1)
    finally:
        G.ggml_free(ctx)


def test_an_error_inside_a_named_scope_is_reported_every_time(dev):
    """a seam that fails inside a named scope (the host then runs its own code for that node) must
    keep failing -- five computes of one keyed scope with one node of a rejected type (Q4_3: a null slot in the reference,
    Ggml.cs:248) return the error five times; the scope is never captured or replayed, the good node's data is right every time."""
    L = _lib.lib()
    rng = np.random.default_rng(77)
    ctx = G.ggml_init(8 * 1024 * 1024)
    try:
        K, M, N = 128, 64, 3
        X = G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        W = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
        Wbad = G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M)
        for w in (W, Wbad):
            G.tensor_bytes(w)[:] = O.quantize_row(G.Q4_0, rng.standard_normal((M, K)).astype(np.float32)).reshape(-1)
        y1, y2 = G.ggml_mul_mat(ctx, W, X), G.ggml_mul_mat(ctx, Wbad, X)
        Wbad.contents.type = 5                                  # GGML_TYPE_Q4_3: no quantize_fns slot, the seam rejects it
        p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
        c0 = _counters()
        for it in range(5):
            G.tensor_f32(X)[:] = rng.standard_normal((N, K)).astype(np.float32).reshape(1, 1, N, K)
            L.ggml_hip_invalidate_range(X.contents.data, N * K * 4)
            G.tensor_f32(y1)[:] = -5.0
            _lib.check(L.ggml_hip_graph_begin_keyed(0xBADC0DE), "begin")
            rc1 = L.ggml_hip_compute_forward_mul_mat(C.byref(p), y1.contents.src0, y1.contents.src1, y1)
            rc2 = L.ggml_hip_compute_forward_mul_mat(C.byref(p), y2.contents.src0, y2.contents.src1, y2)
            _lib.check(L.ggml_hip_graph_end(), "end")
            assert rc1 == 0, (it, rc1)
            assert rc2 != 0, f"compute {it}: the rejected node reported success"
            ref = O.mul_mat(O.Q4_0, np.array(G.tensor_bytes(W)), np.array(G.tensor_f32(X)).reshape(N, K), M, K, N)[0, 0]
            got = np.array(G.tensor_f32(y1)).reshape(N, M)
            O.assert_mul_mat_close(got, ref, K, f"compute {it}")
        c1 = _counters()
        assert c1[1] == c0[1] and c1[2] == c0[2], f"a scope with a failing node was captured / replayed: {c0} -> {c1}"
    finally:
        G.ggml_free(ctx)


def test_a_foreign_threads_seam_during_a_capture_ends_it_and_both_results_are_right(dev):
    """thread B's seam arrives on a slot between two seams of thread A's CAPTURING scope.  The capture is ended by B (begun
    in relaxed mode: the ending thread need not be the beginning one) and issued live, B's product runs, A carries on live: every
    result is right every time, nothing of such a run is kept as a captured graph, and the stream is left usable."""
    import threading
    L = _lib.lib()
    rng = np.random.default_rng(99)
    ctx = G.ggml_init(16 * 1024 * 1024)
    try:
        K, M, N = 256, 96, 4
        X, Xb = G.ggml_new_tensor_2d(ctx, G.F32, K, N), G.ggml_new_tensor_2d(ctx, G.F32, K, N)
        Ws = [G.ggml_new_tensor_2d(ctx, G.Q4_0, K, M) for _ in range(3)]
        for w in Ws:
            G.tensor_bytes(w)[:] = O.quantize_row(G.Q4_0, rng.standard_normal((M, K)).astype(np.float32)).reshape(-1)
        y1, y2, yb = G.ggml_mul_mat(ctx, Ws[0], X), G.ggml_mul_mat(ctx, Ws[1], X), G.ggml_mul_mat(ctx, Ws[2], Xb)
        p = _lib.ggml_compute_params(_lib.GGML_TASK_COMPUTE, 0, 1, 0, None)
        c0 = _counters()
        for it in range(6):
            for t in (X, Xb):
                G.tensor_f32(t)[:] = rng.standard_normal((N, K)).astype(np.float32).reshape(1, 1, N, K)
                L.ggml_hip_invalidate_range(t.contents.data, N * K * 4)
            for y in (y1, y2, yb):
                G.tensor_f32(y)[:] = -7.0
            rcb = []

            def foreign():
                rcb.append(L.ggml_hip_compute_forward_mul_mat(C.byref(p), yb.contents.src0, yb.contents.src1, yb))

            _lib.check(L.ggml_hip_graph_begin_keyed(0xF0E1), "begin")
            rc1 = L.ggml_hip_compute_forward_mul_mat(C.byref(p), y1.contents.src0, y1.contents.src1, y1)
            th = threading.Thread(target=foreign)
            th.start()
            th.join()
            rc2 = L.ggml_hip_compute_forward_mul_mat(C.byref(p), y2.contents.src0, y2.contents.src1, y2)
            _lib.check(L.ggml_hip_graph_end(), f"end (compute {it})")
            assert rc1 == 0 and rc2 == 0 and rcb == [0], (it, rc1, rc2, rcb)
            xs = np.array(G.tensor_f32(X)).reshape(N, K)
            for y, w, xin, name in ((y1, Ws[0], xs, "A first"), (y2, Ws[1], xs, "A second"), (yb, Ws[2], np.array(G.tensor_f32(Xb)).reshape(N, K), "B")):
                ref = O.mul_mat(O.Q4_0, np.array(G.tensor_bytes(w)), xin, M, K, N)[0, 0]
                O.assert_mul_mat_close(np.array(G.tensor_f32(y)).reshape(N, M), ref, K, f"compute {it}, {name}")
        c1 = _counters()
        assert c1[1] == c0[1] and c1[2] == c0[2], f"a scope that a foreign seam cut into was kept as a graph: {c0} -> {c1}"
        # the slot is healthy afterwards: the same scope without the intruder still computes
        _lib.check(L.ggml_hip_graph_begin_keyed(0xF0E2), "begin")
        assert L.ggml_hip_compute_forward_mul_mat(C.byref(p), y1.contents.src0, y1.contents.src1, y1) == 0
        _lib.check(L.ggml_hip_graph_end(), "end")
    finally:
        G.ggml_free(ctx)
