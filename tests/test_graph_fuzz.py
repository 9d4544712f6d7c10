"""GPU test (-m gpu): random graphs through the host mirror's ggml_graph_compute -- which reorders nodes (the silu hoist, the
grouping of projections that share their input), fuses neighbours, and observes / captures / replays the
named scope -- against the SAME graph run node by node through the single seams in the graph's own order (Ggml.cs:7559-7619
builds that order, 3553-3670 runs it).  Every node's host data must be identical, bit for bit, on every compute.

The generator mixes free-standing random nodes with the shapes the mirror looks for (q / k / v groups, rms_norm -> mul ->
projection chains, gated feed-forwards, mul_mat -> add / scale), in-place scale views included, so that the dependency checks
of the reordering passes see aliases as well as plain producer / consumer edges."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import ggml_mirror as G
from ggmlsharp_amd import _lib

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

WIDTHS = (64, 96, 128, 256)
TYPES = (G.Q4_0, G.Q4_1, G.Q5_0, G.Q8_0)


@pytest.fixture(scope="module")
def dev():
    from ggmlsharp_amd import device
    device.init(0)
    return device


class _Builder:
    def __init__(self, ctx, rng, N):
        self.ctx, self.rng, self.N = ctx, rng, N
        self.pool = {w: [] for w in WIDTHS}     # f32 tensors [w, N] by width
        self.leaves = []                        # (tensor, width) of the f32 leaves: new contents before every compute
        self.weights = {}
        self.used = set()                       # tensors somebody consumes
        self.made = []                          # every op node, in creation order
        for w in WIDTHS:
            for _ in range(2):
                self.pool[w].append(self.leaf(w))
        self.scalars = []
        for v in (0.5, -1.25):
            s = G.ggml_new_tensor_1d(ctx, G.F32, 1)
            G.tensor_f32(s)[:] = v
            self.scalars.append(s)

    def leaf(self, w):
        t = G.ggml_new_tensor_2d(self.ctx, G.F32, w, self.N)
        self.leaves.append((t, w))
        return t

    def weight(self, ty, K, M, fresh=False):
        key = (ty, K, M)
        if fresh or key not in self.weights:
            t = G.ggml_new_tensor_2d(self.ctx, ty, K, M)
            G.tensor_bytes(t)[:] = O.quantize_row(ty, (self.rng.standard_normal((M, K)) * 0.3).astype(np.float32)).reshape(-1)
            if fresh:
                return t
            self.weights[key] = t
        return self.weights[key]

    def pick(self, w=None):
        if w is None:
            w = int(self.rng.choice(WIDTHS))
        ts = self.pool[w]
        # recent tensors more often than old ones: chains, not a star around the leaves
        i = len(ts) - 1 - int(min(self.rng.geometric(0.45) - 1, len(ts) - 1))
        return ts[i], w

    def add_node(self, t, w, *srcs):
        self.pool[w].append(t)
        self.made.append(t)
        for s in srcs:
            self.used.add(C.addressof(s.contents))
        return t

    def mm(self, a, K, M=None, ty=None, fresh=False):
        M = M or int(self.rng.choice(WIDTHS))
        ty = ty if ty is not None else TYPES[int(self.rng.integers(len(TYPES)))]
        return self.add_node(G.ggml_mul_mat(self.ctx, self.weight(ty, K, M, fresh), a), M, a), M

    def step(self):
        r = self.rng.random()
        ctx = self.ctx
        if r < 0.14:                                             # q / k / v: 2..4 projections of one input, one type
            a, K = self.pick()
            ty = TYPES[int(self.rng.integers(len(TYPES)))]
            for _ in range(int(self.rng.integers(2, 5))):
                self.mm(a, K, ty=ty, fresh=True)
        elif r < 0.26:                                           # rms_norm -> mul -> 1..3 projections [-> add]
            a, K = self.pick()
            g, _ = self.pick(K)
            n = self.add_node(G.ggml_rms_norm(ctx, a), K, a)
            y = self.add_node(G.ggml_mul(ctx, n, g) if self.rng.random() < 0.7 else G.ggml_mul(ctx, g, n), K, n, g)
            ty = TYPES[int(self.rng.integers(len(TYPES)))]
            for _ in range(int(self.rng.integers(1, 4))):
                o, M = self.mm(y, K, ty=ty, fresh=True)
            if self.rng.random() < 0.5:
                b, _ = self.pick(M)
                if b is not o:
                    self.add_node(G.ggml_add(ctx, o, b), M, o, b)
        elif r < 0.38:                                           # gated feed-forward
            a, K = self.pick()
            F = int(self.rng.choice(WIDTHS))
            ty = TYPES[int(self.rng.integers(len(TYPES)))]
            u, _ = self.mm(a, K, F, ty, fresh=True)
            gt, _ = self.mm(a, K, F, ty, fresh=True)
            s = self.add_node(G.ggml_silu(ctx, u), F, u)
            p = self.add_node(G.ggml_mul(ctx, s, gt), F, s, gt)
            o, M = self.mm(p, F)
            if self.rng.random() < 0.6:
                b, _ = self.pick(M)
                if b is not o:
                    self.add_node(G.ggml_add(ctx, o, b), M, o, b)
        elif r < 0.50:
            a, K = self.pick()
            self.mm(a, K)
        elif r < 0.62:
            a, w = self.pick()
            b, _ = self.pick(w)
            self.add_node(G.ggml_add(ctx, a, b), w, a, b)
        elif r < 0.72:
            a, w = self.pick()
            b, _ = self.pick(w)
            if a is not b:
                self.add_node(G.ggml_mul(ctx, a, b), w, a, b)
        elif r < 0.80:
            a, w = self.pick()
            self.add_node(G.ggml_rms_norm(ctx, a), w, a)
        elif r < 0.88:
            a, w = self.pick()
            self.add_node(G.ggml_silu(ctx, a), w, a)
        else:                                                    # scale: a VIEW of its input (Ggml.cs:8264-8265), op nodes only
            a, w = self.pick()
            if any(a is m for m in self.made):
                s = self.scalars[int(self.rng.integers(len(self.scalars)))]
                self.add_node(G.ggml_scale(ctx, a, s), w, a)

    def root(self):
        """one tensor that depends on every node nobody consumed yet"""
        acc = None
        for t in list(self.made):
            if C.addressof(t.contents) in self.used:
                continue
            w = int(t.contents.ne[0])
            if w != 64:
                t, _ = self.mm(t, w, 64)
            acc = t if acc is None else self.add_node(G.ggml_add(self.ctx, acc, t), 64, acc, t)
        return acc

    def new_leaf_data(self, rng):
        for t, w in self.leaves:
            G.tensor_f32(t)[:] = rng.standard_normal((self.N, w)).astype(np.float32).reshape(1, 1, self.N, w)
            _lib.lib().ggml_hip_invalidate_range(t.contents.data, self.N * w * 4)


def _node_by_node(gf):
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
        _lib.check(rc, f"node {i} (op {op})")


def _snapshot(gf):
    return [np.array(G.tensor_f32(gf.nodes[i]), copy=True) for i in range(gf.n_nodes)]


_BATCHES = (1, 1, 2, 3, 4, 4, 6, 9, 20, 40, 150)
_SEEN = []          # per graph: named scopes (observed, captured, replayed, refused) over its five computes


def _counters():
    v = [C.c_uint64() for _ in range(4)]
    _lib.lib().ggml_hip_debug_scope_counters(*[C.byref(x) for x in v])
    return tuple(int(x.value) for x in v)


@pytest.mark.parametrize("seed,N", [(s, _BATCHES[s % len(_BATCHES)]) for s in range(1, 67)])
def test_random_graph_matches_its_nodes_one_by_one(dev, seed, N):
    rng = np.random.default_rng(1000 + seed)
    ctx = G.ggml_init(256 * 1024 * 1024)
    try:
        b = _Builder(ctx, rng, N)
        for _ in range(int(rng.integers(8, 22))):
            b.step()
        root = b.root()
        if root is None:
            pytest.skip("the generator made no node")
        gf = G.ggml_build_forward(root)
        assert 0 < gf.n_nodes < 4000
        ops = [gf.nodes[i].contents.op for i in range(gf.n_nodes)]
        c0 = _counters()
        for it in range(5):
            data = np.random.default_rng(50 * seed + it)
            b.new_leaf_data(data)
            G.ggml_graph_compute(ctx, gf)
            got = _snapshot(gf)
            b.new_leaf_data(np.random.default_rng(50 * seed + it))        # (in-place nodes may have rewritten what they view)
            _node_by_node(gf)
            ref = _snapshot(gf)
            for i, (x, y) in enumerate(zip(got, ref)):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), \
                    f"seed {seed}, N {N}, compute {it}: node {i} of {gf.n_nodes} (op {ops[i]}) differs; ops = {ops}"
        c1 = _counters()
        _SEEN.append(tuple(b_ - a_ for a_, b_ in zip(c0, c1)))
    finally:
        G.ggml_free(ctx)


def test_the_random_graphs_were_captured_and_replayed(dev):
    """(runs after the graphs above) the comparison is only worth its name if the computes it checked went through the
    captured / replayed path too, not just the live one"""
    if not _SEEN:
        pytest.skip("no graph ran")
    tot = [sum(x[k] for x in _SEEN) for k in range(4)]
    print("named scopes over the random graphs: observed %d, captured %d, replayed %d, refused %d" % tuple(tot))
    assert tot[1] >= len(_SEEN) // 2 and tot[2] >= len(_SEEN), tot
