#!/usr/bin/env python3
"""Developer soak test of the replayed graph scope: thousands of computes of one decoder-layer-shaped graph with new leaf contents
each time; every 250th compute every node is compared bit for bit with the node-by-node seams."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_graph_scope as T  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "support"))
import ggml_mirror as G  # noqa: E402  (test support: the host mirror)
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.default_rng(5)
ctx = G.ggml_init(128 << 20)
gf, (x, g1, g2, S), nodes = T._layer(ctx, rng, 512, 384, 640, 2)
bad = 0
for it in range(n):
    G.tensor_f32(x)[:] = rng.standard_normal((2, 512)).astype(np.float32).reshape(1, 1, 2, 512)
    G.ggml_graph_compute(ctx, gf)
    if it % 250 == 249:
        got = T._snapshot(nodes)
        T._node_by_node(gf)
        for i, (a, b) in enumerate(zip(got, T._snapshot(nodes))):
            if not np.array_equal(a, b):
                bad += 1
                print("MISMATCH at compute", it, "node", i, flush=True)
        print(it + 1, "computes, bad", bad, T._counters(), flush=True)
G.ggml_free(ctx)
print("bad:", bad)
sys.exit(1 if bad else 0)
