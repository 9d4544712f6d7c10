#!/usr/bin/env python3
"""The drop-in decode layer (bench.py dropin_decode_layer) with the host mirror's scopes NAMED (observed -> captured -> replayed: the
product's behaviour) against a mirror built with -DGGML_MIRROR_PLAIN_SCOPE (every graph issued live on the stream, never captured):
what does the replay's fixed cost (24.6 us per graph compute, tools/experiments/node_cost.py) buy or cost at 8 nodes per graph?
usage: python tools/experiments/layer_plain_vs_replay.py [plain]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ggmlsharp_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == "plain":
    _lib.MIRROR_PATH = os.path.join(_lib.PKG_DIR, "lib", "dbg", "libggml_hostmirror_plain.so")   # (a variant of tests/support/ggml_host.cpp built by hand)
import bench
from ggmlsharp_amd import device
device.init(0)
for N in (1, 4):
    r = bench.dropin_decode_layer(N, 200)
    print(sys.argv[1:] or ["named"], "batch", N, r["us_per_graph"], "us per graph (p10", r["p10_us"], "p90", r["p90_us"], ")", r["named_scopes"], flush=True)
