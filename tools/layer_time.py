#!/usr/bin/env python3
"""Drop-in decode latency: bench.py's dropin_decode_layer (one LLaMA-7B-shaped decoder layer through ggml_graph_compute of the
host mirror, host tensors in a registered pool) at a batch size given on the command line.
usage: python tools/layer_time.py [N]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print(json.dumps(bench.dropin_decode_layer(N, 100)))
