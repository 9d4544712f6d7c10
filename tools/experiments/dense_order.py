#!/usr/bin/env python3
"""dense F16 / F32 4096^3 through bench.py's own dense_config, cold and again after a second of quantized side configs (does the place in
bench.py's sequence move the number?).  Developer tool, GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402
device.init(0)
for rnd in range(2):
    for wt in (1, 0):
        r = bench.dense_config(device, wt, 4096, 4096, 4096, iters=20 if wt else 5)
        print("round", rnd, "F16" if wt else "F32", r["ms_per_step"], flush=True)
    if rnd == 0:
        for (M, K, N, t) in ((4096, 11008, 1024, 8), (4096, 11008, 2048, 113), (4096, 11008, 512, 112), (32000, 4096, 512, 2)):
            r = bench.side_config(device, M, K, N, copies=3, iters=40, qtype=t)
            print("  side", M, K, N, t, r["ms_per_step"], flush=True)
