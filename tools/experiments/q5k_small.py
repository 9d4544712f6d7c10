#!/usr/bin/env python3
"""Q5_K at decode and batched-decode sizes through bench.py's own side_config (developer tool, GPU box; GGML_HIP_LIB selects a variant build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ggmlsharp_amd import device
device.init(0)
for qt, name in ((bench.Q5_K, "Q5_K"), (bench.Q5_0, "Q5_0")):
    for (M, K, N) in ((4096, 4096, 1), (4096, 4096, 4), (4096, 4096, 8), (4096, 4096, 16), (4096, 4096, 64), (4096, 11008, 1), (11008, 4096, 1)):
        r = bench.side_config(device, M, K, N, copies=8, iters=100, qtype=qt)
        print(f"{name} {M}x{K}x{N}: step {r['ms_per_step'] * 1e3:6.1f} us", flush=True)
