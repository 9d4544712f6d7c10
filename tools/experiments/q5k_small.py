#!/usr/bin/env python3
"""Q5_K at batched-decode sizes through bench.py's own side_config (developer tool, GPU box; GGML_HIP_LIB selects a variant build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ggmlsharp_amd import device
device.init(0)
for (M, K, N) in ((4096, 4096, 16), (4096, 4096, 32), (4096, 4096, 64), (4096, 11008, 32), (11008, 4096, 64)):
    r = bench.side_config(device, M, K, N, copies=8, iters=100, qtype=bench.Q5_K)
    print(f"Q5_K {M}x{K}x{N}: step {r['ms_per_step'] * 1e3:6.1f} us", flush=True)
