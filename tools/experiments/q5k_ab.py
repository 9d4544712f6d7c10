#!/usr/bin/env python3
"""Q5_K (and its Q5_0 stand-in) at config 4's size through bench.py's own side_config, for the library GGML_HIP_LIB selects:
same-box A/B of min-term forms.  (developer tool, GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ggmlsharp_amd import device
device.init(0)
for name, qt in (("Q5_K", bench.Q5_K), ("Q5_0", bench.Q5_0), ("Q5_K", bench.Q5_K), ("Q5_0", bench.Q5_0)):
    r = bench.side_config(device, 4096, 11008, 512, copies=8, iters=60, qtype=qt)
    print(f"{os.environ.get('GGML_HIP_LIB', 'product').split('_')[-1]:14s} {name}: step {r['ms_per_step'] * 1e3:6.1f} us  kernel {r['roofline'].get('kernel_ms', 0) * 1e3:6.1f} us", flush=True)
