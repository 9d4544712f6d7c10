#!/usr/bin/env python3
"""Batched-decode sizes with the weights cold in HBM (bench.py's side_config: 32 weight copies taken in turn inside one replayed
hipGraph): whole call and COMPUTE kernel per shape.  usage: python tools/small_batch_time.py [M:K:N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(4096, 4096, n) for n in (9, 16, 32, 64, 128)] + [(11008, 4096, 32), (4096, 11008, 32)]
for (M, K, N) in shapes:
    r = bench.side_config(device, M, K, N, copies=32 if M * K <= 4096 * 11008 else 8, iters=60)
    print(f"{M:6d} x {K:6d} x {N:4d}: call {r['ms_per_step'] * 1e3:7.2f} us, compute kernel {(r.get('compute_kernel_ms') or 0) * 1e3:7.2f} us, "
          f"{r['algorithmic_GBs']:7.1f} GB/s", flush=True)
