#!/usr/bin/env python3
"""Seam 1 with host tensors (the drop-in path): ms per ggml_graph_compute, next to this box's PCIe rates one way and both ways."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402


def bidir_probe():
    n = 64 << 20
    h1 = torch.empty(n, dtype=torch.uint8).pin_memory()
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d1 = torch.empty(n, dtype=torch.uint8, device="cuda")
    d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        s1.wait_event(a)
        s2.wait_event(a)
        with torch.cuda.stream(s1):
            d1.copy_(h1, non_blocking=True)
        with torch.cuda.stream(s2):
            h2.copy_(d2, non_blocking=True)
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    t = float(np.median(ts))
    return {"both_ways_GBs_total": round(2 * n / t / 1e6, 1), "ms_for_64MiB_each_way": round(t, 3)}


if __name__ == "__main__":
    device.init(0)
    print(json.dumps({"one_way": bench.pcie_probe(), "both_ways": bidir_probe()}))
    for (M, K, N, it) in ((4096, 4096, 4096, 10), (4096, 4096, 512, 30), (4096, 4096, 128, 30), (11008, 4096, 512, 20)):
        print(json.dumps(bench.seam1_host_config(M, K, N, it)), flush=True)
