#!/usr/bin/env python3
"""The types x batch-sizes grid of profiles/r04_grid_quant.txt: COMPUTE launch time of every quantized type over batch sizes under the plan as
built (tools/kbench.py's event timing, 30 back-to-back launches).  Developer tool, GPU box: python tools/grid_quant.py > gpurun_out/grid.txt"""
import io
import os
import re
import sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kbench  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402

device.init(0)
TYPES = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q4_2"]
SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096)]
NS = [1, 4, 5, 9, 16, 17, 32, 64, 65, 128, 256, 257, 512, 768, 1024, 2048, 3072, 4096]
print("# COMPUTE launch time in us (tools/grid_quant.py = tools/kbench.py, HIP events over 30 back-to-back launches after 3 warm-up launches, real quantised data, one MI355X box,")
print("# one gpurun call, the final plan of round 4) of every quantized type over batch sizes: which kernel form serves what is decided by csrc/plan.cpp from (type, K, N).")
print("# Not a contract measurement (short runs on a chip that has not reached its steady clock read high, an isolated outlier is a one-off stall): the table is for")
print("# spotting forms that are out of line with their neighbours -- DESIGN.md 10.2d.  INIT (4-20 us) not included.")
print("M     K      N   " + "".join(f"{t:>9}" for t in TYPES), flush=True)
for (M, K) in SHAPES:
    for N in NS:
        row = []
        for t in TYPES:
            buf = io.StringIO()
            with redirect_stdout(buf):
                kbench.run(t, M, K, N, 30, check=False)
            m = re.search(r"compute\s+([0-9.]+) us", buf.getvalue())
            row.append(float(m.group(1)) if m else float("nan"))
        print(f"{M:5d} {K:5d} {N:5d}  " + "".join(f"{v:9.1f}" for v in row), flush=True)
