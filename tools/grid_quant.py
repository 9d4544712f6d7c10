#!/usr/bin/env python3
"""The types x batch-sizes grid behind the plan's thresholds (csrc/plan.cpp): GPU-side COMPUTE time of every quantized type over batch sizes
under the plan as built.  r5: timed as REPLAYED hipGraphs of 64 launches (tools/kbench.py --graph) -- round 4's table used per-call events,
which are host-bound near 8 us per call and read flat below that -- and re-checkable:

    python tools/grid_quant.py > gpurun_out/grid.txt                          # measure (GPU box)
    python tools/grid_quant.py --check profiles/r05_grid_quant.txt --tolerance 8%   # re-measure every cell of a committed table, list the
                                                                              # cells that moved by more than the tolerance, exit 1 if any got SLOWER

A cell that got slower beyond the tolerance is a regression of the plan or of a kernel; boxes of one day differ by +-4 %, hence 8 %."""
import argparse
import io
import os
import re
import sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kbench  # noqa: E402
from ggmlsharp_amd import device  # noqa: E402

TYPES = ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0", "q4_2"]
SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096), (1024, 4096), (32000, 4096)]   # (r5: + a short matrix -- a grouped-query k / v projection -- and a vocabulary-sized one: the family follows M)
NS = [1, 4, 5, 9, 16, 17, 32, 64, 65, 128, 129, 256, 257, 512, 768, 1024, 2048, 4096]


def measure(t, M, K, N):
    copies = max(1, min(16, (300 << 20) // (M * K)))        # rotate weight copies past the Infinity Cache where the matrix is small
    buf = io.StringIO()
    with redirect_stdout(buf):
        kbench.run(t, M, K, N, 5, check=False, copies=copies)
    return kbench.RESULT.get("graph_compute_us", float("nan"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", help="a table written by this tool: re-measure its cells and compare")
    ap.add_argument("--tolerance", default="8%")
    ap.add_argument("--quick", action="store_true", help="--check only a spread of cells (every third row)")
    a = ap.parse_args()
    tol = float(a.tolerance.rstrip("%")) / 100.0
    device.init(0)
    kbench.GRAPH = True
    if a.check:
        rows = [ln.split() for ln in open(a.check) if ln.strip() and not ln.startswith(("#", "M "))]
        worse = better = cells = 0
        for i, r in enumerate(rows):
            if a.quick and i % 3:
                continue
            M, K, N = int(r[0]), int(r[1]), int(r[2])
            for t, old in zip(TYPES, r[3:]):
                old = float(old)
                new = measure(t, M, K, N)
                cells += 1
                if new > old * (1 + tol):
                    worse += 1
                    print(f"SLOWER  {t} {M} x {K} x {N}: {old:.1f} -> {new:.1f} us ({(new / old - 1) * 100:+.0f} %)", flush=True)
                elif new < old * (1 - tol):
                    better += 1
                    print(f"faster  {t} {M} x {K} x {N}: {old:.1f} -> {new:.1f} us ({(new / old - 1) * 100:+.0f} %)", flush=True)
        print(f"grid check: {cells} cells, {worse} slower and {better} faster beyond {a.tolerance}")
        sys.exit(1 if worse else 0)
    print("# COMPUTE time in us per launch, GPU side: replayed hipGraphs of 64 back-to-back COMPUTE launches rotating over weight copies (tools/grid_quant.py = tools/kbench.py --graph),")
    print("# real quantised data, one MI355X box, one gpurun call, under the plan as built (csrc/plan.cpp).  INIT not included.  Re-check with --check FILE --tolerance 8%.")
    print("M     K      N   " + "".join(f"{t:>9}" for t in TYPES), flush=True)
    for (M, K) in SHAPES:
        for N in NS:
            row = [measure(t, M, K, N) for t in TYPES]
            print(f"{M:5d} {K:5d} {N:5d}  " + "".join(f"{v:9.1f}" for v in row), flush=True)


if __name__ == "__main__":
    main()
