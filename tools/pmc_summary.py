#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc / --kernel-trace CSVs under gpurun_out/pmc_<tag>_{a,b} (developer tool)."""
import collections
import csv
import glob
import re
import sys

tag = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else "gemm|gemv|quantize_act")


def short(n):
    m = re.search(r"(\w+_kernel<[^>]*>|\w+_kernel)", n)
    return m.group(1) if m else n[:40]


for part in "ab":
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{part}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            if not pat.search(r["Kernel_Name"]):
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
        for k, d in agg.items():
            print(k, "vgpr/lds/grid/wg", meta[k])
            for c, v in sorted(d.items()):
                print(f"    {c:28s} {sum(v) / len(v) / 1e6:12.3f} M")
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{part}/*/*_kernel_trace.csv"):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in d.items():
            print(f"  [{part}] dur_us {k}: {sum(v) / len(v) / 1e3:.1f} (n={len(v)})")
