#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>_{stats,p1..pN} (tools/pmc_cfgs.sh <tag> ...) into profiles/<out>.txt.
usage: python tools/summarize_cfgs.py <tag> <out-name> [kernel-name-regex]"""
import collections
import csv
import glob
import os
import re
import sys

tag, out = sys.argv[1], sys.argv[2]
PAT = re.compile(sys.argv[3] if len(sys.argv) > 3 else "gemm_q|gemv_|quantize_act|dense16|convert_act|layer_")


def short(n):
    m = re.search(r"(\w+_kernel<[^>]*>|\w+_kernel)", n)
    return m.group(1) if m else n[:60]


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


lines = [f"# rocprofv3 kernel-trace stats + --pmc passes (each pass its own run, --kernel-trace only), tag {tag}, MI355X (gfx950), ROCm 7.2.",
         "# Counter values: average per dispatch, in millions (FETCH_SIZE / WRITE_SIZE in KB, not millions).",
         "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles.",
         "# HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md).", ""]
for f in newest(f"gpurun_out/pmc_{tag}_stats/*/*_kernel_stats.csv"):
    lines.append("## kernel-trace --stats (kbench: 3 warm-up + 20 timed launches of INIT and of COMPUTE per config)")
    for r in csv.DictReader(open(f)):
        if PAT.search(r["Name"]):
            lines.append(f"{short(r['Name']):70s} calls {r['Calls']:>4s}  avg_us {float(r['AverageNs']) / 1e3:8.1f}  min_us {float(r['MinNs']) / 1e3:8.1f}  max_us {float(r['MaxNs']) / 1e3:8.1f}")
    lines.append("")
for f in newest(f"gpurun_out/pmc_{tag}_stats.log"):
    lines.append("## kbench's own HIP-event timings in the same (profiled) run")
    lines += [ln.rstrip() for ln in open(f) if re.match(r"^(q\d|f16|f32)", ln)]
    lines.append("")
hbm = collections.defaultdict(dict)          # kernel -> {"FETCH_SIZE": KB, "WRITE_SIZE": KB, "dur_us": ...}
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_p*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in newest(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if PAT.search(r["Kernel_Name"]):
                k = short(r["Kernel_Name"]) + f" grid {r['Grid_Size']}"
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = f"vgpr {r['VGPR_Count']} agpr {r.get('Accum_VGPR_Count', '?')} lds {r['LDS_Block_Size']} wg {r['Workgroup_Size']}"
    dur = collections.defaultdict(list)
    for f in newest(d + "*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if PAT.search(r["Kernel_Name"]):
                dur[short(r["Kernel_Name"]) + f" grid {r['Grid_Size_X']}"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines.append(f"## pass {d.rstrip('/').split('_')[-1]}")
    for k, c in sorted(agg.items()):
        du = dur.get(k, [0])
        lines.append(f"{k}  {meta[k]}  dur_us {sum(du) / len(du) / 1e3:.1f} (n={len(du)})")
        for name, v in sorted(c.items()):
            a = sum(v) / len(v)
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                lines.append(f"    {name:28s} {a:14.1f} KB")
                hbm[k][name] = a
                hbm[k]["dur_us"] = sum(du) / len(du) / 1e3
            else:
                lines.append(f"    {name:28s} {a / 1e6:14.4f} M")
    lines.append("")
lines.append("## HBM-side traffic against the 8 TB/s roofline (per launch: 2 * FETCH_SIZE + WRITE_SIZE, over the kernel's duration in the FETCH_SIZE pass)")
for k, v in sorted(hbm.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and v.get("dur_us", 0) > 0:
        b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        lines.append(f"{k}: {b / 1e6:8.1f} MB per launch / {v['dur_us']:.1f} us = {b / v['dur_us'] / 1e3:7.1f} GB/s = {b / v['dur_us'] / 1e3 / 8000:.3f} of 8 TB/s")
lines.append("")
out = out[:-4] if out.endswith(".txt") else out
open(f"profiles/{out}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
