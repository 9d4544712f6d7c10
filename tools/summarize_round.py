#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag> + gpurun_out/pmc_<tag>_* (tools/profile_round.sh <tag>) into the committed profiles/ files.
usage: python tools/summarize_round.py r02"""
import sys
T = sys.argv[1] if len(sys.argv) > 1 else "r02"
import collections
import csv
import glob
import json
import re
import shutil

PAT = re.compile("gemm_q|gemv_q|gemv_fused|quantize_act|dense16|dense32|convert_act")


def short(n):
    m = re.search(r"(\w+_kernel<[^>]*>|\w+_kernel)", n)
    return m.group(1) if m else n[:40]


import os


def newest(pattern):
    """gpurun merges every run's files into gpurun_out/: take the latest of each kind"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


shutil.copy(newest(f"gpurun_out/prof_{T}/*/*_kernel_stats.csv")[0], f"profiles/{T}_bench_kernel_stats.csv")
# the JSON line bench.py printed while rocprofv3 --stats was attached (its own HIP-event timing of the dominant kernel is the
# number to compare with the stats average: both are taken in the same, profiled, run) and the un-profiled line
for src, dst in ((f"gpurun_out/prof_{T}.log", f"profiles/{T}_bench_under_rocprof.json"), (f"gpurun_out/bench_{T}.json", f"profiles/{T}_bench.json")):
    js = [ln for ln in open(src).read().splitlines() if ln.startswith('{"metric"')]
    if js:
        open(dst, "w").write(js[-1] + "\n")
lines = [f"# rocprofv3 --pmc summaries, {T}, MI355X (gfx950), ROCm 7.2.  Averages per dispatch, in millions.",
         "# Collected by tools/profile_round.sh around tools/kbench.py --cfg q4_0:4096:4096:4096 q4_0:4096:4096:1:32 (separate passes",
         "# per counter set, --kernel-trace only).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.",
         "# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads half the bytes of a 16-B/lane stream",
         "# (MI355X_MICROARCH.md, HBM) -- the batch-1 mat-vec confirms it against its 10.5 MB of algorithmic bytes.", ""]
traffic = {}
for d in sorted(glob.glob(f"gpurun_out/pmc_{T}_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in newest(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if PAT.search(r["Kernel_Name"]):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = (r["VGPR_Count"], r.get("Accum_VGPR_Count", "?"), r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
    dur = collections.defaultdict(list)
    for f in newest(d + "*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if PAT.search(r["Kernel_Name"]):
                dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines.append(f"## pass {d.split('pmc_' + T + '_')[1].strip('/')}")
    for k, c in sorted(agg.items()):
        lines.append(f"{k}  vgpr/agpr/lds/grid/wg {meta[k]}  dur_us {sum(dur[k]) / len(dur[k]) / 1e3:.1f} (n={len(dur[k])})")
        for name, v in sorted(c.items()):
            lines.append(f"    {name:28s} {sum(v) / len(v) / 1e6:12.4f} M")
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic.setdefault(k, {})[name + "_KB"] = round(sum(v) / len(v), 1)
    lines.append("")
open(f"profiles/{T}_pmc_summary.txt", "w").write("\n".join(lines))
out = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC (separate --pmc passes). bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: "
                   "FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream; the batch-1 kernel confirms it)."}
# matched by prefix so that an added template parameter does not silently drop a kernel from the file
prefixes = [("gemm_qmx_kernel<2, 2, 4, 4, 1,", "gemm_qmx_kernel<Q4_0,2,4,4,1> M=4096 K=4096 N=4096"),
            ("gemv_fused_kernel<2, 1,", "gemv_fused_kernel<Q4_0,1> M=4096 K=4096 N=1"),
            ("quantize_act_bf6_kernel", "quantize_act_bf6_kernel N=4096 K=4096")]
names = {k: nm for k in traffic for (pre, nm) in prefixes if k.startswith(pre)}
assert len(names) == len(prefixes), (sorted(traffic), names)
for k, t in traffic.items():
    if k in names and "FETCH_SIZE_KB" in t and "WRITE_SIZE_KB" in t:
        out[names[k]] = dict(t, traffic_bytes=int((2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024))
json.dump(out, open(f"profiles/{T}_traffic.json", "w"), indent=1)
print(open(f"profiles/{T}_pmc_summary.txt").read())
print(json.dumps(out, indent=1))
