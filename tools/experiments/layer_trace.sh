#!/bin/bash
# usage (GPU box): bash tools/experiments/layer_trace.sh  -- kernel trace of the drop-in decode layer at batch 1: per-kernel durations and the gaps between them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/layer_trace -- python3 $R/tools/layer_time.py 1 > $R/gpurun_out/layer_trace.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, os, collections
f = sorted(glob.glob("$R/gpurun_out/layer_trace/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
import re
def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:40]
# the last 20 graph computes: find a repeating period by the scatter_copy kernel (the scope's last node)
names = [short(r["Kernel_Name"]) for r in rows]
ends = [i for i, n in enumerate(names) if n.startswith("scatter_copy")]
per = []
for a, b in zip(ends[-21:-1], ends[-20:]):
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"]); t1 = int(seg[-1]["End_Timestamp"])
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    per.append((t1 - t0, dur, len(seg)))
import statistics
print("graph computes analysed:", len(per), "nodes per compute:", per[-1][2])
print("span first-start -> last-end (us): median %.1f" % (statistics.median(p[0] for p in per) / 1e3))
print("sum of kernel durations (us): median %.1f" % (statistics.median(p[1] for p in per) / 1e3))
seg = rows[ends[-2] + 1:ends[-1] + 1]
prev = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("  gap %5.2f us  dur %6.2f us  grid %8s wg %4s  %s" % (gap, (e - s) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"], short(r["Kernel_Name"])))
    prev = e
PY
