#!/bin/bash
# r5: K3p's weight loads with the nt cache policy (dev build -DK3P_W_AUX=2) against the default: time (replayed graphs) and FETCH_SIZE
cd "$(dirname "$0")/../.."
R=$PWD
CFG="q4_0:32000:4096:512:3 q4_0:4096:4096:512:32 q8_0:4096:11008:512:6 q4_0:11008:4096:512:8"
for v in dev0 devnt; do
  echo "== $v"
  GGML_HIP_LIB=$R/ggmlsharp_amd/lib/libggml_hip_$v.so python tools/kbench.py --no-check --graph --iters 30 --cfg $CFG 2>&1 | grep -v amdgpu.ids
  export TMPDIR=/tmp
  (cd /tmp && GGML_HIP_LIB=$R/ggmlsharp_amd/lib/libggml_hip_$v.so rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/nt_$v -- python3 $R/tools/kbench.py --cfg q4_0:32000:4096:512:3 q4_0:4096:4096:512:32 --iters 3 --no-check > $R/gpurun_out/nt_$v.log 2>&1)
  python3 - $R/gpurun_out/nt_$v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mid_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[(r["Kernel_Name"][:40], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  FETCH_SIZE", k, "launches", len(v), "mean KB raw", round(sum(v) / len(v), 1), "=> x2 corrected MB", round(2 * sum(v) / len(v) / 1024, 1))
PY
done
