#!/bin/bash
# usage: tools/pmc.sh <tag> <kbench cfg...>   (run on the GPU box through gpurun) -- collects SQ counters in two passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/tools/kbench.py --cfg "$@" --iters 3 --no-check > $R/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/tools/kbench.py --cfg "$@" --iters 3 --no-check > $R/gpurun_out/pmc_${tag}_b.log 2>&1
python3 - <<PY
import csv, glob, collections
for part in "ab":
    for f in glob.glob("$R/gpurun_out/pmc_${tag}_%s/*/*_counter_collection.csv" % part):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if "gemm" in k or "gemv" in k or "quantize_act" in k:
                print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "(millions)")
    for f in glob.glob("$R/gpurun_out/pmc_${tag}_%s/*/*_kernel_trace.csv" % part):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in d.items():
            if "gemm" in k or "gemv" in k or "quantize_act" in k:
                print("  dur_us", k, round(sum(v) / len(v) / 1e3, 1), "VGPR/LDS see trace")
PY
