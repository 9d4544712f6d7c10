#!/bin/bash
# r5: Q4_0 at 65..128 src1 rows -- the plan (K3s behind K >= 11008, the staged four-way forms else) | K3s whatever M | K3p whatever M (dev switch GGML_HIP_MX_DUAL_WGS)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_0:2048:4096:128:32 q4_0:4096:4096:96:24 q4_0:4096:4096:128:24 q4_0:8192:4096:128:12 q4_0:11008:4096:65:8 q4_0:11008:4096:128:8 q4_0:16384:4096:128:6 q4_0:32000:4096:96:4 q4_0:32000:4096:128:4 q4_0:4096:11008:128:8 q4_0:8192:8192:128:6 q4_0:11008:11008:128:3"}
for v in 0 1000000 1; do
  echo "== GGML_HIP_MX_DUAL_WGS=$v (0: the plan; 1000000: K3s whatever M; 1: K3p whatever M)"
  GGML_HIP_MX_DUAL_WGS=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]\|rror"
done
