#!/bin/bash
# r5: tall matrices at 33..64 src1 rows -- K3s (the plan) | K3p from 192 workgroups of 64-row tiles (one column tile: 12288 rows and more) (dev switch GGML_HIP_K3_DUAL_NMIN)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q8_0:16384:4096:64:4 q8_0:32000:4096:33:3 q8_0:32000:4096:64:3 q5_1:32000:4096:64:3 q4_2:32000:4096:64:3 q4_1:32000:4096:64:3 q4_0:16384:4096:64:4 q4_0:32000:4096:48:3 q4_0:32000:4096:64:3 q8_0:14336:4096:64:4 q4_0:28672:8192:64:2 q8_0:28672:8192:64:2 q8_0:12288:11008:64:2"}
for v in 65 33; do
  echo "== GGML_HIP_K3_DUAL_NMIN=$v"
  GGML_HIP_K3_DUAL_NMIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]\|rror"
done
