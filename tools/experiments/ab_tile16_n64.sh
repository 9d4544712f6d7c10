#!/bin/bash
# r5: 33..64 src1 rows on a short matrix -- 32-row tiles (the plan) | 16-row tiles x 32 columns per workgroup, two column groups (dev switch GGML_HIP_K3S_16_NMAX=64)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_0:1024:4096:64:32 q4_0:2048:4096:64:32 q4_0:2048:4096:48:32 q8_0:2048:4096:64:32 q8_0:1024:4096:64:32 q4_0:2048:8192:64:12 q5_1:2048:4096:64:24 q4_0:1536:1536:64:32 q4_1:2048:4096:64:32"}
for v in 32 64; do
  echo "== GGML_HIP_K3S_16_NMAX=$v"
  GGML_HIP_K3S_16_NMAX=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]\|rror"
done
