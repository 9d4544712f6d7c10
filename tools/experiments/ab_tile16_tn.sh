#!/bin/bash
# r5: 17..32 src1 rows on 16-row tiles -- two 16-column slices per workgroup (the plan) | one slice per workgroup and twice the workgroups (dev switch GGML_HIP_K3S_16_TN=16)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_0:1024:4096:32:32 q4_0:2048:4096:32:32 q4_0:2048:4096:24:32 q8_0:2048:4096:32:32 q8_0:1024:4096:32:32 q4_0:2048:8192:32:12 q5_1:2048:4096:32:24 q4_0:1536:1536:32:32"}
for v in 0 16; do
  echo "== GGML_HIP_K3S_16_TN=$v"
  GGML_HIP_K3S_16_TN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]\|rror"
done
