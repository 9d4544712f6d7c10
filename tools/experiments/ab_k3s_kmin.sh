#!/bin/bash
# r5: the batched-decode forms behind a SHORT K (1024 <= K < 2048: a small model's hidden size) -- the staged forms (the plan) | K3s from 32 k-blocks (dev switch GGML_HIP_K3S_KMIN)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_0:1536:1536:16:32 q4_0:1536:1536:32:32 q4_0:1536:1536:64:32 q4_0:8960:1536:16:16 q4_0:8960:1536:32:16 q8_0:1536:1536:32:32 q8_0:8960:1536:32:16 q4_0:4096:1024:32:32 q8_0:4096:1024:32:32 q5_1:8960:1536:16:16 q4_0:2048:1792:32:32"}
for v in 64 32; do
  echo "== GGML_HIP_K3S_KMIN=$v"
  GGML_HIP_K3S_KMIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]\|rror"
done
