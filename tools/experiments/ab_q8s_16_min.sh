#!/bin/bash
# r5: the min-term types (Q5_1; Q5_K / Q4_K in its form) on 16-row batched-decode tiles (dev switch GGML_HIP_Q8S_16_MIN: 0 = the 32-row form)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q5_1:4096:4096:9:24 q5_1:4096:4096:16:24 q5_1:4096:4096:32:24 q5_1:4096:11008:16:10 q5_1:4096:11008:32:10 q5_1:2048:8192:32:12 q5_1:4096:2048:16:32 q5_1:4096:28672:16:4"}
KQ=${KQ:-"q5_k:4096:4096:16:24 q4_k:4096:4096:32:24 q4_k:4096:11008:16:10"}
for v in 0 1; do
  echo "== GGML_HIP_Q8S_16_MIN=$v (0: 32-row tiles; 1: 16-row tiles)"
  GGML_HIP_Q8S_16_MIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad\|FAIL\|rror"
  GGML_HIP_Q8S_16_MIN=$v python tools/kbench.py --graph --iters 20 --no-check --cfg $KQ 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|FAIL\|rror"
done
