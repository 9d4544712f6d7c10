#!/bin/bash
# r5: the int8 batched-decode form with FOUR tiles per workgroup beyond 512 tile groups (dev switch GGML_HIP_Q8S_TILES: 2 = the former rule's two)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q8_0:11008:4096:64:6 q5_0:11008:4096:64:6 q8_0:32000:4096:32:3 q8_0:32000:4096:64:3 q8_0:8192:8192:96:4 q8_0:28672:8192:32:2 q8_0:11008:4096:128:6 q8_0:32000:4096:128:3"}
for v in 2 0; do
  echo "== GGML_HIP_Q8S_TILES=$v (2: two tiles per workgroup; 0: the plan's rule)"
  GGML_HIP_Q8S_TILES=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]"
done
