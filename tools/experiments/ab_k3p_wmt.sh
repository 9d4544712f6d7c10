#!/bin/bash
# r5: K3p-int8 on 64-row wave tiles (where 128-row tiles leave CUs idle) against the 128-row tiles everywhere (dev build switch GGML_HIP_K3P_WMT)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q8_0:4096:4096:192:16 q8_0:4096:4096:256:16 q5_0:4096:4096:129:16 q5_1:4096:4096:256:16 q8_0:4096:11008:192:6 q8_0:4096:11008:256:6 q5_1:4096:11008:256:6 q8_0:2048:4096:512:32 q4_1:2048:4096:300:32 q8_0:4096:28672:192:3 q8_0:1024:4096:512:32"}
for v in 4 0; do
  echo "== GGML_HIP_K3P_WMT=$v (4: 128-row wave tiles everywhere; 0: the plan's rule)"
  GGML_HIP_K3P_WMT=$v python tools/kbench.py --no-check --graph --iters 30 --cfg $CFG 2>&1 | grep -v amdgpu.ids
done
