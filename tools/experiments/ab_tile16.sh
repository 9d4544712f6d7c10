#!/bin/bash
# r5: the batched-decode forms on 16-row tiles against the 32-row tiles (developer switches of the dev build: workgroup limit of the 16-row form)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_0:4096:4096:32:32 q8_0:4096:4096:32:24 q5_0:4096:4096:32:24 q4_0:4096:4096:16:32 q8_0:4096:4096:8:24 q4_0:4096:4096:64:32 q8_0:4096:4096:64:24 q4_0:4096:11008:32:12 q8_0:4096:11008:64:8 q4_0:8192:8192:32:8 q4_0:11008:4096:32:12 q8_0:11008:4096:64:8 q4_0:2048:8192:32:32"}
for lim in ${LIMS:-0 256 100000}; do
  echo "== 16-row tiles up to $lim workgroups"
  GGML_HIP_K3S_16_WGS=$lim GGML_HIP_Q8S_16_WGS=$lim python tools/kbench.py --no-check --graph --iters 50 --cfg $CFG 2>&1 | grep -v amdgpu.ids
done
