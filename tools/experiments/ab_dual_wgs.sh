#!/bin/bash
# r5: the workgroup count from which K3p takes over in the shared range (dev switches GGML_HIP_K3_DUAL_WGS / GGML_HIP_MX_DUAL_WGS): 192 (the plan) | 160 | 128
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q8_0:11008:4096:64:6 q5_1:11008:4096:64:6 q4_2:11008:4096:64:6 q4_0:11008:4096:64:8 q8_0:10240:4096:48:6 q8_0:5120:4096:128:12 q8_0:5120:13824:96:4 q4_0:5120:5120:128:12 q8_0:8192:4096:64:8 q8_0:4096:4096:128:16 q4_0:8192:4096:64:8 q8_0:2560:4096:256:16 q4_0:2560:4096:256:16"}
for v in 192 160 128; do
  echo "== DUAL_WGS=$v"
  GGML_HIP_K3_DUAL_WGS=$v GGML_HIP_MX_DUAL_WGS=$v python tools/kbench.py --graph --iters 20 --no-check --cfg $CFG 2>&1 | grep "graph-replayed"
done
