#!/bin/bash
# r5: K3p for the two-scale types (Q4_2, Q6_K in its form): staged int8 kernel | gemm_q8_mid_kernel<Q4_2> (dev switch GGML_HIP_K3P_2SC_NMIN)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_2:4096:4096:129:8 q4_2:4096:4096:256:8 q4_2:4096:4096:512:8 q4_2:4096:4096:1024:8 q4_2:4096:4096:2048:8 q4_2:4096:4096:4096:8 q4_2:4096:11008:512:4 q4_2:11008:4096:512:4 q4_2:11008:4096:192:4 q4_2:32000:4096:512:2 q4_2:4096:11008:2048:4"}
KQ=${KQ:-"q6_k:4096:4096:512:8 q6_k:4096:11008:512:4 q6_k:4096:4096:192:8"}
for v in 100000 129; do
  echo "== GGML_HIP_K3P_2SC_NMIN=$v (100000: the staged int8 kernel; 129: K3p from 129 rows)"
  GGML_HIP_K3P_2SC_NMIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]\|FAIL\|rror"
  GGML_HIP_K3P_2SC_NMIN=$v python tools/kbench.py --graph --iters 20 --no-check --cfg $KQ 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|FAIL\|rror"
done
