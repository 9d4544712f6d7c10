#!/bin/bash
# r5: Q4_1 at 33..64 src1 rows -- its MX batched-decode form (the plan: the int8 pair from 65) | the int8 pair from 33 (dev switch GGML_HIP_Q41_PAIR_MIN)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_1:2048:4096:64:24 q4_1:4096:4096:48:16 q4_1:4096:4096:64:16 q4_1:8192:4096:64:8 q4_1:11008:4096:64:6 q4_1:32000:4096:48:3 q4_1:32000:4096:64:3 q4_1:4096:8192:64:8"}
for v in 65 33; do
  echo "== GGML_HIP_Q41_PAIR_MIN=$v"
  GGML_HIP_Q41_PAIR_MIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]"
done
