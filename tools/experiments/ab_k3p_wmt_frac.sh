#!/bin/bash
# r5: K3p wave tile height on grids of a FRACTIONAL number of rounds (a few tiles more than one round of 128-row tiles): 128-row | 64-row forced
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q8_0:11008:4096:192:6 q8_0:11008:4096:256:6 q8_0:11008:4096:129:6 q8_0:9000:4096:256:8 q5_1:11008:4096:192:6 q8_0:8192:8192:192:4 q8_0:11008:4096:384:6 q8_0:14336:4096:256:6 q4_0:11008:4096:320:8 q8_0:5504:4096:512:8"}
for v in 4 2; do
  echo "== GGML_HIP_K3P_WMT=$v"
  GGML_HIP_K3P_WMT=$v python tools/kbench.py --no-check --graph --iters 30 --cfg $CFG 2>&1 | grep "graph-replayed"
done
