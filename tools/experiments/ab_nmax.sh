#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for nmax in 512 2048 512 2048; do
  echo "== nmax $nmax" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3P_NMAX=$nmax timeout -k 10 400 python tools/kbench.py --cfg q4_1:4096:4096:768 q4_1:4096:4096:1024 q4_1:4096:4096:2048 q4_1:4096:11008:1024 q4_1:4096:11008:2048 q5_1:4096:4096:1024 q5_1:4096:4096:2048 q5_0:4096:4096:1024 q5_0:4096:4096:2048 q8_0:2048:2048:1024 q8_0:8192:8192:1024 q8_0:4096:14336:1024 q8_0:14336:4096:1024 --iters 60 --no-check >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
