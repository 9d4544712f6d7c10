#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for nmax in 2048 4096 2048 4096; do
  echo "== nmax $nmax" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3P_NMAX=$nmax timeout -k 10 400 python tools/kbench.py --cfg q8_0:4096:4096:2560 q8_0:4096:4096:3072 q8_0:4096:4096:3584 q8_0:4096:11008:3072 q8_0:11008:4096:3072 q5_0:4096:4096:3072 q5_0:4096:11008:3072 q8_0:8192:8192:3072 --iters 40 --no-check >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
