#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for nmax in 512 2048 512 2048; do
  echo "== mx nmax $nmax" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3P_MX_NMAX=$nmax timeout -k 10 400 python tools/kbench.py --cfg q4_0:4096:4096:576 q4_0:4096:4096:640 q4_0:4096:4096:768 q4_0:4096:4096:1024 q4_0:4096:4096:1536 q4_0:4096:4096:2048 q4_0:4096:11008:768 q4_0:4096:11008:1024 q4_0:4096:11008:2048 q4_0:11008:4096:1024 q4_0:8192:8192:1024 q4_0:2048:2048:1024 --iters 100 --no-check >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
