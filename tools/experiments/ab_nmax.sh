#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for t in 384 257 129 384 257 129; do
  echo "== T128 $t" >> gpurun_out/ab_nmax.log
  GGML_HIP_MX_T128=$t timeout -k 10 400 python tools/kbench.py --cfg q4_0:4096:4096:768 q4_0:4096:4096:1024 q4_0:4096:4096:1100 q4_0:4096:4096:1280 q4_0:4096:11008:1024 q4_0:4096:11008:1280 q4_0:8192:8192:600 q4_0:2048:8192:2048 q4_1:4096:4096:2048 q4_1:4096:4096:1280 --iters 60 --no-check >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
