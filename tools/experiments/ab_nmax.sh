#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for var in 0 5 7 0 5 7; do
  echo "== D16_TILE $var" >> gpurun_out/ab_nmax.log
  GGML_HIP_D16_TILE=$var timeout -k 10 400 python tools/kbench.py --cfg f16:11008:4096:257 f16:11008:4096:384 f16:11008:4096:512 f16:4096:11008:512 f16:8192:8192:512 f16:4096:4096:512 f16:4096:4096:257 f16:14336:4096:512 f16:32000:4096:512 --iters 40 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
