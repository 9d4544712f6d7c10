#!/bin/bash
# r5: 16-row tiles beyond 64 src1 rows wherever the workgroups fit one round (dev switch GGML_HIP_K3S_16_NMAX: 64 | 512)
cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_0:1024:4096:128:32 q4_0:1024:4096:96:32 q4_0:512:4096:256:32 q4_0:512:4096:128:32 q8_0:1024:4096:128:32 q8_0:512:4096:256:32 q4_0:1024:11008:128:16 q5_1:1024:4096:128:32 q4_2:1024:4096:96:32 q4_0:256:4096:512:32"
for v in 64 512; do
  echo "== GGML_HIP_K3S_16_NMAX=$v"
  GGML_HIP_K3S_16_NMAX=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]\|rror"
done
