#!/bin/bash
# r5: K3s behind K = 512 .. 1023 (two or three k-blocks per wave): wins at K = 768 (Q4_0 768 x 768 x 16 4.4 -> 3.4 us, Q5_1 3072 x 768 x 16 6.0 -> 4.2), loses at K = 512 (2048 x 512 x 32 3.9 -> 4.3): 1024 stays the bound
cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_0:768:768:16:32 q4_0:3072:768:32:32 q4_0:2048:512:32:32 q4_0:4096:512:64:32 q8_0:3072:768:32:32 q8_0:2048:512:16:32 q5_1:3072:768:16:32 q4_0:768:768:64:32"
for v in 32 16; do
  echo "== GGML_HIP_K3S_KMIN=$v"
  GGML_HIP_K3S_KMIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]\|rror"
done
