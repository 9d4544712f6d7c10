#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for nmin in 257 129 257 129; do
  echo "== K3P_NMIN $nmin" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3P_NMIN=$nmin timeout -k 10 400 python tools/kbench.py --cfg q4_1:4096:4096:160 q4_1:4096:4096:192 q4_1:4096:4096:256 q4_1:4096:11008:160 q4_1:4096:11008:192 q4_1:11008:4096:160 q4_1:11008:4096:192 q4_1:8192:8192:192 q8_0:4096:4096:129 q8_0:11008:4096:129 q8_0:4096:11008:129 q8_0:8192:8192:129 q8_0:14336:4096:140 --iters 60 --no-check >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
