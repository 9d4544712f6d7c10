#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for n in 17 5 17 5; do
  echo "== KS_NMIN $n" >> gpurun_out/ab_nmax.log
  GGML_HIP_D32_KS_NMIN=$n timeout -k 10 400 python tools/kbench.py --cfg f32:4096:4096:5 f32:4096:4096:8 f32:4096:4096:9 f32:4096:4096:16 f32:4096:11008:8 f32:11008:4096:8 f32:11008:4096:16 f32:32000:4096:8 f32:2048:2048:8 --iters 40 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
