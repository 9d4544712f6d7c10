#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in 128 256 128 256; do
  echo "== D16_S4_NMAX $v" >> gpurun_out/ab_nmax.log
  GGML_HIP_D16_S4_NMAX=$v timeout -k 10 400 python tools/kbench.py --cfg f16:4096:4096:129 f16:4096:4096:192 f16:4096:4096:256 f16:4096:11008:160 f16:4096:11008:256 f16:11008:4096:160 f16:11008:4096:256 f16:8192:8192:192 f16:2048:2048:256 --iters 60 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
