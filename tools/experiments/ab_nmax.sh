#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/ab_nmax.log
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in 64 128 64 128; do
  echo "== K3S_NMAX $v (the one-scale int8 types on the batched-decode form up to this many rows; beyond: the staged forms) -- long K" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3S_NMAX=$v timeout -k 10 400 python tools/kbench.py --no-check --cfg q8_0:4096:8192:128 q8_0:8192:8192:96 q8_0:11008:11008:128 q8_0:32000:8192:128 q8_0:5120:13824:128 q8_0:8192:28672:128 q5_1:5120:13824:96 q5_0:8192:28672:128 q8_0:13824:5120:128 q8_0:28672:8192:128 --iters 20 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
