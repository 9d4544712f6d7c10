#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/ab_nmax.log
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in 2 4 2 4; do
  echo "== K3S_COLS $v (Q4_0 / Q4_1 on the batched-decode MX form up to 32 x this many rows whatever K) -- with the XCD-aware tile order" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3S_COLS=$v timeout -k 10 400 python tools/kbench.py --no-check --cfg q4_0:4096:4096:128 q4_0:11008:4096:128 q4_0:32000:4096:128 q4_0:13824:5120:128 q4_0:28672:8192:128 q4_0:8192:8192:128 q4_1:11008:4096:96 q4_1:4096:4096:128 --iters 20 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
