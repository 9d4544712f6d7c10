#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/ab_nmax.log
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in 0 100000 0 100000; do
  echo "== K3P_NMAX $v (0: the plan's own bounds -- Q8_0 / Q5_0 up to 3072 rows; 100000: K3p whatever N)" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3P_NMAX=$v timeout -k 10 400 python tools/kbench.py --no-check --cfg q8_0:4096:11008:4096 q8_0:4096:11008:8192 q5_0:4096:11008:4096 q8_0:4096:8192:4096 q8_0:8192:8192:4096 q8_0:4096:6144:4096 q8_0:4096:28672:4096 q5_0:4096:28672:8192 q8_0:11008:4096:4096 --iters 20 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
