#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/ab_nmax.log
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in 17 9 5 17 9 5; do
  echo "== K3S_NMIN_2SC $v (Q4_2 on the batched-decode form from this many rows; below: its mat-vec) -- INIT + COMPUTE, and the graph-replayed whole call up to 8 rows" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3S_NMIN_2SC=$v timeout -k 10 400 python tools/kbench.py --cfg q4_2:4096:4096:5 q4_2:4096:4096:8 q4_2:4096:4096:9 q4_2:4096:4096:16 q4_2:4096:11008:8 q4_2:4096:11008:16 q4_2:11008:4096:8 q4_2:11008:4096:16 q4_2:32000:4096:8 q4_2:32000:4096:16 --iters 20 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
