#!/bin/bash
# developer A/B on one GPU box (through gpurun): the dev library (make -C ggmlsharp_amd/csrc dev) under one of plan.cpp's developer switches,
# alternating arms inside ONE call (boxes differ by several per cent).  Edit the switch, its values and the kbench shapes below; the
# round-4 bounds of plan.cpp (K3p ranges, tile-count thresholds, the dense forms) were all measured with it.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/ab_nmax.log
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in 9 5 9 5; do
  echo "== K3S_NMIN_KQ $v (Q5_K / Q4_K on the batched-decode form from this many rows; below: INIT + the mat-vec) -- the graph-replayed whole calls" >> gpurun_out/ab_nmax.log
  GGML_HIP_K3S_NMIN_KQ=$v timeout -k 10 400 python tools/kbench.py --no-check --cfg q5_k:4096:4096:5 q5_k:4096:4096:8 q5_k:4096:11008:5 q5_k:4096:11008:8 q5_k:11008:4096:8 q5_k:32000:4096:8 q4_k:8192:8192:6 --iters 20 >> gpurun_out/ab_nmax.log 2>&1 || exit 1
done
