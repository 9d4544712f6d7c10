#!/bin/bash
# the batched-decode forms' XCD-aware workgroup -> tile order (common.h k3s_tile_of): shapes with two and more column tiles, small and large M
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/k3s_map.log
timeout -k 10 500 python tools/kbench.py --cfg q8_0:4096:4096:64 q8_0:11008:4096:64 q8_0:32000:4096:64 q4_0:11008:4096:64 q4_0:32000:4096:64 q5_1:11008:4096:48 q8_0:4096:11008:64 q8_0:4096:11008:128 q4_0:4096:11008:128 q4_2:11008:4096:128 q4_2:32000:4096:128 q4_2:4096:4096:256 q8_0:8192:8192:64 q8_0:4100:4096:64 --iters 30 > gpurun_out/k3s_map.log 2>&1 || exit 2
echo ok
