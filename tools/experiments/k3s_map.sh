#!/bin/bash
# the batched-decode forms' XCD-aware workgroup -> tile order (common.h k3s_tile_of): one column tile (identity order) must not have moved
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/k3s_map.log
timeout -k 10 500 python tools/kbench.py --no-check --cfg q4_0:4096:4096:32 q4_1:4096:4096:32 q4_0:4096:4096:16 q4_0:4096:4096:32 q8_0:4096:4096:32 q4_0:4096:4096:33 q4_0:4096:4096:32 q4_1:4096:4096:32 --iters 60 > gpurun_out/k3s_map.log 2>&1 || exit 2
echo ok
