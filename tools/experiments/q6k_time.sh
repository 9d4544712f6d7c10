#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/q6k_time.log
timeout -k 10 500 python tools/kbench.py --no-check --cfg q6_k:4096:4096:1 q6_k:4096:4096:8 q6_k:4096:4096:32 q6_k:4096:4096:64 q6_k:4096:4096:128 q6_k:4096:4096:512 q6_k:4096:11008:512 q6_k:4096:4096:4096 q4_2:4096:4096:1 q4_2:4096:4096:512 q4_2:4096:11008:512 q5_k:4096:4096:1 q5_k:4096:11008:512 q6_k:4096:1024:1 q6_k:32000:4096:1 --iters 60 > gpurun_out/q6k_time.log 2>&1 || exit 2
echo ok
