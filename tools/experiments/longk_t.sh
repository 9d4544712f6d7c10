#!/bin/bash
# K3p's sliced forms (K > 20480), timings only: the int8 types either side of the one-table limit (r4; see longk.sh for the parity sweep)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/longk_t.log
timeout -k 10 500 python tools/kbench.py --cfg q8_0:4096:20480:512 q8_0:4096:20512:512 q8_0:4096:28672:512 q5_0:4096:20480:512 q5_0:4096:20512:512 q5_0:4096:28672:512 q5_1:4096:28672:512 q8_0:4096:28672:2048 q5_0:4096:28672:2048 --iters 60 > gpurun_out/longk_t.log 2>&1 || exit 2
echo ok
