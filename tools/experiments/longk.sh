#!/bin/bash
# K3p beyond one table (K > 19968): parity sweep, then timings at the K values either side of the old limit
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/longk_*.log
timeout -k 10 900 python tools/sweep_k3p.py 21 140 > gpurun_out/longk_sweep.log 2>&1 || exit 1
timeout -k 10 500 python tools/kbench.py --cfg q4_0:4096:4096:512 q8_0:4096:11008:512 q5_1:4096:11008:512 q4_0:4096:19968:512 q4_0:4096:20480:512 q4_0:4096:20512:512 q4_0:4096:28672:512 q8_0:4096:19968:512 q8_0:4096:20480:512 q8_0:4096:20512:512 q8_0:4096:28672:512 q5_1:4096:20480:512 q5_1:4096:28672:512 q8_0:8192:28672:512 q4_0:8192:28672:512 q8_0:4096:53248:384 --iters 60 > gpurun_out/longk_bench.log 2>&1 || exit 2
echo ok
