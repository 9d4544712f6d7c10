#!/bin/bash
# the k-quant extension types: the GPU tests of tests/test_kquants.py, then decode-step timings (graph-replayed whole calls) -- r4, kquants.hip / gemv.hip K8
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out && rm -f gpurun_out/q6k_time.log gpurun_out/kq_tests.log
timeout -k 10 700 python -m pytest tests/test_kquants.py -x -q -m gpu > gpurun_out/kq_tests.log 2>&1 || { echo FAILED >> gpurun_out/kq_tests.log; exit 1; }
timeout -k 10 500 python tools/kbench.py --no-check --cfg q6_k:4096:4096:1 q6_k:4096:4096:4 q6_k:4096:11008:1 q6_k:32000:4096:1 q6_k:4096:1024:1 q5_k:4096:4096:1 q4_k:4096:4096:1 q5_k:4096:4096:2 q5_k:4096:4096:4 q5_k:4096:11008:1 q5_k:11008:4096:1 q5_k:32000:4096:1 q5_1:4096:4096:1 q5_1:4096:4096:2 q5_1:4096:4096:4 q5_1:4096:11008:1 q5_1:32000:4096:1 q4_0:4096:4096:1 --iters 60 > gpurun_out/q6k_time.log 2>&1 || exit 2
echo ok
