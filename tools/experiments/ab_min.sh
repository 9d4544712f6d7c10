#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kquants.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/ab_min_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ab_min_tests.log
for v in minh minh3 minh1 minp minh minh3 minh1 minp; do
  echo "== $v" >> gpurun_out/ab_min_time.log
  GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so timeout -k 10 300 python tools/kbench.py --cfg q5_1:4096:11008:512 q5_1:4096:4096:512 q4_1:4096:4096:512 q4_1:4096:11008:512 q5_1:32000:4096:512 --iters 200 >> gpurun_out/ab_min_time.log 2>&1 || exit 1
done
for v in pre1 pre3; do echo "== $v q51"; timeout -k 10 120 tools/bin/k3p_trace_$v 4096 11008 512 q51 || exit 1; done > gpurun_out/k3p_trace_min.log 2>&1
