#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kquants.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_fused.py -x -q -m gpu > gpurun_out/ab_sb_tests.log 2>&1 || { echo FAILED >> gpurun_out/ab_sb_tests.log; exit 1; }
for v in mxsb1 mxsb0 mxsb1 mxsb0 mxsb1 mxsb0; do
  echo "== $v" >> gpurun_out/ab_sb.log
  GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so timeout -k 10 300 python tools/kbench.py --cfg q4_0:4096:4096:512 q4_0:4000:4096:512 q4_0:32000:4096:512 q4_0:4096:11008:512 q4_0:4096:4096:300 --iters 300 >> gpurun_out/ab_sb.log 2>&1 || exit 1
done
