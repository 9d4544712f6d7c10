#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kquants.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_fused.py -x -q -m gpu > gpurun_out/ab_tab_tests.log 2>&1 || { echo FAILED >> gpurun_out/ab_tab_tests.log; exit 1; }
for v in dma1 dma0 dma1 dma0; do
  echo "== $v" >> gpurun_out/ab_tab.log
  GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so timeout -k 10 300 python tools/kbench.py --cfg q8_0:4096:11008:512 q5_0:4096:11008:512 q5_1:4096:11008:512 q4_0:4096:4096:512 q4_0:32000:4096:512 q4_0:4096:11008:512 q8_0:4096:4096:300 --iters 200 >> gpurun_out/ab_tab.log 2>&1 || exit 1
done
(echo "== mx"; timeout -k 10 120 tools/bin/k3p_trace 4096 4096 512 mx; echo "== i8"; timeout -k 10 120 tools/bin/k3p_trace 4096 11008 512 i8; echo "== q51"; timeout -k 10 120 tools/bin/k3p_trace 4096 11008 512 q51) > gpurun_out/k3p_trace_dma.log 2>&1
