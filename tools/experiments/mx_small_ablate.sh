#!/bin/bash
# developer experiment: small-batch (9..128 rows) MX mat-mat forms for variant builds of gemm_qmx.hip against the product library
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; echo product; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; echo $v; fi
  for cfg in q4_0:4096:4096:16 q4_0:4096:4096:64 q4_0:11008:4096:32 q4_0:4096:11008:32 q4_0:32000:4096:32 q4_0:4096:4096:128 q4_1:4096:4096:32; do
    python tools/kbench.py --cfg $cfg --iters 30 --no-check 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
