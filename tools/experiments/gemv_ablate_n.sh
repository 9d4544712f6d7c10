#!/bin/bash
# developer experiment: batches of 2..4 rows for variant builds of gemv.hip against the product library
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; fi
  python tools/gemv_time.py q4_0:4096:4096:2 q4_0:11008:4096:2 q4_0:32000:4096:2 q4_0:4096:4096:4 q4_0:11008:4096:4 q4_0:32000:4096:4 q8_0:11008:4096:2 2>&1 | grep -v amdgpu.ids
done
