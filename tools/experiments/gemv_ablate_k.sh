#!/bin/bash
# developer experiment: multi-chunk K shapes for variant builds of gemv.hip against the product library
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; fi
  python tools/gemv_time.py q4_0:4096:11008:1 q4_0:11008:11008:1 q4_0:32000:8192:1 q8_0:4096:11008:1 q4_0:4096:8192:1 2>&1 | grep -v amdgpu.ids
done
