#!/bin/bash
# developer experiment: time variant builds of gemv.hip (tools/build_variant.sh NAME gemv.hip "-D...") against the product library
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; fi
  python tools/gemv_time.py q4_0:4096:4096:1 q4_0:32000:4096:1 q4_0:65536:4096:1 2>&1 | grep -v amdgpu.ids
done
