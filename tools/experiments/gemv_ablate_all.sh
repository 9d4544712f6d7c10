#!/bin/bash
# developer experiment: the whole tools/gemv_time.py shape list for variant builds of gemv.hip against the product library
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; fi
  python tools/gemv_time.py 2>&1 | grep -v amdgpu.ids
done
