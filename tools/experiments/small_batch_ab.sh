#!/bin/bash
# developer experiment: tools/small_batch_time.py for the product library and variant builds (tools/build_variant.sh)
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; echo product; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; echo $v; fi
  python tools/small_batch_time.py 2>&1 | grep " x "
done
