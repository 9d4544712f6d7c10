#!/bin/bash
# developer experiment: tools/seam1_time.py for variant builds of seams.cpp (build_variant.sh NAME seams.cpp "-D...") against the product library
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; echo product; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; echo $v; fi
  python tools/seam1_time.py 2>&1 | grep -o '"workload": "[^"]*"\|"ms_per_step": [0-9.]*' | paste - -
done
