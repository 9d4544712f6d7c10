#!/bin/bash
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for v in minstag minloop; do
  echo "== $v" >> gpurun_out/ab_min_prec.log
  GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so timeout -k 10 300 python tools/experiments/min_term_precision.py >> gpurun_out/ab_min_prec.log 2>&1 || exit 1
done
