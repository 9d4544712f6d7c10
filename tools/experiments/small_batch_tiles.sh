#!/bin/bash
# developer experiment: the small-batch MX tile forms (GGML_HIP_MX_TILE: 13 = 32-row, 15 = 64-row, 12 = 128-row tiles, 25 = the
# 64-column form for N <= 32) with the weights cold in HBM, on the developer library (make dev)
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
for v in ${MX_TILE_VALUES:-0 13 15 12 25}; do
  echo "GGML_HIP_MX_TILE=$v"
  GGML_HIP_MX_TILE=$v python tools/small_batch_time.py "$@" 2>&1 | grep " x "
done
