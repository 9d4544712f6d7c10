#!/bin/bash
# A/B of the f16 kernel's mid-size tile configurations (developer tool, GPU box)
set -e
CFG="q8_0:4096:4096:640 q8_0:4096:4096:1024 q8_0:4096:4096:2048 q8_0:11008:4096:1024 q8_0:4096:11008:1024 q8_0:32000:4096:1024 q5_0:4096:4096:1024 q5_0:11008:4096:1024 q5_1:4096:4096:1024 q8_0:16384:4096:640"
for mid in 1 2 3; do
  echo "== GGML_HIP_Q16_MID=$mid"
  GGML_HIP_Q16_MID=$mid python tools/ab_kernels.py $CFG
done
