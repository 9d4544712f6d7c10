#!/bin/bash
# r5: Q4_0 at 129..256 src1 rows -- the staged forms (the plan) | the stage-free forms by M (dev switch GGML_HIP_MX_DUAL_NMAX=256)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q4_0:1024:4096:256:32 q4_0:2048:4096:192:32 q4_0:2048:4096:256:32 q4_0:4096:4096:129:24 q4_0:4096:4096:192:24 q4_0:4096:4096:256:24 q4_0:8192:4096:192:12 q4_0:11008:4096:192:8 q4_0:11008:4096:256:8 q4_0:32000:4096:192:4 q4_0:32000:4096:256:4 q4_0:4096:11008:192:8 q4_0:4096:11008:256:8 q4_0:8192:8192:192:6 q4_0:1024:11008:256:16"}
for v in 128 256; do
  echo "== GGML_HIP_MX_DUAL_NMAX=$v"
  GGML_HIP_MX_DUAL_NMAX=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]\|rror"
done
