#!/bin/bash
# r5: short matrices (a GQA model's k / v projections: 1024 x 4096) at 257..512 src1 rows -- K3p (the plan) | K3s where K3p's grid has fewer than 192 workgroups
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG=${CFG:-"q8_0:1024:4096:384:32 q8_0:1024:4096:512:32 q8_0:512:4096:512:32 q8_0:1024:11008:512:16 q8_0:1536:4096:512:32 q5_1:1024:4096:512:32 q4_0:1024:4096:384:32 q4_0:1024:4096:512:32 q4_0:512:4096:512:32 q4_0:1024:11008:512:16 q4_0:1536:4096:512:32"}
echo "== the plan"
python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]\|rror"
echo "== K3s up to 512 rows below 192 workgroups"
GGML_HIP_K3S_NMAX=512 GGML_HIP_MX_DUAL_NMAX=512 python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad [1-9]\|rror"
