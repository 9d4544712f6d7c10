#!/bin/bash
# r5: where K3s-int8 and K3p-int8 could both serve (65..256 src1 rows) -- which wins by M?  (dev switches GGML_HIP_K3P_NMIN / GGML_HIP_K3S_NMAX)
cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
A=${A:-"q8_0:4096:4096:96:16 q8_0:4096:4096:128:16 q8_0:8192:4096:96:8 q8_0:8192:4096:128:8 q8_0:11008:4096:65:6 q8_0:11008:4096:128:6 q8_0:16384:4096:128:4 q8_0:32000:4096:65:3 q8_0:32000:4096:128:3 q8_0:4096:11008:128:6 q8_0:8192:8192:128:4 q5_1:11008:4096:128:6 q5_1:32000:4096:96:3 q4_2:32000:4096:128:3"}
B=${B:-"q8_0:1024:4096:256:32 q8_0:2048:4096:192:32 q8_0:2048:4096:256:32 q8_0:1024:11008:256:16 q8_0:2048:8192:160:12 q5_1:2048:4096:256:32"}
echo "== 65..128 rows: K3s (the plan) | K3p from 65"
python tools/kbench.py --graph --iters 20 --no-check --cfg $A 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|rror"
GGML_HIP_K3P_NMIN=65 GGML_HIP_K3P_2SC_NMIN=65 GGML_HIP_K3S_NMAX=64 GGML_HIP_K3S_NMAX_2SC=64 python tools/kbench.py --graph --iters 20 --no-check --cfg $A 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|rror"
echo "== 129..256 rows on short matrices: K3p (the plan) | K3s up to 256"
python tools/kbench.py --graph --iters 20 --no-check --cfg $B 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|rror"
GGML_HIP_K3P_NMIN=257 GGML_HIP_K3S_NMAX=256 python tools/kbench.py --graph --iters 20 --no-check --cfg $B 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|rror"
