#!/bin/bash
# r5: 16-row batched-decode tiles beyond one round of the chip (dev switches GGML_HIP_Q8S_16_WGS / GGML_HIP_K3S_16_WGS): 256 | 384 workgroups allowed -- Q4_0 5120 x 5120 x 16 / 32 15.0 | 16.6, 14.9 | 22.0 us,
# Q8_0 16.7 | 17.0, 16.8 | 23.5, 6144 x 4096 x 32 11.6 | 15.5: one round stays the limit
cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_0:5120:5120:16:16 q4_0:5120:5120:32:16 q8_0:5120:5120:16:12 q8_0:5120:5120:32:12 q4_0:6144:4096:16:16 q8_0:6144:4096:32:12 q4_0:8192:4096:16:12 q5_1:5120:5120:16:12"
for v in 256 384 512; do
  echo "== 16-row tiles up to $v workgroups"
  GGML_HIP_Q8S_16_WGS=$v GGML_HIP_K3S_16_WGS=$v python tools/kbench.py --graph --iters 20 --no-check --cfg $CFG 2>&1 | grep "graph-replayed"
done
