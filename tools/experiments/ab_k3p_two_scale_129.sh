cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_2:4096:4096:129:8 q4_2:4096:4096:192:8 q4_2:4096:4096:256:8 q4_2:4096:11008:192:4 q4_2:4096:11008:256:4 q4_2:11008:4096:129:4 q4_2:11008:4096:256:4 q4_2:32000:4096:192:2 q4_2:8192:8192:160:2 q4_2:2048:4096:256:8"
for v in 256 128; do
  echo "== GGML_HIP_K3S_NMAX_2SC=$v (256: K3s up to 256 rows; 128: K3p from 129)"
  GGML_HIP_K3S_NMAX_2SC=$v GGML_HIP_K3P_2SC_NMIN=129 python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep -v amdgpu.ids | grep "graph-replayed\|bad\|FAIL\|rror"
done
