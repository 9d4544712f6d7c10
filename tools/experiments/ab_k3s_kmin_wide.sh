cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_0:1536:1536:128:32 q4_0:1536:1536:256:32 q4_0:1536:1536:512:32 q4_0:8960:1536:128:16 q4_0:8960:1536:512:16 q8_0:1536:1536:128:32 q8_0:1536:1536:512:32 q8_0:8960:1536:128:16 q8_0:8960:1536:512:16 q5_1:1536:1536:256:32 q4_0:4096:1024:128:32 q8_0:4096:1024:512:32"
for v in 64 32; do
  echo "== GGML_HIP_K3S_KMIN=$v"
  GGML_HIP_K3S_KMIN=$v python tools/kbench.py --graph --iters 20 --cfg $CFG 2>&1 | grep "graph-replayed\|bad [1-9]\|rror"
done
