cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_0:1024:4096:768:32 q4_0:1024:4096:1024:32 q4_0:1024:4096:2048:32 q4_0:2048:4096:1024:24 q4_0:512:4096:1024:32 q4_0:1024:11008:1024:12"
for v in 0 31 32; do
  echo "== GGML_HIP_MX_TILE=$v (0: the plan; 31: never the 64 x 64 form; 32: always)"
  GGML_HIP_MX_TILE=$v python tools/kbench.py --graph --iters 20 --no-check --cfg $CFG 2>&1 | grep "graph-replayed"
done
