cd "$(dirname "$0")/../.."
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q8_0:4096:11008:512:6 q5_0:4096:11008:512:6 q8_0:4096:4096:512:16 q8_0:4096:4096:320:16 q8_0:4096:4096:384:16 q8_0:8192:4096:192:8 q8_0:11008:4096:512:6 q8_0:4096:11008:1024:6 q5_1:4096:11008:512:6 q8_0:4000:4096:512:16"
for v in 4 2; do
  echo "== GGML_HIP_K3P_WMT=$v"
  GGML_HIP_K3P_WMT=$v python tools/kbench.py --no-check --graph --iters 30 --cfg $CFG 2>&1 | grep "graph-replayed"
done
