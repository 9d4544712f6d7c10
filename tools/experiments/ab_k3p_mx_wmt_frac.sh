cd /root/repo
export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/libggml_hip_dev.so
CFG="q4_0:11008:4096:129:8 q4_0:11008:4096:192:8 q4_0:11008:4096:256:8 q4_0:11008:4096:320:8 q4_0:5504:4096:512:12 q4_0:9000:4096:256:8 q4_0:8192:8192:192:4 q4_0:14336:4096:256:6 q4_0:4096:4096:320:24 q4_0:4096:4096:512:24"
for v in 0 2; do
  echo "== GGML_HIP_K3P_WMT=$v (0: the plan; 2: 64-row wave tiles)"
  GGML_HIP_K3P_WMT=$v python tools/kbench.py --graph --iters 20 --no-check --cfg $CFG 2>&1 | grep "graph-replayed"
done
