for v in nt1 nt0 nt1 nt0; do
  if [ $v = nt1 ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_nt0.so; fi
  echo "== $v"; timeout -k 5 100 python tools/kbench.py --cfg q4_0:4096:4096:1:32 q4_0:32000:4096:1:8 q8_0:4096:4096:1:16 --no-check 2>&1 | grep "graph-replayed"
done
