for v in ${VARS:-base gv8 base}; do
  if [ $v = base ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; fi
  echo "== $v"; timeout -k 5 100 python tools/kbench.py --cfg q4_0:4096:4096:1:32 q4_0:32000:4096:1:8 q4_0:4096:4096:4:32 q4_0:4096:4096:8:32 q8_0:32000:4096:1:8 q4_0:4096:11008:1:16 2>&1 | grep "graph-replayed\|bad"
done
