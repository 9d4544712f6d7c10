for v in base NODRAIN RING2 RING8 base; do
  if [ $v = base ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_d16$v.so; fi
  echo "== $v"; timeout -k 5 100 python tools/kbench.py --cfg f16:4096:4096:4096 f16:8192:8192:8192 2>&1 | grep "^f16"
done
