for d in ${DBGS:-0 1 2 4 8 16 32}; do
  if [ $d = 0 ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_dbg$d.so; fi
  echo "== dbg $d"; timeout -k 5 100 python tools/kbench.py --cfg ${CFGS:-q4_0:4096:4096:4096} --no-check 2>&1 | grep "^q4_0"
done
