for v in ${VARS:-base k1nostore base}; do
  if [ $v = base ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; fi
  echo "== $v"; timeout -k 5 100 python tools/kbench.py --cfg q4_0:4096:4096:4096 q4_0:4096:4096:512 --no-check 2>&1 | grep "^q4_0"
done
