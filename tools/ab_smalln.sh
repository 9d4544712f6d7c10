for v in new old; do
  unset GGML_HIP_MX_TILE; if [ $v = old ]; then export GGML_HIP_MX_TILE=9; fi
  echo "== $v"; timeout -k 5 200 python tools/kbench.py --cfg q4_0:4096:4096:17 q4_0:4096:4096:64 q4_0:4096:4096:128 q4_0:11008:4096:64 q4_0:4096:11008:64 q4_0:32000:4096:64 q4_0:32000:4096:128 q4_1:4096:4096:64 2>&1 | grep "^q4"
done
