for v in new old; do
  unset GGML_HIP_MX_TILE GGML_HIP_Q16_OLD128; if [ $v = old ]; then export GGML_HIP_MX_TILE=9 GGML_HIP_Q16_OLD128=1; fi
  echo "== $v"; timeout -k 5 200 python tools/kbench.py --cfg ${CFGS:-q8_0:4096:4096:17 q8_0:4096:4096:64 q5_0:4096:4096:128 q8_0:4096:11008:64 q5_0:11008:4096:64 q8_0:32000:4096:64} 2>&1 | grep "^q"
done
