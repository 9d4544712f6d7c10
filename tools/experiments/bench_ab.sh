#!/bin/bash
# developer experiment: bench.py (headline + side configs, weights cold) for the product library and variant builds
# usage: tools/experiments/bench_ab.sh VARIANT ...   (libraries from tools/build_variant.sh)
for v in "" "$@"; do
  if [ -z "$v" ]; then unset GGML_HIP_LIB; echo product; else export GGML_HIP_LIB=$PWD/ggmlsharp_amd/lib/dbg/libggml_hip_$v.so; echo $v; fi
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{\"metric\""):
        d=json.loads(line); oc=d["other_configs"]
        print("  value %.1f  ms %.4f  kernel_us %s" % (d["value"]/1e3, d["ms_per_step"], d["roofline"].get("kernel_us")))
        for k in ("batch32","prompt512","q8_0_ffn512","q5_0_ffn512","vocab512","dense_f16"):
            if k in oc: print("  %-14s %.5f ms (kernel %s)" % (k, oc[k]["ms_per_step"], oc[k].get("compute_kernel_ms")))
'
done
