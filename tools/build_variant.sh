#!/bin/bash
# usage: tools/build_variant.sh NAME FILE.hip "-DFOO=1 ..."   -> ggmlsharp_amd/lib/dbg/libggml_hip_NAME.so
# Developer A/B builds: the product objects with ONE source recompiled under -DGGML_HIP_DEV + extra flags (select the
# library with GGML_HIP_LIB=...).  `make dev` builds libggml_hip_dev.so: every source with the developer switches.
set -e
cd "$(dirname "$0")/../ggmlsharp_amd/csrc"
NAME=$1; FILE=$2; EXTRA=$3
make -s all
mkdir -p ../lib/dbg ../lib/obj_dbg
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -DGGML_HIP_DEV -Wno-unused-function -Wno-unused-variable"
case $FILE in gemm_q16.hip|gemm_qmx.hip|gemm_q8s.hip|gemm_qmp.hip) FL="$FL -fno-slp-vectorize";; esac
case $FILE in *.cpp) FL="$FL -x hip";; esac
/opt/rocm/bin/hipcc $FL $EXTRA -c $FILE -o ../lib/obj_dbg/$NAME.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../lib/obj_dbg/$NAME.o -ldl -Wl,--version-script=exports.map -o ../lib/dbg/libggml_hip_$NAME.so
echo built $NAME
