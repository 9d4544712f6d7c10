#!/bin/bash
# usage: tools/isa_stats.sh FILE.hip [extra flags]  -> per-kernel VGPRs / scratch / v_mov / MFMA counts of the gfx950 ISA
set -e
cd "$(dirname "$0")/../ggmlsharp_amd/csrc"
F=$1; shift
mkdir -p ../../tools/bin
FL="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FL "$@" --cuda-device-only -S $F -o ../../tools/bin/${F%.hip}.s
python3 - ../../tools/bin/${F%.hip}.s <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body); sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body)
    acc = re.search(r'\.amdhsa_accum_offset (\d+)', body)
    targs = re.search(r'ILi(\d+)E' * 3, name)
    short = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', name).split('ILi')[0] + '<' + ','.join(re.findall(r'Li(\d+)E', name)) + '>'
    print(f"{short:48s} vgpr {vg.group(1) if vg else '?':>4} accum_off {acc.group(1) if acc else '?':>4} scratch {sc.group(1) if sc else '?':>5} "
          f"v_mov {len(re.findall(r'v_mov_b32', body)):4d} v_accvgpr {len(re.findall(r'v_accvgpr', body)):4d} mfma {len(re.findall(r'v_mfma', body)):4d} lines {body.count(chr(10))}")
PY
