#!/usr/bin/env python3
"""Every loop (backward branch) of one kernel in a gfx950 .s file with its instruction counts: which loop holds the scratch traffic?
usage: tools/loop_stats.py tools/bin/FILE.s KERNEL_NAME_SUBSTRING [min_lines]"""
import re, sys
txt = open(sys.argv[1]).read()
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
    if sys.argv[2] not in m.group(1):
        continue
    body = m.group(2).split('\n')
    labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
    print(m.group(1)[:90])
    for i, l in enumerate(body):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i and i - labels[mm.group(1)] >= minl:
            seg = body[labels[mm.group(1)]:i + 1]
            cnt = lambda p: sum(1 for x in seg if re.search(p, x))
            valu, ds = cnt(r'^\s+v_') - cnt('v_mfma'), cnt(r'^\s+ds_')
            print(f"  lines {labels[mm.group(1)]:5d}-{i:5d}: mfma {cnt('v_mfma'):3d} (i8 {cnt('mfma_i32')}, bf16 {cnt('_bf16')}, f32 {cnt('x2_f32')})  valu {valu:4d}  "
                  f"vmem {cnt('buffer_load|global_load'):3d}  ds {ds:3d}  scratch {cnt('scratch_'):3d}")
