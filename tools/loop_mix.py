#!/usr/bin/env python3
"""Instruction mix of the K loop (the innermost loop that holds MFMAs) of one kernel in a gfx950 .s file (tools/isa_stats.sh writes it).
usage: tools/loop_mix.py tools/bin/FILE.s KERNEL_SUBSTRING [--dump]"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
    if pat not in m.group(1):
        continue
    lines = m.group(2).split('\n')
    labels = {mm.group(1): i for i, l in enumerate(lines) if (mm := re.match(r'^(\.LBB\d+_\d+):', l))}
    loops = []
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((i - labels[mm.group(1)], labels[mm.group(1)], i))
    loops = [lp for lp in loops if any('v_mfma' in l for l in lines[lp[1]:lp[2]])]      # the innermost loop that holds MFMAs
    loops.sort()
    n, a, b = loops[0]
    seg = lines[a:b + 1]
    c = Counter()
    for l in seg:
        t = l.strip().split(' ')[0]
        if re.match(r'^(v_|s_|buffer|ds_|scratch|global)', t):
            c[t] += 1
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print(f"{m.group(1)[:70]}: loop of {n} lines; VALU {valu}, MFMA {sum(v for k, v in c.items() if 'mfma' in k)}, "
          f"vmem {sum(v for k, v in c.items() if k.startswith('buffer') or k.startswith('global'))}, ds {sum(v for k, v in c.items() if k.startswith('ds_'))}, "
          f"scratch {sum(v for k, v in c.items() if k.startswith('scratch'))}, s_nop {c.get('s_nop', 0)}, s_waitcnt {c.get('s_waitcnt', 0)}")
    print("  ", ", ".join(f"{k} {v}" for k, v in c.most_common(30)))
    if "--dump" in sys.argv:
        print("\n".join(seg))
