#!/usr/bin/env python3
"""Audit of the gfx950 ISA (a .s from tools/isa_stats.sh) for MFMA-result hazards that hipcc does not pad around inline asm:
for every MFMA, the number of instruction slots until the first later instruction that reads or writes any of its D
registers (except an MFMA taking D whole as C).  Needed: passes + 4 slots (8-pass 12, 16-pass 20).  Linear scan (falls
through branches), so loop-carried distances are over-estimates; developer tool.  usage: mfma_hazard_audit.py file.s [kernel-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_scale_f32_32x32x64_f8f6f4": 8, "v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_32x32x16_bf16": 8,
          "v_mfma_i32_32x32x32_i8": 8, "v_mfma_i32_32x32x16_i8": 8, "v_mfma_f32_32x32x8_f16": 16}
def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out
worst = {}
for km in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = km.group(1), km.group(2)
    if flt not in name: continue
    short = 'k<' + ','.join(re.findall(r'Li(\d+)E', name)) + '>'
    ins = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':')]
    ins = [l for l in ins if not l.startswith(';;')]
    for idx, l in enumerate(ins):
        op = l.split()[0]
        if not op.startswith('v_mfma'): continue
        ops = l[len(op):].split(',')
        d = regs(ops[0])
        need = PASSES.get(op, 8) + 4
        dist = None
        for k in range(idx + 1, min(idx + 1 + 40, len(ins))):
            l2 = ins[k]; op2 = l2.split()[0]
            if op2.startswith('s_nop'):
                continue
            r2 = regs(l2)
            if r2 & d:
                if op2.startswith('v_mfma'):
                    o2 = l2[len(op2):].split(',')
                    if regs(o2[0]) == d and len(o2) >= 4 and regs(o2[3]) == d and not (regs(o2[1]) & d) and not (regs(o2[2]) & d):
                        dist = None; break     # accumulate chain
                # slots = instructions in between, s_nop N counted as N + 1
                slots = 0
                for q in ins[idx + 1:k]:
                    slots += (int(q.split()[1]) + 1) if q.startswith('s_nop') else 1
                dist = (slots, l2[:60])
                break
        if dist is not None and dist[0] < need:
            key = (short, op)
            if key not in worst or dist[0] < worst[key][0]: worst[key] = (dist[0], need, dist[1])
for (short, op), (d, need, l2) in sorted(worst.items()):
    print(f"{short:36s} {op:36s} first touch of D after {d:2d} slots (need {need}): {l2}")
print("kernels flagged:", len({k[0] for k in worst}))
