#!/usr/bin/env python3
"""Audit of the gfx950 ISA (a .s from tools/isa_stats.sh) for MFMA-result hazards that hipcc does not pad: an MFMA's D
registers may not be read or written for passes + 4 issue slots (8-pass 12, 16-pass 20), and hipcc only guarantees that
for instructions it generated itself -- not for the body of an inline-asm statement (between ;;#ASMSTART and ;;#ASMEND).
For every MFMA this reports the slots until the first later instruction INSIDE an asm body that touches its D registers
(an MFMA that itself sits inside an asm body is invisible to hipcc, so every toucher counts for it; --all reports hipcc's own
touchers for the others too; an MFMA taking D whole as C is an accumulate chain and exempt).
Linear scan (falls through branches, 48 instructions ahead), so loop-carried distances are over-estimates.
usage: mfma_hazard_audit.py file.s [--all]      exit status 1 if an asm toucher is too close"""
import re
import sys

PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_scale_f32_32x32x64_f8f6f4": 8, "v_mfma_f32_32x32x16_f16": 8,
          "v_mfma_f32_32x32x16_bf16": 8, "v_mfma_i32_32x32x32_i8": 8, "v_mfma_i32_32x32x16_i8": 8, "v_mfma_f32_32x32x8_f16": 16}


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def audit(path, include_compiler=False):
    """-> list of (kernel, mfma opcode, slots, needed, toucher text, toucher is in asm)"""
    txt = open(path).read()
    found = {}
    for km in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        name, body = km.group(1), km.group(2)
        short = 'k<' + ','.join(re.findall(r'Li(\d+)E', name)) + '>'
        ins, in_asm = [], False
        for raw in body.split('\n'):
            l = raw.strip()
            if l.startswith(';;#ASMSTART'):
                in_asm = True
                continue
            if l.startswith(';;#ASMEND'):
                in_asm = False
                continue
            if not l or l.startswith((';', '.')) or l.endswith(':'):
                continue
            ins.append((l, in_asm))
        for idx, (l, mf_asm) in enumerate(ins):
            op = l.split()[0]
            if not op.startswith('v_mfma'):
                continue
            d = regs(l[len(op):].split(',')[0])
            need = PASSES.get(op, 8) + 4
            slots = 0
            for k in range(idx + 1, min(idx + 49, len(ins))):
                l2, asm2 = ins[k]
                op2 = l2.split()[0]
                if op2.startswith('s_nop'):
                    slots += int(l2.split()[1]) + 1
                    continue
                if regs(l2) & d:
                    if op2.startswith('v_mfma'):
                        o2 = l2[len(op2):].split(',')
                        if len(o2) >= 4 and regs(o2[0]) == d and regs(o2[3]) == d and not (regs(o2[1]) & d) and not (regs(o2[2]) & d):
                            break                    # accumulate chain: the pipe orders it
                    if asm2 or mf_asm or include_compiler:     # (an MFMA inside an asm body is invisible to hipcc: everything counts)
                        if slots < need:
                            key = (short, op, asm2)
                            if key not in found or slots < found[key][0]:
                                found[key] = (slots, need, l2[:60])
                        break
                    # a compiler-generated toucher: hipcc has padded it; what follows is ordered behind it
                    break
                slots += 1
    return [(k[0], k[1], v[0], v[1], v[2], k[2]) for k, v in sorted(found.items())]


if __name__ == "__main__":
    res = audit(sys.argv[1], "--all" in sys.argv)
    for kern, op, slots, need, text, in_asm in res:
        print(f"{kern:36s} {op:36s} D touched after {slots:2d} slots (need {need}) by {'ASM ' if in_asm else 'hipcc'}: {text}")
    bad = [r for r in res if r[5] or "--all" not in sys.argv]
    print(f"MFMA results touched too early inside inline asm: {len(bad)}")
    sys.exit(1 if bad else 0)
