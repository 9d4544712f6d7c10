#!/usr/bin/env python3
"""Static check of the relaxed stage drains (gemm_qmx.hip, dense16.hip): an inline-asm `s_waitcnt vmcnt(N)` with N > 0 stands for
"the LDS-DMA pieces of the next stage have landed; only the N weight loads issued behind the last piece may still be in flight".
Vector-memory operations complete in issue order, so that holds iff at least N vector-memory instructions stand between the last
`buffer_load ... lds` and the wait in the emitted code -- hipcc may move loads, so the count is checked on the ISA, per kernel.
usage: drain_audit.py FILE.s   (prints one line per drain, exit status 1 on a short one)"""
import re
import sys

VMEM = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "scratch_load", "scratch_store", "flat_")


def audit(path):
    """-> list of (kernel, N, vmem instructions since the last LDS-DMA or None when no DMA precedes the wait)"""
    out = []
    txt = open(path).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
        name = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', m.group(1)).split('ILi')[0] + '<' + ','.join(re.findall(r'Li(\d+)E', m.group(1))) + '>'
        lines = [l.strip() for l in m.group(2).split('\n') if l.strip()]
        inasm = False
        for idx, l in enumerate(lines):
            if l.startswith(';;#ASMSTART'):
                inasm = True
                continue
            if l.startswith(';;#ASMEND'):
                inasm = False
                continue
            w = re.match(r's_waitcnt vmcnt\((\d+)\)$', l)
            if not (inasm and w and int(w.group(1)) > 0):
                continue
            # Walk back over the emitted text.  A label that a FORWARD branch above it jumps to closes a conditional block (the wave-uniform
            # `if` around the scale pieces): its register loads are not counted (a wave that skips the block has not issued them), a DMA piece
            # inside it ends the walk (a wave that runs the block has it as its youngest piece).  A label nothing above jumps to is a loop head.
            n, cnt, found, k, skip_to = int(w.group(1)), 0, False, idx - 1, None
            while k >= 0:
                t = lines[k]
                if t.startswith('buffer_load') and ' lds' in t:
                    found = True
                    break
                if t.endswith(':') and not t.startswith(';'):
                    lab = t[:-1]
                    if skip_to is None and any(re.match(r's_c?branch\w*\s+' + re.escape(lab) + r'$', lines[q]) for q in range(max(0, k - 400), k)):
                        skip_to = lab
                    elif skip_to is None:
                        break
                elif skip_to is not None and re.match(r's_c?branch\w*\s+' + re.escape(skip_to) + r'$', t):
                    skip_to = None
                elif t.startswith(VMEM) and skip_to is None:
                    cnt += 1
                k -= 1
            out.append((name, n, cnt if found else None))
    return out


if __name__ == "__main__":
    bad = 0
    for name, n, cnt in audit(sys.argv[1]):
        ok = cnt is not None and cnt >= n
        bad += not ok
        print(f"{name:44s} vmcnt({n}) behind {cnt} vector-memory instructions since the last LDS-DMA piece: {'ok' if ok else 'SHORT'}")
    sys.exit(1 if bad else 0)
