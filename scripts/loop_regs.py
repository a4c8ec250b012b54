#!/usr/bin/env python3
"""diagnostics: distinct VGPRs referenced inside each inner loop of one kernel in a hipcc -S listing"""
import re, sys
txt = open(sys.argv[1]).read()
name = sys.argv[2]
i0 = txt.index(name + ':'); i1 = txt.index('s_endpgm', i0)
lines = txt[i0:i1].split('\n')
def regs_of(body):
    regs = set()
    for l in body:
        l = l.split(';')[0]
        for mm in re.finditer(r'v\[(\d+):(\d+)\]', l):
            regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        for mm in re.finditer(r'\bv(\d+)\b', l):
            regs.add(int(mm.group(1)))
    return regs
print('whole kernel: lines', len(lines), 'vgprs', len(regs_of(lines)))
for i, l in enumerate(lines):
    if 'Inner Loop Header' in l:
        lab = lines[i - 1].split(':')[0]
        for j in range(i, len(lines)):
            if 's_cbranch' in lines[j] and lab + '\n' in lines[j] + '\n' and lines[j].strip().endswith(lab):
                body = lines[i:j]
                print(lab, 'lines', len(body), 'mfma', sum('v_mfma' in x for x in body), 'loads', sum('buffer_load' in x or 'global_load' in x for x in body), 'vgprs', len(regs_of(body)))
                break
