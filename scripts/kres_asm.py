#!/usr/bin/env python3
"""kernel resource usage from a device-only assembly listing (hipcc --cuda-device-only -S):
   scripts/kres_asm.py file.s [mangled-name regex] [--dump DIR]   -> vgpr / sgpr / scratch / occupancy / code bytes / LDS-DMA count"""
import re, sys, subprocess, os
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else '.'
dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
for m in re.finditer(r'^(_Z\w+):\s*; @', txt, re.M):
    name = m.group(1)
    if not re.search(flt, name):
        continue
    j = txt.index('.end_amdhsa_kernel', m.start()) if '.end_amdhsa_kernel' in txt[m.start():] else len(txt)
    tail = txt[j:j + 8000]
    g = lambda k: (re.search(r'; %s: (\d+)' % k, tail) or [None, '?'])[1]
    body = txt[m.start():j]
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    print('%-72s vgpr %s sgpr %s scratch %s occ %s codeB %s | lines %d lds-dma %d vmcnt0 %d' % (
        d[:72], g('NumVgprs'), g('NumSgprs'), g('ScratchSize'), g('Occupancy'), g('codeLenInByte'),
        body.count('\n'), len(re.findall(r'offen.* lds', body)), body.count('s_waitcnt vmcnt(0)')))
    if dump:
        os.makedirs(dump, exist_ok=True)
        open(os.path.join(dump, re.sub(r'\W', '_', d)[:80] + '.s'), 'w').write(body)
