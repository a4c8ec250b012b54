#!/bin/bash
# kernel resource usage + prologue shape of an object file's gfx950 kernels:  scripts/kres.sh <file.o> <mangled-name filter>
# prints vgpr / sgpr / spills / kernarg bytes, bytes of code and s_waitcnt lgkmcnt before the first weight (nt) load
set -e
O=$(realpath $1); F=${2:-.}
T=$(mktemp -d); cd $T; cp $O x.o
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1
CO=$(ls x.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $CO > notes.txt
/opt/rocm/lib/llvm/bin/llvm-objdump -d $CO > k.s
python3 - "$F" <<'PY'
import re,sys,subprocess
flt=sys.argv[1]
txt=open('notes.txt').read()
res={}
for b in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',b).group(1)
    g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,b).group(1)
    res[name]=(g('vgpr_count'),g('sgpr_count'),g('vgpr_spill_count'),g('sgpr_spill_count'),g('kernarg_segment_size'))
lines=open('k.s').read().split('\n')
cur=None; start=0; info={}
for ln in lines:
    m=re.match(r'^([0-9a-f]+) <(\S+)>:',ln)
    if m:
        cur=m.group(2); start=int(m.group(1),16); info[cur]=dict(first=None,lgkm=0,wl=0,end=start); continue
    if cur is None: continue
    a=re.search(r'// ([0-9A-F]+):',ln)
    if a: info[cur]['end']=int(a.group(1),16)
    if info[cur]['first'] is None:
        if 's_waitcnt lgkmcnt' in ln: info[cur]['lgkm']+=1
        if 'v_writelane' in ln: info[cur]['wl']+=1
        if 'buffer_load_dwordx4' in ln and ' nt' in ln and a: info[cur]['first']=int(a.group(1),16)-start
for name,(v,sg,vs,ss,ka) in res.items():
    if not re.search(flt,name): continue
    d=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    i=info.get(name,{})
    print('%-78s vgpr %3s sgpr %3s spill v%s s%s kernarg %s | first nt load @%s B, lgkm waits %s, writelanes %s, code %s B'%(d[:78],v,sg,vs,ss,ka,i.get('first'),i.get('lgkm'),i.get('wl'),i.get('end',0)-0 if False else ''))
PY
rm -rf $T
