#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05q
one() { python bench.py --secondary-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c4=d.get('config4',{}); c5=d.get('config5',{})
print('$1 config4 us/layer', c4.get('us_per_layer'), 'frac', c4.get('frac_of_8TBps'), 'tok/s', c4.get('tokens_per_s'))"; }
for i in 1 2 3; do
NS_LIB_PATH=/root/repo/gpurun_tmp/libns_hip_aperm0.so one "A re-ordered in LDS (round 4)  "
one "A fragments permuted at the read"
done 2>&1 | tee gpurun_out/r05q/aperm_ab.txt
