#!/bin/bash
cd /root/repo
for nw in 0 2 4 8 16; do
NS_GV_NW=$nw python bench.py --secondary-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c4=d.get('config4',{}); c5=d.get('config5',{})
print('NS_GV_NW=$nw config5 ms', c5.get('ms_per_step'), 'us/layer', c5.get('us_per_layer'), 'lm_head', c5.get('us_lm_head'), '| config4 us/layer', c4.get('us_per_layer'))"
done
