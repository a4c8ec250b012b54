#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05q
timeout 900 python -m pytest tests/test_gpu_gemvs.py tests/test_gpu_configs.py -x -q > gpurun_out/r05q/pytest.txt 2>&1; tail -4 gpurun_out/r05q/pytest.txt | cut -c1-300
for i in 1 2 3; do python bench.py --secondary-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c4=d.get('config4',{}); c5=d.get('config5',{})
print('config4 us/layer', c4.get('us_per_layer'), 'frac', c4.get('frac_of_8TBps'), 'tok/s', c4.get('tokens_per_s'), '| config5 ms', c5.get('ms_per_step'), 'parity', max(c4.get('parity_rel_l2_vs_oracle',{}).values()))"; done 2>&1 | tee gpurun_out/r05q/secondary.txt
