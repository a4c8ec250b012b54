#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05s
timeout 600 python -m pytest tests/test_gpu_gemm3.py -x -q 2>&1 | tail -2
NS_G3_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_gemm3.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
for pz in 0 1; do
NS_G3_PERSIST=$pz timeout 300 python scripts/r05/prefill_probe.py 2048 40 --no-lib 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
row = []
for q in ('int4', 'int8'):
    r = d[q]
    row.append(q + ' ' + ' '.join('%s %.0f/%.0f' % (k.replace('x4096','').replace('4096x','K'), v['tflops'], v['tflops_f32_only']) for k, v in r.items() if 'tflops_f32_only' in v) + ' gate/up %.0f ffn %.0f' % (r['gate_up_pairs_fp16_out']['tflops'], r['ffn_fused_entry']['tflops']))
print('persist $pz | ' + ' | '.join(row))"
done; done | tee gpurun_out/r05s/persist_ab.txt
