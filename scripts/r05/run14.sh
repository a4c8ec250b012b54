#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05o
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_norm_link.py tests/test_gpu_whole_token_7b.py tests/test_gpu_decoder_layer.py -x -q > gpurun_out/r05o/pytest.txt 2>&1; tail -15 gpurun_out/r05o/pytest.txt | cut -c1-300
for i in 1 2; do
NS_BENCH_MERGE_IN_WO=0 python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two launches', d['full_token'])"
python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge in WO  ', d['full_token'])"
done
