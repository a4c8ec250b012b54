#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05p
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_whole_token_7b.py tests/test_gpu_decoder_layer.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r05p/pytest_attention.txt 2>&1; tail -5 gpurun_out/r05p/pytest_attention.txt | cut -c1-300
timeout 600 python scripts/r05/attn_stream_ab.py 2048 512 4096 8192 128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05p/attn_stream_ab.txt | cut -c1-330
for i in 1 2; do
NS_ATTN_STREAM=0 python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('registers', d['full_token'])"
python bench.py --full-token-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lds rings', d['full_token'])"
done 2>&1 | tee gpurun_out/r05p/full_token_ab.txt
