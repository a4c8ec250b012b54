#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_gpu_llama_model.py -x -q > gpurun_out/r05r/pytest_llama.txt 2>&1; tail -5 gpurun_out/r05r/pytest_llama.txt | cut -c1-300
for i in 1 2; do
NS_ROUTE_LINKS=0 NS_ROUTE_TIMING=1 timeout 600 python scripts/dev_llama7b.py device 64 512 2>&1 | grep -v amdgpu.ids | tail -4 | sed 's/^/no carried norms: /'
NS_ROUTE_TIMING=1 NS_ROUTE_DEBUG=1 timeout 600 python scripts/dev_llama7b.py device 64 512 2>&1 | grep -v "amdgpu.ids\|token ended" | tail -6 | sed 's/^/carried norms:    /'
done 2>&1 | tee gpurun_out/r05r/dev7b_512.txt
