#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05r
timeout 900 python -m pytest tests/test_gpu_route_replay.py tests/test_gpu_device_lazy.py tests/test_gpu_llama_model.py tests/test_gpu_device_backend.py -x -q 2>&1 | tail -4
for ctx in 512 2048; do
for i in 1 2; do
NS_ROUTE_LINKS=0 NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"\|\"route\"" | cut -c1-400 | sed "s/^/n_ctx $ctx norms launched: /"
NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"\|\"route\"" | cut -c1-400 | sed "s/^/n_ctx $ctx norms carried:  /"
done; done | tee gpurun_out/r05r/ab2.txt
