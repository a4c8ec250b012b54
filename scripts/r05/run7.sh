#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out/r05g
NS_ROUTE_TIMING=1 timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05g/timing_ctx512.txt 2>&1
grep -E "route timing|replay" gpurun_out/r05g/timing_ctx512.txt | cut -c1-400
NS_ROUTE_TIMING=1 NS_ROUTE_SEG=64 timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05g/timing_seg64_ctx512.txt 2>&1
grep -E "route timing|replay" gpurun_out/r05g/timing_seg64_ctx512.txt | cut -c1-400
NS_ROUTE_TIMING=1 NS_ROUTE_SEG=12 timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05g/timing_seg12_ctx512.txt 2>&1
grep -E "route timing|replay" gpurun_out/r05g/timing_seg12_ctx512.txt | cut -c1-400
