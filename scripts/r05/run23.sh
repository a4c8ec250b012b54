#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05r
for ctx in 2048 512; do
for i in 1 2; do
for s0 in 64 6 16; do
NS_ROUTE_SEG0=$s0 NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"" | cut -c1-400 | sed "s/^/n_ctx $ctx first segment $s0: /"
done; done; done | tee gpurun_out/r05r/seg0.txt | grep -o "n_ctx.*segment [0-9]*\|GPU span.*\|tokens_per_s_median[^,]*"
