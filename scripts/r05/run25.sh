#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05r
for ctx in 2048 512; do
for i in 1 2; do
NS_MHA_INLAUNCH=0 NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"" | cut -c1-400 | sed "s/^/n_ctx $ctx merge launch:    /"
NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"" | cut -c1-400 | sed "s/^/n_ctx $ctx merge in launch: /"
done; done | tee gpurun_out/r05r/inlaunch.txt | grep -o "n_ctx [0-9]* merge[a-z ]*: route.*\|tokens_per_s_median[^,]*"
# a long prompt: 1500 tokens then 64 single-token evals at n_ctx 2048 (12+ live ranges)
for i in 1; do
NS_MHA_INLAUNCH=0 NS_ROUTE_TIMING=1 timeout 600 python scripts/dev_llama7b.py device 64 2048 1500 2>&1 | grep "route timing\|\"replay\"" | cut -c1-400 | sed "s/^/prompt 1500 merge launch:    /"
NS_ROUTE_TIMING=1 timeout 600 python scripts/dev_llama7b.py device 64 2048 1500 2>&1 | grep "route timing\|\"replay\"" | cut -c1-400 | sed "s/^/prompt 1500 merge in launch: /"
done | tee -a gpurun_out/r05r/inlaunch.txt | grep -o "prompt.*: route.*\|tokens_per_s_median[^,]*"
