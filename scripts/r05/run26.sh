#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r05r
timeout 600 python -m pytest tests/test_gpu_route_replay.py -x -q 2>&1 | tail -2
for ctx in 2048 512; do
for i in 1 2; do
NS_ROUTE_ATTN_FUSE=0 NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"\|\"route\"" | cut -c1-600 | sed "s/^/n_ctx $ctx rope+append launch:   /"
NS_ROUTE_TIMING=1 timeout 300 python scripts/dev_llama7b.py device 64 $ctx 2>&1 | grep "route timing\|\"replay\"\|\"route\"" | cut -c1-600 | sed "s/^/n_ctx $ctx inside the attention: /"
done; done > gpurun_out/r05r/attnfuse.txt
sed 's/{"replay": {"tokens_replayed": \([0-9]*\).*"captured_launches": \([0-9]*\).*"us_median": \([0-9.]*\), "tokens_per_s_median": \([0-9.]*\).*/replayed \1, captured launches \2, median \3 us = \4 tok\/s/; s/{"route".*"tokens": \(\[[^]]*\]\).*/tokens \1/' gpurun_out/r05r/attnfuse.txt | grep -v tokens | cut -c1-200
