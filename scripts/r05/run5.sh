#!/bin/bash
set -x
mkdir -p gpurun_out/r05h
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_llama_model.py tests/test_gpu_reference_graph.py tests/test_gpu_device_lazy.py tests/test_gpu_device_mha.py -x -q > gpurun_out/r05h/pytest.txt 2>&1
tail -5 gpurun_out/r05h/pytest.txt
NS_ROUTE_TIMING=1 timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05h/dev7b_replay_ctx512.txt 2>&1
grep -E "replay|route" gpurun_out/r05h/dev7b_replay_ctx512.txt | cut -c1-700
NS_ROUTE_TIMING=1 NS_ROUTE_ROPE_APPEND=0 timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05h/dev7b_replay_noappend_ctx512.txt 2>&1
grep -E "replay|route" gpurun_out/r05h/dev7b_replay_noappend_ctx512.txt | cut -c1-700
NS_ROUTE_TIMING=1 NS_DEV7B_PROMPT=1500 timeout 600 python scripts/dev_llama7b.py device 64 2048 > gpurun_out/r05h/dev7b_replay_ctx2048.txt 2>&1
grep -E "replay|route" gpurun_out/r05h/dev7b_replay_ctx2048.txt | cut -c1-700
NS_DEVICE_REPLAY=0 NS_DEV7B_PROMPT=1500 timeout 600 python scripts/dev_llama7b.py device 64 2048 > gpurun_out/r05h/dev7b_eager_ctx2048.txt 2>&1
grep -E "replay|route" gpurun_out/r05h/dev7b_eager_ctx2048.txt | cut -c1-700
