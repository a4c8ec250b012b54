#!/bin/bash
# round 5, GPU visit 4: replay of the reference's device route — parity tests, then the 7B-shaped model through model_eval
set -x
mkdir -p gpurun_out/r05d
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_llama_model.py tests/test_gpu_device_lazy.py tests/test_gpu_device_mha.py tests/test_gpu_tp_first_contact.py -x -q > gpurun_out/r05d/pytest.txt 2>&1
tail -15 gpurun_out/r05d/pytest.txt
timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05d/dev7b_replay_ctx512.txt 2>&1
tail -4 gpurun_out/r05d/dev7b_replay_ctx512.txt
NS_DEVICE_REPLAY=0 timeout 600 python scripts/dev_llama7b.py device 64 512 > gpurun_out/r05d/dev7b_eager_ctx512.txt 2>&1
tail -3 gpurun_out/r05d/dev7b_eager_ctx512.txt
NS_DEV7B_PROMPT=1900 timeout 600 python scripts/dev_llama7b.py device 64 2048 > gpurun_out/r05d/dev7b_replay_ctx2048.txt 2>&1
tail -3 gpurun_out/r05d/dev7b_replay_ctx2048.txt
NS_DEVICE_REPLAY=0 NS_DEV7B_PROMPT=1900 timeout 600 python scripts/dev_llama7b.py device 64 2048 > gpurun_out/r05d/dev7b_eager_ctx2048.txt 2>&1
tail -3 gpurun_out/r05d/dev7b_eager_ctx2048.txt
